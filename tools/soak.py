"""Stability soak (development aid): mixed searches and encodes for a few minutes; checks results stay identical
and device memory does not grow.  python tools/soak.py [seconds]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from claude_semantic_search_amd.flat_index import IndexFlatIP
from claude_semantic_search_amd.mpnet_encoder import MpnetEncoder
from claude_semantic_search_amd import synth

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
ix = IndexFlatIP(768)
ix.reserve(4_000_000)
ix.add_synthetic(4_000_000, seed=7)
enc = MpnetEncoder(synthetic_seed=1, compute="bf16")
q = synth.rows(600, 768, 99)
allow = np.random.default_rng(0).random(4_000_000) < 0.3
ref = {n: ix.search(q[:n], 10, normalize=True) for n in (1, 3, 8, 40, 600)}
refm = ix.search(q[:40], 10, normalize=True, allow=allow)
ids = [list(range(4, 4 + n)) for n in (5, 60, 384, 17)]
ids = [[0] + s[:382] + [2] for s in ids]
eref = enc.encode_ids(ids)
# a batch of > 1024 tokens: the LayerNorm-folded GEMM path (integer-atomic row statistics: bit-reproducible)
big = [[0] + [4 + (7 * i + j) % 30000 for j in range(n)] + [2] for i, n in enumerate((382, 200, 1, 333, 77, 250, 129, 64))]
bref = enc.encode_ids(big)
# a small clustered index whose queries overflow their candidate buffers: the device-side exact fix-up
cent = synth.rows(40, 768, 11)
cx = IndexFlatIP(768)
cx.add(np.repeat(cent, 6000, axis=0) + 0.01 * synth.rows(240_000, 768, 12), normalize=True)
cx.set_search_mode("coarse")
cq = cent[:24] + 0.005 * synth.rows(24, 768, 13)
cref = cx.search(cq, 10, normalize=True)
assert cx.last_flagged() > 0
free0 = torch.cuda.mem_get_info()[0]
t0 = time.time(); it = 0
while time.time() - t0 < secs:
    for n in (1, 3, 8, 40, 600):
        D, I = ix.search(q[:n], 10, normalize=True)
        assert (I == ref[n][1]).all() and np.array_equal(D, ref[n][0]), f"search nq={n} changed at iteration {it}"
    D, I = ix.search(q[:40], 10, normalize=True, allow=allow)
    assert (I == refm[1]).all()
    e = enc.encode_ids(ids)
    assert np.array_equal(e, eref), f"encode changed at iteration {it}"
    assert np.array_equal(enc.encode_ids(big), bref), f"folded-path encode changed at iteration {it}"
    D, I = cx.search(cq, 10, normalize=True)
    # (6000 near-duplicates per centre: scores tie within an fp32 rounding, and the 24-query search alternates between the
    # int8-MFMA sweep + exact fix-up and, for the 16 searches after it flagged, the bf16 batch scan + second pass, whose
    # exact scores are summed in another order: equal within 1e-6, ids may swap inside ties -- tools/cx_probe.py)
    assert np.allclose(D, cref[0], atol=1e-6, rtol=0), f"fix-up search changed at iteration {it}"
    it += 1
    if it % 50 == 0:
        print(f"iteration {it}, {time.time() - t0:.0f}s, free HBM delta {(free0 - torch.cuda.mem_get_info()[0]) / 1e6:.1f} MB", flush=True)
print(f"soak ok: {it} iterations in {time.time() - t0:.0f}s; free HBM delta {(free0 - torch.cuda.mem_get_info()[0]) / 1e6:.1f} MB")
