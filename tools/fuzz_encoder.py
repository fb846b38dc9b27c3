#!/usr/bin/env python3
"""Differential fuzz of the encoder (development aid): random ragged batches through the bf16 product mode (small-batch
kernels below 1024 tokens, LayerNorm-folded GEMM path above) against the fp32 verification mode of the same library
(itself held against the oracle by the tests): per-row cosine >= 1 - 1e-3, repeat runs bit-identical.
python tools/fuzz_encoder.py [seconds] [seed]"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np

from claude_semantic_search_amd.mpnet_encoder import MpnetEncoder

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
layers = 3
e16 = MpnetEncoder(synthetic_seed=11, compute="bf16", cfg_overrides={"num_layers": layers})
e32 = MpnetEncoder(synthetic_seed=11, compute="fp32", cfg_overrides={"num_layers": layers})
vocab = e16.cfg["vocab"]
t0 = time.time()
it = 0
worst = 1.0
while time.time() - t0 < secs:
    style = rng.choice(["few_long", "many_short", "mixed", "boundary"])
    if style == "few_long":
        lens = rng.integers(200, 385, size=int(rng.integers(1, 12))).tolist()
    elif style == "many_short":
        lens = rng.integers(1, 9, size=int(rng.integers(100, 600))).tolist()
    elif style == "mixed":
        lens = np.clip(rng.geometric(0.01, size=int(rng.integers(2, 40))), 1, 384).tolist()
    else:   # token totals around the path switch (1024) and the 256-row tile edges
        total = int(rng.choice([1023, 1024, 1025, 1279, 1280, 1281, 2048, 2049]))
        lens = []
        while total > 0:
            l = int(min(total, rng.integers(1, 385)))
            lens.append(l)
            total -= l
    batch = [[0] + rng.integers(4, vocab - 1, size=max(l - 2, 0)).tolist() + ([2] if l > 1 else []) for l in lens]
    batch = [s[:384] for s in batch]
    a = e16.encode_ids(batch)
    b = e32.encode_ids(batch)
    cos = (a * b).sum(1)
    tag = f"it={it} style={style} B={len(batch)} T={sum(map(len, batch))}"
    assert np.isfinite(a).all() and cos.min() > 1 - 1e-3, f"min cos {cos.min()}: " + tag
    assert np.array_equal(a, e16.encode_ids(batch)), "not reproducible: " + tag
    worst = min(worst, float(cos.min()))
    it += 1
    if it % 25 == 0:
        print(f"{it} batches, {time.time() - t0:.0f}s, worst cos so far {worst:.6f} (last: {tag})", flush=True)
print(f"fuzz ok: {it} batches in {time.time() - t0:.0f}s, worst cos {worst:.6f}")
