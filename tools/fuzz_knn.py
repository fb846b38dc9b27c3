#!/usr/bin/env python3
"""Differential fuzz of the search paths (development aid): random index shapes / data / batch sizes, the candidate
path (coarse scan or sweep + fp32 rescoring + device fix-up) against the exact fp32 kernels of the same library.
Scores must agree to 2e-5 (both are fp32 sums in different orders), ids wherever the exact scores of neighbouring
ranks differ by more than 1e-5.  python tools/fuzz_knn.py [seconds] [seed]"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np

from claude_semantic_search_amd import synth
from claude_semantic_search_amd.flat_index import IndexFlat

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time()
it = 0
while time.time() - t0 < secs:
    d = int(rng.choice([64, 128, 384, 768]))
    n = int(rng.choice([1500, 20_000, 90_000, 400_000]))
    metric = int(rng.integers(0, 2))
    norm = bool(rng.integers(0, 2)) if metric == 0 else False
    kind = rng.choice(["uniform", "clustered", "dups", "wide_norms"])
    seed = int(rng.integers(1, 1 << 30))
    x = synth.rows(n, d, seed)
    if kind == "clustered":
        c = synth.rows(max(2, n // 300), d, seed + 1)
        x = c[np.arange(n) % c.shape[0]] + 0.03 * x
    elif kind == "dups":
        x[n // 3: n // 3 + min(n // 4, 7000)] = x[7]
    elif kind == "wide_norms":
        x = x * np.exp(rng.normal(0, 0.7, size=(n, 1))).astype(np.float32)
    nq = int(rng.choice([1, 2, 4, 7, 16, 40, 255, 256, 700]))
    k = int(rng.choice([1, 10, 100]))
    q = synth.rows(nq, d, seed + 2)
    if kind in ("clustered", "dups") and nq > 1:
        q[: nq // 2] = x[rng.integers(0, n, size=nq // 2)] + 0.01 * q[: nq // 2]
    allow = (rng.random(n) < 0.6) if rng.random() < 0.25 else None
    ix = IndexFlat(d, metric)
    noshadow = rng.random() < 0.2
    if noshadow:
        ix.set_shadow(0)      # candidate scores from the fp32 rows (split operands) + fp32 rescoring
    ix.add(x, normalize=norm)
    res = {}
    for mode in ("exact_fp32", "coarse"):
        ix.set_search_mode(mode)
        res[mode] = ix.search(q, k, normalize=norm, allow=allow) if allow is not None else ix.search(q, k, normalize=norm)
    De, Ie = res["exact_fp32"]
    Dc, Ic = res["coarse"]
    tag = f"it={it} d={d} n={n} metric={metric} norm={norm} kind={kind} nq={nq} k={k} masked={allow is not None} noshadow={noshadow} seed={seed}"
    valid = Ie >= 0
    assert ((Ic >= 0) == valid).all(), "padding differs: " + tag
    scale = max(1.0, float(np.abs(De[valid]).max())) if valid.any() else 1.0
    assert np.allclose(Dc[valid], De[valid], atol=2e-5 * scale, rtol=0), f"scores differ by {np.abs(Dc[valid] - De[valid]).max()}: " + tag
    gaps = np.abs(np.diff(De.astype(np.float64), axis=1))
    safe = valid.copy()
    safe[:, 1:] &= gaps > 1e-5 * scale
    safe[:, :-1] &= gaps > 1e-5 * scale
    safe[:, -1] = False          # the rank behind the last slot is unknown
    bad = safe & (Ic != Ie)
    assert not bad.any(), f"{int(bad.sum())} id mismatches: " + tag
    ix.close()
    it += 1
    if it % 20 == 0:
        print(f"{it} cases, {time.time() - t0:.0f}s (last: {tag})", flush=True)
print(f"fuzz ok: {it} cases in {time.time() - t0:.0f}s")
