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
    elif kind == "dups":   # > 4096 copies: flagged -> second coarse pass; > 32768: on to the exact sweep
        x[n // 3: n // 3 + min(n // 4, int(rng.choice([7000, 7000, 40_000])))] = x[7]
    elif kind == "wide_norms":
        x = x * np.exp(rng.normal(0, 0.7, size=(n, 1))).astype(np.float32)
    nq = int(rng.choice([1, 2, 4, 7, 16, 40, 255, 256, 700]))
    k = int(rng.choice([1, 10, 100, 100, 129, 300, 700]))     # > 128: passes of 128 over the rows not returned yet
    if k > 128 and nq > 7:
        nq = int(rng.choice([1, 2, 5]))                        # (k > 128 goes query by query)
    q = synth.rows(nq, d, seed + 2)
    if kind in ("clustered", "dups") and nq > 1:
        q[: nq // 2] = x[rng.integers(0, n, size=nq // 2)] + 0.01 * q[: nq // 2]
    allow = (rng.random(n) < 0.6) if rng.random() < 0.25 else None
    ix = IndexFlat(d, metric)
    noshadow = rng.random() < 0.2
    i8only = (not noshadow) and metric == 0 and d % 256 == 0 and rng.random() < 0.3
    if i8only:
        ix.set_shadow("int8")  # int8 rows without bf16 rows: int8 sweep / scan, else bf16 scratch ranges / exact kernels
    if noshadow:
        ix.set_shadow(0)      # batches: bf16 rows rounded range by range into scratch memory ("coarse"), or split operands ("split")
        ix.set_range_rows(int(rng.choice([0, 0, 4096, 33_000])))
    ix.add(x, normalize=norm)
    res = {}
    for mode in ("exact_fp32", "coarse") + (("split",) if noshadow and rng.random() < 0.5 else ()):
        ix.set_search_mode(mode)
        res[mode] = ix.search(q, k, normalize=norm, allow=allow) if allow is not None else ix.search(q, k, normalize=norm)
    De, Ie = res["exact_fp32"]
    tag0 = f"it={it} d={d} n={n} metric={metric} norm={norm} kind={kind} nq={nq} k={k} masked={allow is not None} noshadow={noshadow} i8only={i8only} seed={seed}"
    valid = Ie >= 0
    scale = max(1.0, float(np.abs(De[valid]).max())) if valid.any() else 1.0
    gaps = np.abs(np.diff(De.astype(np.float64), axis=1))
    safe = valid.copy()
    safe[:, 1:] &= gaps > 1e-5 * scale
    safe[:, :-1] &= gaps > 1e-5 * scale
    safe[:, -1] = False          # the rank behind the last slot is unknown
    for mode in res:
        if mode == "exact_fp32":
            continue
        Dc, Ic = res[mode]
        tag = f"[{mode}] " + tag0
        assert ((Ic >= 0) == valid).all(), "padding differs: " + tag
        assert np.allclose(Dc[valid], De[valid], atol=2e-5 * scale, rtol=0), f"scores differ by {np.abs(Dc[valid] - De[valid]).max()}: " + tag
        bad = safe & (Ic != Ie)
        assert not bad.any(), f"{int(bad.sum())} id mismatches: " + tag
    tag = tag0
    ix.close()
    it += 1
    if it % 20 == 0:
        print(f"{it} cases, {time.time() - t0:.0f}s (last: {tag})", flush=True)
print(f"fuzz ok: {it} cases in {time.time() - t0:.0f}s")
