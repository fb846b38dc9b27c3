"""Shared comparison helper: HIP top-k vs the CPU oracle on identical inputs.

Bar (BASELINE.json north_star): same top-k ids, scores within 1e-3 fp32.  Ids
are compared exactly wherever the oracle's fp64 score gap to the neighbouring
ranks exceeds 1e-6 (near ties may legitimately swap: SURVEY.md 8d)."""
import numpy as np

SCORE_TOL = 1e-3


def assert_topk_matches(D, I, D_ref, I_ref, D64_ref, what="", D64_next=None, tie_eps=1e-6):
    """``D64_next``: fp64 score of the oracle's rank k + 1 (when known): the last slot is then compared only where
    its gap to that rank also exceeds ``tie_eps``.  ``tie_eps``: fp64 gap below which neighbouring ranks may swap
    (1e-6 for blocked fp32 sums; a sequential 768-term fmaf chain, the fp32-input MFMA scan, rounds ~1e-6 itself)."""
    assert D.shape == D_ref.shape and I.shape == I_ref.shape, what
    valid = I_ref >= 0
    assert ((I >= 0) == valid).all(), f"{what}: padding differs"
    assert np.allclose(D[valid], D_ref[valid], atol=SCORE_TOL, rtol=0), \
        f"{what}: max score diff {np.abs(D[valid] - D_ref[valid]).max()}"
    gaps = np.abs(np.diff(D64_ref, axis=1))
    safe = valid.copy()
    safe[:, 1:] &= gaps > tie_eps
    safe[:, :-1] &= gaps > tie_eps
    if D64_next is not None:
        safe[:, -1] &= np.abs(D64_ref[:, -1] - np.asarray(D64_next)) > tie_eps
    bad = safe & (I != I_ref)
    assert not bad.any(), f"{what}: {int(bad.sum())} id mismatches outside near ties, e.g. {np.argwhere(bad)[:5].tolist()}"
    # near-tie slots: the id sets must still agree as multisets per row when the row has no pad
    for r in range(I.shape[0]):
        if valid[r].all() and not (I[r] == I_ref[r]).all():
            # allow swaps only; the last slot may differ when tied with rank k+1
            a, b = set(I[r][:-1].tolist()), set(I_ref[r].tolist())
            assert len(a - b) <= 1, f"{what}: row {r} differs beyond a near-tie swap"
