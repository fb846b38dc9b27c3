"""GPU: BASELINE.json full-size configurations checked through size-independent
properties (the CPU oracle cannot sweep 10M rows in test time):

  * a query that IS row i must come back first with score ~1 (unit rows);
  * scores descending, ids unique and in range;
  * fp64 re-score of the returned rows (exported from the device) matches D to 1e-3;
  * no row of a random 100k-row sample beats the k-th returned score;
  * the single-query path (bf16 sweep + rescoring) and the batched path (bf16 MFMA scan + rescoring) agree.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, D, K = 10_000_000, 768, 10


@pytest.fixture(scope="module")
def big_index():
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    ix = IndexFlatIP(D)
    ix.reserve(N)
    ix.add_synthetic(N, seed=4, first_row=0, normalize=True)
    assert ix.ntotal == N
    yield ix
    ix.close()


def _check(ix, q, Dq, Iq, probe_ids):
    nq = q.shape[0]
    assert ((Iq >= 0) & (Iq < N)).all()
    assert (np.diff(Dq, axis=1) <= 0).all()
    for r in range(nq):
        assert len(set(Iq[r].tolist())) == K
    if probe_ids is not None:
        assert (Iq[:, 0] == probe_ids).all() and np.allclose(Dq[:, 0], 1.0, atol=1e-5)
    # fp64 re-score of returned rows
    for r in range(0, nq, max(1, nq // 16)):
        rows = np.stack([ix.reconstruct(int(i)) for i in Iq[r]])
        s64 = rows.astype(np.float64) @ q[r].astype(np.float64)
        assert np.abs(s64 - Dq[r]).max() < 1e-3
    # sampled rows never beat the k-th score
    sample = ix.reconstruct_n(3_000_000, 100_000)
    best = (q.astype(np.float32) @ sample.T).max(axis=1)
    in_sample = ((Iq >= 3_000_000) & (Iq < 3_100_000)).any(axis=1)
    assert (best[~in_sample] <= Dq[~in_sample, K - 1] + 1e-5).all()


def test_10m_single_query_and_small_batch(big_index):
    ids = np.array([0, 1, 4_999_999, 9_999_999, 1234567], dtype=np.int64)
    q = np.stack([big_index.reconstruct(int(i)) for i in ids])
    for nq in (1, 5):
        Dq, Iq = big_index.search(q[:nq], K)
        _check(big_index, q[:nq], Dq, Iq, ids[:nq])


def test_10m_query_batch_mfma_path_and_agreement(big_index):
    rng = np.random.default_rng(0)
    ids = rng.integers(0, N, size=200)
    q = np.stack([big_index.reconstruct(int(i)) for i in ids])
    Db, Ib = big_index.search(q, K)              # > 4 queries: MFMA coarse scan + exact rescoring
    _check(big_index, q, Db, Ib, ids)
    Ds, Is = big_index.search(q[:4], K)          # <= 4 queries: HBM-bound bf16 sweep + exact rescoring
    assert np.abs(Ds - Db[:4]).max() < 1e-5
    same = Is == Ib[:4]
    assert same.mean() > 0.97                    # near-ties may swap between the two arithmetic orders
    # the parity mode (every score formed in fp32 by the scan kernels) returns the same answer at full size
    big_index.set_search_mode("exact_fp32")
    try:
        De, Ie = big_index.search(q[:40], K)     # split-operand MFMA scan
        D1, I1 = big_index.search(q[:2], K)      # fp32 sweep
    finally:
        big_index.set_search_mode("auto")
    assert np.abs(De - Db[:40]).max() < 1e-5 and (Ie == Ib[:40]).mean() > 0.97
    assert np.abs(D1 - Db[:2]).max() < 1e-5 and (I1 == Ib[:2]).mean() > 0.9


def test_1m_clustered_rows_auto_mode_equals_exact_mode():
    """Dense candidate bands at scale: 1 M rows in 2 000 tight clusters (cosine spread ~1e-3 inside a cluster), queries
    near cluster centres and random ones, 1 and 300 queries.  The product path (bf16 candidate scan + fp32
    rescoring, flagged queries re-run exactly) must return what the exact fp32 kernels return."""
    from claude_semantic_search_amd import synth
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    n, nc = 1_000_000, 2000
    cent = synth.rows(nc, D, 71)
    ix = IndexFlatIP(D)
    ix.reserve(n)
    for c0 in range(0, n, 100_000):
        ids = (np.arange(c0, c0 + 100_000) % nc)
        ix.add(cent[ids] + 0.05 * synth.rows(100_000, D, 72 + c0 // 100_000), normalize=True)
    q = np.concatenate([cent[:200] + 0.02 * synth.rows(200, D, 90), synth.rows(100, D, 91)])
    for nq in (1, 300):
        ix.set_search_mode("coarse")
        Da, Ia = ix.search(q[:nq], K, normalize=True)
        ix.set_search_mode("exact_fp32")
        De, Ie = ix.search(q[:nq], K, normalize=True)
        assert np.abs(Da - De).max() < 1e-5
        assert (Ia == Ie).mean() > 0.995          # fp32 near-ties inside a cluster may swap between summation orders
        assert (np.sort(Ia, axis=1) == np.sort(Ie, axis=1)).mean() > 0.995
    ix.close()
