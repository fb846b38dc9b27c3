"""GPU: BASELINE.json full-size configurations.

``test_10m_bench_queries_match_the_cpu_oracle`` checks the benchmark's own inputs (the 10 M-row synthetic index of
seed 4 and the first queries of seed 5, exactly what ``bench.py`` searches) against the CPU oracle, which
regenerates the rows from ``css_synth.h`` in 1 M-row chunks on the host cores and merges the per-chunk lists.
The other tests use size-independent properties:

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
        De, Ie = big_index.search(q[:40], K)     # fp32-input MFMA scan
        D1, I1 = big_index.search(q[:2], K)      # fp32 sweep
    finally:
        big_index.set_search_mode("auto")
    assert np.abs(De - Db[:40]).max() < 1e-5 and (Ie == Ib[:40]).mean() > 0.97
    assert np.abs(D1 - Db[:2]).max() < 1e-5 and (I1 == Ib[:2]).mean() > 0.9


def test_1m_clustered_rows_every_mode_matches_the_cpu_oracle():
    """Dense candidate bands at scale: 1 M rows in 2 000 tight clusters (cosine spread ~1e-3 inside a cluster), queries
    near cluster centres and random ones, 1 and 300 queries.  The product path (bf16 candidate scan + fp32
    rescoring, flagged queries fixed up exactly on the device) and the exact fp32 kernels are both held against
    the CPU oracle: same ids wherever the fp64 gap to the neighbouring rank exceeds 1e-6 (inside a cluster fp32
    summation orders legitimately swap closer neighbours), scores within 1e-3."""
    import os

    from claude_semantic_search_amd import synth
    from claude_semantic_search_amd.flat_index import IndexFlatIP
    from knn_checks import assert_topk_matches
    from oracle import knn_oracle as ko

    n, nc = 1_000_000, 2000
    cent = synth.rows(nc, D, 71)
    ix = IndexFlatIP(D)
    ix.reserve(n)
    ko.set_threads(min(os.cpu_count() or 1, 32))
    ref = ko.FlatIndexOracle(D)
    ref._xb = np.empty((n, D), dtype=np.float32)
    for c0 in range(0, n, 100_000):
        ids = (np.arange(c0, c0 + 100_000) % nc)
        x = cent[ids] + 0.05 * synth.rows(100_000, D, 72 + c0 // 100_000)
        ix.add(x, normalize=True)
        ref._xb[c0:c0 + 100_000] = ko.normalize_rows(x)
    q = np.concatenate([cent[:200] + 0.02 * synth.rows(200, D, 90), synth.rows(100, D, 91)])
    qn = ko.normalize_rows(q)
    Dr, Ir = ko.search_blas(ref._xb, qn, K + 1)     # one rank more: the gap behind the last returned slot
    D64 = ref.rescore64(qn, Ir)
    for nq in (1, 300):
        for mode in ("coarse", "exact_fp32"):
            ix.set_search_mode(mode)
            Da, Ia = ix.search(q[:nq], K, normalize=True)
            # (the fp32-input MFMA scan of the exact mode sums 768 terms in one sequential fmaf chain: ~1e-6 of
            # rounding on unit vectors, so inside these dense clusters ranks closer than 4e-6 may swap)
            assert_topk_matches(Da, Ia, Dr[:nq, :K], Ir[:nq, :K], D64[:nq, :K], f"1M clustered [{mode}] nq={nq}",
                                D64_next=D64[:nq, K], tie_eps=4e-6 if mode == "exact_fp32" and nq > 16 else 1e-6)
    ix.close()


# ---- the benchmark's own inputs at BASELINE configs[3], against the CPU oracle -----------------------------------
NQ_BENCH, NQ_ORACLE = 1000, 256


@pytest.fixture(scope="module")
def bench_queries_and_oracle():
    """bench.py's query batch (seed 5) and the oracle's top-10 of its first NQ_ORACLE queries over the 10 M rows
    of seed 4: rows regenerated on the host from css_synth.h, normalised as add_chunks does, 1 M at a time."""
    import os

    from claude_semantic_search_amd import synth
    from oracle import knn_oracle as ko

    ko.set_threads(min(os.cpu_count() or 1, 32))
    q = synth.rows(NQ_BENCH, D, 5)
    qn = ko.normalize_rows(q)
    Dr, Ir, D64 = ko.search_synth_chunked(N, D, 4, qn[:NQ_ORACLE], K, metric=0, normalize=True, chunk=1_000_000)
    return q, Dr, Ir, D64


def test_10m_bench_queries_match_the_cpu_oracle(big_index, bench_queries_and_oracle):
    from knn_checks import assert_topk_matches

    q, Dr, Ir, D64 = bench_queries_and_oracle
    assert np.isfinite(D64).all() and (np.diff(Dr, axis=1) <= 0).all()
    try:
        for mode in ("auto", "coarse", "exact_fp32"):
            big_index.set_search_mode(mode)
            # the batch bench.py times (auto / coarse: the bf16 MFMA cascade; exact: the fp32-input MFMA scan,
            # given a smaller batch -- it costs 0.2 s per 128 queries at this size)
            nqb = NQ_BENCH if mode != "exact_fp32" else 40
            Db, Ib = big_index.search(q[:nqb], K, normalize=True)
            n = min(nqb, NQ_ORACLE)
            assert_topk_matches(Db[:n], Ib[:n], Dr[:n], Ir[:n], D64[:n], f"10M batched x{nqb} [{mode}]")
            # the reference's own call shape, one query per call (bf16 sweep cascade / fp32 sweep), and a 4-query call
            for r in (0, 7, NQ_ORACLE - 1):
                D1, I1 = big_index.search(q[r:r + 1], K, normalize=True)
                assert_topk_matches(D1, I1, Dr[r:r + 1], Ir[r:r + 1], D64[r:r + 1], f"10M single query {r} [{mode}]")
            D4, I4 = big_index.search(q[8:12], K, normalize=True)
            assert_topk_matches(D4, I4, Dr[8:12], Ir[8:12], D64[8:12], f"10M 4 queries [{mode}]")
    finally:
        big_index.set_search_mode("auto")


def test_config0_ten_thousand_chunks_encode_index_search_chain():
    """BASELINE configs[0] at its stated size: 10 000 synthetic chunks with the chunker's length
    mix (chars ~ U[100, 2000] -> L = clip(round(chars / 4) + 2, 2, 384), ~25 % at the 384 cap), encoded by the HIP
    encoder (12 layers, bf16 product mode, batches of 256, length-sorted), added to the HIP flat-IP index, 100 queries
    = rows {0, 100, ...}.  Parity in two independent halves (SURVEY "hard parts" (ii)): (a) a sample of the
    embeddings against the fp32 CPU oracle encoder, cos >= 1 - 1e-3; (b) the HIP top-10 against the CPU oracle kNN run
    on the HIP embedding matrix itself -- same ids outside fp64 near-ties, scores within 1e-3."""
    from knn_checks import assert_topk_matches
    from oracle import knn_oracle as ko
    from oracle import mpnet_oracle as mo
    from claude_semantic_search_amd import synth
    from claude_semantic_search_amd.flat_index import IndexFlatIP
    from claude_semantic_search_amd.mpnet_encoder import MpnetEncoder

    n, bs = 10_000, 256
    cfg = mo.MpnetCfg()
    chars = 100 + synth.uint(1, np.arange(n, dtype=np.uint64), 0, 1901)
    lens = np.clip(np.round(chars / 4).astype(np.int64) + 2, 2, 384)
    assert 0.2 < (lens == 384).mean() < 0.3 and 230 < lens.mean() < 270
    batch = mo.synth_batch(cfg, lens.tolist(), seed=1)
    enc = MpnetEncoder(synthetic_seed=1, compute="bf16")
    emb = np.empty((n, 768), dtype=np.float32)
    order = np.argsort(-lens, kind="stable")
    for s in range(0, n, bs):
        idx = order[s:s + bs]
        emb[idx] = enc.encode_ids([batch[i] for i in idx])
    enc.close()
    assert np.isfinite(emb).all() and np.abs(np.linalg.norm(emb, axis=1) - 1).max() < 1e-3
    # (a) encoder parity on a sample spread over lengths and batches (incl. the shortest, the longest, first, last)
    sample = sorted(set([0, n - 1, int(order[0]), int(order[-1])] + list(range(37, n, n // 28))))[:32]
    w = mo.synth_weights(cfg, 1)
    ref = mo.encode(w, cfg, [batch[i] for i in sample])
    cos = (emb[sample] * ref).sum(1)
    assert cos.min() > 1 - 1e-3, cos
    # (b) kNN parity on the HIP embedding matrix
    ix = IndexFlatIP(768)
    ix.add(emb, normalize=True)
    q = emb[::100]
    o = ko.FlatIndexOracle(768, 0)
    xr, qr = ko.normalize_rows(emb), ko.normalize_rows(q)
    o.add(xr)
    for k in (10, 100):
        Dr1, Ir1 = o.search(qr, k + 1)                     # rank k + 1 too: the last slot may tie with it
        D641 = o.rescore64(qr, Ir1)
        Dr, Ir, D64 = Dr1[:, :k], Ir1[:, :k], D641[:, :k]
        for mode in ("auto", "coarse", "exact_fp32"):
            ix.set_search_mode(mode)
            Dh, Ih = ix.search(q, k, normalize=True)
            assert_topk_matches(Dh, Ih, Dr, Ir, D64, f"config0 k={k} [{mode}]", D64_next=D641[:, k])
            assert (Ih[:, 0] == np.arange(0, n, 100)).all() and np.abs(Dh[:, 0] - 1).max() < 1e-5
    ix.close()


def test_10m_clustered_rows_second_pass_matches_the_cpu_oracle():
    """dense candidate bands at the benchmark's size.  10 M rows in 2 000 tight clusters = 5 000
    near-identical rows per cluster, more than the cascade's 4 096-slot candidate buffers hold: every query near a
    cluster centre is flagged and settled by the SECOND coarse pass (threshold from the exactly rescored buffered
    candidates, 32 768-slot buffers) -- none may fall through to the exact fp32 sweep, which round 2 used for all
    of them (8 queries per 30 GB pass).  Rows are generated on the device (synth.rows_torch, bit-identical to the
    host generator: spot-checked), exported 1 M at a time, and the CPU oracle answers per chunk + merge."""
    import torch

    from claude_semantic_search_amd import synth
    from claude_semantic_search_amd.flat_index import IndexFlatIP
    from knn_checks import assert_topk_matches
    from oracle import knn_oracle as ko

    n, nc, chunk = 10_000_000, 2000, 250_000
    dev = torch.device("cuda:0")
    cent = synth.rows(nc, D, 71)
    cent_d = torch.from_numpy(cent).to(dev)
    ix = IndexFlatIP(D)
    ix.reserve(n)
    for c0 in range(0, n, chunk):
        ids = torch.arange(c0, c0 + chunk, device=dev) % nc
        rows = cent_d[ids] + 0.05 * synth.rows_torch(chunk, D, 72 + c0 // chunk, device=dev)
        ix.add_dev(rows.data_ptr(), chunk, normalize=True)
        torch.cuda.synchronize()
    # the device generator is the host generator: three rows of the last chunk, regenerated with numpy (bit for bit
    # before the ingest kernel normalises them; its sum of squares has another summation order than the oracle's)
    c0 = n - chunk
    host = cent[(np.arange(c0 + 7, c0 + 10) % nc)] + 0.05 * synth.rows(3, D, 72 + c0 // chunk, first_row=7)
    assert np.array_equal(rows[7:10].cpu().numpy(), host)
    assert np.allclose(ix.reconstruct_n(c0 + 7, 3), ko.normalize_rows(host), rtol=0, atol=2e-8)
    del cent_d, rows
    q = np.concatenate([cent[5:21] + 0.02 * synth.rows(16, D, 90), synth.rows(8, D, 91)])
    qn = ko.normalize_rows(q)
    nq = q.shape[0]
    parts_d, parts_i = [], []
    for r0 in range(0, n, 1_000_000):
        xb = ix.reconstruct_n(r0, 1_000_000)
        d_, i_ = ko.search_blas(xb, qn, K + 1)
        parts_d.append(d_)
        parts_i.append(np.where(i_ >= 0, i_ + r0, -1))
    Dr, Ir = ko.merge_topk(np.stack(parts_d), np.stack(parts_i), 0)
    Dr, Ir = Dr[:, :K + 1], Ir[:, :K + 1]
    D64 = np.stack([np.stack([ix.reconstruct(int(i)) for i in Ir[r]]).astype(np.float64) @ qn[r].astype(np.float64)
                    for r in range(nq)])
    for mode in ("coarse", "auto"):
        ix.set_search_mode(mode)
        Da, Ia = ix.search(q, K, normalize=True)
        assert_topk_matches(Da, Ia, Dr[:, :K], Ir[:, :K], D64[:, :K], f"10M clustered [{mode}]", D64_next=D64[:, K])
        assert ix.last_flagged() >= 16 and ix.last_swept() == 0, (ix.last_flagged(), ix.last_swept())
    # the single-query path (bf16 sweep cascade; a flagged query there is re-run by the exact sweep)
    Da, Ia = ix.search(q[:1], K, normalize=True)
    assert_topk_matches(Da, Ia, Dr[:1, :K], Ir[:1, :K], D64[:1, :K], "10M clustered nq=1", D64_next=D64[:1, K])
    ix.close()


def test_10m_index_without_shadow_rows_answers_like_the_shadowed_one(big_index):
    """An index without bf16 shadow rows (the state of a shard beyond ~38 M rows per GPU) rounds its rows to bf16 one
    row range at a time and runs the cascade of shadowed indexes per range: at 10 M rows (7.7e9 elements: beyond one
    dispatch's 2^32 work-items, which an earlier version of the conversion kernel silently truncated) the batched
    answers must be those of the shadowed index over the same rows -- same bf16 values, same error band, same fp32
    rescoring -- with one range and with three merged ranges."""
    from claude_semantic_search_amd import synth
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    q = synth.rows(300, D, 5)
    Dref, Iref = big_index.search(q, K, normalize=True)
    ix = IndexFlatIP(D)
    ix.set_shadow(False)
    ix.reserve(N)
    ix.add_synthetic(N, seed=4, first_row=0, normalize=True)
    for rows in (0, 4_000_000):
        ix.set_range_rows(rows)
        Dn, In = ix.search(q, K, normalize=True)
        assert np.array_equal(In, Iref) and np.array_equal(Dn, Dref), rows
    ix.close()


def test_rows_beyond_2_pow_26_of_one_add_are_ingested():
    """The ingest kernel runs one wave per row and a dispatch carries at most 2^32 work-items = 2^26 rows: one add of more
    rows than that used to leave the rows beyond 2^26 unwritten, silently (an 80 M-row index on one GPU answered from
    garbage rows through the exact sweep, 12 s per batch).  Narrow rows keep the test at 18 GB: 70 M rows x 4 floats,
    rows around and far beyond 2^26 read back and compared with the host generator."""
    from claude_semantic_search_amd import synth
    from claude_semantic_search_amd.flat_index import IndexFlatIP
    from oracle import knn_oracle as ko

    n, d = 70_000_000, 4
    ix = IndexFlatIP(d)
    ix.set_shadow(False)
    ix.reserve(n)
    ix.add_synthetic(n, seed=9, first_row=0, normalize=True)
    assert ix.ntotal == n
    for r0 in (0, (1 << 24) - 2, (1 << 26) - 2, (1 << 26) + 1_000_003, n - 5):
        got = ix.reconstruct_n(r0, 4)
        want = ko.normalize_rows(synth.rows(4, d, 9, first_row=r0))
        assert np.allclose(got, want, rtol=0, atol=3e-7), (r0, got, want)
    # and a search sees them: the query equal to a row far beyond 2^26 finds that row first
    q = ix.reconstruct_n((1 << 26) + 12_345, 1)
    D_, I_ = ix.search(q, 1, normalize=False)
    assert abs(float(D_[0, 0]) - 1.0) < 1e-5
    ix.close()


@pytest.mark.gpu
def test_first_search_with_more_queries_than_before_uses_fresh_query_error_norms():
    """Until round 4 launch_scan_coarse took the pointer to the int8 queries' error norms BEFORE the buffer behind it grew:
    the selects of the first int8 search with more queries than any before on that index read the freed, shorter buffer --
    zeros on a fresh device (a band without the query term), stale bytes of freed indexes otherwise: in the third index
    of one process (10 M, 3 M, 1 M rows; 256 then 1000 queries each) 744 of 1000 queries were flagged, the search took
    33 ms and the index backed off to the bf16 rows.  Results stayed exact (the flagged queries end in the fix-up), so the
    symptom is the flagged count: zero for every search of that sequence now."""
    import torch
    from claude_semantic_search_amd import synth
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    st = torch.cuda.current_stream().cuda_stream
    for rows in (10_000_000, 3_000_000, 1_000_000):
        ix = IndexFlatIP(768)
        ix.reserve(rows)
        ix.add_synthetic(rows, seed=7)
        for nq in (256, 1000):
            q = torch.from_numpy(synth.rows(nq, 768, 99)).cuda()
            D = torch.empty((nq, 10), dtype=torch.float32, device="cuda")
            I = torch.empty((nq, 10), dtype=torch.int64, device="cuda")
            for it in range(2):
                ix.search_dev(q.data_ptr(), nq, 10, D.data_ptr(), I.data_ptr(), st, normalize=True)
                torch.cuda.synchronize()
                assert ix.last_flagged() == 0, f"{rows} rows, {nq} queries, search {it}: {ix.last_flagged()} queries flagged"
            assert (I >= 0).all() and (I < rows).all() and (D[:, :-1] >= D[:, 1:]).all()
        ix.close()
