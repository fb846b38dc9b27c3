"""GPU parity tests: hand-written HIP flat index (through the C ABI) vs the CPU oracle."""
import json
from pathlib import Path

import numpy as np
import pytest

from claude_semantic_search_amd import synth
from tests_support import oracle_index, hip_index  # noqa: F401  (fixtures)
from knn_checks import assert_topk_matches

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"


def _run_case(n, d, nq, k, metric, normalize, seed=1, tie_eps=1e-6, with_next=False):
    from oracle import knn_oracle as ko
    from claude_semantic_search_amd.flat_index import IndexFlat

    x = synth.rows(n, d, seed)
    q = synth.rows(nq, d, seed + 1000)
    hip = IndexFlat(d, metric)
    hip.add(x, normalize=normalize)
    assert hip.ntotal == n
    ref = ko.FlatIndexOracle(d, metric)
    xr, qr = (ko.normalize_rows(x), ko.normalize_rows(q)) if normalize else (x, q)
    ref.add(xr)
    Dr, Ir = ref.search(qr, k)
    D64 = ref.rescore64(qr, np.where(Ir < 0, 0, Ir))
    D64_next = None
    if with_next and k < n:   # tens of thousands of result slots: rank k may tie with rank k + 1, which the lists do not show
        _, I1 = ref.search(qr, k + 1)
        D64_next = ref.rescore64(qr, I1[:, k:k + 1])[:, 0]
    # both search paths against the oracle: "exact_fp32" (every score formed in fp32 by the scan kernels: the
    # parity mode) and "auto" (the product default: bf16 candidate scan + exact fp32 rescoring where available)
    for mode in ("exact_fp32", "coarse", "auto"):
        hip.set_search_mode(mode)
        D, I = hip.search(q, k, normalize=normalize)
        assert_topk_matches(D, I, Dr, Ir, D64, f"[{mode}] n={n} d={d} nq={nq} k={k} metric={metric}", D64_next=D64_next, tie_eps=tie_eps)
    hip.close()


@pytest.mark.parametrize("case", json.loads((GOLD / "knn_reference_cases.json").read_text())["cases"],
                         ids=lambda c: c["name"])
def test_reference_known_answers_on_hip(case):
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    ix = IndexFlatIP(4)
    ix.add(np.array(case["rows"], np.float32), normalize=True)
    D, I = ix.search(np.array(case["query"], np.float32), min(100, ix.ntotal), normalize=True)
    assert I[0].tolist() == case["expected_ids"]
    assert np.allclose(D[0], case["expected_sims"], atol=1e-6)


def test_env_case_on_hip():
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    env = json.loads((GOLD / "knn_reference_cases.json").read_text())["env_case"]
    v = np.random.default_rng(env["seed"]).random((env["n"], env["d"])).astype(np.float32)
    ix = IndexFlatIP(env["d"])
    assert ix.d == 128 and ix.ntotal == 0
    ix.add(v)
    assert ix.ntotal == 10
    D, I = ix.search(v[0:1], 5)
    assert D.shape == (1, 5) and I.shape == (1, 5) and I[0][0] == 0


@pytest.mark.parametrize("n", [1, 3, 4, 5, 63, 64, 65, 257, 1000])
def test_ragged_sizes_ip(n):
    _run_case(n, 768, 3, min(10, 128), 0, True, seed=n)


@pytest.mark.parametrize("d", [1, 4, 63, 64, 65, 128, 384, 768, 1024])
def test_dims(d):
    _run_case(777, d, 2, 10, 0, True, seed=d)


@pytest.mark.parametrize("nq", [1, 2, 3, 5, 8, 9, 16, 17, 33])
@pytest.mark.parametrize("metric", [0, 1])
def test_query_counts(nq, metric):
    _run_case(5000, 768, nq, 10, metric, metric == 0, seed=nq)


@pytest.mark.parametrize("k", [1, 2, 10, 64, 100, 128])
def test_k_values(k):
    _run_case(20000, 768, 2, k, 0, True, seed=k)


@pytest.mark.parametrize("k", [129, 500, 2048])
def test_k_beyond_one_kernel_pass(k):
    """k > 128 (the reference passes k' = min(max_results, ntotal) for ANY max_results, src/storage.py:432): passes of
    128 over the rows not returned yet + one sort; every search mode, single queries and a small batch, both metrics."""
    _run_case(20000, 768, 1, k, 0, True, seed=k)
    # (squared L2 of raw N(0,1) rows: distances ~1500, one fp32 ulp there is 1.2e-4 -- ranks closer than that may swap)
    _run_case(6000, 768, 3, k, 1, False, seed=k + 1, tie_eps=1e-3)
    _run_case(3000, 100, 2, k, 0, True, seed=k + 2)
    if k == 500:
        _run_case(300, 768, 2, k, 0, True, seed=7)      # fewer rows than k: padded like faiss


def test_k_beyond_one_pass_with_mask_duplicates_and_device_api():
    import torch

    from oracle import knn_oracle as ko
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    n, d, k = 9000, 768, 300
    base = synth.rows(n // 3, d, 31)
    x = np.concatenate([base, base, base])                          # every row three times: ties inside and across passes
    q = synth.rows(2, d, 32)
    ix = IndexFlatIP(d)
    ix.add(x, normalize=True)
    allow = (np.arange(n) % 5) != 0
    ref = ko.FlatIndexOracle(d, 0)
    xr = ko.normalize_rows(x)
    ref.add(xr[allow])
    back = np.nonzero(allow)[0]
    Dr, Ir = ref.search(ko.normalize_rows(q), k)
    Ir = back[Ir]
    D, I = ix.search(q, k, normalize=True, allow=allow)
    # ties: a triplet that straddles a pass boundary (or rank k) is split lowest ids first, so the id SETS are the oracle's;
    # inside the list two passes may have formed equal rows' scores in different summation orders (1 ulp): order by score
    assert np.abs(D - Dr).max() < 1e-5 and (np.diff(D, axis=1) <= 0).all()
    assert all(set(I[r].tolist()) == set(Ir[r].tolist()) for r in range(2))
    # device-pointer twin, output rows strided by k
    qd = torch.from_numpy(q).cuda()
    Dd = torch.empty((2, k), dtype=torch.float32, device="cuda")
    Id = torch.empty((2, k), dtype=torch.int64, device="cuda")
    bits = torch.from_numpy(__import__("claude_semantic_search_amd.flat_index", fromlist=["x"]).pack_allow_bits(allow, n).view(np.int32)).cuda()
    ix.search_dev(qd.data_ptr(), 2, k, Dd.data_ptr(), Id.data_ptr(), torch.cuda.current_stream().cuda_stream, normalize=True,
                  allow_bits_ptr=bits.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(Id.cpu().numpy(), I) and np.array_equal(Dd.cpu().numpy(), D)
    with pytest.raises(ValueError):
        ix.search(q, 2049)
    ix.close()


def test_merge_of_shard_lists_with_k_beyond_128():
    import ctypes

    import torch

    from claude_semantic_search_amd import _native as nat
    from claude_semantic_search_amd.flat_index import IndexFlat
    from claude_semantic_search_amd.sharded import packed_layout

    n, d, G, nq, k = 4000, 128, 3, 3, 700
    x = synth.rows(n, d, 41)
    x[100:140] = x[60:100]                                          # ties across shards
    q = synth.rows(nq, d, 42)
    for metric in (0, 1):
        whole = IndexFlat(d, metric)
        whole.add(x, normalize=metric == 0)
        D, I = whole.search(q, k, normalize=metric == 0)
        ib, db, record = packed_layout(nq, k)
        recv = torch.zeros((G, record), dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        qd = torch.from_numpy(q).cuda()
        bounds = [0, 90, 1500, n]                                    # shard 0 holds fewer rows than k: pads inside its list
        keep = []
        for g in range(G):
            s_ = IndexFlat(d, metric)
            s_.add(x[bounds[g]:bounds[g + 1]], normalize=metric == 0)
            s_.set_id_base(bounds[g])
            s_.search_dev(qd.data_ptr(), nq, k, recv[g, ib:db].view(torch.float32).data_ptr(), recv[g, :ib].view(torch.int64).data_ptr(),
                          st, normalize=metric == 0)
            keep.append(s_)
        Do = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        Io = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        nat.check(nat.lib().css_merge_topk_packed_dev(ctypes.c_void_p(recv.data_ptr()), G, record, nq, k, metric,
                                                      ctypes.c_void_p(Do.data_ptr()), ctypes.c_void_p(Io.data_ptr()), 0, ctypes.c_void_p(st)))
        torch.cuda.synchronize()
        assert np.array_equal(Io.cpu().numpy(), I) and np.abs(Do.cpu().numpy() - D).max() < 1e-5


def test_k_larger_than_ntotal_pads():
    _run_case(7, 768, 2, 100, 0, True)
    _run_case(7, 768, 2, 100, 1, False)


def test_l2_raw_rows():
    _run_case(10000, 768, 4, 10, 1, False)


def test_config2_100k_768_top10_single_query_path():
    # BASELINE config 2 shape, the reference's real call shape nq=1 (src/storage.py:429) and a small batch
    _run_case(100_000, 768, 1, 100, 0, True, seed=2)
    _run_case(100_000, 768, 16, 10, 0, True, seed=3)


def test_duplicates_and_ties_lower_id_first():
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    base = synth.rows(50, 768, 5)
    x = np.concatenate([base, base, base], axis=0)  # every row appears 3 times
    ix = IndexFlatIP(768)
    ix.add(x, normalize=True)
    D, I = ix.search(base[7:8], 9, normalize=True)
    assert I[0][:3].tolist() == [7, 57, 107]
    assert D[0][0] == D[0][1] == D[0][2]
    # constant vectors incl. an all-zero row (tests/test_incremental_indexing.py:88,129 of the reference)
    c = np.ones((20, 768), np.float32)
    c[5] = 0.0
    ix2 = IndexFlatIP(768)
    ix2.add(c, normalize=True)
    D, I = ix2.search(np.ones((1, 768), np.float32), 20, normalize=True)
    assert I[0][:19].tolist() == [i for i in range(20) if i != 5] and I[0][19] == 5
    assert abs(D[0][0] - 1.0) < 1e-5 and D[0][19] == 0.0


def test_committed_goldens_on_hip():
    from claude_semantic_search_amd.flat_index import IndexFlat

    for name, metric, norm in (("knn_synth_ip.npz", 0, True), ("knn_synth_l2.npz", 1, False)):
        g = np.load(GOLD / name)
        ix = IndexFlat(int(g["d"]), metric)
        ix.add_synthetic(int(g["n"]), int(g["seed_x"]), 0, normalize=norm)
        q = synth.rows(int(g["nq"]), int(g["d"]), int(g["seed_q"]))
        for mode in ("exact_fp32", "coarse", "auto"):
            ix.set_search_mode(mode)
            D, I = ix.search(q, int(g["k"]), normalize=norm)
            assert_topk_matches(D, I, g["D"], g["I"].astype(np.int64), g["D64"], f"{name} [{mode}]")
            D1, I1 = ix.search(q[:3], int(g["k"]), normalize=norm)   # the few-query path of the same mode
            assert_topk_matches(D1, I1, g["D"][:3], g["I"][:3].astype(np.int64), g["D64"][:3], f"{name} [{mode}, 3 queries]")


def test_device_synthetic_rows_equal_host_generator():
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    ix = IndexFlatIP(768)
    ix.add_synthetic(300, 99, first_row=12345, normalize=False)
    assert np.array_equal(ix.reconstruct_n(0, 300), synth.rows(300, 768, 99, first_row=12345))
    ix.add_synthetic(10, 98, first_row=0, normalize=False)
    assert np.array_equal(ix.reconstruct_n(300, 10), synth.rows(10, 768, 98))


def test_incremental_adds_and_growth():
    from oracle import knn_oracle as ko
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    ix = IndexFlatIP(768)
    ref = ko.FlatIndexOracle(768)
    q = ko.normalize_rows(synth.rows(3, 768, 500))
    total = 0
    for step, n in enumerate([1, 10, 100, 1500, 3000]):
        x = synth.rows(n, 768, 600 + step)
        ix.add(x, normalize=True)
        ref.add(ko.normalize_rows(x))
        total += n
        assert ix.ntotal == total
        D, I = ix.search(q, 10)
        Dr, Ir = ref.search(q, min(10, total) if total >= 10 else 10)
        D64 = ref.rescore64(q, np.where(Ir < 0, 0, Ir))
        assert_topk_matches(D, I, Dr, Ir, D64, f"after {total} rows")
    ix.reset()
    assert ix.ntotal == 0


def test_int8_rows_follow_growth_reset_and_shadow_policy():
    """The int8 rows of the 1..4-query sweep live and die with the bf16 shadow rows: carried over when the row table
    is reallocated, restarted after a reset, absent without shadow rows -- and single-query results stay the oracle's."""
    from oracle import knn_oracle as ko
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    ix = IndexFlatIP(768)
    ix.set_search_mode("coarse")
    ref = ko.FlatIndexOracle(768)
    q = ko.normalize_rows(synth.rows(3, 768, 510))
    total = 0
    for step, n in enumerate([700, 900, 5000, 20000]):     # each add outgrows the capacity: three reallocations
        x = synth.rows(n, 768, 610 + step)
        ix.add(x, normalize=True)
        ref.add(ko.normalize_rows(x))
        total += n
        assert ix.shadow_info() == {"bf16": True, "int8": True}
        for nq in (1, 3):
            D, I = ix.search(q[:nq], 10)
            Dr, Ir = ref.search(q[:nq], 10)
            assert_topk_matches(D, I, Dr, Ir, ref.rescore64(q[:nq], Ir), f"{total} rows, {nq} queries")
    ix.reset()
    x = synth.rows(3000, 768, 620)
    ix.add(x, normalize=True)
    ref2 = ko.FlatIndexOracle(768)
    ref2.add(ko.normalize_rows(x))
    assert ix.shadow_info() == {"bf16": True, "int8": True}
    D, I = ix.search(q[:1], 10)
    Dr, Ir = ref2.search(q[:1], 10)
    assert_topk_matches(D, I, Dr, Ir, ref2.rescore64(q[:1], Ir), "after reset")
    ix.close()
    ns = IndexFlatIP(768)
    ns.set_shadow(False)
    ns.add(x, normalize=True)
    assert ns.shadow_info() == {"bf16": False, "int8": False}
    ns.close()


def test_int8_only_shadow_rows_every_query_shape_matches_the_oracle():
    """An index that keeps int8 rows but NO bf16 rows (what a 38-46 M-row shard of 768 floats gets automatically: 5 bytes
    per element fit in 80 % of the HBM, 6 do not; forced here with set_shadow("int8")): single queries sweep the int8 rows,
    batches scan them with int8 MFMA, flagged queries (no bf16 rows for a second pass) go to the exact sweep; shapes the
    int8 scan does not serve -- k beyond its select's range, an int8 choice the feedback has backed off -- take the bf16
    scratch ranges / exact kernels of a shadow-less index.  Rows follow reallocation and reset; masks, duplicates floods
    and k > 128 included; same answers as the oracle throughout."""
    from oracle import knn_oracle as ko
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    ix = IndexFlatIP(768)
    ix.set_shadow("int8")
    ref = ko.FlatIndexOracle(768)
    q = ko.normalize_rows(synth.rows(300, 768, 710))
    for step, n in enumerate([900, 6000, 30000]):          # reallocations carry the int8 rows over
        x = synth.rows(n, 768, 720 + step)
        if step == 2:
            x[5000:9000] = x[4999]                            # a duplicate flood: band overflow -> exact sweep (no second pass)
        ix.add(x, normalize=True)
        ref.add(ko.normalize_rows(x))
        assert ix.shadow_info() == {"bf16": False, "int8": True}
    qq = np.concatenate([q, ref._xb[4999:5000] + 0.01 * q[:1]])     # ... and a query inside the flood
    for mode in ("coarse", "auto", "exact_fp32"):
        ix.set_search_mode(mode)
        for nq, k in ((1, 10), (1, 100), (3, 10), (40, 10), (301, 10), (301, 100), (2, 300)):
            D, I = ix.search(qq[-nq:], k)
            Dr, Ir = ref.search(qq[-nq:], k)
            assert_topk_matches(D, I, Dr, Ir, ref.rescore64(qq[-nq:], Ir), f"int8-only [{mode}] nq={nq} k={k}")
    ix.set_search_mode("coarse")
    allow = (np.arange(ix.ntotal) % 4) != 2
    sub = np.flatnonzero(allow)
    o2 = ko.FlatIndexOracle(768)
    o2.add(ref._xb[sub])
    for nq in (1, 64):
        D, I = ix.search(q[:nq], 10, allow=allow)
        Dr, Ir = o2.search(q[:nq], 10)
        assert_topk_matches(D, I, Dr, sub[Ir], o2.rescore64(q[:nq], Ir), f"int8-only masked nq={nq}")
    ix.reset()
    x = synth.rows(4000, 768, 730)
    ix.add(x, normalize=True)
    assert ix.shadow_info() == {"bf16": False, "int8": True}
    r2 = ko.FlatIndexOracle(768)
    r2.add(ko.normalize_rows(x))
    D, I = ix.search(q[:5], 10)
    Dr, Ir = r2.search(q[:5], 10)
    assert_topk_matches(D, I, Dr, Ir, r2.rescore64(q[:5], Ir), "int8-only after reset")
    ix.close()
    # from 300 k rows on batches with k <= 32 take the int8 MFMA scan by themselves
    big = IndexFlatIP(768)
    big.set_shadow("int8")
    big.add_synthetic(320_000, seed=740, first_row=0, normalize=True)
    rb = ko.FlatIndexOracle(768)
    rb.add(ko.normalize_rows(ko.synth_rows(320_000, 768, 740)))
    for nq in (64, 1):
        D, I = big.search(q[:nq], 10)
        Dr, Ir = rb.search(q[:nq], 10)
        assert_topk_matches(D, I, Dr, Ir, rb.rescore64(q[:nq], Ir), f"int8-only 320 k rows nq={nq}")
    assert big.shadow_info() == {"bf16": False, "int8": True}
    big.close()
    l2 = __import__("claude_semantic_search_amd.flat_index", fromlist=["x"]).IndexFlatL2(768)
    l2.set_shadow("int8")                                      # squared L2 has no int8 scan: no copy at all
    l2.add(x)
    assert l2.shadow_info() == {"bf16": False, "int8": False}
    l2.close()


def test_id_base_and_merge_parts_match_whole():
    """Row-partitioned shards + merge == one index, on one GPU."""
    import ctypes

    import torch

    from claude_semantic_search_amd import _native as nat
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    n, d, nq, k, G = 8000, 768, 12, 10, 4
    x = synth.rows(n, d, 21)
    q = synth.rows(nq, d, 22)
    whole = IndexFlatIP(d)
    whole.add(x, normalize=True)
    D, I = whole.search(q, k, normalize=True)
    qd = torch.from_numpy(q).cuda()
    Dp = torch.empty((G, nq, k), dtype=torch.float32, device="cuda")
    Ip = torch.empty((G, nq, k), dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    shards = []
    for g in range(G):
        s = IndexFlatIP(d)
        lo, hi = g * n // G, (g + 1) * n // G
        s.add(x[lo:hi], normalize=True)
        s.set_id_base(lo)
        s.search_dev(qd.data_ptr(), nq, k, Dp[g].data_ptr(), Ip[g].data_ptr(), st, normalize=True)
        shards.append(s)
    Do = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    Io = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    nat.check(nat.lib().css_merge_topk_dev(ctypes.c_void_p(Dp.data_ptr()), ctypes.c_void_p(Ip.data_ptr()), G, nq, k, 0,
                                           ctypes.c_void_p(Do.data_ptr()), ctypes.c_void_p(Io.data_ptr()), 0,
                                           ctypes.c_void_p(st)))
    torch.cuda.synchronize()
    assert np.array_equal(Io.cpu().numpy(), I)
    assert np.array_equal(Do.cpu().numpy(), D)


# ---- query batches > 16 take the MFMA kernel (k <= 64) -------------------------
@pytest.mark.parametrize("n,nq,k,metric", [
    (1, 20, 5, 0), (127, 17, 10, 0), (128, 40, 10, 0), (129, 129, 64, 0), (130, 129, 64, 1),
    (3000, 1000, 10, 0), (20000, 300, 10, 0), (20000, 300, 10, 1), (50000, 128, 1, 0), (40000, 257, 33, 0),
])
def test_mfma_batch_path(n, nq, k, metric):
    _run_case(n, 768, nq, k, metric, metric == 0, seed=n + nq)


@pytest.mark.parametrize("d", [4, 64, 96, 128, 1024])
def test_mfma_batch_dims(d):
    _run_case(4000, d, 64, 10, 0, True, seed=d)


def test_sixteen_query_tiles_share_the_row_tiles_of_a_stage():
    """4200 queries = a chunk of 4096 (16 query tiles: the most that share a row tile) and a ragged rest, on 40 000 rows:
    under CSS_KNN_SCAN=i8 (tests/test_knn_i8_forced_gpu.py) the main stage of the first chunk has 78 x 16 (row tile, query
    tile) pairs -- enough for k_scan_qreg_i8 by itself, four row-tile streams per XCD."""
    _run_case(40_000, 768, 4200, 10, 0, True, seed=77, with_next=True)


@pytest.mark.parametrize("d", [200, 256, 512, 768])
def test_batch_scan_row_widths_of_the_register_resident_query_kernel(d):
    """Rows of 256 / 512 / 768 padded columns: the int8 batch scan's later stages keep the queries in registers
    (k_scan_qreg_i8<4 | 8 | 12>; css_index.hip: qreg_applies) -- reached here when tests/test_knn_i8_forced_gpu.py re-runs
    this file with CSS_KNN_SCAN=i8 (with and without CSS_KNN_QREG=0); in the plain run the bf16 scan answers.  300
    queries = two query tiles, the second one ragged; 9000 rows = a ragged last row tile."""
    _run_case(9000, d, 300, 10, 0, True, seed=300 + d)


def test_config2_100k_768_1000_queries_top10():
    # BASELINE.json configs[1]: 100k x 768 index, 1000 queries, top-10 vs the CPU oracle
    _run_case(100_000, 768, 1000, 10, 0, True, seed=2)


def test_mfma_duplicates_ties():
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    base = synth.rows(300, 768, 5)
    x = np.concatenate([base, base], axis=0)
    ix = IndexFlatIP(768)
    ix.add(x, normalize=True)
    D, I = ix.search(base[:40], 4, normalize=True)
    for r in range(40):
        assert I[r][:2].tolist() == [r, r + 300] and D[r][0] == D[r][1]


# ---- coarse bf16 scan + exact rescoring (the default batched path): adversarial inputs ----------
def _check_against_oracle(x, q, k, normalize, what):
    from oracle import knn_oracle as ko
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    ix = IndexFlatIP(x.shape[1])
    ix.add(x, normalize=normalize)
    ix.set_search_mode("coarse")          # (auto would pick the exact kernels for an index this small)
    D, I = ix.search(q, k, normalize=normalize)
    ref = ko.FlatIndexOracle(x.shape[1], 0)
    xr, qr = (ko.normalize_rows(x), ko.normalize_rows(q)) if normalize else (x, q)
    ref.add(xr)
    Dr, Ir = ref.search(qr, k)
    D64 = ref.rescore64(qr, np.where(Ir < 0, 0, Ir))
    assert_topk_matches(D, I, Dr, Ir, D64, what)
    ix.close()
    return D, I


def test_coarse_band_overflow_falls_back_to_exact():
    # 6000 copies of one row: every copy ties for the top -> the candidate band overflows the
    # rescoring buffer and the flagged queries are re-run on the exact path (lowest ids win the tie)
    base = synth.rows(3000, 768, 21)
    x = np.concatenate([base[:1500], np.repeat(base[7:8], 6000, axis=0), base[1500:]], axis=0)
    q = np.concatenate([base[7:8] + 0.01 * synth.rows(3, 768, 22), synth.rows(37, 768, 23)], axis=0)
    D, I = _check_against_oracle(x, q, 10, True, "duplicate flood")
    assert I[0].tolist() == [7] + list(range(1500, 1509))


def test_coarse_clustered_rows_dense_bands():
    # tight clusters (cosine spread ~1e-3 inside a cluster): all 400 members of the query's cluster
    # fall inside the coarse error band of the k-th best and are rescored exactly
    cent = synth.rows(20, 768, 31)
    x = np.repeat(cent, 400, axis=0) + 0.05 * synth.rows(8000, 768, 32)
    q = cent[:18] + 0.02 * synth.rows(18, 768, 33)
    q = np.concatenate([q, synth.rows(14, 768, 34)], axis=0)
    _check_against_oracle(x, q, 10, True, "clustered")


def _to_bf16_midpoints(a):
    """Every element moved to the exact midpoint between two neighbouring bf16 values: round-to-nearest-even then
    loses a full half ulp per element, the largest rounding error a bf16 copy can have."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    return ((u & np.uint32(0xFFFF0000)) | np.uint32(0x8000)).view(np.float32)


def test_coarse_measured_error_band_holds_for_worst_case_roundings():
    # The candidate band is drawn from the rounding errors MEASURED at ingest and query preparation (cz_eps), not from
    # the unit roundoff.  Worst case for that: operands on bf16 midpoints, a dense cluster whose exact scores differ by
    # far less than the coarse error, and the badly rounding rows arriving in a LATER add than rows that round exactly.
    rng = np.random.default_rng(7)
    exact_rows = synth.rows(20000, 768, 51)
    exact_rows /= np.linalg.norm(exact_rows, axis=1, keepdims=True)
    exact_rows = (exact_rows.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)      # bf16-representable: error 0
    cent = synth.rows(6, 768, 52)
    cent /= np.linalg.norm(cent, axis=1, keepdims=True)
    cluster = np.repeat(cent, 300, axis=0) + 2e-4 * synth.rows(1800, 768, 53)               # exact scores ~1e-4 apart
    noisy = np.concatenate([cluster, synth.rows(6000, 768, 54) / np.sqrt(768.0)], axis=0)
    noisy = _to_bf16_midpoints(noisy[rng.permutation(len(noisy))])
    q = _to_bf16_midpoints(np.concatenate([cent + 0.01 * synth.rows(6, 768, 55), synth.rows(42, 768, 56) / np.sqrt(768.0)]))
    from oracle import knn_oracle as ko
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    ix = IndexFlatIP(768)
    ix.set_search_mode("coarse")
    ref = ko.FlatIndexOracle(768, 0)
    for part, what in ((exact_rows, "rows without rounding error"), (noisy, "worst-case roundings added later")):
        ix.add(part, normalize=False)
        ref.add(part)
        for k in (10, 100):
            D, I = ix.search(q, k, normalize=False)
            Dr, Ir = ref.search(q, k)
            assert_topk_matches(D, I, Dr, Ir, ref.rescore64(q, np.where(Ir < 0, 0, Ir)), f"{what}, k={k}")
            D3, I3 = ix.search(q[:3], k, normalize=False)      # the 1..4-query sweep: fp32 queries, bf16 rows
            assert_topk_matches(D3, I3, Dr[:3], Ir[:3], ref.rescore64(q[:3], np.where(Ir[:3] < 0, 0, Ir[:3])), f"{what}, k={k}, 3 queries")
    ix.close()


def _bf16_rne(a):
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)


def _adversarial_pool(q, n_pool, seed, spread, grid="bf16"):
    """Rows near the direction of q whose rounding errors are ALIGNED with q (the Cauchy-Schwarz worst case the error
    band has to cover; random errors stay 20-30 x below it): x = xh + s * 0.24 cell(xh) * sign(q), xh on the grid of
    the reduced-precision copy (bf16 values; or, grid = "int8", the per-row-scaled int8 values of the few-query
    sweep), so the copy of x is xh and x.q = xh.q + s * D.  The rows with the highest coarse scores get s = -1
    (decoys), the rows just below them s = +1: the true top-k are rows whose coarse score lies ~1.6 D under the k-th
    best coarse score, behind every decoy.  Returns (rows, D, coarse scores in float64)."""
    d = q.shape[0]
    t = 1.0 + spread * (2.0 * np.random.default_rng(seed).random((n_pool, 1)) - 1.0)
    base = (q[None, :] * t + 0.004 * synth.rows(n_pool, d, seed + 1) / np.sqrt(d)).astype(np.float32)
    q64 = q.astype(np.float64)
    if grid == "bf16":
        xh = _bf16_rne(base).astype(np.float64)
        cell = np.spacing(np.abs(xh.astype(np.float32))).astype(np.float64) * 65536.0
    else:
        amax = np.abs(base).max(axis=1, keepdims=True)
        s8 = (amax / np.float32(127.0)).astype(np.float32)
        kq = np.clip(np.rint(base / s8), -127, 127).astype(np.float64)
        xh = s8.astype(np.float64) * kq
        cell = np.where(np.abs(kq) == 127, 0.0, 1.0) * s8.astype(np.float64)      # the row's largest element stays put
    ch = xh @ q64
    D = float(np.median(0.24 * (cell * np.abs(q64)[None, :]).sum(axis=1)))
    s_ = np.where(ch >= ch.max() - 1.6 * D, -1.0, 1.0)
    x = (xh + s_[:, None] * 0.24 * cell * np.sign(q64)[None, :]).astype(np.float32)
    if grid == "bf16":
        assert np.array_equal(_bf16_rne(x), xh.astype(np.float32)), "construction: the perturbed rows must round back to the grid rows"
    else:   # what k_ingest_rows does: s = max|x| / 127, byte = 128 + rint(x * (127 / max|x|))
        am = np.abs(x).max(axis=1, keepdims=True)
        k2 = np.rint((x * (np.float32(127.0) / am)).astype(np.float32))
        assert np.array_equal(k2, kq.astype(np.float32)), "construction: the perturbed rows must quantise back to the grid rows"
        ch = ((am / np.float32(127.0)).astype(np.float32).astype(np.float64) * kq) @ q64
    return x, D, ch


@pytest.mark.parametrize("n_pool,spread,grid", [(600, 0.005, "bf16"), (6000, 0.005, "bf16"), (1500, 0.02, "int8")])
def test_coarse_error_band_covers_rounding_errors_aligned_with_the_query(n_pool, spread, grid):
    # (a build whose measured bound is scaled by 0.2 -- -DCZ_EPS_TEST_SCALE=0.2f -- fails this test: the true top-k
    # rows fall out of a band drawn that narrow; 600 rows: the band is rescored directly, 6000: flagged -> second pass;
    # "int8": the grid of the 1..4-query sweep's rows, checked through that sweep)
    from oracle import knn_oracle as ko
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    qs, pools = [], []
    for j in range(3):
        qj = synth.rows(1, 768, 61 + j)[0]
        qj = _bf16_rne(qj / np.linalg.norm(qj))
        xj, D, ch = _adversarial_pool(qj, n_pool, 70 + 10 * j, spread, grid)
        exact = xj.astype(np.float64) @ qj.astype(np.float64)
        top = np.argsort(-exact)[:10]
        coarse_rank = (ch[None, :] > ch[top][:, None]).sum(axis=1)
        assert coarse_rank.min() >= 10, "construction: every true top-10 row must sit behind >= 10 decoys in coarse order"
        assert (ch.max() - ch[top]).min() > 1.2 * D
        qs.append(qj)
        pools.append(xj)
    filler = synth.rows(20000, 768, 99) / np.sqrt(768.0)
    x = np.concatenate([filler[:7000], pools[0], filler[7000:15000], pools[1], pools[2], filler[15000:]], axis=0)
    q = np.stack(qs + [synth.rows(1, 768, 100 + j)[0] / np.sqrt(768.0) for j in range(29)]).astype(np.float32)
    ix = IndexFlatIP(768)
    ix.add(x, normalize=False)
    ix.set_search_mode("coarse")
    ref = ko.FlatIndexOracle(768, 0)
    ref.add(x)
    # 32 queries: the MFMA cascade (bf16 rows; the int8 rows under CSS_KNN_SCAN=i8, tests/test_knn_i8_forced_gpu.py);
    # 3 queries: the 1..4-query sweep (int8 rows)
    for nq in (32, 3):
        D_, I_ = ix.search(q[:nq], 10, normalize=False)
        Dr, Ir = ref.search(q[:nq], 10)
        assert_topk_matches(D_, I_, Dr, Ir, ref.rescore64(q[:nq], np.where(Ir < 0, 0, Ir)), f"aligned rounding errors, nq={nq}")
    ix.close()


def test_rows_with_dominant_dimensions_stay_exact_whichever_copy_is_read():
    # sentence embeddings tend to carry a few dimensions far larger than the rest: they set the int8 scale of a row, the
    # other 766 elements round to almost nothing, the measured band gets wide (queries overflow their buffers and end in
    # the fix-up).  Results must stay exact, and the index falls back to the bf16 rows for the searches that follow.
    from oracle import knn_oracle as ko
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    x = synth.rows(30000, 768, 141)
    x[:, 5] *= 25.0
    x[:, 77] = 12.0 + x[:, 77]
    q = synth.rows(40, 768, 142)
    q[:, 5] *= 25.0
    q[:, 77] = 12.0 + q[:, 77]
    ix = IndexFlatIP(768)
    ix.add(x, normalize=True)
    ix.set_search_mode("coarse")
    ref = ko.FlatIndexOracle(768, 0)
    ref.add(ko.normalize_rows(x))
    qn = ko.normalize_rows(q)
    for rep in range(3):                       # (the first search of each kind reads the int8 rows, the next ones adapt)
        for nq in (1, 3, 40):
            D, I = ix.search(q[:nq], 10, normalize=True)
            Dr, Ir = ref.search(qn[:nq], 10)
            assert_topk_matches(D, I, Dr, Ir, ref.rescore64(qn[:nq], Ir), f"dominant dimensions, nq={nq}, rep={rep}")
    ix.close()


def test_coarse_raw_inner_product_wide_norms():
    # un-normalised rows with norms spread over 3 decades: the error bound scales with max ||row||
    x = synth.rows(12000, 768, 41) * (10.0 ** (3.0 * np.random.default_rng(1).random((12000, 1)) - 1.5)).astype(np.float32)
    q = synth.rows(48, 768, 42) * 3.0
    from oracle import knn_oracle as ko
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    ix = IndexFlatIP(768)
    ix.add(x, normalize=False)
    ix.set_search_mode("coarse")
    D, I = ix.search(q, 10, normalize=False)
    ref = ko.FlatIndexOracle(768, 0)
    ref.add(x)
    Dr, Ir = ref.search(q, 10)
    assert (I == Ir).mean() > 0.999 and np.allclose(D, Dr, rtol=1e-5, atol=1e-3)
    ix.close()


@pytest.mark.parametrize("nq", [1, 3, 40, 300])
def test_coarse_l2_raw_rows_and_duplicates(nq):
    # squared-L2 through the candidate path: un-normalised rows with norms over a decade, 40 exact duplicates of
    # one row (ties -> lowest ids first), queries that ARE rows (distance 0)
    from oracle import knn_oracle as ko
    from claude_semantic_search_amd.flat_index import IndexFlatL2

    rng = np.random.default_rng(8)
    x = synth.rows(20000, 768, 81) * (10.0 ** rng.uniform(-0.5, 0.5, (20000, 1))).astype(np.float32)
    x[5000:5040] = x[123]
    q = np.concatenate([x[[123, 7, 19999]], synth.rows(max(nq - 3, 0), 768, 82)])[:nq]
    ix = IndexFlatL2(768)
    ix.add(x)
    ref = ko.FlatIndexOracle(768, 1)
    ref.add(x)
    Dr, Ir = ref.search(q, 10)
    D64 = ref.rescore64(q, Ir)
    gaps = np.abs(np.diff(D64, axis=1))
    safe = np.ones_like(Ir, dtype=bool)      # ids must agree wherever the fp64 gap exceeds 1e-5 of the norm scale (~8e3)
    safe[:, 1:] &= gaps > 0.1
    safe[:, :-1] &= gaps > 0.1
    safe[:, -1] = False                      # (the gap to rank k+1 is not known here)
    for mode in ("exact_fp32", "coarse"):
        ix.set_search_mode(mode)
        D, I = ix.search(q, 10)
        # squared norms here reach ~8e3: the bar is relative to them (fp32 has 7 digits; the batched exact kernel
        # uses faiss' expanded form ||q||^2 - 2 x.q + ||x||^2, whose self distance is ~1e-6 ||x||^2, not 0)
        assert np.allclose(D, Dr, rtol=2e-6, atol=1e-2), f"[{mode}] {np.abs(D - Dr).max()}"
        assert (I[safe] == Ir[safe]).all(), f"[{mode}] id mismatch"
        assert I[0, 0] == 123 and D[0, 0] < 1e-2 and I[0, 1] == 5000
    ix.close()


def test_candidate_path_really_runs_for_both_metrics():
    # in-library timing scopes show which kernels ran: on ordinary data the coarse cascade answers alone
    # (no exact re-run), for inner product and for squared L2
    from claude_semantic_search_amd import _native as nat
    from claude_semantic_search_amd.flat_index import IndexFlat

    for metric, norm in ((0, True), (1, False)):
        ix = IndexFlat(768, metric)
        ix.add(synth.rows(30000, 768, 91), normalize=norm)
        ix.set_search_mode("coarse")
        for nq, scope in ((64, "knn_coarse_cascade"), (2, "knn_sweep_cascade")):
            nat.prof_reset()
            nat.prof_enable(True)
            ix.search(synth.rows(nq, 768, 92), 10, normalize=norm)
            nat.prof_enable(False)
            assert nat.prof_read(scope)[1] == 1
            assert nat.prof_read("knn_scan_mfma")[1] == 0 and nat.prof_read("knn_scan_small")[1] == 0
        nat.prof_reset()
        ix.close()


@pytest.mark.parametrize("nq", [3, 7, 12, 16, 17, 32])
def test_three_to_sixteen_queries_sweep_on_the_int8_mfma(nq):
    """3..16 inner-product queries on an index with int8 rows, from 50 k rows on (k <= 32; 3 and 4 queries reach the
    candidate path from 100 k rows): the cascade stages run on the
    int8 MFMA with the queries as the register operand (k_sweep_mfma_i8; css_index.hip: mfma_sweep_applies) -- the
    reference's search (src/storage.py:429) returns the same ids whatever the batch size, so must this.  The timing
    scopes show that the path under test is the one that ran; 33 queries go through the batch scan."""
    from oracle import knn_oracle as ko
    from claude_semantic_search_amd import _native as nat
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    n, d, k = 100_000, 768, 10
    x = ko.normalize_rows(synth.rows(n, d, 41))
    ix = IndexFlatIP(d)
    ix.add(x)
    ref = ko.FlatIndexOracle(d, 0)
    ref.add(x)
    # (17 .. 32 queries: two fragment sets per lane; 33 queries go through the batch scan)
    for count, scope in ((nq, "knn_sweep_mfma_main"), (33, "knn_scan_coarse_main")):
        q = synth.rows(count, d, 42 + count)
        nat.prof_reset()
        nat.prof_enable(True)
        D, I = ix.search(q, k, normalize=True)
        nat.prof_enable(False)
        assert nat.prof_read(scope)[1] == 1, f"{count} queries did not take {scope}"
        qr = ko.normalize_rows(q)
        Dr, Ir = ref.search(qr, k)
        assert_topk_matches(D, I, Dr, Ir, ref.rescore64(qr, Ir), f"nq={count} on the int8 MFMA sweep / batch scan")
    nat.prof_reset()
    ix.close()


@pytest.mark.parametrize("n,nq,k", [(9000, 40, 100), (9000, 33, 128), (700, 4200, 10), (257, 513, 3)])
def test_coarse_large_k_and_query_chunks(n, nq, k):
    _run_case(n, 768, nq, k, 0, True, seed=n + k)


def test_coarse_and_split_paths_agree(monkeypatch):
    # same batch through the single-query sweeps (exact fp32 VALU path): identical ids, scores to 1e-5
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    x = synth.rows(30000, 768, 51)
    q = synth.rows(64, 768, 52)
    ix = IndexFlatIP(768)
    ix.add(x, normalize=True)
    ix.set_search_mode("coarse")
    D, I = ix.search(q, 10, normalize=True)
    for r in range(0, 64, 9):
        d1, i1 = ix.search(q[r:r + 1], 10, normalize=True)
        assert i1[0].tolist() == I[r].tolist() and np.allclose(d1[0], D[r], atol=1e-5)
    ix.close()


# ---- masked search (filter / tombstone push-down): every kernel family, both search modes ----------
@pytest.mark.parametrize("metric,nq,density", [(0, 1, 0.05), (0, 3, 0.5), (0, 40, 0.05), (0, 300, 0.3),
                                               (1, 2, 0.2), (1, 40, 0.2)])
def test_masked_search_matches_oracle_on_the_allowed_rows(metric, nq, density):
    from oracle import knn_oracle as ko
    from claude_semantic_search_amd.flat_index import IndexFlat

    n, d, k = 30000, 768, 10
    x = synth.rows(n, d, 61)
    q = synth.rows(nq, d, 62)
    allow = np.random.default_rng(5).random(n) < density
    norm = metric == 0
    ix = IndexFlat(d, metric)
    ix.add(x, normalize=norm)
    sub = np.flatnonzero(allow)
    ref = ko.FlatIndexOracle(d, metric)
    xr, qr = (ko.normalize_rows(x), ko.normalize_rows(q)) if norm else (x, q)
    ref.add(xr[sub])
    Dr, Ir = ref.search(qr, k)
    D64 = ref.rescore64(qr, Ir)
    Ir = sub[Ir]
    for mode in ("exact_fp32", "coarse", "auto"):
        ix.set_search_mode(mode)
        D, I = ix.search(q, k, normalize=norm, allow=allow)
        assert allow[I].all()
        assert_topk_matches(D, I, Dr, Ir, D64, f"masked [{mode}] metric={metric} nq={nq}")
    ix.close()


def test_masked_search_with_few_or_no_allowed_rows_pads():
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    x = synth.rows(5000, 768, 63)
    ix = IndexFlatIP(768)
    ix.add(x, normalize=True)
    allow = np.zeros(5000, dtype=bool)
    allow[[7, 4000, 4999]] = True
    ix.set_search_mode("coarse")
    for nq in (1, 20):
        D, I = ix.search(x[:nq], 5, normalize=True, allow=allow)
        assert (np.sort(I[:, :3], axis=1) == np.array([7, 4000, 4999])).all() and (I[:, 3:] == -1).all()
        assert (D[:, 3:] == np.finfo(np.float32).min).all()
        D, I = ix.search(x[:nq], 5, normalize=True, allow=np.zeros(5000, dtype=bool))
        assert (I == -1).all()
    with pytest.raises(ValueError):
        ix.search(x[:1], 5, allow=np.zeros(10, dtype=bool))
    ix.close()


def test_invalid_arguments_raise():
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    ix = IndexFlatIP(8)
    with pytest.raises(ValueError):
        ix.add(np.zeros((2, 7), np.float32))
    with pytest.raises(ValueError):
        ix.search(np.zeros((1, 8), np.float32), 0)
    with pytest.raises(ValueError):
        ix.search(np.zeros((1, 8), np.float32), 2049)
    D, I = ix.search(np.zeros((1, 8), np.float32), 3)  # empty index: padded
    assert I[0].tolist() == [-1, -1, -1]


# ---- device-side exact fix-up of flagged queries (no host round trip on the candidate path) ----------
@pytest.mark.parametrize("metric", [0, 1])
def test_many_flagged_queries_are_fixed_up_on_the_device(metric):
    # 40 queries next to a flood of 5000 identical rows: every one of them overflows its band, so the fix-up
    # walks several chunks of flagged queries and many blocks merge into each query's global list
    from oracle import knn_oracle as ko
    from claude_semantic_search_amd import _native as nat
    from claude_semantic_search_amd.flat_index import IndexFlat

    base = synth.rows(4000, 768, 71)
    x = np.concatenate([base[:2000], np.repeat(base[11:12], 5000, axis=0), base[2000:]], axis=0)
    q = np.concatenate([base[11:12] + 0.01 * synth.rows(40, 768, 72), synth.rows(25, 768, 73)], axis=0)
    norm = metric == 0
    ix = IndexFlat(768, metric)
    ix.add(x, normalize=norm)
    ix.set_search_mode("coarse")
    ref = ko.FlatIndexOracle(768, metric)
    xr, qr = (ko.normalize_rows(x), ko.normalize_rows(q)) if norm else (x, q)
    ref.add(xr)
    for nq in (65, 3):          # MFMA cascade and the few-query sweep cascade
        nat.prof_reset()
        nat.prof_enable(True)
        D, I = ix.search(q[:nq], 10, normalize=norm)
        nat.prof_enable(False)
        assert nat.prof_read("knn_fix_scan")[1] >= 1 and nat.prof_read("knn_scan_small")[1] == 0
        Dr, Ir = ref.search(qr[:nq], 10)
        D64 = ref.rescore64(qr[:nq], Ir)
        assert_topk_matches(D, I, Dr, Ir, D64, f"flood metric={metric} nq={nq}")
        # exact ties inside the flood: the lowest ids win (rows 2000.. are the copies; row 11 is the original)
        assert I[0].tolist() == [11] + list(range(2000, 2009))
    nat.prof_reset()
    ix.close()


# ---- indexes without bf16 shadow rows: candidate scores from the fp32 rows (split operands) + fp32 rescoring ----
@pytest.mark.parametrize("n,nq,k,metric", [(20000, 300, 10, 0), (20000, 300, 10, 1), (9000, 40, 60, 0),
                                           (3000, 129, 1, 0), (50000, 17, 10, 0)])
def test_no_shadow_batches_match_oracle(n, nq, k, metric):
    from oracle import knn_oracle as ko
    from claude_semantic_search_amd import _native as nat
    from claude_semantic_search_amd.flat_index import IndexFlat

    norm = metric == 0
    x = synth.rows(n, 768, 100 + n % 97)
    q = synth.rows(nq, 768, 200 + nq)
    ix = IndexFlat(768, metric)
    ix.set_shadow(False)
    ix.add(x, normalize=norm)
    ref = ko.FlatIndexOracle(768, metric)
    xr, qr = (ko.normalize_rows(x), ko.normalize_rows(q)) if norm else (x, q)
    ref.add(xr)
    Dr, Ir = ref.search(qr, k)
    D64 = ref.rescore64(qr, Ir)
    # auto / coarse: bf16 rows rounded on the fly, range by range, through the cascade of shadowed indexes; split: the
    # split-operand scan (the fallback when no scratch memory is left); exact_fp32: the fp32-input MFMA scan
    for mode, scope in (("auto", "knn_noshadow_ranges"), ("coarse", "knn_noshadow_ranges"), ("split", "knn_split_cascade"),
                        ("exact_fp32", "knn_scan_mfma")):
        ix.set_search_mode(mode)
        nat.prof_reset()
        nat.prof_enable(True)
        D, I = ix.search(q, k, normalize=norm)
        nat.prof_enable(False)
        assert nat.prof_read(scope)[1] == 1, f"[{mode}] expected the {scope} path"
        assert (nat.prof_read("knn_coarse_cascade")[1] >= 1) == (scope == "knn_noshadow_ranges")
        assert (nat.prof_read("knn_split_cascade")[1] >= 1) == (scope == "knn_split_cascade")
        assert_topk_matches(D, I, Dr, Ir, D64, f"no shadow [{mode}] n={n} nq={nq} k={k} metric={metric}")
    nat.prof_reset()
    with pytest.raises(RuntimeError):
        ix.set_shadow(True)      # only on an empty index
    ix.close()


@pytest.mark.parametrize("metric", [0, 1])
def test_no_shadow_row_ranges_merge_like_one_index(metric):
    """Shadow-less batches round the rows to bf16 one row range at a time and merge the per-range top-k lists: 70 000
    rows in ranges of 16 384 (5 ranges, the last one ragged), 2 500 rows (one range), a masked search across the
    ranges and id_base, against the oracle; the answers must not depend on the range size."""
    from oracle import knn_oracle as ko
    from claude_semantic_search_amd import _native as nat
    from claude_semantic_search_amd.flat_index import IndexFlat

    norm = metric == 0
    n, nq, k = 70_000, 130, 10
    x = synth.rows(n, 768, 501)
    q = synth.rows(nq, 768, 502)
    ix = IndexFlat(768, metric)
    ix.set_shadow(False)
    ix.add(x, normalize=norm)
    ix.set_id_base(1000)
    ref = ko.FlatIndexOracle(768, metric)
    xr, qr = (ko.normalize_rows(x), ko.normalize_rows(q)) if norm else (x, q)
    ref.add(xr)
    Dr, Ir = ref.search(qr, k)
    D64 = ref.rescore64(qr, Ir)
    got = []
    for rows in (16_384, 0):
        ix.set_range_rows(rows)
        nat.prof_reset()
        nat.prof_enable(True)
        D, I = ix.search(q, k, normalize=norm)
        nat.prof_enable(False)
        assert nat.prof_read("knn_noshadow_ranges")[1] == 1
        # (the scratch rows are bf16, or int8 where the int8 scan is chosen / forced: one conversion per range)
        assert nat.prof_read("knn_rows_to_bf16")[1] + nat.prof_read("knn_rows_to_i8")[1] == (5 if rows else 1)
        assert nat.prof_read("knn_merge_parts")[1] == (1 if rows else 0)
        assert_topk_matches(D, I - 1000, Dr, Ir, D64, f"no shadow, ranges of {rows} rows, metric={metric}")
        got.append((D, I))
    assert np.array_equal(got[0][1], got[1][1]) and np.array_equal(got[0][0], got[1][0])
    nat.prof_reset()
    # masked: only rows whose number is 1 mod 3 (bitmap in local row numbering, cut at range boundaries)
    allow = (np.arange(n) % 3) == 1
    ix.set_range_rows(16_384)
    Dm, Im = ix.search(q, k, normalize=norm, allow=allow)
    sub = np.flatnonzero(allow)
    o = ko.FlatIndexOracle(768, metric)
    o.add(xr[sub])
    Ds, Is = o.search(qr, k)
    assert_topk_matches(Dm, Im - 1000, Ds, sub[Is], o.rescore64(qr, Is), f"no shadow, masked, metric={metric}")
    ix.close()


def test_no_shadow_large_batches_are_chunked_and_k_above_60_takes_the_mfma_scan():
    """the split-operand candidate path sized its 32 KiB-per-query candidate buffers for the whole batch
    (100 k queries = 3.3 GB) -- it now walks the batch in chunks of 4096 like the bf16 cascade; and shadow-less
    batches with k = 61..64 (no room for the split scan's extra ranks) take the fp32-input MFMA scan, not 16-query
    VALU sweeps."""
    from oracle import knn_oracle as ko
    from claude_semantic_search_amd import _native as nat
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    x = synth.rows(6000, 768, 311)
    ix = IndexFlatIP(768)
    ix.set_shadow(False)
    ix.add(x, normalize=True)
    ix.set_search_mode("split")
    ref = ko.FlatIndexOracle(768, 0)
    ref.add(ko.normalize_rows(x))
    q = synth.rows(4096 + 4096 + 130, 768, 312)            # two full chunks + a ragged one
    qr = ko.normalize_rows(q)
    nat.prof_reset()
    nat.prof_enable(True)
    D, I = ix.search(q, 10, normalize=True)
    nat.prof_enable(False)
    assert nat.prof_read("knn_split_cascade")[1] == 3
    Dr, Ir = ref.search(qr, 10)
    assert_topk_matches(D, I, Dr, Ir, ref.rescore64(qr, Ir), "no shadow, 8322 queries")
    nat.prof_reset()
    nat.prof_enable(True)
    D, I = ix.search(q[:70], 63, normalize=True)
    nat.prof_enable(False)
    assert nat.prof_read("knn_scan_mfma")[1] == 1 and nat.prof_read("knn_scan_small")[1] == 0
    Dr, Ir = ref.search(qr[:70], 63)
    assert_topk_matches(D, I, Dr, Ir, ref.rescore64(qr[:70], Ir), "no shadow, k = 63")
    nat.prof_reset()
    ix.close()


def test_no_shadow_band_beyond_the_kept_ranks_is_fixed_up():
    # 30 exact copies of the best row: the k + 4 ranks the split scan keeps all tie, the band cannot be shown
    # closed, the query is flagged and the exact fix-up returns the lowest ids
    from claude_semantic_search_amd import _native as nat
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    base = synth.rows(6000, 768, 75)
    x = np.concatenate([base[:3000], np.repeat(base[5:6], 30, axis=0), base[3000:]], axis=0)
    q = np.concatenate([base[5:6], synth.rows(39, 768, 76)], axis=0)
    ix = IndexFlatIP(768)
    ix.set_shadow(False)
    ix.add(x, normalize=True)
    ix.set_search_mode("split")
    nat.prof_reset()
    nat.prof_enable(True)
    D, I = ix.search(q, 10, normalize=True)
    nat.prof_enable(False)
    assert nat.prof_read("knn_split_cascade")[1] == 1 and nat.prof_read("knn_fix_scan")[1] == 1
    assert I[0].tolist() == [5] + list(range(3000, 3009)) and np.all(np.abs(D[0] - 1.0) < 1e-6)
    nat.prof_reset()
    _ = _check_against_oracle  # (ordinary queries of the same batch are covered by the parametrised test above)
    ix.close()


def test_search_right_behind_an_asynchronous_ingest_on_another_stream():
    # css_index_add_synthetic only enqueues on the caller's stream; a search on a different stream must still
    # see the rows, norms and max-norm scalar (ordered by an event inside the library)
    import torch

    from claude_semantic_search_amd.flat_index import IndexFlatIP

    side = torch.cuda.Stream()
    n = 3_000_000
    ix = IndexFlatIP(768)
    ix.reserve(n)
    ix.add_synthetic(n, seed=44, first_row=0, normalize=True, stream=side.cuda_stream)   # ~6 ms of device work
    q = synth.rows(40, 768, 45)
    D, I = ix.search(q, 10, normalize=True)              # host API: runs on the index's own stream
    torch.cuda.synchronize()
    D2, I2 = ix.search(q, 10, normalize=True)
    assert np.array_equal(I, I2) and np.array_equal(D, D2)
    rows = synth.rows(3, 768, 44, first_row=n - 3)
    rows /= (np.linalg.norm(rows, axis=1, keepdims=True) + 1e-8)
    d1, i1 = ix.search(rows, 1)
    assert i1[:, 0].tolist() == [n - 3, n - 2, n - 1] and np.all(np.abs(d1 - 1.0) < 1e-5)
    ix.close()


def test_searches_of_one_index_on_two_streams_do_not_share_workspaces_concurrently():
    """All searches of an index use one set of device workspaces (padded queries, thresholds, candidate buffers,
    partial lists).  css_index_search_dev only enqueues, so two searches given DIFFERENT streams used to be able
    to run at the same time and overwrite each other's workspaces: the library now chains them with
    an event.  Two different query batches are enqueued back to back on two streams, several rounds, in the
    candidate path and in the exact path, and compared with the same searches run alone."""
    import torch

    from claude_semantic_search_amd.flat_index import IndexFlatIP

    n, d, k = 1_500_000, 768, 10
    ix = IndexFlatIP(d)
    ix.reserve(n)
    ix.add_synthetic(n, seed=61, first_row=0, normalize=True)
    dev = torch.device("cuda:0")
    qa = torch.from_numpy(synth.rows(300, d, 62)).to(dev)
    qb = torch.from_numpy(synth.rows(37, d, 63)).to(dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def run(q, stream):
        D = torch.empty((q.shape[0], k), dtype=torch.float32, device=dev)
        I = torch.empty((q.shape[0], k), dtype=torch.int64, device=dev)
        ix.search_dev(q.data_ptr(), q.shape[0], k, D.data_ptr(), I.data_ptr(), stream=stream.cuda_stream, normalize=True)
        return D, I

    for mode in ("auto", "exact_fp32"):
        ix.set_search_mode(mode)
        torch.cuda.synchronize()
        Da, Ia = run(qa, s1)
        torch.cuda.synchronize()
        Db, Ib = run(qb, s2)
        torch.cuda.synchronize()
        for _ in range(4):
            Da2, Ia2 = run(qa, s1)          # ~1-3 ms of device work, still running when the next call is enqueued
            Db2, Ib2 = run(qb, s2)
            Da3, Ia3 = run(qa, s1)
            torch.cuda.synchronize()
            assert torch.equal(Ia, Ia2) and torch.equal(Da, Da2), mode
            assert torch.equal(Ib, Ib2) and torch.equal(Db, Db2), mode
            assert torch.equal(Ia, Ia3) and torch.equal(Da, Da3), mode
    ix.close()


def test_two_host_threads_search_own_and_shared_indexes(tmp_path):
    """The concurrency contract of css_hip.h through the C ABI without Python in the way: tests/native/cabi_threads.cc
    (the driver of the CPU sanitizer builds, tests/test_cabi_sanitizers.py) linked against the product library; two
    threads each build and search their own index and both search a shared one, 30 rounds, answers must not move."""
    import subprocess
    from pathlib import Path

    csrc = Path(__file__).resolve().parents[1] / "claude_semantic_search_amd" / "csrc"
    r = subprocess.run(["make", "-C", str(csrc), "threads"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    vocab = tmp_path / "vocab.txt"
    vocab.write_text("\n".join(["<s>", "<pad>", "</s>", "<unk>", "fix", "the", "python", "error", "vector", "search",
                                "kernel", "##s", "on", "a", "gpu", "claude", "session", "index", "test"]) + "\n")
    r = subprocess.run([str(csrc / "san" / "cabi_threads"), str(vocab), "gpu"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "cabi_threads: ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
