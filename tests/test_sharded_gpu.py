"""GPU: the product wiring of the row-sharded index -- packed exchange records, the HIP merge that reads them, and a
two-rank rehearsal (both ranks on cuda:0, gloo moving the records) of exactly what ``bench.py --gpus 2`` runs."""
import ctypes
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_packed_records_merge_equals_one_index():
    import torch

    from claude_semantic_search_amd import _native as nat
    from claude_semantic_search_amd import synth
    from claude_semantic_search_amd.flat_index import IndexFlat
    from claude_semantic_search_amd.sharded import packed_layout

    for metric, norm, nq, k in ((0, True, 13, 10), (1, False, 7, 3), (0, True, 1, 1)):
        n, d, G = 9000, 768, 3
        x = synth.rows(n, d, 21 + metric)
        q = synth.rows(nq, d, 22)
        whole = IndexFlat(d, metric)
        whole.add(x, normalize=norm)
        D, I = whole.search(q, k, normalize=norm)
        qd = torch.from_numpy(q).cuda()
        ib, db, record = packed_layout(nq, k)
        assert record % 16 == 0 and record >= db
        recv = torch.zeros((G, record), dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        shards = []
        for g in range(G):
            s = IndexFlat(d, metric)
            lo, hi = g * n // G, (g + 1) * n // G
            s.add(x[lo:hi], normalize=norm)
            s.set_id_base(lo)
            Iv = recv[g, :ib].view(torch.int64)
            Dv = recv[g, ib:db].view(torch.float32)
            s.search_dev(qd.data_ptr(), nq, k, Dv.data_ptr(), Iv.data_ptr(), st, normalize=norm)
            shards.append(s)
        Do = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        Io = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        nat.check(nat.lib().css_merge_topk_packed_dev(ctypes.c_void_p(recv.data_ptr()), G, record, nq, k, metric,
                                                      ctypes.c_void_p(Do.data_ptr()), ctypes.c_void_p(Io.data_ptr()), 0,
                                                      ctypes.c_void_p(st)))
        torch.cuda.synchronize()
        assert np.array_equal(Io.cpu().numpy(), I) and np.array_equal(Do.cpu().numpy(), D)
    with pytest.raises(nat.CssError):
        nat.check(nat.lib().css_merge_topk_packed_dev(ctypes.c_void_p(recv.data_ptr()), G, 8, nq, k, 0,
                                                      ctypes.c_void_p(Do.data_ptr()), ctypes.c_void_p(Io.data_ptr()), 0,
                                                      ctypes.c_void_p(st)))


def _rank(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from claude_semantic_search_amd import synth
        from claude_semantic_search_amd.sharded import ShardedFlatIndex

        torch.cuda.set_device(0)
        sh = ShardedFlatIndex(768, 0, device_index=0)
        sh.add_synthetic_global(300_000, seed=4, normalize=True)
        sh.add_routed(synth.rows(500, 768, 8), normalize=True)           # second segment on rank 0
        sh.add_global(synth.rows(1000, 768, 9), normalize=True)
        q = torch.from_numpy(synth.rows(40, 768, 5)).cuda()
        D, I = sh.search_tensors(q, 10, normalize=True)                   # MFMA cascade on each shard
        D1, I1 = sh.search_tensors(q[:1], 10, normalize=True)             # single-query path
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, f"g{rank}.npz"), D=D.cpu().numpy(), I=I.cpu().numpy(), D1=D1.cpu().numpy(),
                 I1=I1.cpu().numpy(), sizes=np.array(sh.shard_sizes))
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_the_cpu_oracle(tmp_path):
    import torch.multiprocessing as mp

    from knn_checks import assert_topk_matches
    from oracle import knn_oracle as ko

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    x = np.concatenate([ko.synth_rows(300_000, 768, 4), ko.synth_rows(500, 768, 8), ko.synth_rows(1000, 768, 9)])
    ref = ko.FlatIndexOracle(768, 0)
    ref.add(ko.normalize_rows(x))
    qn = ko.normalize_rows(ko.synth_rows(40, 768, 5))
    Dr, Ir = ref.search(qn, 10)
    D64 = ref.rescore64(qn, Ir)
    for r in range(2):
        g = np.load(tmp_path / f"g{r}.npz")
        assert g["sizes"].tolist() == [150_000 + 500 + 500, 150_000 + 500]
        assert_topk_matches(g["D"], g["I"], Dr, Ir, D64, f"rank {r}")
        assert_topk_matches(g["D1"], g["I1"], Dr[:1], Ir[:1], D64[:1], f"rank {r} single query")


def _rank_masked(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from claude_semantic_search_amd import synth
        from claude_semantic_search_amd.sharded import ShardedFlatIndex

        torch.cuda.set_device(0)
        sh = ShardedFlatIndex(768, 0, device_index=0)
        sh.add_synthetic_global(200_000, seed=4, normalize=True)
        sh.add_routed(synth.rows(700, 768, 8), normalize=True)
        sh.add_global(synth.rows(1300, 768, 9), normalize=True)
        n = sh.ntotal_global
        allow = (np.arange(n) % 3) != 1
        q = synth.rows(40, 768, 5)
        D1, I1 = sh.search(q, 10, normalize=True, allow=allow)             # masked MFMA cascade on each shard
        sh.mark_deleted([int(i) for i in I1[:, 0]] + [5, 200_100, n - 1])
        D2, I2 = sh.search(q, 10, normalize=True, allow=allow)             # mask AND tombstones
        D3, I3 = sh.search(q[:2], 300, normalize=True)                     # tombstones only, k beyond one kernel pass
        rows = sh.reconstruct_n(199_990, 720)                              # spans the adds / both shards
        np.savez(os.path.join(out_dir, f"m{rank}.npz"), D1=D1, I1=I1, D2=D2, I2=I2, D3=D3, I3=I3, rows=rows)
    finally:
        dist.destroy_process_group()


def test_two_ranks_masks_tombstones_and_large_k_equal_the_cpu_oracle(tmp_path):
    import torch.multiprocessing as mp

    from knn_checks import assert_topk_matches
    from oracle import knn_oracle as ko

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_rank_masked, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    x = ko.normalize_rows(np.concatenate([ko.synth_rows(200_000, 768, 4), ko.synth_rows(700, 768, 8), ko.synth_rows(1300, 768, 9)]))
    n = x.shape[0]
    qn = ko.normalize_rows(ko.synth_rows(40, 768, 5))

    def oracle(live, k, nq):
        sub = np.flatnonzero(live)
        o = ko.FlatIndexOracle(768, 0)
        o.add(x[sub])
        D, I = o.search(qn[:nq], k)
        return D, sub[I], o.rescore64(qn[:nq], I)

    allow = (np.arange(n) % 3) != 1
    g0 = np.load(tmp_path / "m0.npz")
    Dr1, Ir1, D641 = oracle(allow, 10, 40)
    dead = np.zeros(n, dtype=bool)
    dead[[int(i) for i in g0["I1"][:, 0]] + [5, 200_100, n - 1]] = True
    Dr2, Ir2, D642 = oracle(allow & ~dead, 10, 40)
    Dr3, Ir3, D643 = oracle(~dead, 300, 2)
    for r in range(2):
        g = np.load(tmp_path / f"m{r}.npz")
        assert_topk_matches(g["D1"], g["I1"], Dr1, Ir1, D641, f"rank {r} masked")
        assert_topk_matches(g["D2"], g["I2"], Dr2, Ir2, D642, f"rank {r} masked + tombstones")
        assert_topk_matches(g["D3"], g["I3"], Dr3, Ir3, D643, f"rank {r} k = 300")
        assert not dead[g["I2"]].any() and allow[g["I2"]].all() and not dead[g["I3"]].any()
        assert np.allclose(g["rows"], x[199_990:200_710], rtol=0, atol=3e-7)


def test_flagged_query_count_is_reported():
    from claude_semantic_search_amd import synth
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    base = synth.rows(4000, 768, 71)
    x = np.concatenate([base[:2000], np.repeat(base[11:12], 5000, axis=0), base[2000:]], axis=0)
    ix = IndexFlatIP(768)
    ix.add(x, normalize=True)
    ix.set_search_mode("coarse")
    assert ix.last_flagged() == 0
    ix.search(np.concatenate([base[11:12] + 0.01 * synth.rows(9, 768, 72), synth.rows(30, 768, 73)]), 10, normalize=True)
    assert 9 <= ix.last_flagged() <= 12
    ix.search(synth.rows(30, 768, 74), 10, normalize=True)
    assert ix.last_flagged() <= 1
    ix.close()


def _nccl_single_rank(port, out_path):
    """The RCCL calls of the N > 1 path with ONE rank (all a one-GPU box allows: RCCL refuses two ranks on one
    device): init_process_group("nccl", device_id=...), barrier(device_ids=...), all_reduce of a device tensor, and the
    search's all_gather_into_tensor of a device uint8 record + the HIP merge that reads it."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import torch
    import torch.distributed as dist

    from claude_semantic_search_amd import synth
    from claude_semantic_search_amd.sharded import ShardedFlatIndex

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=dev)
    try:
        dist.barrier(device_ids=[0])
        t = torch.tensor([3.5], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == 3.5
        sh = ShardedFlatIndex(768, 0, device_index=0)
        sh.add_synthetic_global(30_000, seed=4, normalize=True, stream=torch.cuda.current_stream().cuda_stream)
        q = torch.from_numpy(synth.rows(9, 768, 5)).to(dev)
        D0, I0 = sh.search_tensors(q, 10, normalize=True)          # world 1: no exchange
        D0, I0 = D0.clone(), I0.clone()
        sh.exchange_when_single = True
        D1, I1 = sh.search_tensors(q, 10, normalize=True)          # packed all-gather over RCCL + packed merge
        torch.cuda.synchronize()
        assert torch.equal(I0, I1) and torch.equal(D0, D1)
        dist.barrier(device_ids=[0])
        open(out_path, "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_rccl_calls_of_the_sharded_path_run_with_one_rank(tmp_path):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "ok.txt"
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_nccl_single_rank, args=(port, str(out)))
    p.start()
    p.join(300)
    assert p.exitcode == 0 and out.read_text() == "ok"
