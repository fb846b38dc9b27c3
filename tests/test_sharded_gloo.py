"""CPU, world_size 2 over gloo: the multi-GPU search skeleton (row partition ->
local top-k -> one packed all-gather -> merge) equals one big index, also after several adds of every kind.  Device pieces are
replaced by oracle-backed doubles defined here; the product wiring is covered on
the GPU by tests/test_knn_gpu.py::test_id_base_and_merge_parts_match_whole."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _FakeLocal:
    def __init__(self, d, metric):
        from oracle import knn_oracle as ko

        self.ko, self.d, self.metric, self.base = ko, d, metric, 0
        self.o = ko.FlatIndexOracle(d, metric)

    ntotal = property(lambda self: self.o.ntotal)

    def set_id_base(self, b):
        self.base = int(b)

    def reserve(self, n):
        pass

    def close(self):
        pass

    def add(self, x, normalize=False):
        self.o.add(self.ko.normalize_rows(x) if normalize else x)

    def add_synthetic(self, n, seed, first_row=0, normalize=True, stream=0):
        self.add(self.ko.synth_rows(n, self.d, seed, first_row), normalize)

    def search(self, q, k, normalize=False, allow=None):
        qn = self.ko.normalize_rows(q) if normalize else q
        if allow is None:
            D, I = self.o.search(qn, k)
            return D, np.where(I >= 0, I + self.base, -1)
        sub = np.flatnonzero(np.asarray(allow, dtype=bool))      # masked search = the oracle over the allowed rows
        o = self.ko.FlatIndexOracle(self.d, self.metric)
        if sub.size:
            o.add(self.o._xb[sub])
        D, I = o.search(qn, k)
        return D, np.where(I >= 0, sub[np.clip(I, 0, max(sub.size - 1, 0))] + self.base if sub.size else -1, -1)

    def reconstruct_n(self, row0, n):
        return self.o._xb[row0:row0 + n].copy()


def _merge(metric):
    def f(Dg, Ig, k):
        from oracle import knn_oracle as ko

        D, I = ko.merge_topk(Dg.numpy(), Ig.numpy(), metric)
        return torch.from_numpy(D), torch.from_numpy(I)
    return f


def _worker(rank, world, port, metric, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from claude_semantic_search_amd import synth
        from claude_semantic_search_amd.sharded import ShardedFlatIndex, shard_bounds

        d, n, nq, k = 64, 3001, 9, 10
        sh = ShardedFlatIndex(d, metric, index_factory=lambda: _FakeLocal(d, metric), merge=_merge(metric))
        assert (sh.rank, sh.world) == (rank, world)
        sh.add_synthetic_global(n, seed=4, normalize=(metric == 0))
        lo, hi = shard_bounds(n, world, rank)
        assert sh.local.ntotal == hi - lo and sh.ntotal_global == n
        q = synth.rows(nq, d, 5)
        D, I = sh.search(q, k, normalize=(metric == 0))
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), D=D, I=I)
    finally:
        dist.destroy_process_group()


def _worker_incremental(rank, world, port, out_dir):
    """Several adds of each kind: global ids must be those of ONE index that received the same calls in order."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from claude_semantic_search_amd import synth
        from claude_semantic_search_amd.sharded import ShardedFlatIndex

        d, k = 64, 10
        sh = ShardedFlatIndex(d, 0, index_factory=lambda: _FakeLocal(d, 0), merge=_merge(0))
        sh.add_global(synth.rows(10, d, 31), normalize=True)          # rows 0..9:   5 + 5
        sh.add_global(synth.rows(10, d, 32), normalize=True)          # rows 10..19: 5 + 5 (second segment per shard)
        t1 = sh.add_routed(synth.rows(7, d, 33), normalize=True)      # rows 20..26 -> rank 0 (tie: lowest rank)
        t2 = sh.add_routed(synth.rows(3, d, 34), normalize=True)      # rows 27..29 -> rank 1 (least full)
        sh.add_synthetic_global(401, seed=35, normalize=True)         # rows 30..430
        t3 = sh.add_routed(synth.rows(2, d, 36), normalize=True)      # rows 431..432
        assert (t1, t2) == (0, 1) and t3 in (0, 1)
        assert sh.ntotal_global == 433 and sum(sh.shard_sizes) == 433 and sh.local.ntotal == sh.shard_sizes[rank]
        assert len(sh.segments) >= 3
        q = synth.rows(21, d, 37)
        D, I = sh.search(q, k, normalize=True)
        np.savez(os.path.join(out_dir, f"inc{rank}.npz"), D=D, I=I)
    finally:
        dist.destroy_process_group()


def _worker_masked(rank, world, port, out_dir):
    """Allow-masks and tombstones through the shard group (global numbering, several segments per shard), k beyond one
    kernel pass, and rows read back out of the shards."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from claude_semantic_search_amd import synth
        from claude_semantic_search_amd.sharded import ShardedFlatIndex

        d = 64
        sh = ShardedFlatIndex(d, 0, index_factory=lambda: _FakeLocal(d, 0), merge=_merge(0))
        sh.add_global(synth.rows(300, d, 51), normalize=True)
        sh.add_routed(synth.rows(41, d, 52), normalize=True)
        sh.add_global(synth.rows(200, d, 53), normalize=True)
        n = sh.ntotal_global
        allow = (np.arange(n) % 3) != 1
        q = synth.rows(6, d, 54)
        D1, I1 = sh.search(q, 10, normalize=True, allow=allow)
        sh.mark_deleted([int(i) for i in I1[:, 0]] + [5, 340, n - 1])          # the best hit of every query + rows of every add
        D2, I2 = sh.search(q, 10, normalize=True, allow=allow)                # mask AND tombstones
        D3, I3 = sh.search(q, 300, normalize=True)                            # tombstones only, k beyond 128
        rows = sh.reconstruct_n(290, 70)                                       # spans all three adds / both shards
        np.savez(os.path.join(out_dir, f"m{rank}.npz"), D1=D1, I1=I1, D2=D2, I2=I2, D3=D3, I3=I3, rows=rows)
    finally:
        dist.destroy_process_group()


def test_two_rank_masks_tombstones_large_k_and_row_export(tmp_path):
    from oracle import knn_oracle as ko

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker_masked, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    d = 64
    x = ko.normalize_rows(np.concatenate([ko.synth_rows(300, d, 51), ko.synth_rows(41, d, 52), ko.synth_rows(200, d, 53)]))
    n = x.shape[0]
    qn = ko.normalize_rows(ko.synth_rows(6, d, 54))

    def oracle(live, k):
        sub = np.flatnonzero(live)
        o = ko.FlatIndexOracle(d, 0)
        o.add(x[sub])
        D, I = o.search(qn, k)
        return D, np.where(I >= 0, sub[np.clip(I, 0, sub.size - 1)], -1)

    allow = (np.arange(n) % 3) != 1
    Dr1, Ir1 = oracle(allow, 10)
    dead = np.zeros(n, dtype=bool)
    dead[[int(i) for i in Ir1[:, 0]] + [5, 340, n - 1]] = True
    Dr2, Ir2 = oracle(allow & ~dead, 10)
    Dr3, Ir3 = oracle(~dead, 300)
    for r in range(2):
        g = np.load(tmp_path / f"m{r}.npz")
        for a, b in ((g["I1"], Ir1), (g["I2"], Ir2), (g["I3"], Ir3), (g["D1"], Dr1), (g["D2"], Dr2), (g["D3"], Dr3)):
            assert np.array_equal(a, b), f"rank {r}"
        assert np.array_equal(g["rows"], x[290:360])


def test_two_rank_incremental_adds_number_rows_like_one_index(tmp_path):
    from oracle import knn_oracle as ko

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker_incremental, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    d, k = 64, 10
    parts = [ko.synth_rows(10, d, 31), ko.synth_rows(10, d, 32), ko.synth_rows(7, d, 33), ko.synth_rows(3, d, 34),
             ko.synth_rows(401, d, 35), ko.synth_rows(2, d, 36)]
    ref = ko.FlatIndexOracle(d, 0)
    ref.add(ko.normalize_rows(np.concatenate(parts)))
    Dr, Ir = ref.search(ko.normalize_rows(ko.synth_rows(21, d, 37)), k)
    for r in range(2):
        got = np.load(tmp_path / f"inc{r}.npz")
        assert np.array_equal(got["I"], Ir), f"rank {r}"
        assert np.array_equal(got["D"], Dr), f"rank {r}"
        assert len(set(got["I"].ravel().tolist())) > k     # (ids are not all from one shard)


@pytest.mark.parametrize("metric", [0, 1])
def test_two_rank_sharded_search_equals_single_index(tmp_path, metric):
    from oracle import knn_oracle as ko

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, metric, str(tmp_path)), nprocs=2, join=True)
    d, n, nq, k = 64, 3001, 9, 10
    x = ko.synth_rows(n, d, 4)
    q = ko.synth_rows(nq, d, 5)
    ref = ko.FlatIndexOracle(d, metric)
    if metric == 0:
        x, q = ko.normalize_rows(x), ko.normalize_rows(q)
    ref.add(x)
    Dr, Ir = ref.search(q, k)
    for r in range(2):
        got = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(got["I"], Ir), f"rank {r}"
        assert np.array_equal(got["D"], Dr), f"rank {r}"


def test_shard_bounds_cover_and_balance():
    from claude_semantic_search_amd.sharded import shard_bounds

    for n in (0, 1, 7, 8, 10_000_000, 80_000_001):
        for w in (1, 2, 4, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
