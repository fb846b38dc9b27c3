"""Row-partitioned flat index across the GPUs of one node.

One process per GPU (``torch.distributed``; backend "nccl" is RCCL on ROCm).  Every rank keeps its rows in its
own HBM as an ordinary ``IndexFlat``.  A search is: every rank sweeps its shard for the (replicated) query batch
-> ONE all-gather of the per-shard top-k, ids and scores packed into a single buffer of ``nq*k*12`` bytes per
rank (latency bound, nothing bulky ever crosses xGMI) -> every rank merges the ``G*k`` candidates per query by
(score, id).  There is no reference counterpart (the reference pins faiss to device 0, ``src/storage.py:283``);
semantics are those of one big ``IndexFlat`` and are tested as such.

Rows enter through collective calls (every rank passes the same arguments, no communication needed):

* ``add_global(x)`` / ``add_synthetic_global(n)``: the call's rows are split into ``world`` contiguous blocks,
  rank ``r`` keeps block ``r``;
* ``add_routed(x)``: an incremental add (a file's worth of chunks) goes whole to the least-full shard.

Global ids are insertion order over the calls, exactly as one ``IndexFlat`` would number them
(``src/storage.py:358-365``).  A shard therefore holds a list of segments ``(local_row0, global_row0, n)``; with one
segment the library's ``id_base`` does the translation inside the search kernels, with several the local ids are
mapped through the segment table after the local search.

The communication skeleton is backend agnostic (the CPU tests run it over gloo with test doubles for the device
pieces); the product wiring uses device buffers end to end.
"""
from __future__ import annotations

import ctypes
from typing import Callable, List, Optional, Tuple

import numpy as np


def shard_bounds(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced row partition: rows [lo, hi) of rank `rank`."""
    return rank * n_total // world, (rank + 1) * n_total // world


def packed_layout(nq: int, k: int) -> Tuple[int, int, int]:
    """Per-rank exchange record: ``[nq*k int64 ids][nq*k float32 scores]`` padded to 16 bytes.
    Returns (bytes of the id part, bytes of ids + scores, padded record size)."""
    n = nq * k
    return 8 * n, 12 * n, (12 * n + 15) // 16 * 16


class ShardedFlatIndex:
    """``IndexFlat`` semantics over ``world`` row shards.

    ``local_index`` must offer ``ntotal``, ``add(x, normalize=)``, ``add_synthetic``, ``set_id_base`` and
    ``search_dev``/``search``; ``merge`` merges ``[world, nq, k]`` candidate tensors.  Defaults are the HIP
    implementations; tests substitute doubles.
    """

    def __init__(self, d: int, metric: int = 0, group=None, device_index: Optional[int] = None,
                 index_factory: Optional[Callable] = None, merge: Optional[Callable] = None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.d, self.metric = int(d), int(metric)
        self.device_index = device_index
        if index_factory is None:
            from .flat_index import IndexFlat

            index_factory = lambda: IndexFlat(self.d, self.metric, device=device_index or 0)  # noqa: E731
        self.local = index_factory()
        self._merge = merge      # None: the HIP merge straight from the packed exchange buffer
        #: run the all-gather + merge even with ONE rank (tests: the RCCL calls of the N > 1 path on a one-GPU box)
        self.exchange_when_single = False
        self.ntotal_global = 0
        self.shard_sizes = [0] * self.world          # replicated bookkeeping (all adds are collective calls)
        self.segments: List[Tuple[int, int, int]] = []   # (local_row0, global_row0, n) of THIS shard
        self._seg_tensors = None
        self._dead: Optional[np.ndarray] = None          # tombstones in global numbering (replicated; mark_deleted)

    # -- building ----------------------------------------------------------
    def _append_segment(self, n_local: int, global_row0: int) -> None:
        if n_local <= 0:
            return
        local_row0 = self.shard_sizes[self.rank]
        if self.segments and self.segments[-1][0] + self.segments[-1][2] == local_row0 and \
                self.segments[-1][1] + self.segments[-1][2] == global_row0:
            l0, g0, n0 = self.segments[-1]
            self.segments[-1] = (l0, g0, n0 + n_local)       # contiguous in both numberings: one segment
        else:
            self.segments.append((local_row0, global_row0, n_local))
        # one segment: the kernels add id_base themselves; several: local ids, translated after the search
        self.local.set_id_base(self.segments[0][1] - self.segments[0][0] if len(self.segments) == 1 else 0)
        self._seg_tensors = None

    def _account(self, counts: List[int], n_call: int) -> None:
        for r, c in enumerate(counts):
            self.shard_sizes[r] += c
        self.ntotal_global += n_call

    def add_global(self, x: np.ndarray, normalize: bool = False) -> None:
        """Every rank passes the same full ``x``; rank r keeps the r-th contiguous block of the call's rows."""
        n = int(x.shape[0])
        bounds = [shard_bounds(n, self.world, r) for r in range(self.world)]
        lo, hi = bounds[self.rank]
        if hi > lo:
            self.local.add(np.ascontiguousarray(x[lo:hi]), normalize=normalize)
        self._append_segment(hi - lo, self.ntotal_global + lo)
        self._account([b[1] - b[0] for b in bounds], n)

    def add_routed(self, x: np.ndarray, normalize: bool = False) -> int:
        """Incremental add: the whole call goes to the least-full shard (lowest rank on ties).  Returns that rank."""
        n = int(x.shape[0])
        target = min(range(self.world), key=lambda r: (self.shard_sizes[r], r))
        if n and target == self.rank:
            self.local.add(np.ascontiguousarray(x), normalize=normalize)
            self._append_segment(n, self.ntotal_global)
        counts = [0] * self.world
        counts[target] = n
        self._account(counts, n)
        return target

    def add_synthetic_global(self, n_total: int, seed: int, normalize: bool = True, stream: int = 0) -> None:
        """Rows of the virtual synthetic index (``css_synth.h``, row r is the same vector wherever it lives)."""
        bounds = [shard_bounds(n_total, self.world, r) for r in range(self.world)]
        lo, hi = bounds[self.rank]
        if self.shard_sizes[self.rank] == 0:
            self.local.reserve(hi - lo)
        if hi > lo:
            self.local.add_synthetic(hi - lo, seed, first_row=lo, normalize=normalize, stream=stream)
        self._append_segment(hi - lo, self.ntotal_global + lo)
        self._account([b[1] - b[0] for b in bounds], n_total)

    # -- id translation ------------------------------------------------------
    def _to_global(self, I):
        """Local row numbers -> global ids through the segment table (only needed with several segments)."""
        import torch

        if len(self.segments) <= 1:
            return I
        if self._seg_tensors is None or self._seg_tensors[0].device != I.device:
            l0 = torch.tensor([s[0] for s in self.segments], dtype=torch.int64, device=I.device)
            g0 = torch.tensor([s[1] for s in self.segments], dtype=torch.int64, device=I.device)
            self._seg_tensors = (l0, g0)
        l0, g0 = self._seg_tensors
        seg = (torch.bucketize(I, l0, right=True) - 1).clamp_(min=0)
        return torch.where(I >= 0, I - l0[seg] + g0[seg], I)

    # -- searching ---------------------------------------------------------
    def _merge_packed_hip(self, recv, nq: int, k: int, record: int):
        import torch

        from . import _native as nat

        Dm = torch.empty((nq, k), dtype=torch.float32, device=recv.device)
        Im = torch.empty((nq, k), dtype=torch.int64, device=recv.device)
        st = torch.cuda.current_stream().cuda_stream
        nat.check(nat.lib().css_merge_topk_packed_dev(ctypes.c_void_p(recv.data_ptr()), self.world, record, nq, k,
                                                      self.metric, ctypes.c_void_p(Dm.data_ptr()),
                                                      ctypes.c_void_p(Im.data_ptr()), recv.device.index or 0,
                                                      ctypes.c_void_p(st)))
        return Dm, Im

    # -- allow-masks and tombstones (filter push-down, src/storage.py:438-492, :611-635) --------------------
    def local_rows_of(self, global_mask: np.ndarray) -> np.ndarray:
        """This shard's part of a per-row array in GLOBAL numbering (``ntotal_global`` entries, the same on every
        rank), in local row order -- through the segment table."""
        g = np.asarray(global_mask)
        if g.shape[0] != self.ntotal_global:
            raise ValueError(f"mask has {g.shape[0]} entries, the sharded index {self.ntotal_global} rows")
        out = np.zeros(self.shard_sizes[self.rank], dtype=g.dtype)
        for l0, g0, n in self.segments:
            out[l0:l0 + n] = g[g0:g0 + n]
        return out

    def mark_deleted(self, global_ids) -> None:
        """Tombstones: rows that no search may return any more (collective: every rank passes the same ids).  The rows
        stay in HBM, as in the reference, until the caller compacts (``HybridStorage._rebuild_faiss_index``)."""
        ids = np.asarray(list(global_ids), dtype=np.int64)
        if ids.size == 0:
            return
        if self._dead is None or self._dead.shape[0] < self.ntotal_global:
            grown = np.zeros(self.ntotal_global, dtype=bool)
            if self._dead is not None:
                grown[: self._dead.shape[0]] = self._dead
            self._dead = grown
        self._dead[ids] = True

    def _local_allow(self, allow) -> Optional[np.ndarray]:
        """Boolean mask over THIS shard's rows: the caller's global allow-mask minus the tombstones (None = every row)."""
        dead = None
        if self._dead is not None and self._dead.any():
            dead = np.zeros(self.ntotal_global, dtype=bool)
            dead[: self._dead.shape[0]] = self._dead
        if allow is None and dead is None:
            return None
        g = np.ones(self.ntotal_global, dtype=bool) if allow is None else np.asarray(allow, dtype=bool)
        if dead is not None:
            g = g & ~dead
        return self.local_rows_of(g)

    def search_tensors(self, q, k: int, normalize: bool = False, allow=None):
        """``q``: [nq, d] float32 tensor on this rank's device (same on all ranks).  ``allow``: optional boolean array
        over the GLOBAL rows (same on all ranks): only rows marked True can be returned; every rank cuts its shard's
        part out and hands it to the local masked search (``css_index_search_masked_dev``).
        Returns merged ``(D, I)`` tensors (global ids) on every rank."""
        import torch

        nq = q.shape[0]
        ib, db, record = packed_layout(nq, k)
        # the local result is written straight into this rank's exchange record: [ids | scores]
        send = torch.empty(record, dtype=torch.uint8, device=q.device)
        I = send[:ib].view(torch.int64).view(nq, k)
        D = send[ib:db].view(torch.float32).view(nq, k)
        loc = self._local_allow(allow)
        if q.is_cuda:
            st = torch.cuda.current_stream().cuda_stream
            bits = None
            if loc is not None and loc.shape[0]:
                from .flat_index import pack_allow_bits

                bits = torch.from_numpy(pack_allow_bits(loc, loc.shape[0]).view(np.int32)).to(q.device)
            self.local.search_dev(q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), st, normalize=normalize,
                                  allow_bits_ptr=bits.data_ptr() if bits is not None else 0)
            if bits is not None:
                bits.record_stream(torch.cuda.current_stream())
        else:  # CPU doubles (tests)
            if loc is not None:
                d_np, i_np = self.local.search(q.numpy(), k, normalize=normalize, allow=loc)
            else:
                d_np, i_np = self.local.search(q.numpy(), k, normalize=normalize)
            D.copy_(torch.from_numpy(d_np))
            I.copy_(torch.from_numpy(i_np))
        if len(self.segments) > 1:
            I.copy_(self._to_global(I))
        if self.world == 1 and not self.exchange_when_single:
            return D, I
        # THE exchange step of the path: one all-gather of nq*k*12 bytes per rank
        recv_flat = torch.empty(self.world * record, dtype=torch.uint8, device=q.device)
        if q.is_cuda and self.dist.get_backend(self.group) == "gloo":
            # rehearsal of the N > 1 path on a box without RCCL peers: gloo moves host memory
            host = torch.empty(self.world * record, dtype=torch.uint8)
            self.dist.all_gather_into_tensor(host, send.cpu(), group=self.group)
            recv_flat.copy_(host)
        else:
            self.dist.all_gather_into_tensor(recv_flat, send, group=self.group)
        recv = recv_flat.view(self.world, record)
        if self._merge is None and q.is_cuda:
            return self._merge_packed_hip(recv, nq, k, record)
        if self._merge is None:
            raise RuntimeError("ShardedFlatIndex: the merge of the gathered top-k lists runs on the HIP device "
                               "(css_merge_topk_packed_dev); queries on the CPU need merge=<callable> (test doubles only)")
        Ig = recv[:, :ib].view(torch.int64).view(self.world, nq, k)
        Dg = recv[:, ib:db].view(torch.float32).view(self.world, nq, k)
        return self._merge(Dg.contiguous(), Ig.contiguous(), k)

    def search(self, q: np.ndarray, k: int, normalize: bool = False, allow=None):
        import torch

        qt = torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32).reshape(-1, self.d))
        if self.device_index is not None and torch.cuda.is_available():
            qt = qt.to(f"cuda:{self.device_index}")
        D, I = self.search_tensors(qt, k, normalize, allow=allow)
        return D.cpu().numpy(), I.cpu().numpy()

    # -- rows back out (index files, compaction) ---------------------------------------------------------------
    def reconstruct_n(self, row0: int, n: int) -> np.ndarray:
        """Rows ``[row0, row0 + n)`` in GLOBAL numbering on every rank (collective).  Each row lives on exactly one
        shard: every rank fills in the rows it owns and ONE sum all-reduce of the zero-filled blocks completes them
        (save / backup / compaction paths only -- never on the search path)."""
        import torch

        out = np.zeros((int(n), self.d), dtype=np.float32)
        for l0, g0, m in self.segments:
            lo, hi = max(g0, row0), min(g0 + m, row0 + n)
            if hi > lo:
                out[lo - row0:hi - row0] = self.local.reconstruct_n(l0 + (lo - g0), hi - lo)
        if self.world > 1 and n:
            t = torch.from_numpy(out)
            if self.dist.get_backend(self.group) != "gloo":
                t = t.to(f"cuda:{self.device_index or 0}")
            self.dist.all_reduce(t, group=self.group)
            out = t.cpu().numpy()
        return out

    def add_file_rows(self, path: str, offset: int, n: int, normalize: bool = False, chunk_rows: int = 1 << 18) -> None:
        """``add_global`` of ``n`` fp32 rows stored row-major at byte ``offset`` of ``path`` (the payload of an index
        file): every rank reads only its own block (collective call, no communication)."""
        bounds = [shard_bounds(n, self.world, r) for r in range(self.world)]
        lo, hi = bounds[self.rank]
        if hi > lo:
            if self.shard_sizes[self.rank] == 0:
                self.local.reserve(hi - lo)
            with open(path, "rb") as f:
                for r0 in range(lo, hi, chunk_rows):
                    m = min(chunk_rows, hi - r0)
                    f.seek(offset + r0 * self.d * 4)
                    buf = f.read(m * self.d * 4)
                    if len(buf) != m * self.d * 4:
                        raise RuntimeError("truncated index file (in rows)")
                    self.local.add(np.frombuffer(buf, dtype=np.float32).reshape(m, self.d), normalize=normalize)
        self._append_segment(hi - lo, self.ntotal_global + lo)
        self._account([b[1] - b[0] for b in bounds], n)


class ShardedIndexFacade:
    """The members of ``flat_index.IndexFlat`` that ``HybridStorage`` touches, over a ``ShardedFlatIndex``: the
    reference's ``Storage.search()`` behind 1..8 GPUs (``StorageConfig.sharded``).  SPMD: every rank of the process
    group runs the same ``HybridStorage`` calls with the same arguments (adds, searches, deletes, saves); an ``add`` of
    a file's chunks goes whole to the least-full shard (``add_routed``), a search is the local masked search + ONE
    all-gather + merge, and ids are those of one ``IndexFlat`` that received the same calls."""

    def __init__(self, d: int, metric: int = 0, device: int = 0, group=None, index_factory=None, merge=None):
        self.sh = ShardedFlatIndex(d, metric, group=group, device_index=device, index_factory=index_factory, merge=merge)
        self.d, self.metric_type, self.device, self.is_trained = int(d), int(metric), int(device), True

    ntotal = property(lambda self: self.sh.ntotal_global)

    def reserve(self, n: int) -> None:   # (shards grow on their own; a global reserve would over-allocate every shard)
        pass

    def add(self, x, normalize: bool = False) -> None:
        a = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, self.d)
        if a.shape[0]:
            self.sh.add_routed(a, normalize=normalize)

    def search(self, q, k: int, normalize: bool = False, allow=None):
        return self.sh.search(np.asarray(q, dtype=np.float32), int(k), normalize=normalize, allow=allow)

    def reconstruct_n(self, row0: int = 0, n: Optional[int] = None) -> np.ndarray:
        return self.sh.reconstruct_n(int(row0), self.ntotal - int(row0) if n is None else int(n))

    def reconstruct(self, i: int) -> np.ndarray:
        return self.reconstruct_n(int(i), 1)[0]

    def mark_deleted(self, ids) -> None:
        self.sh.mark_deleted(ids)

    def set_search_mode(self, mode: str) -> None:
        self.sh.local.set_search_mode(mode)

    def close(self) -> None:
        self.sh.local.close()


def read_index_sharded(path: str, device: int = 0, group=None, index_factory=None, merge=None) -> ShardedIndexFacade:
    """``flat_index.read_index`` for a shard group: the header is validated by the same code, then every rank reads
    its own contiguous block of the rows (global ids = file order, as ``faiss.read_index`` numbers them)."""
    from . import flat_index as fi

    d, n, metric, offset = fi.read_index_header(path)
    ix = ShardedIndexFacade(d, metric, device=device, group=group, index_factory=index_factory, merge=merge)
    if n:
        ix.sh.add_file_rows(path, offset, n)
    return ix
