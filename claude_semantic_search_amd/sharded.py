"""Row-partitioned flat index across the GPUs of one node.

One process per GPU (``torch.distributed``; backend "nccl" is RCCL on ROCm).
Rank ``r`` of ``G`` owns the contiguous global rows ``[r*N/G, (r+1)*N/G)`` in its
own HBM as an ordinary ``IndexFlat`` with ``id_base`` = first global row.  A
search is: every rank sweeps its shard for the (replicated) query batch -> ONE
all-gather of the per-shard top-k (``nq*k*12`` bytes per rank, latency bound,
nothing bulky ever crosses xGMI) -> every rank merges the ``G*k`` candidates per
query by (score, id).  There is no reference counterpart (the reference pins
faiss to device 0, ``src/storage.py:283``); semantics are those of one big
``IndexFlat`` and are tested as such.

The communication skeleton is backend agnostic (the CPU tests run it over gloo
with test doubles for the device pieces); the product wiring uses device
buffers end to end.
"""
from __future__ import annotations

import ctypes
from typing import Callable, Optional, Tuple

import numpy as np


def shard_bounds(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced row partition: rows [lo, hi) of rank `rank`."""
    return rank * n_total // world, (rank + 1) * n_total // world


class ShardedFlatIndex:
    """``IndexFlat`` semantics over ``world`` row shards.

    ``local_index`` must offer ``ntotal``, ``add(x, normalize=)``,
    ``add_synthetic``, ``set_id_base`` and ``search_dev``/``search``;
    ``merge`` merges ``[world, nq, k]`` candidate tensors.  Defaults are the HIP
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
        self._merge = merge or self._merge_hip
        self.ntotal_global = 0

    # -- building ----------------------------------------------------------
    def add_global(self, x: np.ndarray, normalize: bool = False) -> None:
        """Every rank passes the same full ``x``; each keeps its own row block."""
        lo, hi = shard_bounds(x.shape[0], self.world, self.rank)
        if self.local.ntotal == 0:
            self.local.set_id_base(self.ntotal_global + lo)
        self.local.add(np.ascontiguousarray(x[lo:hi]), normalize=normalize)
        self.ntotal_global += x.shape[0]

    def add_synthetic_global(self, n_total: int, seed: int, normalize: bool = True, stream: int = 0) -> None:
        lo, hi = shard_bounds(n_total, self.world, self.rank)
        self.local.reserve(hi - lo)
        self.local.add_synthetic(hi - lo, seed, first_row=lo, normalize=normalize, stream=stream)
        self.local.set_id_base(lo)
        self.ntotal_global = n_total

    # -- searching ---------------------------------------------------------
    def _merge_hip(self, Dg, Ig, k: int):
        import torch

        from . import _native as nat

        nq = Dg.shape[1]
        Dm = torch.empty((nq, k), dtype=torch.float32, device=Dg.device)
        Im = torch.empty((nq, k), dtype=torch.int64, device=Dg.device)
        st = torch.cuda.current_stream().cuda_stream
        nat.check(nat.lib().css_merge_topk_dev(ctypes.c_void_p(Dg.data_ptr()), ctypes.c_void_p(Ig.data_ptr()),
                                               self.world, nq, k, self.metric, ctypes.c_void_p(Dm.data_ptr()),
                                               ctypes.c_void_p(Im.data_ptr()), Dg.device.index or 0,
                                               ctypes.c_void_p(st)))
        return Dm, Im

    def search_tensors(self, q, k: int, normalize: bool = False):
        """``q``: [nq, d] float32 tensor on this rank's device (same on all ranks).
        Returns merged ``(D, I)`` tensors (global ids) on every rank."""
        import torch

        nq = q.shape[0]
        D = torch.empty((nq, k), dtype=torch.float32, device=q.device)
        I = torch.empty((nq, k), dtype=torch.int64, device=q.device)
        if q.is_cuda:
            st = torch.cuda.current_stream().cuda_stream
            self.local.search_dev(q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), st, normalize=normalize)
        else:  # CPU doubles (tests)
            d_np, i_np = self.local.search(q.numpy(), k, normalize=normalize)
            D.copy_(torch.from_numpy(d_np))
            I.copy_(torch.from_numpy(i_np))
        if self.world == 1:
            return D, I
        Dg = torch.empty((self.world, nq, k), dtype=torch.float32, device=q.device)
        Ig = torch.empty((self.world, nq, k), dtype=torch.int64, device=q.device)
        # the single exchange step of the path: per-shard top-k, 12 bytes per (query, slot)
        # (concatenated [world*nq, k] view: the only output form every backend accepts)
        self.dist.all_gather_into_tensor(Dg.view(self.world * nq, k), D, group=self.group)
        self.dist.all_gather_into_tensor(Ig.view(self.world * nq, k), I, group=self.group)
        return self._merge(Dg, Ig, k)

    def search(self, q: np.ndarray, k: int, normalize: bool = False):
        import torch

        qt = torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32).reshape(-1, self.d))
        if self.device_index is not None and torch.cuda.is_available():
            qt = qt.to(f"cuda:{self.device_index}")
        D, I = self.search_tensors(qt, k, normalize)
        return D.cpu().numpy(), I.cpu().numpy()
