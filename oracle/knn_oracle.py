"""TEST INFRASTRUCTURE, NOT PRODUCT CODE -- CPU oracle for the flat exact kNN.

Python face of ``oracle/knn_oracle.c`` (ctypes) plus a numpy/BLAS restatement of
faiss-cpu's batched path.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module; nothing under
``claude_semantic_search_amd/`` does.

What is restated (reference file:line, relative to the reference checkout):

* ``HybridStorage.add_chunks`` numeric part, ``src/storage.py:343-359``:
  ``x = np.array(.., float32)``; ``x / (||x|| + 1e-8)`` per row; ids are
  ``ntotal .. ntotal+n-1``.
* ``HybridStorage.search`` numeric part, ``src/storage.py:424-436``:
  ``q / (||q|| + 1e-8)``; ``reshape(1,-1).astype(float32)``;
  ``k' = min(max_results, ntotal)``; ``index.search(q, k')``.
* ``faiss.IndexFlatIP`` / ``IndexFlatL2`` (third party, faiss-cpu>=1.11.0 per the
  reference's ``pyproject.toml:9``; not installed, restated from its published
  semantics -- SURVEY.md App. B): exact fp32 inner product, descending; exact
  squared L2, ascending; int64 ids; ``-1`` padding.

Pinned by the reference's own known-answer tests via
``tests/golden/knn_reference_cases.json`` (see ``tests/test_oracle_knn.py``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

METRIC_IP = 0
METRIC_L2 = 1

_HERE = Path(__file__).resolve().parent
_LIB = None


def _lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = _HERE / "_build" / "libknn_oracle.so"
    if not so.exists():
        subprocess.run(["make", "-C", str(_HERE)], check=True, capture_output=True)
    lib = ctypes.CDLL(str(so))
    f32p = ctypes.POINTER(ctypes.c_float)
    i64p = ctypes.POINTER(ctypes.c_int64)
    f64p = ctypes.POINTER(ctypes.c_double)
    lib.knn_oracle_normalize_rows.argtypes = [f32p, ctypes.c_int64, ctypes.c_int]
    lib.knn_oracle_normalize_rows.restype = None
    lib.knn_oracle_search.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, f32p, ctypes.c_int64,
                                      ctypes.c_int, ctypes.c_int, f32p, i64p]
    lib.knn_oracle_search.restype = None
    lib.knn_oracle_rescore64.argtypes = [f32p, ctypes.c_int, f32p, ctypes.c_int64, ctypes.c_int,
                                         ctypes.c_int, i64p, f64p]
    lib.knn_oracle_rescore64.restype = None
    lib.knn_oracle_merge.argtypes = [f32p, i64p, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                     ctypes.c_int, f32p, i64p]
    lib.knn_oracle_merge.restype = None
    lib.knn_oracle_fold_block.argtypes = [f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                          ctypes.c_int, f32p, i64p]
    lib.knn_oracle_fold_block.restype = None
    lib.knn_oracle_num_threads.restype = ctypes.c_int
    lib.knn_oracle_set_threads.argtypes = [ctypes.c_int]
    lib.knn_oracle_synth_rows.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_uint64, ctypes.c_int64]
    lib.knn_oracle_synth_rows.restype = None
    _LIB = lib
    return lib


def _f32(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i64(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


def num_threads() -> int:
    return int(_lib().knn_oracle_num_threads())


def set_threads(t: int) -> None:
    _lib().knn_oracle_set_threads(int(t))


def normalize_rows(x: np.ndarray) -> np.ndarray:
    """``x / (||x||_2 + 1e-8)`` row-wise in fp32 (``src/storage.py:347-350``)."""
    x = np.array(x, dtype=np.float32, order="C", copy=True)
    if x.ndim == 1:
        x = x.reshape(1, -1)
    _lib().knn_oracle_normalize_rows(_f32(x), x.shape[0], x.shape[1])
    return x


def normalize_rows_numpy(x: np.ndarray) -> np.ndarray:
    """Literal numpy form of the reference lines, for cross-checking the C one."""
    x = np.array(x, dtype=np.float32)
    norms = np.linalg.norm(x, axis=1, keepdims=True)
    return (x / (norms + 1e-8)).astype(np.float32)


def synth_rows(n: int, d: int, seed: int, first_row: int = 0) -> np.ndarray:
    x = np.empty((n, d), dtype=np.float32)
    _lib().knn_oracle_synth_rows(_f32(x), n, d, ctypes.c_uint64(seed), first_row)
    return x


class FlatIndexOracle:
    """Duck-types the members of ``faiss.IndexFlatIP/L2`` the reference touches
    (``.d``, ``.ntotal``, ``.add``, ``.search``, ``.reset``)."""

    def __init__(self, d: int, metric: int = METRIC_IP):
        self.d = int(d)
        self.metric = int(metric)
        self._xb = np.zeros((0, self.d), dtype=np.float32)

    @property
    def ntotal(self) -> int:
        return int(self._xb.shape[0])

    def reset(self) -> None:
        self._xb = np.zeros((0, self.d), dtype=np.float32)

    def add(self, x: np.ndarray) -> None:
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, self.d)
        self._xb = np.ascontiguousarray(np.concatenate([self._xb, x], axis=0))

    def search(self, q: np.ndarray, k: int):
        q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1, self.d)
        nq = q.shape[0]
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        _lib().knn_oracle_search(_f32(self._xb), self.ntotal, self.d, _f32(q), nq, int(k),
                                 self.metric, _f32(D), _i64(I))
        return D, I

    def rescore64(self, q: np.ndarray, I: np.ndarray) -> np.ndarray:
        q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1, self.d)
        I = np.ascontiguousarray(I, dtype=np.int64)
        D64 = np.empty(I.shape, dtype=np.float64)
        _lib().knn_oracle_rescore64(_f32(self._xb), self.d, _f32(q), q.shape[0], I.shape[1],
                                    self.metric, _i64(I), D64.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        return D64


def merge_topk(D_parts: np.ndarray, I_parts: np.ndarray, metric: int = METRIC_IP):
    """[nparts, nq, k] per-shard lists -> global top-k (SURVEY.md 8e exchange)."""
    D_parts = np.ascontiguousarray(D_parts, dtype=np.float32)
    I_parts = np.ascontiguousarray(I_parts, dtype=np.int64)
    nparts, nq, k = D_parts.shape
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    _lib().knn_oracle_merge(_f32(D_parts), _i64(I_parts), nparts, nq, k, metric, _f32(D), _i64(I))
    return D, I


def search_blas(xb: np.ndarray, q: np.ndarray, k: int, metric: int = METRIC_IP, block: int = 16384):
    """faiss-cpu's batched path (nq >= 20): blocked SGEMM, then every score of the block is offered to the
    query's result heap.  The SGEMM is numpy's (the BLAS numpy links, all cores); the heap phase is
    ``knn_oracle_fold_block`` (OpenMP over queries, one compare per score in the common case).  Used by
    bench.py as the ``cpu_baseline`` of kind "port" for query batches."""
    xb = np.ascontiguousarray(xb, dtype=np.float32)
    q = np.ascontiguousarray(q, dtype=np.float32)
    nq = q.shape[0]
    n = xb.shape[0]
    D = np.full((nq, k), -np.finfo(np.float32).max if metric == METRIC_IP else np.finfo(np.float32).max, np.float32)
    I = np.full((nq, k), -1, np.int64)
    qn = (q * q).sum(1)[:, None] if metric == METRIC_L2 else None
    lib = _lib()
    for r0 in range(0, n, block):
        blk = xb[r0:r0 + block]
        s = q @ blk.T
        if metric == METRIC_L2:
            s = qn + (blk * blk).sum(1)[None, :] - 2.0 * s
            np.maximum(s, 0, out=s)
        s = np.ascontiguousarray(s, dtype=np.float32)
        lib.knn_oracle_fold_block(_f32(s), nq, blk.shape[0], r0, int(k), metric, _f32(D), _i64(I))
    return D, I


def search_synth_chunked(n_total: int, d: int, seed: int, q: np.ndarray, k: int, metric: int = METRIC_IP,
                         normalize: bool = True, chunk: int = 1_000_000, first_row: int = 0):
    """Exact top-k over a SYNTHETIC index too large to hold: rows ``first_row .. first_row + n_total`` of the
    ``css_synth.h`` generator are regenerated ``chunk`` rows at a time (optionally row-normalised as
    ``add_chunks`` does), searched with the plain C sweep, and the per-chunk lists merged with
    ``knn_oracle_merge``.  Returns ``(D, I, D64)`` with global ids and the fp64 scores of the returned rows."""
    q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1, d)
    nq = q.shape[0]
    Dp, Ip, D64p = [], [], []
    lib = _lib()
    for r0 in range(0, n_total, chunk):
        m = min(chunk, n_total - r0)
        x = synth_rows(m, d, seed, first_row=first_row + r0)
        if normalize:
            lib.knn_oracle_normalize_rows(_f32(x), m, d)
        ref = FlatIndexOracle(d, metric)
        ref._xb = x
        kk = min(k, m)
        Dc, Ic = ref.search(q, kk)
        D64c = ref.rescore64(q, Ic)
        if kk < k:
            padv = -np.finfo(np.float32).max if metric == METRIC_IP else np.finfo(np.float32).max
            Dc = np.concatenate([Dc, np.full((nq, k - kk), padv, np.float32)], axis=1)
            Ic = np.concatenate([Ic, np.full((nq, k - kk), -1, np.int64)], axis=1)
            D64c = np.concatenate([D64c, np.full((nq, k - kk), float(padv))], axis=1)
        Dp.append(Dc)
        Ip.append(np.where(Ic >= 0, Ic + r0, -1))
        D64p.append(D64c)
        del ref, x
    Dp, Ip, D64p = np.stack(Dp), np.stack(Ip), np.stack(D64p)
    D, I = merge_topk(Dp, Ip, metric)
    D64 = np.empty(I.shape, dtype=np.float64)
    for r in range(nq):
        lut = {int(i): float(v) for i, v in zip(Ip[:, r].ravel(), D64p[:, r].ravel()) if i >= 0}
        D64[r] = [lut.get(int(i), np.nan) for i in I[r]]
    return D, I, D64
