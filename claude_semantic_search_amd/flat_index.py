"""Flat exact index on the MI355X, duck-typing what the reference uses of faiss.

The reference touches exactly these faiss members: the classes
``IndexFlatIP`` / ``IndexFlatL2`` (``src/storage.py:256-258``), ``index.ntotal``
(``:358``, ``:421``), ``index.add(x)`` (``:359``), ``index.search(q, k)``
(``:436``), ``faiss.read_index`` / ``write_index`` (``:306``, ``:879-884``,
``:895``, ``:913``) and the device toggles ``index_cpu_to_gpu`` /
``index_gpu_to_cpu`` / ``StandardGpuResources`` / ``get_num_gpus`` (``:274-296``,
``src/gpu_utils.py:117-118``).  This module provides the same names over
``libcss_hip.so``; the index always lives in HBM (there is no CPU index).
"""
from __future__ import annotations

import ctypes
import struct
from typing import Optional, Tuple

import numpy as np

from . import _native as nat

METRIC_INNER_PRODUCT = nat.METRIC_IP
METRIC_L2 = nat.METRIC_L2


def _as_f32_2d(x, d: int, what: str) -> np.ndarray:
    a = np.ascontiguousarray(x, dtype=np.float32)
    if a.ndim == 1:
        a = a.reshape(1, -1)
    if a.ndim != 2 or a.shape[1] != d:
        raise ValueError(f"{what}: expected shape (n, {d}), got {tuple(np.shape(x))}")
    return a


MAX_K = nat.MAX_K


def pack_allow_bits(allow, ntotal: int) -> np.ndarray:
    """Boolean row mask -> the uint32 bitmap of ``css_index_search_masked`` (bit r & 31 of word r >> 5)."""
    m = np.asarray(allow)
    if m.dtype != np.bool_ or m.ndim != 1 or m.shape[0] != ntotal:
        raise ValueError(f"allow must be a boolean array of ntotal={ntotal} entries")
    by = np.packbits(m, bitorder="little")
    words = (ntotal + 31) // 32
    out = np.zeros(words * 4, dtype=np.uint8)
    out[: by.shape[0]] = by
    return out.view("<u4")


class IndexFlat:
    """Exact brute-force index in HBM (``faiss.IndexFlat`` semantics)."""

    def __init__(self, d: int, metric: int = METRIC_INNER_PRODUCT, device: int = 0):
        self.d = int(d)
        self.metric_type = int(metric)
        self.device = int(device)
        self.is_trained = True
        h = ctypes.c_void_p()
        nat.check(nat.lib().css_index_create(self.d, self.metric_type, self.device, ctypes.byref(h)))
        self._h: Optional[ctypes.c_void_p] = h

    # -- lifetime ---------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) is not None:
            nat.lib().css_index_free(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover - interpreter shutdown order
        try:
            self.close()
        except Exception:
            pass

    def _handle(self):
        if self._h is None:
            raise RuntimeError("index has been freed")
        return self._h

    # -- faiss surface ----------------------------------------------------
    @property
    def ntotal(self) -> int:
        n = ctypes.c_int64(0)
        nat.check(nat.lib().css_index_ntotal(self._handle(), ctypes.byref(n)))
        return int(n.value)

    def reset(self) -> None:
        nat.check(nat.lib().css_index_reset(self._handle()))

    def reserve(self, n: int) -> None:
        nat.check(nat.lib().css_index_reserve(self._handle(), int(n)))

    def add(self, x, normalize: bool = False) -> None:
        """Append rows; ids are ``ntotal .. ntotal+n-1`` (``src/storage.py:358-365``).
        ``normalize=True`` fuses the reference's ``x / (||x|| + 1e-8)``
        (``src/storage.py:347-350``) into the ingest kernel."""
        a = _as_f32_2d(x, self.d, "add")
        if a.shape[0] == 0:
            return
        nat.check(nat.lib().css_index_add(self._handle(), a.ctypes.data, a.shape[0], 1 if normalize else 0))

    def add_dev(self, x_ptr: int, n: int, normalize: bool = False, stream: int = 0) -> None:
        """Device-pointer twin of ``add``: ``x_ptr`` = device address of ``[n, d]`` fp32 rows (``tensor.data_ptr()``),
        enqueued on ``stream``; later searches are ordered behind it by the library."""
        if n:
            nat.check(nat.lib().css_index_add_dev(self._handle(), ctypes.c_void_p(x_ptr), int(n), 1 if normalize else 0,
                                                  ctypes.c_void_p(stream)))

    def add_synthetic(self, n: int, seed: int, first_row: int = 0, normalize: bool = True, stream: int = 0) -> None:
        nat.check(nat.lib().css_index_add_synthetic(self._handle(), int(n), ctypes.c_uint64(seed), int(first_row),
                                                    1 if normalize else 0, ctypes.c_void_p(stream)))

    def search(self, q, k: int, normalize: bool = False, allow=None) -> Tuple[np.ndarray, np.ndarray]:
        """``(D[nq,k] float32, I[nq,k] int64)``; IP descending, L2 ascending squared
        distances, ``-1`` padded (``src/storage.py:436``).  ``allow`` (optional boolean array of
        ``ntotal`` entries) restricts the answer to the rows marked True -- the filter / tombstone
        push-down the reference approximates by over-fetching (``src/storage.py:438-492``)."""
        a = _as_f32_2d(q, self.d, "search")
        k = int(k)
        if k < 1 or k > nat.MAX_K:
            raise ValueError(f"k={k} outside [1, {nat.MAX_K}]")
        nq = a.shape[0]
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        bits = None
        if allow is not None:
            bits = pack_allow_bits(allow, self.ntotal)
        if nq:
            nat.check(nat.lib().css_index_search_masked(self._handle(), a.ctypes.data, nq, k, 1 if normalize else 0,
                                                        bits.ctypes.data if bits is not None else None,
                                                        D.ctypes.data, I.ctypes.data))
        return D, I

    def search_dev(self, q_ptr: int, nq: int, k: int, D_ptr: int, I_ptr: int, stream: int = 0,
                   normalize: bool = False, allow_bits_ptr: int = 0) -> None:
        """Device-pointer twin: ``q_ptr``/``D_ptr``/``I_ptr`` are device addresses
        (e.g. ``tensor.data_ptr()``), enqueued on ``stream`` (a ``hipStream_t``).
        ``allow_bits_ptr``: optional device bitmap (``pack_allow_bits`` layout) of a masked search."""
        nat.check(nat.lib().css_index_search_masked_dev(self._handle(), ctypes.c_void_p(q_ptr), int(nq), int(k),
                                                        1 if normalize else 0,
                                                        ctypes.c_void_p(allow_bits_ptr) if allow_bits_ptr else None,
                                                        ctypes.c_void_p(D_ptr), ctypes.c_void_p(I_ptr),
                                                        ctypes.c_void_p(stream)))

    def set_search_mode(self, mode: str) -> None:
        """``"auto"`` (default: bf16 candidate scan + exact fp32 rescoring where the index keeps shadow
        rows and is large enough for it to pay), ``"exact_fp32"`` (every score formed in fp32 by the
        scan kernels), ``"coarse"`` (the candidate path whatever the index size) or ``"split"`` (batches: candidates
        from split-operand products of the fp32 rows, the fallback of shadow-less indexes; verification)."""
        modes = {"auto": 0, "exact_fp32": 1, "coarse": 2, "split": 3}
        if mode not in modes:
            raise ValueError(f"unknown search mode {mode!r}")
        nat.check(nat.lib().css_index_set_search_mode(self._handle(), modes[mode]))

    def last_flagged(self) -> int:
        """Diagnostics: queries of the last candidate-path search whose candidate band or buffer overflowed."""
        n = ctypes.c_int64(0)
        nat.check(nat.lib().css_index_last_flagged(self._handle(), ctypes.byref(n)))
        return int(n.value)

    def last_swept(self) -> int:
        """Diagnostics: flagged queries of the last candidate-path search that the second coarse pass could not settle
        and that the exact fp32 sweep re-ran."""
        n = ctypes.c_int64(0)
        nat.check(nat.lib().css_index_last_swept(self._handle(), ctypes.byref(n)))
        return int(n.value)

    def shadow_info(self) -> dict:
        """Diagnostics: which reduced-precision copies of the rows the index holds (``{"bf16": bool, "int8": bool}``)."""
        a, b = ctypes.c_int(0), ctypes.c_int(0)
        nat.check(nat.lib().css_index_shadow_info(self._handle(), ctypes.byref(a), ctypes.byref(b)))
        return {"bf16": bool(a.value), "int8": bool(b.value)}

    def set_shadow(self, policy) -> None:
        """Reduced-precision copies of the rows (operands of the candidate scans): ``None`` = automatic (bf16 + int8
        rows while 7 bytes per element fit in 80 % of the HBM, bf16 only at 6, int8 ONLY at 5 -- shards of ~38-46 M
        rows of 768 floats), ``False`` = never, ``True`` = always bf16, ``"int8"`` = int8 rows only.  Only on an empty
        index; results do not change."""
        p = -1 if policy is None else (2 if policy == "int8" else (1 if policy else 0))
        nat.check(nat.lib().css_index_set_shadow(self._handle(), p))

    def set_range_rows(self, rows: int) -> None:
        """Shadow-less indexes: rows per bf16 scratch range of a batched search (0 = automatic); results do not
        depend on it."""
        nat.check(nat.lib().css_index_set_range_rows(self._handle(), int(rows)))

    def set_id_base(self, base: int) -> None:
        nat.check(nat.lib().css_index_set_id_base(self._handle(), int(base)))

    def reconstruct_n(self, row0: int = 0, n: Optional[int] = None) -> np.ndarray:
        if n is None:
            n = self.ntotal - row0
        out = np.empty((int(n), self.d), dtype=np.float32)
        if n:
            nat.check(nat.lib().css_index_export(self._handle(), int(row0), int(n), out.ctypes.data))
        return out

    def reconstruct(self, i: int) -> np.ndarray:
        return self.reconstruct_n(int(i), 1)[0]


class IndexFlatIP(IndexFlat):
    def __init__(self, d: int, device: int = 0):
        super().__init__(d, METRIC_INNER_PRODUCT, device)


class IndexFlatL2(IndexFlat):
    def __init__(self, d: int, device: int = 0):
        super().__init__(d, METRIC_L2, device)


# ---------------------------------------------------------------------------
# faiss module-level seams
# ---------------------------------------------------------------------------
def get_num_gpus() -> int:
    return nat.device_count()


class StandardGpuResources:
    """Placeholder for ``faiss.StandardGpuResources`` (``src/storage.py:274``):
    libcss_hip owns its streams and workspaces per index."""

    def __init__(self):
        if nat.device_count() <= 0:
            raise RuntimeError("no HIP device")


def index_cpu_to_gpu(resources, device: int, index: IndexFlat) -> IndexFlat:
    """The index already lives in HBM; moving between devices copies the rows."""
    if index.device == int(device):
        return index
    out = IndexFlat(index.d, index.metric_type, int(device))
    if index.ntotal:
        out.add(index.reconstruct_n(0, index.ntotal))
    return out


def index_gpu_to_cpu(index: IndexFlat) -> IndexFlat:
    return index


# On-disk format of faiss' IndexFlat as faiss/impl/index_write.cpp / index_read.cpp lay it out (SURVEY.md 8f rank 1;
# [from knowledge of the public sources], not verifiable offline because faiss is not installed -- no file written by
# real faiss has been read here; tests/test_storage_host.py assembles fixtures byte by byte from THIS description):
#   fourcc  "IxFI" (inner product) | "IxF2" (L2) | "IxFl" (IndexFlat with the metric taken from the header)
#   index header (write_index_header):  int32 d | int64 ntotal | int64 dummy (1 << 20) | int64 dummy (1 << 20)
#                                       | uint8 is_trained | int32 metric_type (0 IP, 1 L2) [| float32 metric_arg if > 1]
#   codes (WRITEXBVECTOR):              uint64 n_floats (= ntotal * d) | n_floats * float32, row-major
# Readers ignore the two dummies; bytes behind the last row are ignored too (the append-on-save of HybridStorage
# relies on that for crash safety).
_FOURCC = {METRIC_INNER_PRODUCT: b"IxFI", METRIC_L2: b"IxF2"}
_FAISS_METRIC = {METRIC_INNER_PRODUCT: 0, METRIC_L2: 1}


def write_index(index: IndexFlat, path: str, chunk_rows: int = 1 << 18) -> None:
    n = index.ntotal
    with open(path, "wb") as f:
        f.write(_FOURCC[index.metric_type])
        f.write(struct.pack("<i", index.d))
        f.write(struct.pack("<q", n))
        f.write(struct.pack("<q", 1 << 20))
        f.write(struct.pack("<q", 1 << 20))
        f.write(struct.pack("<B", 1))
        f.write(struct.pack("<i", _FAISS_METRIC[index.metric_type]))
        f.write(struct.pack("<Q", n * index.d))
        for r0 in range(0, n, chunk_rows):
            m = min(chunk_rows, n - r0)
            f.write(index.reconstruct_n(r0, m).tobytes())


def _need(f, nbytes: int, what: str) -> bytes:
    buf = f.read(nbytes)
    if len(buf) != nbytes:
        raise RuntimeError(f"truncated index file (in {what})")
    return buf


def _read_header(f):
    """Header of an IndexFlat file -> (d, ntotal, metric).  Anything but a self-consistent flat index raises."""
    fourcc = _need(f, 4, "fourcc")
    if fourcc not in (b"IxFI", b"IxF2", b"IxFl"):
        raise RuntimeError(f"unsupported index file (fourcc {fourcc!r}); only IndexFlat (IxFI / IxF2 / IxFl) is implemented")
    (d,) = struct.unpack("<i", _need(f, 4, "d"))
    (n,) = struct.unpack("<q", _need(f, 8, "ntotal"))
    _need(f, 16, "header")                                   # two dummies (1 << 20 each): not interpreted, as in faiss
    (trained,) = struct.unpack("<B", _need(f, 1, "is_trained"))
    (mtype,) = struct.unpack("<i", _need(f, 4, "metric_type"))
    if mtype not in (0, 1):
        raise RuntimeError(f"unsupported metric_type {mtype} in index file (0 = inner product, 1 = L2)")
    metric = METRIC_INNER_PRODUCT if mtype == 0 else METRIC_L2
    if fourcc != b"IxFl" and fourcc != _FOURCC[metric]:
        raise RuntimeError(f"corrupt index file: fourcc {fourcc!r} with metric_type {mtype}")
    if trained != 1:
        raise RuntimeError("corrupt index file: a flat index is always trained")
    (nfl,) = struct.unpack("<Q", _need(f, 8, "vector size"))
    if d <= 0 or n < 0 or nfl != n * d:
        raise RuntimeError(f"corrupt index file: d={d}, ntotal={n}, {nfl} floats")
    return d, n, metric


def read_index_header(path: str):
    """``(d, ntotal, metric, byte offset of the rows)`` of an IndexFlat file whose payload is complete (a sharded
    reader then takes its own block of the rows: ``sharded.read_index_sharded``)."""
    import os

    with open(path, "rb") as f:
        d, n, metric = _read_header(f)
        offset = f.tell()
    if os.path.getsize(path) < offset + n * d * 4:
        raise RuntimeError("truncated index file (in rows)")
    return d, n, metric, offset


def read_index(path: str, device: int = 0, chunk_rows: int = 1 << 18) -> IndexFlat:
    """``faiss.read_index`` for the flat indexes the reference writes (``src/storage.py:301-316``, ``:870-885``).
    Anything else -- another index family, an untrained index, a header that contradicts itself, a file shorter
    than its header promises -- raises ``RuntimeError`` (the reference then starts a fresh index, ``:314-316``)."""
    with open(path, "rb") as f:
        d, n, metric = _read_header(f)
        index = IndexFlat(d, metric, device)
        if n:
            index.reserve(n)
        for r0 in range(0, n, chunk_rows):
            m = min(chunk_rows, n - r0)
            buf = _need(f, m * d * 4, "rows")
            index.add(np.frombuffer(buf, dtype=np.float32).reshape(m, d))
    return index
