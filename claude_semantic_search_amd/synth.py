"""numpy restatement of ``include/css_synth.h`` (bit-identical to the C/HIP form).

Used by bench.py and the tests to regenerate, on the host, any row of a
synthetic index / any synthetic token sequence the device generated.
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
SYNTH_SCALE = np.float32(2.64290629e-05)


def mix64(seed: int, idx: np.ndarray) -> np.ndarray:
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + _GOLDEN * (idx + np.uint64(1))
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def normal(seed: int, idx: np.ndarray) -> np.ndarray:
    h = mix64(seed, idx)
    m = np.uint64(0xFFFF)
    s = ((h & m).astype(np.int64) + ((h >> np.uint64(16)) & m).astype(np.int64)
         + ((h >> np.uint64(32)) & m).astype(np.int64) + (h >> np.uint64(48)).astype(np.int64) - 131070)
    return s.astype(np.float32) * SYNTH_SCALE


def uint(seed: int, idx: np.ndarray, lo: int, hi: int) -> np.ndarray:
    h = mix64(seed, idx)
    return (np.uint64(lo) + (((h >> np.uint64(32)) * np.uint64(hi - lo)) >> np.uint64(32))).astype(np.int64)


def tensor_seed(seed: int, tensor_id: int) -> int:
    return (seed ^ (tensor_id << 40) ^ 0xC55E7E11) & 0xFFFFFFFFFFFFFFFF


def rows(n: int, d: int, seed: int, first_row: int = 0) -> np.ndarray:
    """[n, d] fp32: row r, col c = normal(seed, (first_row + r) * d + c)."""
    idx = (np.arange(n, dtype=np.uint64)[:, None] + np.uint64(first_row)) * np.uint64(d) + np.arange(d, dtype=np.uint64)[None, :]
    return normal(seed, idx)


def rows_torch(n: int, d: int, seed: int, first_row: int = 0, device="cuda"):
    """``rows`` on a torch device, bit-identical (int64 arithmetic wraps like uint64; right shifts are made logical
    by masking).  Lets benchmarks and tests build derived synthetic data (clustered rows) at 10 M-row scale without
    generating 30 GB on the host."""
    import torch

    def lsr(z, sh):
        return (z >> sh) & ((1 << (64 - sh)) - 1)

    def s64(v):   # python int (uint64 value) -> the int64 with the same bits
        v &= 0xFFFFFFFFFFFFFFFF
        return v - (1 << 64) if v >= (1 << 63) else v

    idx = (torch.arange(n, dtype=torch.int64, device=device)[:, None] + first_row) * d + \
        torch.arange(d, dtype=torch.int64, device=device)[None, :]
    z = s64(seed) + s64(int(_GOLDEN)) * (idx + 1)
    z = (z ^ lsr(z, 30)) * s64(int(_M1))
    z = (z ^ lsr(z, 27)) * s64(int(_M2))
    h = z ^ lsr(z, 31)
    ssum = (h & 0xFFFF) + (lsr(h, 16) & 0xFFFF) + (lsr(h, 32) & 0xFFFF) + lsr(h, 48) - 131070
    return ssum.to(torch.float32) * float(SYNTH_SCALE)
