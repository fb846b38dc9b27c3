#!/usr/bin/env python3
"""Experiment: do two independent half batches (128 x 384 each, two encoders, two streams) overlap their GEMM tails?
Compares one 256 x 384 forward with two concurrent 128 x 384 forwards (and two sequential ones)."""
import ctypes
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from claude_semantic_search_amd import _native as nat  # noqa: E402
from claude_semantic_search_amd import synth  # noqa: E402
from claude_semantic_search_amd.mpnet_encoder import MpnetEncoder  # noqa: E402


def make(B, L, dev):
    enc = MpnetEncoder(synthetic_seed=1, compute="bf16", device=0)
    T = B * L
    ids_h = synth.uint(7, np.arange(T, dtype=np.uint64), 4, enc.cfg["vocab"]).astype(np.int32)
    ids_h[0::L] = 0
    ids_h[L - 1::L] = 2
    cu_h = (np.arange(B + 1, dtype=np.int64) * L).astype(np.int32)
    return enc, torch.from_numpy(ids_h).to(dev), torch.from_numpy(cu_h).to(dev), torch.empty((B, 768), device=dev), B, T, L


def fwd(h, stream):
    enc, ids, cu, out, B, T, L = h
    nat.check(nat.lib().css_encoder_forward_dev(enc._h, ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(cu.data_ptr()), B, T, L, 1,
                                                ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(stream.cuda_stream)))


def main():
    dev = torch.device("cuda:0")
    full = make(256, 384, dev)
    h1, h2 = make(128, 384, dev), make(128, 384, dev)
    s0, s1, s2 = torch.cuda.current_stream(), torch.cuda.Stream(), torch.cuda.Stream()
    reps = 10
    for name, run in (("one 256 x 384", lambda: fwd(full, s0)),
                      ("two 128 x 384, one stream", lambda: (fwd(h1, s0), fwd(h2, s0))),
                      ("two 128 x 384, two streams", lambda: (fwd(h1, s1), fwd(h2, s2)))):
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        print(f"{name}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms per 256 sequences")


if __name__ == "__main__":
    main()
