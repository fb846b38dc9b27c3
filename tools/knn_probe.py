#!/usr/bin/env python3
"""Development aid: N searches of the synthetic bench index (seed 4 rows, seed 5 queries) at a chosen size, printing
ms per batch.  Run under `rocprofv3 --kernel-trace --output-format rocpd` and feed the .db to tools/step_timeline.py
for the kernel timeline of one step.   Usage: python tools/knn_probe.py [--rows N] [--nq Q] [--k K] [--reps R] [--mode M]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from claude_semantic_search_amd import synth  # noqa: E402
from claude_semantic_search_amd.flat_index import IndexFlatIP  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--nq", type=int, default=1000)
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--mode", default="auto", help="search mode (flat_index.IndexFlat.set_search_mode): auto | exact_fp32 | coarse | split")
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
st = torch.cuda.current_stream().cuda_stream
ix = IndexFlatIP(768)
ix.reserve(a.rows)
ix.add_synthetic(a.rows, seed=4, first_row=0, normalize=True, stream=st)
ix.set_search_mode(a.mode)
q = torch.from_numpy(synth.rows(a.nq, 768, 5)).to(dev)
D = torch.empty((a.nq, a.k), dtype=torch.float32, device=dev)
I = torch.empty((a.nq, a.k), dtype=torch.int64, device=dev)
for _ in range(3):
    ix.search_dev(q.data_ptr(), a.nq, a.k, D.data_ptr(), I.data_ptr(), st, normalize=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    ix.search_dev(q.data_ptr(), a.nq, a.k, D.data_ptr(), I.data_ptr(), st, normalize=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.reps
print(f"rows={a.rows} nq={a.nq} k={a.k} mode={a.mode}: {dt * 1e3:.3f} ms per batch, {a.nq / dt:.0f} queries/s")
