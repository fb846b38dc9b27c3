"""Single-/few-query search latency at bench scale (development aid): python tools/knn_latency.py [rows]"""
import sys, time
sys.path.insert(0, ".")
import torch
from claude_semantic_search_amd.flat_index import IndexFlatIP
from claude_semantic_search_amd import synth
import numpy as np

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ix = IndexFlatIP(768)
ix.reserve(rows)
ix.add_synthetic(rows, seed=7)
st = torch.cuda.current_stream().cuda_stream
for nq, k in ((1, 10), (1, 100), (2, 10), (4, 10), (8, 10), (9, 10), (16, 10), (32, 10), (256, 10)):
    q = torch.from_numpy(synth.rows(nq, 768, 99)).cuda()
    D = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    I = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    for _ in range(3):
        ix.search_dev(q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), st, normalize=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        ix.search_dev(q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), st, normalize=True)
    torch.cuda.synchronize()
    print(f"nq={nq:4d} k={k:3d}: {(time.perf_counter() - t0) / reps * 1e3:7.3f} ms", flush=True)
