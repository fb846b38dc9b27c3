"""Latency of the candidate path (mode "coarse") vs the exact fp32 kernels by index size and query count: where
CSS_SEARCH_AUTO should switch (css_index.hip, search_dev_enqueue: coarse_pays).  Development aid.
usage: python tools/knn_crossover.py [ip|l2]"""
import sys, time
sys.path.insert(0, ".")
import torch
from claude_semantic_search_amd.flat_index import IndexFlatIP, IndexFlatL2
from claude_semantic_search_amd import synth

st = torch.cuda.current_stream().cuda_stream
metric = sys.argv[1] if len(sys.argv) > 1 else "ip"
for rows in (2_000, 5_000, 10_000, 20_000, 30_000, 50_000, 75_000, 100_000, 1_000_000):
    ix = IndexFlatIP(768) if metric == "ip" else IndexFlatL2(768)
    ix.reserve(rows)
    ix.add_synthetic(rows, seed=7)
    line = [f"N={rows:8d}"]
    for nq, k in ((1, 100), (1, 10), (4, 10), (8, 10), (16, 10)):
        q = torch.from_numpy(synth.rows(nq, 768, 99)).cuda()
        D = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        I = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        res = []
        for mode in ("coarse", "exact_fp32"):
            ix.set_search_mode(mode)
            for _ in range(3):
                ix.search_dev(q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), st, normalize=True)
            torch.cuda.synchronize()
            reps = 30
            t0 = time.perf_counter()
            for _ in range(reps):
                ix.search_dev(q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), st, normalize=True)
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / reps * 1e3)
        line.append(f"nq={nq},k={k}: {res[0]:.3f}/{res[1]:.3f}")
    print("  ".join(line) + "   (coarse/exact ms)", flush=True)
    ix.close()
