"""Latency of the product search path vs the exact fp32 kernels by index size (development aid)."""
import sys, time
sys.path.insert(0, ".")
import torch
from claude_semantic_search_amd.flat_index import IndexFlatIP, IndexFlatL2
from claude_semantic_search_amd import synth

st = torch.cuda.current_stream().cuda_stream
metric = sys.argv[1] if len(sys.argv) > 1 else "ip"
for rows in (10_000, 100_000, 500_000, 1_000_000, 2_000_000, 4_000_000, 10_000_000):
    ix = IndexFlatIP(768) if metric == "ip" else IndexFlatL2(768)
    ix.reserve(rows)
    ix.add_synthetic(rows, seed=7)
    line = [f"N={rows:8d}"]
    for nq, k in ((1, 100), (8, 10), (32, 10), (1000, 10)):
        q = torch.from_numpy(synth.rows(nq, 768, 99)).cuda()
        D = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        I = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        res = []
        for mode in ("auto", "exact_fp32"):
            ix.set_search_mode(mode)
            for _ in range(3):
                ix.search_dev(q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), st, normalize=True)
            torch.cuda.synchronize()
            reps = 20
            t0 = time.perf_counter()
            for _ in range(reps):
                ix.search_dev(q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), st, normalize=True)
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / reps * 1e3)
        line.append(f"nq={nq}: auto {res[0]:.3f} / exact {res[1]:.3f} ms")
    print("  ".join(line), flush=True)
    ix.close()
