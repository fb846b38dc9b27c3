"""Few-query search latency, sweep cascade vs MFMA scan (development aid): python tools/knn_fewq_probe.py rows [rows ...]
Run once per CSS_KNN_SWEEP_MAXQ setting (the switch is read once per process)."""
import os, sys, time
sys.path.insert(0, ".")
import torch
from claude_semantic_search_amd.flat_index import IndexFlatIP
from claude_semantic_search_amd import synth

st = torch.cuda.current_stream().cuda_stream
for rows in [int(a) for a in sys.argv[1:]] or [10_000_000]:
    ix = IndexFlatIP(768)
    ix.reserve(rows)
    ix.add_synthetic(rows, seed=7)
    out = []
    for nq in [int(v) for v in os.environ.get("FEWQ_NQ", "1,2,3,4,5,8,16,17").split(",")]:
        for k in [int(v) for v in os.environ.get("FEWQ_K", "10,100").split(",")]:
            q = torch.from_numpy(synth.rows(nq, 768, 99)).cuda()
            D = torch.empty((nq, k), dtype=torch.float32, device="cuda")
            I = torch.empty((nq, k), dtype=torch.int64, device="cuda")
            for _ in range(3):
                ix.search_dev(q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), st, normalize=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                ix.search_dev(q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), st, normalize=True)
            torch.cuda.synchronize()
            out.append(f"nq{nq} k{k}: {(time.perf_counter() - t0) / 10 * 1e3:.3f}")
    print(f"{rows:>9} rows | " + " | ".join(out), flush=True)
    ix.close()
