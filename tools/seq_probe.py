"""Development aid: first 1000-query search of an index after 256-query searches (workspaces grow)."""
import os, sys
sys.path.insert(0, ".")
import torch
from claude_semantic_search_amd.flat_index import IndexFlatIP
from claude_semantic_search_amd import synth, _native as nat
st = torch.cuda.current_stream().cuda_stream
for rows in [int(a) for a in sys.argv[1:]] or [1_000_000]:
    ix = IndexFlatIP(768); ix.reserve(rows); ix.add_synthetic(rows, seed=7)
    for nq in [int(v) for v in os.environ.get("NQSEQ", "256,1000").split(",")]:
        q = torch.from_numpy(synth.rows(nq, 768, 99)).cuda()
        D = torch.empty((nq, 10), dtype=torch.float32, device="cuda"); I = torch.empty((nq, 10), dtype=torch.int64, device="cuda")
        for it in range(3):
            nat.prof_reset(); nat.prof_enable(True)
            ix.search_dev(q.data_ptr(), nq, 10, D.data_ptr(), I.data_ptr(), st, normalize=True)
            torch.cuda.synchronize(); nat.prof_enable(False)
            print(rows, nq, it, "flagged", ix.last_flagged(), "swept", ix.last_swept(), "cascade ms", round(nat.prof_read("knn_coarse_cascade")[0], 3), flush=True)
    ix.close()
