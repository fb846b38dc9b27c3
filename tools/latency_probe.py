"""Single-query latencies of the product path (the reference's search_conversations shape)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from claude_semantic_search_amd.mpnet_encoder import MpnetEncoder
from claude_semantic_search_amd.flat_index import IndexFlatIP

enc = MpnetEncoder(synthetic_seed=1, compute="bf16")
text = "how do I handle errors in python with try except blocks"
for _ in range(5):
    enc.encode(text)
t0 = time.perf_counter(); n = 200
for _ in range(n):
    q = enc.encode(text)
dt = (time.perf_counter() - t0) / n
print(f"single-query encode ({len(enc.tokenize([text])[0])} tokens): {dt*1e3:.3f} ms")
ids = enc.tokenize([text])
t0 = time.perf_counter()
for _ in range(n):
    enc.encode_ids(ids)
print(f"  encode_ids only: {(time.perf_counter()-t0)/n*1e3:.3f} ms")
for N in (10_000, 100_000, 1_000_000):
    ix = IndexFlatIP(768); ix.add_synthetic(N, 4, 0, True)
    for _ in range(3): ix.search(q, 100)
    t0 = time.perf_counter()
    for _ in range(n): ix.search(q, 100)
    print(f"flat search N={N}, k'=100 (host API, incl. H2D/D2H): {(time.perf_counter()-t0)/n*1e3:.3f} ms")
    ix.close()
