#!/usr/bin/env python3
"""Development aid: bench.py's shadow-less leg alone (ranges of on-the-fly bf16 rows vs the split-operand scan)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

args = bench.parse_args([])
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
print(json.dumps(bench.bench_no_shadow(args, dev, torch.cuda.current_stream().cuda_stream, lambda m: print("[probe]", m, file=sys.stderr), n=n)))
