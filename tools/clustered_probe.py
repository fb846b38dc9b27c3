#!/usr/bin/env python3
"""Development aid: the clustered-rows leg of bench.py alone (queries/s, flagged queries) at a chosen size.
Usage: python tools/clustered_probe.py [--rows N] [--clusters C] [--uniform]   (run under rocprofv3 --kernel-trace for a timeline)"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_000_000)
ap.add_argument("--clusters", type=int, default=2000)
ap.add_argument("--uniform", action="store_true")
a = ap.parse_args()
args = bench.parse_args([])
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
out = bench.bench_clustered(args, dev, torch.cuda.current_stream().cuda_stream, lambda m: print("[probe]", m, file=sys.stderr),
                            n=a.rows, nc=a.clusters, with_uniform=a.uniform)
print(json.dumps(out))
