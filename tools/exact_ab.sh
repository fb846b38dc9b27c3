#!/bin/bash
# A/B timing of the exact fp32 legs (bench.py --legs exact) under different env settings, back to back on one box.
for e in "$@"; do
  if [ "$e" = "-" ]; then e="A=1"; fi
  out=$(env $e timeout -k 10 300 python3 bench.py --no-encoder --no-cpu-baseline --legs exact --allow-debug --steps 3 --warmup 1 2>/dev/null)
  echo "$out" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']
print('%-50s' % '$e', 'exact nq1: %.3f ms frac %.3f | nq256 mfma frac %.3f' % (r['exact_fp32_nq1_latency_ms'], r['exact_fp32_nq1_hbm_frac'], r['exact_fp32_nq256_mfma_frac']))"
done
