#!/bin/bash
# A/B timing of the single-query search (bench.py --legs nq1) under different env settings, back to back on one box.
# usage: tools/nq1_ab.sh "ENV1=.. ENV2=.." ...   (use "-" for the default environment)
for e in "$@"; do
  if [ "$e" = "-" ]; then e=""; fi
  out=$(env $e timeout -k 10 200 python3 bench.py --no-encoder --no-cpu-baseline --legs nq1 --allow-debug 2>/dev/null)
  echo "$out" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); x=d['extra']
print('%-60s' % ('$e' or 'default'), ' '.join('%s: lat %.3f cascade %.3f kernel %.3f ms' % (k, x[k]['latency_ms'], x[k]['cascade_ms'], x[k]['scan_kernel_ms']) for k in ('nq1_k10','nq1_k100')))"
done
