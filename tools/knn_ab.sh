#!/bin/bash
# A/B timing of the batched kNN leg under different env settings, back to back on one box.
# usage: tools/knn_ab.sh "ENV1=.." "ENV2=.." ...   (use "-" for the default environment)
for e in "$@"; do
  if [ "$e" = "-" ]; then e=""; fi
  out=$(env $e timeout -k 10 200 python3 bench.py --no-encoder --no-cpu-baseline --no-extra --steps 5 --warmup 2 --allow-debug 2>/dev/null)
  echo "$out" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline'] or {}
print('%-40s qps=%8.0f ms/step=%7.2f main_ms=%s cascade_ms=%s' % ('$e' or 'default', d['value'], d['ms_per_step'], (r.get('timed_scopes_ms') or {}).get('knn_scan_coarse_main'), r.get('cascade_ms')))"
done
