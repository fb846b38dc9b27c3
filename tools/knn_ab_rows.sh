#!/bin/bash
# usage: tools/knn_ab_rows.sh ROWS "ENV1" "ENV2" ... ("-" = default env): batched-search ms per step at a shard size
rows=$1; shift
for e in "$@"; do
  v="$e"; if [ "$e" = "-" ]; then e=""; fi
  out=$(env $e timeout -k 10 200 python3 bench.py --rows $rows --no-encoder --no-cpu-baseline --no-extra --steps 20 --warmup 3 2>/dev/null)
  echo "$out" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-22s rows=%d qps=%8.0f ms/step=%7.3f' % ('$v', $rows, d['value'], d['ms_per_step']))"
done
