#!/bin/bash
for e in "$@"; do
  if [ "$e" = "-" ]; then e="A=1"; fi
  echo "== $e"; env $e timeout -k 10 200 python3 bench.py --no-cpu-baseline --legs nq1,e2e --allow-debug 2>&1 >/dev/null | grep "e2e"
done
