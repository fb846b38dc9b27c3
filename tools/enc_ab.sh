#!/bin/bash
# usage: tools/enc_ab.sh "ENV1=.. ENV2=.." ...   -> one line per variant with per-kernel ms   (BENCH_ARGS: extra bench.py flags)
for v in "$@"; do
  env $v timeout -k 10 300 python bench.py --only-encoder --enc-fixed-only --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python tools/print_enc_bench.py "$v"
done
