#!/bin/bash
# usage: tools/enc_ab.sh "ENV1=.. ENV2=.." ...   -> one line per variant with per-kernel ms
for v in "$@"; do
  env $v timeout -k 10 300 python bench.py --only-encoder --enc-fixed-only --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
try:
    e=json.loads(sys.stdin.read())['encode']
    print('$v', round(e['ms_per_batch'],2), 'ms', round(e['roofline']['achieved']), 'TF', {k.replace('enc_',''):round(v['ms_per_batch'],2) for k,v in e['kernels'].items()})
except Exception as ex:
    print('$v', 'FAILED', ex)
"
done
