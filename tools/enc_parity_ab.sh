#!/bin/bash
# usage: tools/enc_parity_ab.sh "ENV=.." ... -> ms per batch + min cosine vs the fp32 oracle per variant ("-" = default env)
for v in "$@"; do
  e="$v"; if [ "$v" = "-" ]; then e=""; fi
  env $e timeout -k 10 300 python bench.py --only-encoder 2>/dev/null | python -c "
import json,sys
try:
    e=json.loads(sys.stdin.read())['encode']
    print('%-24s' % '$v', round(e['ms_per_batch'],2), 'ms', round(e['roofline']['achieved']), 'TF  min_cos', e.get('parity_vs_oracle_min_cos'), {k.replace('enc_',''):round(v['ms_per_batch'],2) for k,v in e['kernels'].items()})
except Exception as ex:
    print('$v', 'FAILED', ex)
"
done
