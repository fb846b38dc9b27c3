#!/bin/bash
# Round profile on the GPU box: the default bench, the same command under rocprofv3 --kernel-trace --stats,
# and two separate PMC passes (FETCH_SIZE, WRITE_SIZE) -- never combined with a trace domain.
# Usage (from the repo root, on the GPU box): bash tools/profile_round.sh TAG
set -o pipefail
TAG=${1:-r01}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err || exit 1
echo "[profile] bench done"
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- python3 bench.py > $O/bench_under_rocprof.json 2> $O/stats.err || exit 2
echo "[profile] stats done"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --enc-steps 2 > $O/pmc_fetch.json 2> $O/pmc_fetch.err || exit 3
echo "[profile] fetch done"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --enc-steps 2 > $O/pmc_write.json 2> $O/pmc_write.err || exit 4
echo "[profile] write done"
python3 tools/pmc_summarise.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm_traffic.json \
  --note "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --enc-steps 2" \
  --workload '{"rows_per_gpu": 10000000, "dim": 768, "nq": 1000, "k": 10}'
find $O -name "*kernel_trace.csv" -delete   # large; the stats csv is the committed summary
find $O -name "*counter_collection.csv" -size +20M -delete
ls -la $O $O/stats
