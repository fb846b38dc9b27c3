#!/bin/bash
# Round-4 profile on the GPU box.  One duration population per kernel and file: every rocprofv3 --kernel-trace --stats
# pass runs ONE leg of bench.py, so the per-kernel average of the stats csv is the launch the matching `roofline`
# field describes (profiles/README.md lists which file backs which field).  PMC passes are separate runs with --pmc
# only (never combined with a trace domain); the program follows `--` directly.
# Usage (repo root, GPU box): bash tools/profile_round4.sh TAG [full]
#   full: also the default bench line with every extra (heavy: 80 M rows, clustered 10 M) -> $O/bench.json
set -o pipefail
TAG=${1:-r04}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
LIGHT="--no-cpu-baseline --no-encoder"
stats() {   # name, bench args...
    local name=$1; shift
    timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats_$name -o run --output-format csv -- python3 bench.py "$@" \
        > $O/${name}_under_rocprof.json 2> $O/stats_$name.err || return 1
    cp $(find $O/stats_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
    find $O/stats_$name -name "*kernel_trace.csv" -delete
    echo "[profile] stats $name done"
}
pmc() {     # dir name, counters (quoted), bench args...
    local name=$1 ctrs=$2; shift 2
    timeout -k 10 400 rocprofv3 --pmc $ctrs -d $O/$name -o run --output-format csv -- python3 bench.py "$@" > $O/$name.json 2> $O/$name.err || return 1
    echo "[profile] pmc $name done"
}
if [ "$2" = "full" ]; then
    timeout -k 10 900 python3 bench.py > $O/bench.json 2> $O/bench.err || exit 1
    echo "[profile] full bench done"
fi
# --- kernel-trace stats, one leg each ---------------------------------------------------------------------------
stats headline $LIGHT --legs none || exit 2                       # k_scan_coarse8<false,true,false,4096,true>: 1000 q x 10 M
stats nq1 $LIGHT --legs nq1 --steps 2 --warmup 1 || exit 3        # k_sweep_cascade<1,3,true>: single query, k = 10 and k' = 100
stats exact $LIGHT --legs exact --steps 2 --warmup 1 || exit 4    # k_scan_small<1,..> (nq = 1) and k_scan_mfma (256 queries)
stats encoder --only-encoder --enc-fixed-only --no-cpu-baseline --enc-steps 5 || exit 5
# --- HBM-side traffic (FETCH_SIZE, WRITE_SIZE: separate passes) -----------------------------------------------------
P="--steps 3 --warmup 1 $LIGHT --legs nq1,exact"
pmc pmc_fetch FETCH_SIZE $P || exit 6
pmc pmc_write WRITE_SIZE $P || exit 7
python3 tools/pmc_summarise.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm_traffic.json \
  --note "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on python3 bench.py $P" \
  --workload '{"rows_per_gpu": 10000000, "dim": 768, "nq": 1000, "k": 10}'
E="--only-encoder --enc-fixed-only --no-cpu-baseline --enc-steps 3"
pmc pmc_fetch_enc FETCH_SIZE $E || exit 8
pmc pmc_write_enc WRITE_SIZE $E || exit 9
python3 tools/pmc_summarise.py $O/pmc_fetch_enc $O/pmc_write_enc $O/pmc_hbm_traffic_encoder.json --all \
  --note "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on python3 bench.py $E" \
  --workload '{"enc_batch": 256, "enc_len": 384}'
# --- MFMA utilisation / stalls / clock / LDS conflicts (two passes each) ---------------------------------------------
CA="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
CB="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
Q="--steps 3 --warmup 1 $LIGHT --legs nq1"
pmc pmc_a "$CA" $Q || exit 10
pmc pmc_b "$CB" $Q || exit 11
python3 tools/pmc_counters.py $O/pmc_counters.json $O/pmc_a $O/pmc_b \
  --note "rocprofv3 --pmc (two separate passes) on python3 bench.py $Q" > $O/pmc_counters.txt
pmc pmc_a_enc "$CA" $E || exit 12
pmc pmc_b_enc "$CB" $E || exit 13
python3 tools/pmc_counters.py $O/pmc_counters_encoder.json $O/pmc_a_enc $O/pmc_b_enc \
  --note "rocprofv3 --pmc (two separate passes) on python3 bench.py $E" > $O/pmc_counters_encoder.txt
find $O -name "*counter_collection.csv" -size +20M -delete
ls -la $O
