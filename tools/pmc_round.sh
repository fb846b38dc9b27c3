#!/bin/bash
# Counter passes (MFMA utilisation, wave stall breakdown, LDS conflicts, clock) over bench.py on the GPU box.
# Each pass is its own rocprofv3 run with --pmc only (never combined with a trace domain); the program follows
# `--` directly.  Usage (repo root, GPU box): bash tools/pmc_round.sh TAG [extra bench.py args]
set -o pipefail
TAG=${1:-r02}
shift
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extra --enc-steps 2 $*"
rocprofv3 -L > $O/counters_available.txt 2>&1 || true
timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    -d $O/pmc_a -o run --output-format csv -- python3 bench.py $ARGS > $O/pmc_a.json 2> $O/pmc_a.err || exit 3
echo "[pmc] pass a done"
timeout -k 10 500 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
    -d $O/pmc_b -o run --output-format csv -- python3 bench.py $ARGS > $O/pmc_b.json 2> $O/pmc_b.err || exit 4
echo "[pmc] pass b done"
python3 tools/pmc_counters.py $O/pmc_counters.json $O/pmc_a $O/pmc_b \
  --note "rocprofv3 --pmc (two separate passes) on python3 bench.py $ARGS" > $O/pmc_counters.txt
cat $O/pmc_counters.txt
find $O -name "*counter_collection.csv" -size +20M -delete
