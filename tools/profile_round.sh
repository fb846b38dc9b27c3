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
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-heavy-extra --enc-steps 2 > $O/pmc_fetch.json 2> $O/pmc_fetch.err || exit 3
echo "[profile] fetch done"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-heavy-extra --enc-steps 2 > $O/pmc_write.json 2> $O/pmc_write.err || exit 4
echo "[profile] write done"
python3 tools/pmc_summarise.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm_traffic.json \
  --note "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-heavy-extra --enc-steps 2" \
  --workload '{"rows_per_gpu": 10000000, "dim": 768, "nq": 1000, "k": 10}'
# encoder alone at its fixed 256 x 384 shape (per-forward HBM traffic = sum over its kernels / forwards)
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_enc -o run --output-format csv -- python3 bench.py --only-encoder --enc-fixed-only --no-cpu-baseline --enc-steps 3 > $O/pmc_fetch_enc.json 2> $O/pmc_fetch_enc.err || exit 5
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_enc -o run --output-format csv -- python3 bench.py --only-encoder --enc-fixed-only --no-cpu-baseline --enc-steps 3 > $O/pmc_write_enc.json 2> $O/pmc_write_enc.err || exit 6
python3 tools/pmc_summarise.py $O/pmc_fetch_enc $O/pmc_write_enc $O/pmc_hbm_traffic_encoder.json --all \
  --note "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on python3 bench.py --only-encoder --enc-fixed-only --no-cpu-baseline --enc-steps 3" \
  --workload '{"enc_batch": 256, "enc_len": 384}'
# MFMA utilisation / stall breakdown / clock (two more counter passes)
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    -d $O/pmc_a -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --enc-steps 2 > $O/pmc_a.json 2> $O/pmc_a.err || exit 7
timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
    -d $O/pmc_b -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --enc-steps 2 > $O/pmc_b.json 2> $O/pmc_b.err || exit 8
python3 tools/pmc_counters.py $O/pmc_counters.json $O/pmc_a $O/pmc_b \
  --note "rocprofv3 --pmc (two separate passes) on python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --enc-steps 2" > $O/pmc_counters.txt
# the same two counter passes over the encoder alone at its fixed 256 x 384 shape (every launch full size)
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    -d $O/pmc_a_enc -o run --output-format csv -- python3 bench.py --only-encoder --enc-fixed-only --no-cpu-baseline --enc-steps 3 > $O/pmc_a_enc.json 2> $O/pmc_a_enc.err || exit 9
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
    -d $O/pmc_b_enc -o run --output-format csv -- python3 bench.py --only-encoder --enc-fixed-only --no-cpu-baseline --enc-steps 3 > $O/pmc_b_enc.json 2> $O/pmc_b_enc.err || exit 10
python3 tools/pmc_counters.py $O/pmc_counters_encoder.json $O/pmc_a_enc $O/pmc_b_enc \
  --note "rocprofv3 --pmc (two separate passes) on python3 bench.py --only-encoder --enc-fixed-only --no-cpu-baseline --enc-steps 3" > $O/pmc_counters_encoder.txt
find $O -name "*kernel_trace.csv" -delete   # large; the stats csv is the committed summary
find $O -name "*counter_collection.csv" -size +20M -delete
ls -la $O $O/stats
