#!/bin/bash
# tools/pmc_stall.sh TAG WORKLOAD -- stall / issue breakdown counters for the simulate kernels
set -u
TAG=$1; WL=$2
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_stall -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --workload $WL > $OUT/pmc_stall.log 2>&1 || exit 1
timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_LEVEL_SMEM --output-format csv -d $OUT/pmc_stall2 -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --workload $WL > $OUT/pmc_stall2.log 2>&1 || exit 1
grep -o '{"metric.*' $OUT/pmc_stall.log | head -1
