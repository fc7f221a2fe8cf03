#!/bin/bash
# tools/profile_bench.sh TAG [bench args...] -- rocprofv3 passes over bench.py on the GPU box.
# Pass 1: --kernel-trace --stats (per-kernel time).  Pass 2/3/4: PMC counters, each in its own run
# (SQ issue counters; FETCH_SIZE; WRITE_SIZE) as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
set -u
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --no-cpu-baseline "$@" > $OUT/kt.log 2>&1 || exit 1
grep -o '{"metric.*' $OUT/kt.log | tail -1 > $OUT/bench_under_profiler.json
timeout -k 10 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 "${@:5}" > $OUT/pmc_sq.log 2>&1 || exit 1
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 "${@:5}" > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 "${@:5}" > $OUT/pmc_write.log 2>&1 || exit 1
echo "profile $TAG done"
