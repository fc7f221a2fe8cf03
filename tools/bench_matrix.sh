#!/bin/bash
# tools/bench_matrix.sh OUT.jsonl -- one bench.py line per workload (the DESIGN.md section 7 table), on the GPU box.
OUT=${1:-gpurun_out/bench_matrix.jsonl}
: > $OUT
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '^{"metric' | tail -1 >> $OUT; }
run --workload c2a --steps 20 --warmup 5
run --workload c2a --steps 60 --warmup 5
run --workload c2a --steps 20 --warmup 5 --depth 1
run --workload c2b --steps 20 --warmup 5
run --workload ols_wide --steps 10 --warmup 2
run --workload c3 --steps 20 --warmup 5
run --workload glm_wide --steps 10 --warmup 3
run --workload c4 --steps 20 --warmup 5
run --workload c4 --steps 20 --warmup 5 --sims-per-gpu 1000000
run --workload c5_optimistic --steps 10 --warmup 2
run --workload c5_realistic --steps 6 --warmup 3
run --workload c5_doomer --steps 6 --warmup 3
run --workload lmm_slope --steps 6 --warmup 3
run --workload lmm_crossed --steps 6 --warmup 3
run --workload glmm --steps 6 --warmup 3
run --workload glmm_slope --steps 6 --warmup 3
run --workload glmm_crossed --steps 3 --warmup 1 --depth 1
python - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); r = d["roofline"]
    print("%-14s depth %d steps %2d: %.4g fits/s, %.4f ms/step, launch %.4f ms x%.2f, frac %.3f" % (
        d["config"]["workload"].split(":")[0], d["config"]["batches_in_flight"], d["steps"], d["value"], d["ms_per_step"],
        r["avg_launch_ms"], r["mean_concurrent_launches"], r["frac"]))
PY
