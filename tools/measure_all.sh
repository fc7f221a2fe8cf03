#!/bin/bash
# tools/measure_all.sh TAG -- the whole measurement pass of a round on the GPU box, into gpurun_out/ (copy what is to be
# judged into profiles/ afterwards):  default bench line, one bench line per workload, the 21 frozen cases, whole-call
# latency at the reference's default sizes, BASELINE configs[4] end to end, rocprofv3 + PMC of the headline configs.
# About 8 minutes of box time.  Usage (from this container): gpurun --timeout 1200 -- 'bash tools/measure_all.sh r04_a'
set -u
TAG=${1:?tag}
O=gpurun_out
mkdir -p $O
timeout -k 10 400 python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err || exit 1
timeout -k 10 700 bash tools/bench_matrix.sh $O/${TAG}_bench_matrix.jsonl > $O/${TAG}_bench_matrix.txt 2>&1 || exit 1
timeout -k 10 900 python tools/throughput_cases.py --cpu-fits 4000 --out $O/${TAG}_throughput_cases.jsonl > $O/${TAG}_throughput_cases.log 2>&1 || exit 1
timeout -k 10 300 python tools/latency.py > $O/${TAG}_latency.md 2> $O/${TAG}_latency.err || exit 1
timeout -k 10 400 python tools/c5_sweep.py --check > $O/${TAG}_c5_sweep.log 2>&1 || exit 1
for w in c2a c4 c5_realistic; do
  bash tools/profile_bench.sh ${TAG}_$w --steps 10 --warmup 2 --workload $w || exit 1
done
echo "measure_all $TAG done"
