#!/bin/bash
# tools/profile_cases.sh TAG CASES -- rocprofv3 kernel trace + SQ counters over tools/throughput_cases.py for the given
# frozen case ids (comma separated); summaries land in gpurun_out/prof_TAG
set -u
TAG=$1; CASES=$2
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 tools/throughput_cases.py --no-oracle --cases $CASES --target-ms 200 --out $OUT/cases_kt.jsonl > $OUT/kt.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq -- python3 tools/throughput_cases.py --no-oracle --cases $CASES --target-ms 200 --out $OUT/cases_pmc.jsonl > $OUT/pmc_sq.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/kt/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "mcp::" in r["Name"] and float(r["Percentage"]) > 0.5: print(r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
for f in glob.glob("$OUT/pmc_sq/*/*_counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list)); meta = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "general" not in k and "glm" not in k: continue
        agg[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[(k, r["Grid_Size"])] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
    for k, v in agg.items():
        waves = sum(v["SQ_WAVES"]) / len(v["SQ_WAVES"])
        print(k[0][:70], "grid", k[1], "vgpr/agpr/lds/scratch", meta[k], "per wave:", {c: round(sum(x) / len(x) / waves) for c, x in v.items() if c != "SQ_WAVES"})
PY
