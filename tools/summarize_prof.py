#!/usr/bin/env python3
"""tools/summarize_prof.py TAG ROUND_PREFIX -- condense gpurun_out/prof_TAG into small CSVs under profiles/."""
import collections
import csv
import glob
import os
import shutil
import sys

tag, prefix = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
ks = sorted(glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True)  # newest run first
if ks:
    shutil.copy(ks[0], os.path.join(dst, f"{prefix}_{tag}_kernel_stats.csv"))
bj = os.path.join(src, "bench_under_profiler.json")
if os.path.exists(bj):
    shutil.copy(bj, os.path.join(dst, f"{prefix}_{tag}_bench_under_profiler.json"))
rows = []
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    for f in sorted(glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")), key=os.path.getmtime, reverse=True)[:1]:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        meta = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "mcp::" not in k:
                continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"],
                       r["Workgroup_Size"], r["Grid_Size"])
        for k, v in agg.items():
            for c, vals in sorted(v.items()):
                rows.append((k, c, sum(vals) / len(vals), len(vals)) + meta[k])
with open(os.path.join(dst, f"{prefix}_{tag}_pmc.csv"), "w") as out:
    out.write("kernel,counter,mean_per_dispatch,dispatches,vgpr,agpr,sgpr,lds_bytes,scratch,workgroup,grid\n")
    for r in rows:
        out.write('"%s",%s,%.1f,%d,%s,%s,%s,%s,%s,%s,%s\n' % r)
print(open(os.path.join(dst, f"{prefix}_{tag}_pmc.csv")).read())
if ks:
    print(open(ks[0]).read())
