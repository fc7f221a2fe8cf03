#!/usr/bin/env python3
"""Whole-call latency at the reference's DEFAULT call sizes (configs/config.json:6: n_sims 1 600 for OLS / GLM, 800 for the
mixed estimators) -- the size a real `find_power` call has.  At these sizes a device call is launch latency, not throughput.

Per workload: median wall time of
  * `engine.run_batch` with the per-sim tensors (what `run_batch` returns in the reference, BatchResult), spec marshalled per call;
  * the same, counters only, on a marshalled spec;
  * `_engine.find_power` end to end (msgpack contract -> lowering -> run_batch -> aggregate -> host dict), where a contract
    for the workload exists in tests/golden/contracts;
  * the CPU oracle's `run_batch` on this host's cores (the "port" of bench.py's cpu_baseline), same spec and seeds.
Writes a markdown table and one JSON line per workload to --out."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def med_ms(fn, reps):
    fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(t)), 1e3 * float(np.min(t))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "latency.jsonl"))
    ap.add_argument("--no-oracle", action="store_true")
    a = ap.parse_args()
    from mcpower_amd import engine, workloads as W
    oracle = None
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))
    if not a.no_oracle:
        from oracle import oracle as _o
        _o.use_native_build()
        oracle = _o
    cases = [("C2a OLS n=200 p=5", W.spec_c2a(), 200, 1600), ("C2b OLS n=200 p=6", W.spec_c2b(), 200, 1600),
             ("C3 logit GLM n=500 p=4", W.spec_c3(), 500, 1600), ("C4 LMM 30x20", W.spec_c4(), 600, 800),
             ("glmm_intercept 30x20", W.spec_glmm(), 600, 800), ("glmm_slope 30x20", W.spec_glmm_slope(), 600, 800),
             ("lmm_slope 24x20", W.spec_lmm_slope(), 480, 800), ("lmm_crossed 20x12", W.spec_lmm_crossed(), 480, 800)]
    rows = []
    with open(a.out, "w") as fh:
        for name, spec, n, sims in cases:
            low = spec.to_c()
            full, full_min = med_ms(lambda: engine.run_batch(spec, [n], sims, W.BASE_SEED), a.reps)
            ctr, ctr_min = med_ms(lambda: engine.run_batch(low, [n], sims, W.BASE_SEED, per_sim=False), a.reps)
            r = engine.run_batch(low, [n], sims, W.BASE_SEED, per_sim=False)
            row = {"workload": name, "n_sims": sims, "device_per_sim_ms": full, "device_per_sim_min_ms": full_min,
                   "device_counters_ms": ctr, "device_counters_min_ms": ctr_min, "device_kernel_ms": r.kernel_ms + r.aux_kernel_ms}
            if oracle is not None:
                o_ms, o_min = med_ms(lambda: oracle.run_batch(spec, [n], sims, W.BASE_SEED, n_threads=cores), max(3, a.reps // 6))
                row.update(oracle_ms=o_ms, oracle_min_ms=o_min, oracle_threads=cores)
            rows.append(row)
            fh.write(json.dumps(row) + "\n")
            fh.flush()
    print("| workload | replicates | device call, per-sim tensors (ms) | device call, counters only (ms) | of which kernels (ms) | CPU oracle, %d threads (ms) |" % cores)
    print("|---|---|---|---|---|---|")
    for r in rows:
        print("| %s | %d | %.3f | %.3f | %.3f | %s |" % (r["workload"], r["n_sims"], r["device_per_sim_ms"], r["device_counters_ms"],
                                                        r["device_kernel_ms"], ("%.2f" % r["oracle_ms"]) if "oracle_ms" in r else "-"))


if __name__ == "__main__":
    main()
