#!/usr/bin/env python3
"""The reference's frozen 21-case fits/sec grid (crates/engine-core/src/bin/throughput.rs:268-633) on the device.

For every case id, with the reference's own spec, n and seed:
  * parity  -- the reference's n_sims replicates (capped by --parity-sims) through the C ABI and through the CPU
               oracle: replicates whose decision / convergence flags agree, and the pooled counters;
  * fits/s  -- the reference's convention n_sims / wall (throughput.rs:1111-1113: one optimistic pass, `off` mode),
               median of 3 timed run_batch calls at a replicate count grown until one call takes --target-ms (the
               reference times 500..10 000 replicates on one thread; a device call of that size is launch latency).

Writes one JSON line per case to --out and prints a markdown table."""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# per-thread figures the reference's source records for its own cases (throughput.rs:619, 625, 631)
REFERENCE_NOTED = {"glmm_crossed": 62.0, "glmm_nested": 27.0, "glmm_crossed_nested": 13.0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="", help="comma-separated ids (default: all 21)")
    ap.add_argument("--parity-sims", type=int, default=2000)
    ap.add_argument("--target-ms", type=float, default=400.0)
    ap.add_argument("--max-sims", type=int, default=2_000_000)
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--cpu-fits", type=int, default=0, help="also time the oracle on this many replicates per case (16 threads)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "throughput_cases.jsonl"))
    args = ap.parse_args()

    from mcpower_amd import engine, workloads
    cases = workloads.throughput_cases()
    ids = [c for c in args.cases.split(",") if c] or list(cases)
    oracle = None
    if not args.no_oracle:
        from oracle import oracle as _o
        _o.build()
        oracle = _o
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    rows = []
    with open(args.out, "w") as fh:
        for cid in ids:
            spec, n, ref_sims = cases[cid]
            row = {"case": cid, "n": n, "p": spec.n_predictors, "reference_n_sims": ref_sims}
            try:
                k = min(ref_sims, args.parity_sims)
                dev = engine.run_batch(spec, [n], k, workloads.BASE_SEED)
                row["converged"] = float(dev.converged.mean())
                row["power_x1"] = float(dev.uncorrected[:, 0, 0].mean())
                if oracle is not None:
                    ora = oracle.run_batch(spec, [n], k, workloads.BASE_SEED, n_threads=16)
                    same = (dev.uncorrected == ora.uncorrected).all(axis=(1, 2)) & (dev.converged == ora.converged).all(axis=1)
                    row["parity_sims"] = k
                    row["parity_identical"] = int(same.sum())
                    row["counters_identical"] = bool(np.array_equal(dev.counters, ora.counters))
                # grow the replicate count until one call takes target-ms, then median of 3
                s = max(ref_sims, 1024)
                while True:
                    t0 = time.perf_counter()
                    engine.run_batch(spec, [n], s, workloads.BASE_SEED, per_sim=False)
                    ms = (time.perf_counter() - t0) * 1e3
                    if ms >= args.target_ms or s >= args.max_sims:
                        break
                    s = min(args.max_sims, int(s * min(16.0, max(2.0, 1.2 * args.target_ms / max(ms, 1e-3)))))
                walls = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    engine.run_batch(spec, [n], s, workloads.BASE_SEED, per_sim=False)
                    walls.append(time.perf_counter() - t0)
                row["n_sims"] = s
                row["fits_per_s"] = s / statistics.median(walls)
                if oracle is not None and args.cpu_fits:
                    kc = min(args.cpu_fits, ref_sims * 4)
                    t0 = time.perf_counter()
                    oracle.run_batch(spec, [n], kc, workloads.BASE_SEED, n_threads=16)
                    row["oracle_fits_per_s_16t"] = kc / (time.perf_counter() - t0)
                if cid in REFERENCE_NOTED:
                    row["reference_noted_fits_per_s_1t"] = REFERENCE_NOTED[cid]
            except Exception as e:  # a refused case is a row of the table, not a crash of the sweep
                row["error"] = str(e)[:200]
            fh.write(json.dumps(row) + "\n")
            fh.flush()
            rows.append(row)
            print(json.dumps(row), flush=True)
    print("\n| case | n | p | replicates / call | fits/s (device) | oracle 16 threads | identical replicates |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        if "error" in r:
            print(f"| `{r['case']}` | {r['n']} | {r['p']} | — | refused: {r['error']} | | |")
            continue
        par = f"{r['parity_identical']} / {r['parity_sims']}" if "parity_sims" in r else "—"
        cpu = f"{r['oracle_fits_per_s_16t']:.3g}" if "oracle_fits_per_s_16t" in r else "—"
        print(f"| `{r['case']}` | {r['n']} | {r['p']} | {r['n_sims']} | {r['fits_per_s']:.3g} | {cpu} | {par} |")


if __name__ == "__main__":
    main()
