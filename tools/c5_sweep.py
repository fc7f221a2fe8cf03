#!/usr/bin/env python3
"""BASELINE configs[4] end to end: sample-size sweep N = 50..1000 by 50 under the three robustness scenarios,
through `_engine.find_sample_size` with the replicates sharded over the ranks of the process group (one process
per GPU: `python -m torch.distributed.run --nproc-per-node G tools/c5_sweep.py --sims 10000000`; a plain
`python tools/c5_sweep.py` is the one-GPU case).  Counters only: no per-sim tensor exists anywhere."""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sims", type=int, default=1_250_000, help="replicates per scenario over ALL ranks")
    args = ap.parse_args()
    import msgpack
    import torch
    import torch.distributed as dist
    from mcpower_amd import _engine as E, engine

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    engine.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    base = msgpack.unpackb(open(os.path.join(ROOT, "tests", "golden", "contracts", "ols_contract.msgpack"), "rb").read(),
                           raw=False)[0]
    # ols_multi of the reference bench (throughput.rs:269-272): five predictors, x1 = x2 = 0.18
    base["generation"]["columns"] = [base["generation"]["columns"][0]] * 5
    base["design_generation"]["terms"] = ["Const"] + [{"Direct": {"column": j}} for j in range(5)]
    base["outcome"]["coefficients"] = [0.0, 0.18, 0.18, 0.0, 0.0, 0.0]
    base["test"]["targets"] = [{"kind": "marginal", "term": 1}, {"kind": "marginal", "term": 2}]
    presets = json.load(open(os.path.join(ROOT, "tests", "golden", "scenario_presets.json")))
    contracts = []
    for name in ("optimistic", "realistic", "doomer"):
        c = copy.deepcopy(base)
        c["scenario"].update(presets[name])
        c["scenario"]["name"] = name
        contracts.append(c)
    blob = msgpack.packb(contracts, use_bin_type=True)
    runner = E.sharded_runner(rank, world)
    E.find_sample_size(blob, 80.0, 50, 1000, 4096, 2137, "grid", 50, "fixed", "linear", None, None, _runner=runner)  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = E.find_sample_size(blob, 80.0, 50, 1000, args.sims, 2137, "grid", 50, "fixed", "linear", None, None, _runner=runner)
    dt = time.perf_counter() - t0
    if rank == 0:
        fits = args.sims * 20 * 3
        print(json.dumps({"workload": "c5 sweep: ols_multi, N = 50..1000 by 50, optimistic + realistic + doomer",
                          "sims_per_scenario": args.sims, "n_gpus": world, "wall_s": dt, "prefix_fits_per_s": fits / dt,
                          "required_n": {k: {t: v.get("n_achievable") for t, v in sc["fitted"].items()}
                                         for k, sc in r["scenarios"].items()}}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
