#!/usr/bin/env python3
"""BASELINE configs[4] end to end: sample-size sweep N = 50..1000 by 50 under the three robustness scenarios,
through `_engine.find_sample_size` with the replicates sharded over the ranks of the process group (one process
per GPU: `python -m torch.distributed.run --nproc-per-node G tools/c5_sweep.py --sims 10000000`; a plain
`python tools/c5_sweep.py` is the one-GPU case).  Counters only: no per-sim tensor exists anywhere.

`--check` makes every rank ALSO run the whole sweep unsharded and assert that the sharded result -- every count, rate,
interval and required N of every scenario -- equals it (world 1 and, over gloo with both ranks on one GPU, world 2:
`python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/c5_sweep.py --backend gloo --check
--sims 20000`).  `--backend capi` (default) pools the counters inside the C ABI over RCCL."""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sims", type=int, default=1_250_000, help="replicates per scenario over ALL ranks")
    ap.add_argument("--backend", default="capi", choices=["capi", "gloo"])
    ap.add_argument("--check", action="store_true", help="assert sharded == unsharded on every rank")
    args = ap.parse_args()
    import msgpack
    import torch
    import torch.distributed as dist
    from mcpower_amd import _engine as E, dist as mdist, engine

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local = local if args.backend == "capi" else local % engine.device_count()
    torch.cuda.set_device(local)
    engine.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")  # control plane; the counters travel over RCCL inside the C ABI
        if args.backend == "capi":
            mdist.init_comm(rank, world)
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
    runner = E.sharded_runner(rank, world, allow_host_pool=True)  # world 2 on one GPU is a gloo rehearsal; with a communicator the C ABI pools
    E.find_sample_size(blob, 80.0, 50, 1000, 4096, 2137, "grid", 50, "fixed", "linear", None, None, _runner=runner)  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = E.find_sample_size(blob, 80.0, 50, 1000, args.sims, 2137, "grid", 50, "fixed", "linear", None, None, _runner=runner)
    dt = time.perf_counter() - t0
    checked = None
    if args.check:
        whole = E.find_sample_size(blob, 80.0, 50, 1000, args.sims, 2137, "grid", 50, "fixed", "linear", None, None)
        a = json.loads(json.dumps(r["scenarios"], sort_keys=True, default=str))
        b = json.loads(json.dumps(whole["scenarios"], sort_keys=True, default=str))

        def strip(d):  # per-sim-only fields have no pooled counterpart (tests/test_sharded_engine.py SKIP)
            if isinstance(d, dict):
                return {k: strip(v) for k, v in d.items() if k not in ("boundary_hit", "estimator_extras")}
            if isinstance(d, list):
                return [strip(v) for v in d]
            return d
        assert strip(a) == strip(b), f"rank {rank}: sharded sweep differs from the unsharded one"
        checked = True
    if rank == 0:
        fits = args.sims * 20 * 3
        print(json.dumps({"workload": "c5 sweep: ols_multi, N = 50..1000 by 50, optimistic + realistic + doomer",
                          "sims_per_scenario": args.sims, "n_gpus": world, "backend": args.backend,
                          "sharded_equals_unsharded": checked, "wall_s": dt, "prefix_fits_per_s": fits / dt,
                          "required_n": {k: {t: v.get("n_achievable") for t, v in sc["fitted"].items()}
                                         for k, sc in r["scenarios"].items()}}))
    if world > 1:
        if args.backend == "capi":
            engine.comm_destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
