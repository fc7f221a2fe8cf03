#!/usr/bin/env python3
"""bench.py -- replicate-fits/sec of the simulate path on N MI355X GPUs (one process per GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE run_batch call (reference batch.rs:130) over a batch of 10^6 synthetic replicates
per GPU at BASELINE.json configs[1] (OLS, n=200, p=5: x1,x2,x3 + x1:x2; C2a in BASELINE.md), all
data generated in-kernel from Philox (there are no input tensors; the launch-constant block is
resident in HBM before the timed region).  Replicates shard by sim_id range with the same
base_seed (weak scaling: per-GPU work is fixed) and every step ends with the path's single
collective: one all-reduce (sum) of the rejection counters over RCCL -- issued INSIDE the C-ABI call
(`MCP_ALLREDUCE_COUNTERS`: ncclAllReduce on the device-resident vector, on the launch stream, before
the call's one device-to-host copy).  torch.distributed is the control plane only (rendezvous of the
128-byte communicator id, barriers, the max over ranks of the elapsed time).

Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` for the dominant
kernel of the timed region, `roofline_staged_fit` for the HBM-bound staged fit kernel, and
`cpu_baseline` (the CPU oracle timed on this host's cores, a reported baseline).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# chip constants: /opt/skills/guides/MI355X_MICROARCH.md (chip-level parameters)
PEAK_HBM_GBS = 8000.0          # HBM3E spec; ~6300 GB/s measured achievable
PEAK_VALU_TOPS = 78.6          # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz lane-ops/s (= 157.3 TFLOP/s fp32 FMA / 2)

# algorithmic work per replicate-fit
WORKLOADS = {
    # name: (spec factory, n, ops per fit for the fused kernel, staged bytes read per fit)
    "c2a": ("spec_c2a", 200, 3.76e4, 4 * 200 * 5),   # x1,x2,x3,x1:x2,y stored: 5 f32 columns
    "c2b": ("spec_c2b", 200, 5.26e4, 4 * 200 * 6),   # 5 predictors + y: the 4800 B of BASELINE.md
    "k4": ("spec_k4", 200, 4.3e4, 4 * 200 * 5),
    # the reference bench's widest case (throughput.rs:274-277, :512-517): k = 15, P = 16, n = 200 --
    # 800 Philox calls + 3 200 inverse CDFs + 2 x 200 x 152 f64 MAC: ~1.4e5 ops
    "ols_wide": ("spec_ols_wide", 200, 1.4e5, 4 * 200 * 16),      # tuning aid: 4 plain predictors (p=5, no interaction)
    # C3: logistic IRLS n=500 p=4: 2000 draws + ~5 passes x 500 rows x (exp, log1p, 2 div, 18 fma) ~ 5.0e5 ops
    "c3": ("spec_c3", 500, 5.0e5, 4 * 500 * 4),
    # the reference bench's widest IRLS case (throughput.rs:300-303, :536-541): k = 15, P = 16, n = 200 -- 800 Philox
    # calls + 3 000 inverse CDFs + ~6 passes x 200 rows x (2 x 152 MAC + exp, log1p, 2 div) ~ 5.2e5 ops
    "glm_wide": ("spec_glm_wide", 200, 5.2e5, 4 * 200 * 16),
    # C4: random-intercept REML, 30 x 20: 1800 draws + 30 RE draws + one 600-row pass + ~28 objective evaluations
    "c4": ("spec_c4", 600, 1.6e5, 4 * 600 * 3),
    # general mixed models (reference cases lmm_slope_a / lmm_crossed_a, validation/formulas.R:385-470): 960 / 1440
    # draws + ~300 / ~180 criterion evaluations of ~700 / ~1500 flops (block Cholesky of the penalised normal equations)
    # clustered logistic (reference case glmm_intercept, formulas.R:1092-1100): 1 230 draws + ~300 evaluations of the
    # Laplace deviance (~180 since the stationarity check replaced the unconditional restart), each ~6 Fisher-scoring sweeps over the 600 rows (exp, log1p, ~25 flops per row)
    "glmm": ("spec_glmm", 600, 2.6e7, 4 * 600 * 2),   # ~177 evaluations since the stationarity check (305 before)
    # ... with a random slope (reference case glmm_slope, formulas.R): ~900 evaluations, 2 x 2 penalised systems per cluster
    "glmm_slope": ("spec_glmm_slope", 600, 9.6e7, 4 * 600 * 2),   # ~400 evaluations (664 before)
    # ... and a crossed grouping (reference case glmm_slope_crossed, formulas.R:1130-1140): 54 random effects, joint modes
    "glmm_crossed": ("spec_glmm_crossed", 480, 7.4e8, 4 * 480 * 2),
    "lmm_slope": ("spec_lmm_slope", 480, 1.6e5, 4 * 480 * 2),
    "lmm_crossed": ("spec_lmm_crossed", 480, 2.0e5, 4 * 480 * 3),
    # C5: C2b design on the grid N = 50..1000 (20 prefix fits per replicate on shared draws) under the three
    # frozen scenario presets; ops per FIT = (6000 draws x 47 + 2 x 28000 MAC + 20 solves x 600) / 20 fits
    "c5_optimistic": ("spec_c2b", 1000, 1.75e4, 4 * 1000 * 6 // 20),
    "c5_realistic": ("spec_c2b", 1000, 4.5e4, 4 * 1000 * 6 // 20),
    "c5_doomer": ("spec_c2b", 1000, 3.5e4, 4 * 1000 * 6 // 20),
}


# the same algorithmic counts with an f64 multiply-add taken as ONE lane-op (the convention of PEAK_VALU_TOPS, where a
# v_fma is one issue slot): C2a 200 Philox calls x 50 + 800 inverse CDFs x 22 + 5 000 MAC = 3.26e4; C2b 300 x 50 +
# 1 200 x 22 + 5 600 = 4.70e4
OPS_FMA_AS_ONE = {"c2a": 3.26e4, "c2b": 4.70e4}


def pmc_traffic(workload, kernel_tag):
    """HBM bytes per launch of `kernel_tag` from the newest committed rocprofv3 PMC summary (profiles/*_pmc.csv,
    collected in separate --pmc passes by tools/profile_bench.sh).  gfx950 correction from
    /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE (KiB) reports exactly half of a wide coalesced streaming
    read, so reads = 2 x FETCH_SIZE x 1024; WRITE_SIZE (KiB) is exact for 16-B-per-lane stores."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{workload}_pmc.csv")))
    if not files:
        return None
    fetch = write = None
    for r in csv.DictReader(open(files[-1])):
        if kernel_tag in r["kernel"]:
            if r["counter"] == "FETCH_SIZE":
                fetch = float(r["mean_per_dispatch"])
            if r["counter"] == "WRITE_SIZE":
                write = float(r["mean_per_dispatch"])
    if fetch is None or write is None:
        return None
    return {"bytes_per_launch": 2.0 * fetch * 1024.0 + write * 1024.0, "read_bytes": 2.0 * fetch * 1024.0,
            "write_bytes": write * 1024.0, "source": os.path.basename(files[-1])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2a", choices=sorted(WORKLOADS))
    ap.add_argument("--sims-per-gpu", type=int, default=0, help="default: 10^6 (OLS) / 10^5 (c3) / 10^4 (c4)")
    ap.add_argument("--pipeline", default="fused", choices=["fused", "staged"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="capi", choices=["capi", "nccl", "gloo"],
                    help="capi (default): the collective runs inside the C ABI on RCCL, torch.distributed (gloo) is "
                         "the control plane; nccl: host counter blocks pooled through torch's RCCL process group "
                         "(the round-1 path); gloo: the same through gloo, to rehearse N>1 with several ranks on one GPU")
    ap.add_argument("--depth", type=int, default=3, help="batches in flight per rank (1..4)")
    ap.add_argument("--warm-ms", type=float, default=200.0,
                    help="device warm-up: untimed steps of the same workload are issued after the W warm-up steps until this "
                         "much wall time has gone into warm-up -- an idle MI355X needs ~50 ms of work to reach its "
                         "sustained clocks (C2a: 1.07 ms per step after 5 warm-up steps, 0.95 ms after 50; DESIGN.md "
                         "section 7); 0 = exactly W steps")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from mcpower_amd import dist as mdist
    from mcpower_amd import engine
    from mcpower_amd import workloads as helpers

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if engine.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    dev_index = local_rank
    if args.backend == "gloo" or os.environ.get("MCP_BENCH_SHARE_GPU") == "1":
        dev_index = local_rank % engine.device_count()  # rehearsal of N > 1 on a box with fewer GPUs than ranks
    torch.cuda.set_device(dev_index)
    engine.set_device(dev_index)
    capi = args.backend == "capi" and world > 1
    comm_fallback = False
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":  # host blocks through torch's RCCL process group
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:                       # control plane only
            dist.init_process_group(backend="gloo")
        if capi:
            # the product's communicator: rank 0 creates the id, gloo carries it, every rank joins; mcp_comm_init
            # returns after a one-word all-reduce has crossed the rings, so nothing of the bring-up is timed
            ok = 1
            try:
                mdist.init_comm(rank, world)
            except Exception as e:
                ok = 0
                print(f"[bench] rank {rank}: the C-ABI communicator did not come up ({e}); if any rank reports this, "
                      "every rank pools HOST counter blocks through the gloo group instead", file=sys.stderr, flush=True)
            agreed = torch.tensor([ok], dtype=torch.int64)
            dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
            if int(agreed.item()) == 0:  # all ranks leave the device-side collective together
                if ok:
                    engine.comm_destroy()
                capi = False
                comm_fallback = True
        else:
            t_init = torch.zeros(1, dtype=torch.int64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t_init)
        dist.barrier()

    fac, n_obs, ops_per_fit, bytes_per_fit = WORKLOADS[args.workload]
    spec = getattr(helpers, fac)()
    ns = [n_obs]
    if args.workload.startswith("c5_"):
        spec.scenario = helpers.preset(args.workload[3:])
        ns = helpers.C5_GRID
    S = args.sims_per_gpu or ({"c3": 100_000, "glm_wide": 100_000, "c4": 10_000, "lmm_slope": 40_000, "lmm_crossed": 40_000, "glmm": 40_000, "glmm_slope": 16_000, "glmm_crossed": 4_000}.get(args.workload, 100_000 if args.workload.startswith("c5_") else 1_000_000))
    staged = args.pipeline == "staged"

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    lowered = spec.to_c()  # marshalled once, like the reference holding its SimulationSpec across run_batch calls

    def submit(i):
        # weak scaling: every rank simulates S new replicates per step; ids never repeat across steps
        off = (i * world + rank) * S
        return engine.submit_batch(lowered, ns, S, helpers.BASE_SEED, sim_offset=off, staged=staged, allreduce=capi)

    depth = max(1, min(int(args.depth), 4))

    def run_steps(first, count):
        """`count` steps, software-pipelined `depth` deep: up to `depth` batches are in flight, each on its own stream
        (the host-side lowering and upload of batch i+1 hide behind batch i's kernel, and the partly filled last round
        of batch i's workgroups shares the chip with the head of batch i+1).  With the C-ABI communicator the
        all-reduce of a batch's counters is part of that batch's stream; otherwise the host blocks are pooled one step
        later through the torch process group.  Every batch and every collective completes before this returns."""
        acc = dict(kernel_ms=0.0, aux_ms=0.0, launches=0, pooled_sims=0.0, pooled=None, last=None, spans=[])
        if count <= 0:
            return acc

        def absorb(pend):
            acc["pooled"] = pend.wait() if hasattr(pend, "wait") else pend
            acc["pooled_sims"] += float(acc["pooled"][0, acc["last"].layout["n_sims"]])

        t_sub = t_col = 0.0
        tickets = [submit(first + j) for j in range(min(depth, count))]
        issued = len(tickets)
        reduce_in_flight = None
        for i in range(count):
            ta = time.perf_counter()
            r = tickets.pop(0).collect()
            tb = time.perf_counter()
            if issued < count:
                tickets.append(submit(first + issued))
                issued += 1
            t_sub += time.perf_counter() - tb
            t_col += tb - ta
            acc["last"] = r
            acc["kernel_ms"] += r.kernel_ms
            acc["aux_ms"] += r.aux_kernel_ms
            acc["launches"] += r.n_kernel_launches
            acc["spans"].append((r.kernel_begin_ms, r.kernel_end_ms))
            if capi or world == 1:
                absorb(r.counters)  # already pooled on the device (or there is nothing to pool)
            else:
                if reduce_in_flight is not None:
                    absorb(reduce_in_flight)
                reduce_in_flight = mdist.allreduce_counters_async(r.counters)
        if reduce_in_flight is not None:
            absorb(reduce_in_flight)
        acc["host_submit_ms"], acc["host_collect_ms"] = 1e3 * t_sub / count, 1e3 * t_col / count
        return acc

    def busy_ms(spans):
        """Union of the dominant-kernel intervals: batches in flight overlap on the device, so the kernel-busy time of
        the region is not the sum of the per-launch durations."""
        total, cur_b, cur_e = 0.0, None, None
        for b, e in sorted(spans):
            if cur_e is None or b > cur_e:
                if cur_e is not None:
                    total += cur_e - cur_b
                cur_b, cur_e = b, e
            else:
                cur_e = max(cur_e, e)
        return total + ((cur_e - cur_b) if cur_e is not None else 0.0)

    # W untimed warm-up steps; every batch slot in flight owns its own device arena, so at least `depth` batches must
    # have been issued before the timed region or a first-use hipMalloc lands inside it
    n_warm = max(args.warmup, depth)
    tw = time.perf_counter()
    run_steps(0, n_warm)
    barrier()
    first_ms = 1e3 * (time.perf_counter() - tw)
    if world > 1:
        tt = torch.tensor([first_ms], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        first_ms = float(tt.item())
    if args.warm_ms > 0 and first_ms < 10.0 * args.warm_ms:  # (a first block that long has warmed the device by itself)
        # ... then the same steps, still untimed, until the clocks have ramped.  The first block above also paid the
        # one-time costs (code-object load, first-use hipMalloc of every batch slot), so the step time is taken from a
        # second block; every rank must issue the same number of steps (each carries the collective), so the count
        # comes from the slowest rank
        tb = time.perf_counter()
        run_steps(n_warm, depth)
        barrier()
        n_warm += depth
        per_step = 1e3 * (time.perf_counter() - tb) / depth
        if world > 1:
            tt = torch.tensor([per_step], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            per_step = float(tt.item())
        extra = int(min(2000, args.warm_ms / max(per_step, 1e-3)))
        if extra:
            run_steps(n_warm, extra)
            n_warm += extra
            barrier()
    warm_ms = 1e3 * (time.perf_counter() - tw)
    t0 = time.perf_counter()
    acc = run_steps(n_warm, args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, aux_ms, launches, pooled, r = acc["kernel_ms"], acc["aux_ms"], acc["launches"], acc["pooled"], acc["last"]
    if acc["pooled_sims"] != float(S) * world * args.steps:
        raise SystemExit(f"pooled replicate count {acc['pooled_sims']} != {float(S) * world * args.steps}")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_fits = float(S) * world * args.steps * len(ns)
        value = total_fits / elapsed
        nt = spec.n_targets
        k_unc = pooled[0, r.layout["k_unc"]:r.layout["k_unc"] + nt].astype(np.float64)
        n_pool = float(pooled[0, r.layout["n_sims"]])
        avg_launch_ms = kernel_ms / max(launches, 1)          # HIP events on each launch's stream (host_api.hip)
        kernel_busy_ms = busy_ms(acc["spans"])                # union of those intervals over the timed region
        concurrency = kernel_ms / kernel_busy_ms if kernel_busy_ms > 0 else 1.0
        fits_per_launch = S / max(launches / args.steps, 1)
        out = {
            "metric": "replicate-fits/sec", "value": value, "unit": "fits/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "warmup_steps_issued": n_warm, "warmup_ms": warm_ms,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 data plane / f64 fit", "data": "synthetic",
            "config": {"workload": f"{args.workload}: { {'c3': 'GLM', 'glm_wide': 'GLM', 'c4': 'LMM', 'lmm_slope': 'LMM', 'lmm_crossed': 'LMM', 'glmm': 'GLMM', 'glmm_slope': 'GLMM', 'glmm_crossed': 'GLMM'}.get(args.workload, 'OLS') } n={n_obs} p={spec.n_predictors} "
                                   f"({ {'c2a': 'x1,x2,x3 + x1:x2', 'c2b': 'ols_multi, 5 predictors', 'k4': '4 predictors', 'ols_wide': 'ols_wide of the reference bench, 15 predictors', 'c3': 'logit IRLS, 3 predictors', 'glm_wide': 'glm_wide of the reference bench, logit IRLS, 15 predictors', 'c4': 'random-intercept REML, 30 clusters x 20', 'glmm': 'logit, Laplace, (1 | g), 30 clusters x 20', 'glmm_slope': 'logit, Laplace, (1 + x1 | g), 30 clusters x 20', 'glmm_crossed': 'logit, Laplace, (1 + x1 | g) + (1 | item), 24 x 6 levels', 'lmm_slope': 'REML, (1 + x1 | g), 24 clusters x 20', 'lmm_crossed': 'REML, (1 | g) + (1 | item), 20 x 12 levels'}.get(args.workload, 'ols_multi on the grid N=50..1000 (20 fits per replicate), scenario ' + args.workload[3:]) }), "
                                   f"{S} replicates per GPU per step, seed 2137",
                       "pipeline": args.pipeline, "parallelism": f"replicate-sharded x{world}",
                       "collective": ("1 ncclAllReduce(sum, uint64) of the device-resident counters per step, inside the C ABI"
                                      if capi or world == 1 else f"1 all-reduce(sum) of host counter blocks per step via torch ({'gloo' if args.backend == 'capi' else args.backend})"
                                      + (" -- FALLBACK: the C-ABI RCCL communicator did not come up on every rank (stderr)" if comm_fallback else "")),
                       "batches_in_flight": depth,
                       "host_ms_per_step": {"submit": acc["host_submit_ms"], "collect_wait": acc["host_collect_ms"]}},
            "power_last_step": (k_unc / n_pool).round(5).tolist(),
        }
        is_ols = args.workload not in ("c3", "glm_wide", "c4", "lmm_slope", "lmm_crossed", "glmm", "glmm_slope", "glmm_crossed")
        is_c5 = args.workload.startswith("c5_")
        if not staged:
            fits_per_launch *= len(ns)  # a replicate on a grid is len(ns) prefix fits (throughput.rs:1111-1113)
            # launches overlap (depth > 1: each batch on its own stream), so the rate of the kernel over the timed region
            # is work / kernel-busy time = per-launch work x mean concurrency / average launch duration
            ach = ops_per_fit * fits_per_launch * concurrency / (avg_launch_ms * 1e-3) / 1e12
            ops_1op = OPS_FMA_AS_ONE.get(args.workload)
            out["roofline"] = {"bound": "valu", "kernel": ((f"ols_sim_kernel<{spec.n_predictors},fused>" if spec.n_predictors <= 8 else f"ols_reduced_kernel<{spec.n_predictors}>") if is_ols else (f"glm_sim_kernel<{spec.n_predictors}>" if args.workload in ("c3", "glm_wide") else (f"lmm_general_kernel<{spec.n_predictors}>" if args.workload.startswith("lmm_") else f"glmm_kernel<{spec.n_predictors}>" if args.workload == "glmm" else f"glmm_slopes_kernel<{spec.n_predictors},2>" if args.workload == "glmm_slope" else f"glmm_general_kernel<{spec.n_predictors}>" if args.workload == "glmm_crossed" else f"lmm_prepare_kernel<{spec.n_predictors}> + lmm_optimise_kernel<{spec.n_predictors}>"))),
                               "achieved": ach, "peak": PEAK_VALU_TOPS, "unit": "Top/s", "frac": ach / PEAK_VALU_TOPS,
                               "frac_fma_as_one_op": (ops_1op / ops_per_fit * ach / PEAK_VALU_TOPS) if ops_1op else None,
                               "ops_per_fit_fma_as_one_op": ops_1op,
                               "kernel_busy_ms": kernel_busy_ms, "mean_concurrent_launches": concurrency,
                               "traffic": (pmc_traffic(args.workload, f"ols_sim_kernel<{spec.n_predictors}, 0,")
                                           if (is_ols and not is_c5 and S == 1_000_000) else None),
                               "ops_per_fit": ops_per_fit, "avg_launch_ms": avg_launch_ms,
                               "fits_per_launch": fits_per_launch,
                               "note": "fused generate+fit: no design bytes touch HBM; bound is VALU/integer issue "
                                       "(SURVEY.md 8d); achieved = ops_per_fit x fits_per_launch x mean_concurrent_launches "
                                       "/ avg_launch_ms (launches of batches in flight overlap; avg_launch_ms is the per-launch "
                                       "HIP-event duration rocprofv3 --kernel-trace also reports); ops_per_fit counts an f64 MAC "
                                       "as 2 ops, frac_fma_as_one_op counts it as 1 like the lane-op peak does; "
                                       "HBM-equivalent rate in hbm_equiv_gbs",
                               "hbm_equiv_gbs": bytes_per_fit * fits_per_launch * concurrency / (avg_launch_ms * 1e-3) / 1e9}
        if not staged and is_ols and not is_c5 and spec.n_predictors <= 8:  # the staged pipeline serves the register-resident kernels
            # the HBM-bound kernel of the path: staged fit over a resident design tile (measured outside the timed region)
            rs = engine.run_batch(spec, ns, S, helpers.BASE_SEED, per_sim=False, staged=True)
            rs = engine.run_batch(spec, ns, S, helpers.BASE_SEED, per_sim=False, staged=True)
            fit_ms = rs.kernel_ms / rs.n_kernel_launches
            fpl = S / rs.n_kernel_launches
            gbs = bytes_per_fit * fpl / (fit_ms * 1e-3) / 1e9
            tr = pmc_traffic(args.workload, f"ols_sim_kernel<{spec.n_predictors}, 2,") if S == 1_000_000 else None
            out["roofline_staged_fit"] = {"bound": "hbm", "kernel": f"ols_sim_kernel<{spec.n_predictors},fit>",
                                          "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                          "frac": gbs / PEAK_HBM_GBS, "traffic": tr, "bytes_per_fit": bytes_per_fit,
                                          "algorithmic_bytes_per_launch": bytes_per_fit * fpl,
                                          "avg_launch_ms": fit_ms, "generator_launch_ms": rs.aux_kernel_ms / rs.n_kernel_launches,
                                          "fits_per_launch": fpl}
        if staged:
            gbs = bytes_per_fit * fits_per_launch / (avg_launch_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": f"ols_sim_kernel<{spec.n_predictors},fit>", "achieved": gbs,
                               "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "traffic": None,
                               "bytes_per_fit": bytes_per_fit, "avg_launch_ms": avg_launch_ms,
                               "generator_launch_ms": aux_ms / max(launches, 1), "fits_per_launch": fits_per_launch}
        if not args.no_cpu_baseline and world == 1:
            # the CPU oracle ("port"): same spec, same seeds, on this host's cores, bounded sample
            from oracle import oracle
            flags = oracle.use_native_build()  # -O3 -march=native, built here on the timing host
            # the GPU box hands a one-GPU job a 16-core share of its host (os.cpu_count() reports the whole machine:
            # 256 threads oversubscribing 16 cores ran 30 % slower than 16)
            cores = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16))
            probe = 50_000 if is_ols else 2_000  # iterative fits run 1e2..1e4 per second: keep the probe in seconds
            t1 = time.perf_counter()
            oracle.run_batch(spec, ns, probe, helpers.BASE_SEED, n_threads=cores)
            rate = probe / (time.perf_counter() - t1)
            sample = int(min(max(rate * args.cpu_seconds, probe), 40_000_000))
            t1 = time.perf_counter()
            o = oracle.run_batch(spec, ns, sample, helpers.BASE_SEED, n_threads=cores)
            dt = time.perf_counter() - t1
            g = engine.run_batch(spec, ns, sample, helpers.BASE_SEED, per_sim=False)
            out["cpu_baseline"] = {"value": sample / dt, "unit": "fits/s", "cores": cores, "kind": "port",
                                   "compiler_flags": "gcc " + flags,
                                   "sample": f"{sample} replicates of the same workload (sim_id 0..{sample - 1}), "
                                             f"oracle/ C restatement, {cores} pthreads, {dt:.1f} s",
                                   "power_delta_pp_vs_gpu": float(np.max(np.abs(
                                       o.counters[0, :nt].astype(np.float64) - g.counters[0, :nt].astype(np.float64)))
                                       / sample * 100.0)}
        print(json.dumps(out))
    if world > 1:
        if capi:
            engine.comm_destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
