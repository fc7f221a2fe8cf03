#!/usr/bin/env python3
"""bench.py -- replicate-fits/sec of the simulate path on N MI355X GPUs (one process per GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE run_batch call (reference batch.rs:130) over a batch of 10^6 synthetic replicates
per GPU at BASELINE.json configs[1] (OLS, n=200, p=5: x1,x2,x3 + x1:x2; C2a in BASELINE.md), all
data generated in-kernel from Philox (there are no input tensors; the launch-constant block is
resident in HBM before the timed region).  Replicates shard by sim_id range with the same
base_seed (weak scaling: per-GPU work is fixed; `--scaling strong --total-sims T` fixes the total
instead, BASELINE configs[4]) and every step ends with the path's single collective: one all-reduce
(sum) of the rejection counters over RCCL -- issued INSIDE the C-ABI call (`MCP_ALLREDUCE_COUNTERS`:
ncclAllReduce on the device-resident vector, on the launch stream, before the call's one
device-to-host copy).  torch.distributed (gloo) is the control plane only (rendezvous of the 128-byte
communicator id, barriers, the max over ranks of the elapsed time).  If the C-ABI communicator does
not come up on every rank the run EXITS NON-ZERO (`--allow-host-fallback` pools host counter blocks
through gloo instead and says so in `rccl_ranks` = 0).

Timing: after the warm-up, `--blocks` (default 5) timed regions of EXACTLY K steps each, every one
bracketed by barrier + torch.cuda.synchronize() on both sides and reduced by MAX over ranks; `value`
and `ms_per_step` are the MEDIAN block's, `blocks` carries every block and the spread.

Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` for the dominant
kernel of the timed region (`frac` counts an f64 multiply-add as ONE lane-op, the unit of the peak),
`roofline_staged_fit` for the HBM-bound staged fit kernel, `cpu_baseline` (the CPU oracle timed on
this host's cores, a reported baseline), `rccl` (communicator evidence) and -- default workload,
one GPU -- `configs`: the other BASELINE configurations (C2b, C3, C4, C5 under the three presets),
each measured the same way outside the C2a timed regions with its power delta against the CPU oracle
on a 2 000-replicate head.
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

# Algorithmic work per replicate-fit.  `ops1` counts an f64 multiply-add as ONE
# lane-op (the convention of PEAK_VALU_TOPS, where a v_fma is one issue slot) and is what `roofline.frac` uses;
# `ops2` counts it as two flops (SURVEY 8d's "ops").  `bytes` = staged design bytes read per fit.
#   C2a: 200 Philox calls x 50 + 800 inverse CDFs x 22 + 5 000 MAC;   C2b: 300 x 50 + 1 200 x 22 + 5 600 MAC
#   C3 : 2 000 draws + ~5 passes x 500 rows x (exp, log1p, 2 div, 18 MAC);   C4: 1 830 draws + 600-row pass + ~28 evaluations
#   C5 : per prefix FIT = (6 000 draws x 47 + 28 000 MAC + 20 solves x 600) / 20, presets add the transform / extra residual draws
W = lambda **kw: kw  # noqa: E731
WORKLOADS = {
    "c2a": W(fac="spec_c2a", n=200, ops2=3.76e4, ops1=3.26e4, bytes=4 * 200 * 5, S=1_000_000, kind="OLS", desc="x1,x2,x3 + x1:x2"),
    "c2b": W(fac="spec_c2b", n=200, ops2=5.26e4, ops1=4.70e4, bytes=4 * 200 * 6, S=1_000_000, kind="OLS", desc="ols_multi, 5 predictors"),
    "k4": W(fac="spec_k4", n=200, ops2=4.3e4, ops1=3.9e4, bytes=4 * 200 * 5, S=1_000_000, kind="OLS", desc="4 predictors"),
    # the reference bench's widest case (throughput.rs:274-277): k = 15, P = 16: 800 Philox calls + 3 200 inverse CDFs + 200 x 152 MAC
    "ols_wide": W(fac="spec_ols_wide", n=200, ops2=1.4e5, ops1=1.1e5, bytes=4 * 200 * 16, S=1_000_000, kind="OLS", desc="ols_wide of the reference bench, 15 predictors"),
    "c3": W(fac="spec_c3", n=500, ops2=5.0e5, ops1=4.55e5, bytes=4 * 500 * 4, S=100_000, kind="GLM", desc="logit IRLS, 3 predictors"),
    "glm_wide": W(fac="spec_glm_wide", n=200, ops2=5.2e5, ops1=3.4e5, bytes=4 * 200 * 16, S=100_000, kind="GLM", desc="glm_wide of the reference bench, logit IRLS, 15 predictors"),
    "c4": W(fac="spec_c4", n=600, ops2=1.6e5, ops1=1.5e5, bytes=4 * 600 * 3, S=10_000, kind="LMM", desc="random-intercept REML, 30 clusters x 20"),
    "glmm": W(fac="spec_glmm", n=600, ops2=2.6e7, ops1=2.4e7, bytes=4 * 600 * 2, S=40_000, kind="GLMM", desc="logit, Laplace, (1 | g), 30 clusters x 20"),
    "glmm_slope": W(fac="spec_glmm_slope", n=600, ops2=9.6e7, ops1=9.0e7, bytes=4 * 600 * 2, S=16_000, kind="GLMM", desc="logit, Laplace, (1 + x1 | g), 30 clusters x 20"),
    "glmm_crossed": W(fac="spec_glmm_crossed", n=480, ops2=7.4e8, ops1=6.0e8, bytes=4 * 480 * 2, S=4_000, kind="GLMM", desc="logit, Laplace, (1 + x1 | g) + (1 | item), 24 x 6 levels"),
    "lmm_slope": W(fac="spec_lmm_slope", n=480, ops2=1.6e5, ops1=1.2e5, bytes=4 * 480 * 2, S=40_000, kind="LMM", desc="REML, (1 + x1 | g), 24 clusters x 20"),
    "lmm_crossed": W(fac="spec_lmm_crossed", n=480, ops2=2.0e5, ops1=1.5e5, bytes=4 * 480 * 3, S=40_000, kind="LMM", desc="REML, (1 | g) + (1 | item), 20 x 12 levels"),
    "c5_optimistic": W(fac="spec_c2b", n=1000, ops2=1.75e4, ops1=1.61e4, bytes=4 * 1000 * 6 // 20, S=100_000, kind="OLS", desc=""),
    "c5_realistic": W(fac="spec_c2b", n=1000, ops2=4.5e4, ops1=4.36e4, bytes=4 * 1000 * 6 // 20, S=100_000, kind="OLS", desc=""),
    "c5_doomer": W(fac="spec_c2b", n=1000, ops2=3.5e4, ops1=3.36e4, bytes=4 * 1000 * 6 // 20, S=100_000, kind="OLS", desc=""),
}
# the BASELINE configurations reported beside the headline (name in the JSON line -> workload, steps per block)
CONFIG_MATRIX = (("c2b", "c2b", 20), ("c3", "c3", 20), ("c4", "c4", 60), ("c5_optimistic", "c5_optimistic", 10),
                 ("c5_realistic", "c5_realistic", 5), ("c5_doomer", "c5_doomer", 5))


def kernel_name(workload, p):
    if workload in ("c3", "glm_wide"):
        return f"glm_sim_kernel<{p}>"
    if workload == "c4":
        return f"lmm_prepare_kernel<{p}> + lmm_optimise_kernel<{p}>"
    if workload.startswith("lmm_"):
        return f"lmm_general_kernel<{p}>"
    if workload.startswith("glmm"):
        return {"glmm": f"glmm_kernel<{p}>", "glmm_slope": f"glmm_slopes_kernel<{p},2>"}.get(workload, f"glmm_general_kernel<{p}>")
    if workload in ("c5_realistic", "c5_doomer"):  # general data path: split launch (ols_sim_impl.hpp, SPLIT)
        return f"ols_sim_kernel<{p},rows,general> + ols_fit_rec_kernel<{p}>"
    return f"ols_sim_kernel<{p},fused>" if p <= 12 else f"ols_wide_kernel<{p}>"


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=5, help="timed regions of exactly --steps steps; value = the median block")
    ap.add_argument("--workload", default="c2a", choices=sorted(WORKLOADS))
    ap.add_argument("--sims-per-gpu", type=int, default=0, help="default: 10^6 (OLS) / 10^5 (c3, c5) / 10^4 (c4)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--total-sims", type=int, default=10_000_000, help="--scaling strong: replicates per step over ALL ranks (BASELINE configs[4]: 10^7)")
    ap.add_argument("--pipeline", default="fused", choices=["fused", "staged"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the C2b / C3 / C4 / C5 lines reported beside the default workload")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="capi", choices=["capi", "nccl", "gloo"],
                    help="capi (default): the collective runs inside the C ABI on RCCL, torch.distributed (gloo) is "
                         "the control plane; nccl: host counter blocks pooled through torch's RCCL process group "
                         "(the round-1 path); gloo: the same through gloo, to rehearse N>1 with several ranks on one GPU")
    ap.add_argument("--allow-host-fallback", action="store_true",
                    help="capi only: when the C-ABI communicator does not come up on every rank, pool host counter blocks "
                         "through gloo instead of exiting non-zero (the JSON line then carries rccl_ranks = 0)")
    ap.add_argument("--depth", type=int, default=3, help="batches in flight per rank (1..4)")
    ap.add_argument("--warm-ms", type=float, default=200.0,
                    help="device warm-up: untimed steps of the same workload are issued after the W warm-up steps until this "
                         "much wall time has gone into warm-up -- an idle MI355X needs ~50 ms of work to reach its "
                         "sustained clocks (C2a: 1.07 ms per step after 5 warm-up steps, 0.95 ms after 50; DESIGN.md "
                         "section 7); 0 = exactly W steps")
    args = ap.parse_args()

    # stdout carries ONE JSON line and nothing else: libraries bound later write banners there (RCCL prints its version
    # block to stdout when a communicator comes up), so fd 1 points at stderr for the run and the line goes to the real one
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    # multi-process GPU work on this pool needs dmabuf IPC (the host driver has no legacy IPC: without it RCCL's bring-up fails
    # with hipIpcGetMemHandle: invalid argument); the launcher's environment normally carries it already
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

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
    ctl_dev = "cuda" if args.backend == "nccl" else "cpu"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":  # host blocks through torch's RCCL process group
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:                       # control plane only
            dist.init_process_group(backend="gloo")
        if capi:
            # the product's communicator: rank 0 creates the id, gloo carries it, every rank joins; mcp_comm_init
            # returns after a one-word all-reduce has crossed the rings, so nothing of the bring-up is timed
            ok, why = 1, ""
            try:
                mdist.init_comm(rank, world)
            except Exception as e:
                ok, why = 0, str(e)
                print(f"[bench] rank {rank}: the C-ABI communicator did not come up ({e})", file=sys.stderr, flush=True)
            agreed = torch.tensor([ok], dtype=torch.int64)
            dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
            if int(agreed.item()) == 0:  # all ranks leave the device-side collective together
                if ok:
                    engine.comm_destroy()
                if not args.allow_host_fallback:
                    dist.barrier()
                    dist.destroy_process_group()
                    raise SystemExit(f"[bench] rank {rank}: RCCL communicator missing on at least one rank"
                                     f"{' (' + why + ')' if why else ''}; a number measured without it would not be the path's "
                                     "collective -- rerun with --allow-host-fallback to pool host blocks through gloo instead")
                capi = False
                comm_fallback = True
        else:
            t_init = torch.zeros(1, dtype=torch.int64, device=ctl_dev)
            dist.all_reduce(t_init)
        dist.barrier()
    rccl_ranks = engine.comm_world() if capi else 0

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    depth = max(1, min(int(args.depth), 4))
    strong = args.scaling == "strong"

    def setup(workload, sims):
        wl = WORKLOADS[workload]
        spec = getattr(helpers, wl["fac"])()
        ns = [wl["n"]]
        if workload.startswith("c5_"):
            spec.scenario = helpers.preset(workload[3:])
            ns = helpers.C5_GRID
        return wl, spec, ns, (sims or wl["S"])

    def make_stepper(spec, ns, S, staged, pooled_dev):
        """Returns run_steps(first, count): `count` steps, software-pipelined `depth` deep -- up to `depth` batches in
        flight, each on its own stream (the host-side lowering and upload of batch i+1 hide behind batch i's kernel,
        and the partly filled last round of batch i's workgroups shares the chip with the head of batch i+1).  With
        the C-ABI communicator the all-reduce of a batch's counters is part of that batch's stream; otherwise the
        host blocks are pooled one step later through the torch process group.  Every batch and every collective
        completes before run_steps returns.  Weak scaling: rank r simulates S NEW replicates per step; strong: the
        step's `S` replicates (the whole job's) are split by mcp_shard_range; sim ids never repeat across steps."""
        lowered = spec.to_c()  # marshalled once, like the reference holding its SimulationSpec across run_batch calls

        def submit(i):
            if strong:
                off, cnt = engine.shard_range(S, rank, world)
                return engine.submit_batch(lowered, ns, cnt, helpers.BASE_SEED, sim_offset=i * S + off, staged=staged, allreduce=pooled_dev)
            off = (i * world + rank) * S
            return engine.submit_batch(lowered, ns, S, helpers.BASE_SEED, sim_offset=off, staged=staged, allreduce=pooled_dev)

        def run_steps(first, count):
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
            for _ in range(count):
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
                if pooled_dev or world == 1:
                    absorb(r.counters)  # already pooled on the device (or there is nothing to pool)
                else:
                    if reduce_in_flight is not None:
                        absorb(reduce_in_flight)
                    reduce_in_flight = mdist.allreduce_counters_async(r.counters)
            if reduce_in_flight is not None:
                absorb(reduce_in_flight)
            acc["host_submit_ms"], acc["host_collect_ms"] = 1e3 * t_sub / count, 1e3 * t_col / count
            return acc

        return run_steps

    def measure(workload, sims, steps, n_blocks, warmup, warm_ms, staged=False):
        """Warm up, then `n_blocks` timed regions of exactly `steps` steps, each between two barrier + synchronize
        pairs, each reduced by MAX over ranks.  Returns the median block and all block times."""
        wl, spec, ns, S = setup(workload, sims)
        if strong:
            S = int(args.total_sims)
        run_steps = make_stepper(spec, ns, S, staged, capi)
        # W untimed warm-up steps; every batch slot in flight owns its own device arena, so at least `depth` batches
        # must have been issued before the timed region or a first-use hipMalloc lands inside it
        n_warm = max(warmup, depth)
        tw = time.perf_counter()
        run_steps(0, n_warm)
        barrier()
        first_ms = max_over_ranks(1e3 * (time.perf_counter() - tw))
        if warm_ms > 0 and first_ms < 10.0 * warm_ms:  # (a first block that long has warmed the device by itself)
            # ... then the same steps, still untimed, until the clocks have ramped.  The first block above also paid
            # the one-time costs (code-object load, first-use hipMalloc of every batch slot), so the step time is
            # taken from a second block; every rank must issue the same number of steps (each carries the
            # collective), so the count comes from the slowest rank
            tb = time.perf_counter()
            run_steps(n_warm, depth)
            barrier()
            n_warm += depth
            per_step = max_over_ranks(1e3 * (time.perf_counter() - tb) / depth)
            extra = int(min(2000, warm_ms / max(per_step, 1e-3)))
            if extra:
                run_steps(n_warm, extra)
                n_warm += extra
                barrier()
        warm_total_ms = 1e3 * (time.perf_counter() - tw)
        blocks = []
        nxt = n_warm
        for _ in range(max(1, n_blocks)):
            t0 = time.perf_counter()
            acc = run_steps(nxt, steps)
            barrier()
            elapsed = max_over_ranks(time.perf_counter() - t0)
            nxt += steps
            want = float(S) * (1 if strong else world) * steps
            if acc["pooled_sims"] != want:
                raise SystemExit(f"{workload}: pooled replicate count {acc['pooled_sims']} != {want}")
            blocks.append((elapsed, acc))
        order = sorted(range(len(blocks)), key=lambda i: blocks[i][0])
        med = order[len(order) // 2]
        elapsed, acc = blocks[med]
        total_fits = float(S) * (1 if strong else world) * steps * len(ns)
        ms = [1e3 * b[0] / steps for b in blocks]
        return dict(wl=wl, spec=spec, ns=ns, S=S, steps=steps, elapsed=elapsed, acc=acc, value=total_fits / elapsed,
                    ms_per_step=1e3 * elapsed / steps, n_warm=n_warm, warm_ms=warm_total_ms,
                    blocks={"n": len(blocks), "steps_each": steps, "ms_per_step": [round(v, 5) for v in ms], "median": sorted(ms)[len(ms) // 2],
                            "min": min(ms), "max": max(ms), "spread_pct": 100.0 * (max(ms) - min(ms)) / sorted(ms)[len(ms) // 2]})

    def valu_roofline(workload, m, staged=False):
        wl, spec, ns, S, acc, steps = m["wl"], m["spec"], m["ns"], m["S"], m["acc"], m["steps"]
        launches = max(acc["launches"], 1)
        avg_launch_ms = acc["kernel_ms"] / launches           # HIP events on each launch's stream (host_api.hip)
        kbusy = busy_ms(acc["spans"])                         # union of those intervals over the timed region
        conc = acc["kernel_ms"] / kbusy if kbusy > 0 else 1.0
        per_rank = (S // world) if strong else S
        fits_per_launch = per_rank / max(launches / steps, 1) * len(ns)   # a replicate on a grid is len(ns) prefix fits (throughput.rs:1111-1113)
        # launches overlap (depth > 1: each batch on its own stream), so the rate of the kernel over the timed region is
        # work / kernel-busy time = per-launch work x mean concurrency / average launch duration
        rate = fits_per_launch * conc / (avg_launch_ms * 1e-3)
        ach1, ach2 = wl["ops1"] * rate / 1e12, wl["ops2"] * rate / 1e12
        is_plain_ols = wl["kind"] == "OLS" and not workload.startswith("c5_")
        return {"bound": "valu", "kernel": kernel_name(workload, spec.n_predictors),
                "achieved": ach1, "peak": PEAK_VALU_TOPS, "unit": "Top/s", "frac": ach1 / PEAK_VALU_TOPS,
                "ops_per_fit": wl["ops1"], "frac_mac_as_two_flops": ach2 / PEAK_VALU_TOPS, "ops_per_fit_mac_as_two_flops": wl["ops2"],
                "kernel_busy_ms": kbusy, "mean_concurrent_launches": conc, "avg_launch_ms": avg_launch_ms,
                "fits_per_launch": fits_per_launch,
                "traffic": (pmc_traffic(workload, f"ols_sim_kernel<{spec.n_predictors}, 0,") if (is_plain_ols and S == 1_000_000) else None),
                "hbm_equiv_gbs": wl["bytes"] * rate / 1e9}

    # ------------------------------------------------------------------------------------------------ headline workload
    staged = args.pipeline == "staged"
    m = measure(args.workload, args.sims_per_gpu, args.steps, args.blocks, args.warmup, args.warm_ms, staged)

    if strong and world >= 1:
        # pooled counters of the sharded call == the counters of ONE rank running the same 10^5-replicate head alone
        wl, spec, ns = m["wl"], m["spec"], m["ns"]
        head = 100_000
        off, cnt = engine.shard_range(head, rank, world)
        part = engine.run_batch(spec, ns, cnt, helpers.BASE_SEED, per_sim=False, sim_offset=off, allreduce=capi)
        pooled_head = part.counters if (capi or world == 1) else mdist.allreduce_counters(part.counters)
        if rank == 0:
            alone = engine.run_batch(spec, ns, head, helpers.BASE_SEED, per_sim=False)
            if not np.array_equal(pooled_head, alone.counters):
                raise SystemExit("strong scaling: pooled counters of the sharded head differ from the one-rank run")
        barrier()

    if rank == 0:
        wl, spec, ns, S, acc, r = m["wl"], m["spec"], m["ns"], m["S"], m["acc"], m["acc"]["last"]
        pooled = acc["pooled"]
        nt = spec.n_targets
        k_unc = pooled[0, r.layout["k_unc"]:r.layout["k_unc"] + nt].astype(np.float64)
        n_pool = float(pooled[0, r.layout["n_sims"]])
        desc = wl["desc"] or ("ols_multi on the grid N=50..1000 (20 fits per replicate), scenario " + args.workload[3:])
        out = {
            "metric": "replicate-fits/sec", "value": m["value"], "unit": "fits/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "warmup_steps_issued": m["n_warm"], "warmup_ms": m["warm_ms"],
            "ms_per_step": m["ms_per_step"], "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32 data plane / f64 fit", "data": "synthetic",
            "blocks": m["blocks"], "rccl_ranks": rccl_ranks,
            "config": {"workload": f"{args.workload}: {wl['kind']} n={wl['n']} p={spec.n_predictors} ({desc}), "
                                   + (f"{S} replicates per step over all ranks (strong scaling)" if strong else f"{S} replicates per GPU per step")
                                   + ", seed 2137",
                       "pipeline": args.pipeline, "parallelism": f"replicate-sharded x{world}",
                       "collective": ("none needed (one rank); the RCCL path is exercised by `rccl.world1_selfcheck`" if world == 1 else
                                      "1 ncclAllReduce(sum, uint64) of the device-resident counters per step, inside the C ABI" if capi else
                                      f"1 all-reduce(sum) of HOST counter blocks per step via torch ({'gloo' if args.backend == 'capi' else args.backend})"
                                      + (" -- FALLBACK (--allow-host-fallback): the C-ABI RCCL communicator did not come up on every rank" if comm_fallback else "")),
                       "batches_in_flight": depth,
                       "host_ms_per_step": {"submit": acc["host_submit_ms"], "collect_wait": acc["host_collect_ms"]}},
            "power_last_step": (k_unc / n_pool).round(5).tolist(),
        }
        if not staged:
            out["roofline"] = valu_roofline(args.workload, m)
            out["roofline"]["note"] = ("fused generate+fit: no design bytes touch HBM; bound is VALU/integer issue (SURVEY.md 8d); "
                                       "achieved = ops_per_fit x fits_per_launch x mean_concurrent_launches / avg_launch_ms (launches of "
                                       "batches in flight overlap; avg_launch_ms is the per-launch HIP-event duration rocprofv3 "
                                       "--kernel-trace also reports); ops_per_fit and frac count an f64 multiply-add as ONE lane-op, the "
                                       "unit of the peak; HBM-equivalent rate in hbm_equiv_gbs")
        else:
            launches = max(acc["launches"], 1)
            avg = acc["kernel_ms"] / launches
            fpl = S / max(launches / args.steps, 1)
            gbs = wl["bytes"] * fpl / (avg * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": f"ols_sim_kernel<{spec.n_predictors},fit>", "achieved": gbs,
                               "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "traffic": None,
                               "bytes_per_fit": wl["bytes"], "avg_launch_ms": avg,
                               "generator_launch_ms": acc["aux_ms"] / launches, "fits_per_launch": fpl}
        is_plain_ols = wl["kind"] == "OLS" and not args.workload.startswith("c5_")
        if not staged and is_plain_ols and spec.n_predictors <= 8 and not strong:  # the staged pipeline serves the register-resident kernels
            # the HBM-bound kernel of the path: staged fit over a resident design tile (measured outside the timed regions)
            rs = engine.run_batch(spec, ns, S, helpers.BASE_SEED, per_sim=False, staged=True)
            rs = engine.run_batch(spec, ns, S, helpers.BASE_SEED, per_sim=False, staged=True)
            fit_ms = rs.kernel_ms / rs.n_kernel_launches
            fpl = S / rs.n_kernel_launches
            gbs = wl["bytes"] * fpl / (fit_ms * 1e-3) / 1e9
            out["roofline_staged_fit"] = {"bound": "hbm", "kernel": f"ols_sim_kernel<{spec.n_predictors},fit>",
                                          "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                                          "traffic": pmc_traffic(args.workload, f"ols_sim_kernel<{spec.n_predictors}, 2,") if S == 1_000_000 else None,
                                          "bytes_per_fit": wl["bytes"], "algorithmic_bytes_per_launch": wl["bytes"] * fpl,
                                          "avg_launch_ms": fit_ms, "generator_launch_ms": rs.aux_kernel_ms / rs.n_kernel_launches,
                                          "fits_per_launch": fpl}

        cores = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16))
        oracle = None
        if world == 1 and not (args.no_cpu_baseline and (args.no_configs or args.workload != "c2a")):
            from oracle import oracle  # the checker: cpu_baseline leg and the power deltas of `configs` only
            flags = oracle.use_native_build()  # -O3 -march=native, built here on the timing host

        if world == 1 and args.workload == "c2a" and not args.no_configs and not staged and not strong:
            # the other BASELINE configurations, each measured like the headline (3 blocks, median) with the device warm,
            # and checked against the CPU oracle on a 2 000-replicate head (same spec, same seeds)
            cfgs = {}
            for name, wname, csteps in CONFIG_MATRIX:
                cm = measure(wname, 0, csteps, 3, depth, 30.0)
                cs, cns = cm["spec"], cm["ns"]
                head = 2000
                g = engine.run_batch(cs, cns, head, helpers.BASE_SEED, per_sim=False)
                o = oracle.run_batch(cs, cns, head, helpers.BASE_SEED, n_threads=cores)
                lay = g.layout
                gk = g.counters[:, lay["k_unc"]:lay["k_unc"] + cs.n_targets].astype(np.float64)
                ok_ = o.counters[:, lay["k_unc"]:lay["k_unc"] + cs.n_targets].astype(np.float64)
                rf = valu_roofline(wname, cm)
                cfgs[name] = {"value": cm["value"], "unit": "prefix-fits/s" if len(cns) > 1 else "fits/s",
                              "ms_per_step": cm["ms_per_step"], "replicates_per_step": cm["S"], "grid_points": len(cns),
                              "blocks_ms_per_step": cm["blocks"]["ms_per_step"],
                              "roofline": {k: rf[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "ops_per_fit", "avg_launch_ms", "mean_concurrent_launches")},
                              "power_delta_pp_vs_oracle": float(np.max(np.abs(gk - ok_)) / head * 100.0),
                              "oracle_head_replicates": head}
            out["configs"] = cfgs

        if world == 1:
            # RCCL evidence on one GPU: a world of ONE communicator through the C ABI, pooled counters == local counters
            try:
                mdist.init_comm(0, 1)
                a = engine.run_batch(spec, ns, 20_000, helpers.BASE_SEED, per_sim=False, allreduce=True)
                b = engine.run_batch(spec, ns, 20_000, helpers.BASE_SEED, per_sim=False)
                out["rccl"] = {"ranks": engine.comm_world(), "world1_selfcheck": "pooled == local" if np.array_equal(a.counters, b.counters) else "MISMATCH"}
                engine.comm_destroy()
            except Exception as e:  # a box without RCCL still benches one GPU; the line says so
                out["rccl"] = {"ranks": 0, "world1_selfcheck": f"communicator unavailable: {e}"}
            if out["rccl"]["world1_selfcheck"] == "MISMATCH":
                raise SystemExit("world-of-one RCCL all-reduce changed the counters")
        else:
            out["rccl"] = {"ranks": rccl_ranks}

        if not args.no_cpu_baseline and world == 1:
            # the CPU oracle ("port"): same spec, same seeds, on this host's cores, bounded sample.  The GPU box hands a
            # one-GPU job a 16-core share of its host (os.cpu_count() reports the whole machine: 256 threads
            # oversubscribing 16 cores ran 30 % slower than 16)
            probe = 50_000 if wl["kind"] == "OLS" else 2_000  # iterative fits run 1e2..1e4 per second: keep the probe in seconds
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
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        barrier()
        if capi:
            engine.comm_destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
