"""One process per GPU: shard the replicate range and pool the additive counters.

The reference has no multi-process path; its closest analogue is the browser worker pool,
which runs `single_core_find_power` per worker and pools raw counters with
`merge_power_results` (`/root/reference/crates/engine-orchestrator/src/merge.rs:26-70`,
`ports/wasm/src/seeds.ts:98-103`).  Here rank r of W takes the contiguous slice
[r*S/W, (r+1)*S/W) of sim_ids with the SAME base_seed (so the union is bit-identical to a
single-device run, unlike the browser pool which re-seeds per worker), and the only exchange
on the data path is ONE all-reduce (sum) of the counter vector.

The product form of that collective lives behind the C ABI (`mcp_comm_init` + `MCP_ALLREDUCE_COUNTERS`,
`csrc/host_comm.cpp`): `ncclAllReduce(sum, uint64)` over RCCL / xGMI on the device-resident vector, on the launch
stream, before the call's single device-to-host copy -- no torch, no host bounce.  This module is the thin caller:
`init_comm` carries the 128-byte rendezvous id over the host's control plane (a `torch.distributed` process group of
any backend) and `run_batch_sharded` sets the option.

REHEARSAL ONLY: the `allreduce_counters*` functions below pool HOST counter blocks through a torch process group.  They
are not the product's collective: they exist for the world-size-2 gloo tests on GPU-less boxes (tests/test_dist_cpu.py,
tests/test_sharded_engine.py) and for `bench.py --backend gloo|nccl` / `--allow-host-fallback`.  On a box with a device
`run_batch_sharded` and `_engine.sharded_runner` REFUSE to pool on the host unless told to (`allow_host_pool=True`):
a multi-GPU result must never be mistaken for the RCCL path when it is not.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def shard_range(n_sims: int, rank: int, world: int) -> Tuple[int, int]:
    """(offset, count) of rank's contiguous replicate slice; slices partition [0, n_sims).
    Same rule as `mcp_shard_range` (host_comm.cpp); kept in Python so GPU-less callers need no library."""
    lo = n_sims * rank // world
    hi = n_sims * (rank + 1) // world
    return lo, hi - lo


def init_comm(rank: int, world: int, group=None) -> None:
    """Bring up the C-ABI RCCL communicator of this process (`set_device` first): rank 0 creates the rendezvous id,
    the process group -- control plane only, any backend -- broadcasts its 128 bytes, every rank joins."""
    import torch.distributed as dist

    from . import engine

    box, err = [None], None
    if rank == 0:
        try:
            box[0] = engine.comm_unique_id()
        except Exception as e:  # the other ranks are waiting in the broadcast: tell them before raising
            err = e
    if world > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    if err is not None:
        raise err
    if box[0] is None:
        raise RuntimeError("rank 0 could not create the RCCL rendezvous id")
    engine.comm_init(rank, world, box[0])


def allreduce_counters(counters: np.ndarray, group=None) -> np.ndarray:
    """Sum the (n_ss, stride) uint64 counter block over all ranks (one collective).

    Counts are far below 2^63, so the payload travels as int64 (torch has no uint64 reduction)."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return counters.copy()
    backend = dist.get_backend(group)
    t = torch.from_numpy(counters.astype(np.int64, copy=True))
    if backend == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy().astype(np.uint64)


class PendingCounters:
    """An all-reduce in flight: `wait()` returns the pooled (n_ss, stride) uint64 block."""

    def __init__(self, tensor, work, local):
        self._t, self._work, self._local = tensor, work, local

    def wait(self) -> np.ndarray:
        if self._t is None:
            return self._local
        self._work.wait()
        return self._t.cpu().numpy().astype(np.uint64)


def allreduce_counters_async(counters: np.ndarray, group=None) -> PendingCounters:
    """`allreduce_counters` without the wait: the collective runs on the process group's own stream while the
    caller launches its next batch, so its latency hides behind the next simulate kernel."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return PendingCounters(None, None, counters.copy())
    t = torch.from_numpy(counters.astype(np.int64, copy=True))
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    work = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return PendingCounters(t, work, None)


def wilson_ci(k: int, n: int):
    """Wilson 95% score interval, z = 1.96 (reference aggregation.rs:18-36) -- recomputed from pooled counts."""
    if n == 0:
        return 0.0, 1.0
    z = 1.96
    p = k / n
    z2 = z * z
    den = 1.0 + z2 / n
    center = (p + z2 / (2.0 * n)) / den
    margin = z * ((p * (1.0 - p) / n + z2 / (4.0 * n * n)) ** 0.5) / den
    return max(center - margin, 0.0), min(center + margin, 1.0)


def run_batch_sharded(spec, sample_sizes, n_sims: int, base_seed: int, *, rank: int, world: int, group=None,
                      allow_host_pool: bool = False, **kw):
    """Each rank runs its slice on its own GPU (counters-only); the counts are pooled by the ONE all-reduce, inside the
    call, on the device, through the C-ABI communicator of this world (`init_comm`).  A shard may be empty (fewer
    replicates than ranks): that rank still joins the collective.  Without a communicator of this world the call
    raises; `allow_host_pool=True` (rehearsals) pools the host counter blocks through the torch process group instead.
    Returns (pooled_counters, local BatchResult)."""
    from . import engine

    off, cnt = shard_range(n_sims, rank, world)
    if engine.comm_world() == world:
        local = engine.run_batch(spec, sample_sizes, cnt, base_seed, per_sim=False, sim_offset=off, allreduce=True, **kw)
        return local.counters, local
    if world > 1 and not allow_host_pool:
        raise RuntimeError(f"run_batch_sharded: no C-ABI communicator of world {world} in this process (init_comm first); "
                           "pass allow_host_pool=True to pool host counter blocks through the torch process group instead")
    local = engine.run_batch(spec, sample_sizes, cnt, base_seed, per_sim=False, sim_offset=off, **kw)
    return allreduce_counters(local.counters, group), local
