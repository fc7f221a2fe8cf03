"""The N>1 path on CPU: world_size 2 over gloo.  Each rank takes its replicate slice
(mcpower_amd.dist.shard_range), produces counters for it (the oracle stands in for the device
on this GPU-less box), and the single all-reduce pools them -- the union must equal a
single-process run bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_sims, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    from helpers import BASE_SEED, spec_c2b
    from mcpower_amd import dist as mdist
    from oracle import oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    off, cnt = mdist.shard_range(n_sims, rank, world)
    local = oracle.run_batch(spec_c2b(), [100, 200], cnt, BASE_SEED, sim_offset=off)
    pooled = mdist.allreduce_counters(local.counters)
    np.save(os.path.join(out_dir, f"pooled_{rank}.npy"), pooled)
    # the pipelined form bench.py uses: the collective of step i is collected after step i+1 was issued
    first = mdist.allreduce_counters_async(local.counters)
    second = mdist.allreduce_counters_async(local.counters * np.uint64(2))
    a, b = first.wait(), second.wait()
    assert np.array_equal(a, pooled) and np.array_equal(b, pooled * np.uint64(2))
    dist.destroy_process_group()


def test_shard_ranges_partition_the_replicates(engine):
    """The Python rule and the C ABI's mcp_shard_range (host_comm.cpp) agree and partition [0, S)."""
    from mcpower_amd.dist import shard_range
    for n in (0, 1, 7, 1000, 10**7 + 3, 2**64 - 1):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts == [engine.shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == n
            for (o1, c1), (o2, _) in zip(parts, parts[1:]):
                assert o1 + c1 == o2
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def _comm_worker(rank, world, port, out_dir):
    """Rank / id plumbing of the C-ABI communicator bring-up over a gloo control plane.  No GPU here, so the two
    device-touching entry points are stood in for: id creation hands out a recognisable 128-byte pattern, and the join
    records its arguments and then calls the REAL mcp_comm_init, which must refuse loudly (MCP_ERR_DEVICE)."""
    sys.path.insert(0, ROOT)
    import json
    import torch.distributed as dist
    from mcpower_amd import dist as mdist, engine, spec as S

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    made, joined, status = [], [], []
    real_init = engine.comm_init

    def fake_id():
        made.append(rank)
        return bytes((7 * i + 3) % 256 for i in range(engine.COMM_ID_BYTES))

    def recording_init(r, w, cid):
        joined.append((r, w, cid.hex()))
        try:
            real_init(r, w, cid)
        except S.EngineError as e:
            status.append(int(e.status))

    engine.comm_unique_id, engine.comm_init = fake_id, recording_init
    mdist.init_comm(rank, world)
    off, cnt = engine.shard_range(3001, rank, world)
    with open(os.path.join(out_dir, f"comm_{rank}.json"), "w") as f:
        json.dump({"made": made, "joined": joined, "status": status, "shard": [off, cnt],
                   "world_after": engine.comm_world()}, f)
    dist.destroy_process_group()


def test_two_rank_comm_bringup_plumbing(tmp_path, engine):
    import json
    import torch.multiprocessing as mp
    mp.spawn(_comm_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (json.load(open(tmp_path / f"comm_{r}.json")) for r in (0, 1))
    assert r0["made"] == [0] and r1["made"] == []                      # only rank 0 creates the id
    (a0, w0, id0), (a1, w1, id1) = r0["joined"][0], r1["joined"][0]
    assert (a0, w0, a1, w1) == (0, 2, 1, 2) and id0 == id1 and len(bytes.fromhex(id0)) == 128
    assert r0["status"] == [6] and r1["status"] == [6]                 # no MI355X here: MCP_ERR_DEVICE, loudly
    assert r0["world_after"] == 0 and r1["world_after"] == 0
    assert r0["shard"] == [0, 1500] and r1["shard"] == [1500, 1501]


def test_two_rank_gloo_allreduce_equals_single_run(tmp_path, oracle):
    import torch.multiprocessing as mp

    from helpers import BASE_SEED, spec_c2b
    n_sims = 3001
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_sims, str(tmp_path)), nprocs=2, join=True)
    whole = oracle.run_batch(spec_c2b(), [100, 200], n_sims, BASE_SEED)
    p0 = np.load(tmp_path / "pooled_0.npy")
    p1 = np.load(tmp_path / "pooled_1.npy")
    assert np.array_equal(p0, p1) and np.array_equal(p0, whole.counters)


def test_wilson_ci_from_pooled_counts_matches_reference_formula():
    from mcpower_amd.dist import wilson_ci
    lo, hi = wilson_ci(0, 0)
    assert (lo, hi) == (0.0, 1.0)
    lo, hi = wilson_ci(800, 1600)  # reference: +-1.25 pp SE at 50% power with 1600 sims (engine-architecture.md:49-53)
    assert abs((hi - lo) / 2 - 1.96 * 0.0125) < 1e-4 * 5 and abs((lo + hi) / 2 - 0.5) < 1e-12
    lo, hi = wilson_ci(1600, 1600)
    assert hi > 1.0 - 1e-12 and lo > 0.99
