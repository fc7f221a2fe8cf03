"""The path's one collective behind the C ABI (include/mcpower_hip.h: mcp_comm_* + MCP_ALLREDUCE_COUNTERS; reference
counterpart merge_power_results, crates/engine-orchestrator/src/merge.rs:26-70).  A one-GPU box can hold a world of
one: the all-reduce then has to hand back the local counters bit for bit, through RCCL, on the launch stream, on the
blocking and the submit / collect entry points alike; the cancel verdict travels in the same payload."""
import numpy as np
import pytest

from helpers import BASE_SEED, spec_c2a, spec_c3
from mcpower_amd import dist as mdist, spec as S

pytestmark = pytest.mark.gpu


def test_allreduce_without_a_communicator_is_refused(engine):
    assert engine.comm_world() == 0  # runs before the module's communicator comes up
    with pytest.raises(S.EngineError) as e:
        engine.run_batch(spec_c2a(), [200], 1000, BASE_SEED, per_sim=False, allreduce=True)
    assert e.value.status == S.Status.InvalidSpec and "communicator" in str(e.value)


@pytest.fixture(scope="module")
def comm(engine):
    engine.set_device(0)
    assert engine.comm_rank() == -1 and engine.comm_world() == 0
    mdist.init_comm(0, 1)          # rank 0 of a world of one: id creation + join + bring-up all-reduce
    assert engine.comm_rank() == 0 and engine.comm_world() == 1
    yield engine
    engine.comm_destroy()
    assert engine.comm_world() == 0


@pytest.mark.parametrize("make,ns", [(spec_c2a, [100, 200]), (spec_c3, [250])])
def test_world_of_one_returns_the_local_counters_bit_for_bit(make, ns, comm):
    spec = make()
    plain = comm.run_batch(spec, ns, 20_000, BASE_SEED, per_sim=False, sim_offset=123)
    pooled = comm.run_batch(spec, ns, 20_000, BASE_SEED, per_sim=False, sim_offset=123, allreduce=True)
    assert np.array_equal(plain.counters, pooled.counters) and plain.counters.sum() > 0
    # with the per-sim tensors on: they stay local, the counters are still the pooled ones
    full = comm.run_batch(spec, ns, 2_000, BASE_SEED, allreduce=True)
    ref = comm.run_batch(spec, ns, 2_000, BASE_SEED)
    assert np.array_equal(full.counters, ref.counters) and np.array_equal(full.uncorrected, ref.uncorrected)


def test_submit_collect_pools_in_order(comm):
    spec = spec_c2a().to_c()
    tickets = [comm.submit_batch(spec, [200], 50_000, BASE_SEED, sim_offset=i * 50_000, allreduce=True) for i in range(3)]
    got = [t.collect().counters for t in tickets]
    want = [comm.run_batch(spec, [200], 50_000, BASE_SEED, per_sim=False, sim_offset=i * 50_000).counters for i in range(3)]
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def test_run_batch_sharded_uses_the_communicator(comm):
    spec = spec_c2a()
    pooled, local = mdist.run_batch_sharded(spec, [200], 30_000, BASE_SEED, rank=0, world=1)
    assert np.array_equal(pooled, comm.run_batch(spec, [200], 30_000, BASE_SEED, per_sim=False).counters)


def test_cancel_travels_through_the_collective(comm):
    calls = []

    def sink(cur, tot):
        calls.append(cur)
        return len(calls) < 2

    with pytest.raises(S.EngineError) as e:
        comm.run_batch(spec_c2a(), [200], 200_000, BASE_SEED, sink, per_sim=False, chunk_sims=20_000, allreduce=True)
    assert e.value.status == S.Status.Cancelled and len(calls) == 2
    # the communicator is still usable afterwards
    again = comm.run_batch(spec_c2a(), [200], 10_000, BASE_SEED, per_sim=False, allreduce=True)
    assert int(again.counter("n_sims", 1)[0, 0]) == 10_000




def test_an_empty_shard_joins_the_collective_with_zeros(comm):
    """Fewer replicates than ranks: the rank whose slice is empty launches nothing but still takes part in the
    all-reduce (a refusal there would leave its peers waiting in the collective for ever).  World of one: the pooled
    vector of an empty shard is all zeros, on the blocking and on the submit / collect entry point."""
    spec = spec_c2a()
    r = comm.run_batch(spec, [100, 200], 0, BASE_SEED, per_sim=False, allreduce=True)
    assert r.counters.shape[0] == 2 and int(r.counters.sum()) == 0
    t = comm.submit_batch(spec.to_c(), [200], 0, BASE_SEED, allreduce=True)
    assert int(t.collect().counters.sum()) == 0
    off, cnt = mdist.shard_range(3, 7, 8)
    assert cnt in (0, 1)
    # the communicator is still usable afterwards
    again = comm.run_batch(spec, [200], 10_000, BASE_SEED, per_sim=False, allreduce=True)
    assert int(again.counter("n_sims", 1)[0, 0]) == 10_000
