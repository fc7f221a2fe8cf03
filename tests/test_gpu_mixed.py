"""Device vs oracle on the general mixed-model path (random slopes, crossed / nested extra groupings): the kernel
runs the oracle's algorithm in the oracle's operation order, so decisions agree replicate by replicate and the
captured Wald statistics to rounding (the only non-shared arithmetic is log(): ocml vs glibc)."""
import numpy as np
import pytest

from helpers import BASE_SEED, load_case, mixed_case_names

pytestmark = pytest.mark.gpu


def _atom(spec):
    a = spec.cluster.sizing[1]
    for g in spec.cluster.extra_groupings:
        (_, body), = g["relation"].items()
        a *= list(body.values())[0]
    return a


@pytest.mark.parametrize("name", mixed_case_names())
def test_general_lmm_run_batch_matches_oracle(name, engine, oracle):
    spec, _, a, low = load_case(name)
    n = low["n"]
    half = n // 2
    grid = [half, n] if half % _atom(spec) == 0 else [n]
    spec.scenario.truth_start = name.endswith("_a")          # both start modes are exercised across the cases
    n_sims = 24 if "nested_extra" in name else 48
    dev, stat = engine.run_batch_st_capture(spec, grid, n_sims, BASE_SEED)
    ora = oracle.run_batch(spec, grid, n_sims, BASE_SEED, n_threads=8, stat_capture=True)
    assert dev.converged.mean() > 0.9
    same = ((dev.uncorrected == ora.uncorrected).all(axis=(1, 2)) & (dev.converged == ora.converged).all(axis=1)
            & (dev.boundary_hit == ora.boundary_hit).all(axis=1) & (dev.joint_unc == ora.joint_unc).all(axis=1)
            & (dev.pinned_components == ora.pinned_components).all(axis=1))
    assert same.sum() >= n_sims - 1, (name, np.flatnonzero(~same))
    both = (dev.converged[:, 0] == 1) & (ora.converged[:, 0] == 1) & same
    d, o = stat[both], ora.stat_capture[both]
    assert np.allclose(d, o, rtol=1e-5, atol=1e-8), (name, np.abs(d / o - 1).max())


def test_general_lmm_counters_only_and_shards(engine):
    spec, _, a, low = load_case("lmm_slope_crossed")
    whole = engine.run_batch(spec, [480], 64, BASE_SEED)
    red = engine.run_batch(spec, [480], 64, BASE_SEED, per_sim=False)
    assert np.array_equal(whole.counters, red.counters)
    left = engine.run_batch(spec, [480], 40, BASE_SEED, per_sim=False)
    right = engine.run_batch(spec, [480], 24, BASE_SEED, per_sim=False, sim_offset=40)
    assert np.array_equal(left.counters + right.counters, whole.counters)
    again = engine.run_batch(spec, [480], 64, BASE_SEED)
    assert np.array_equal(again.uncorrected, whole.uncorrected) and np.array_equal(again.pinned_components, whole.pinned_components)
