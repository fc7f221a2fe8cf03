"""Seeded random specs over every family of the path (tests/fuzz_specs.py): the oracle runs them and the host lowering
accepts them (CPU); the device reproduces the oracle's decisions replicate by replicate through the C ABI (GPU)."""
import os

import numpy as np
import pytest

from fuzz_specs import FAMILIES, random_spec
from helpers import BASE_SEED
from mcpower_amd import spec as S

SEEDS = range(int(os.environ.get("MCP_FUZZ_SEEDS", "40")))  # a one-off deeper sweep: MCP_FUZZ_SEEDS=400
FLAGS = ("uncorrected", "corrected", "converged", "overall", "joint_unc", "boundary_hit", "factor_excluded", "pinned_components",
         "posthoc_unc", "posthoc_cor")


@pytest.mark.parametrize("family", FAMILIES)
def test_random_specs_are_accepted_by_oracle_and_host_lowering(family, engine, oracle):
    for seed in SEEDS:
        spec, ns, k, _ = random_spec(seed, family)
        r = oracle.run_batch(spec, ns, 16, BASE_SEED, n_threads=4)
        assert r.uncorrected.shape == (16, len(ns), spec.n_targets)
        try:  # no device here: the lowering (limits, validation) runs first, then the device is asked for
            engine.run_batch(spec, ns, 16, BASE_SEED)
        except S.EngineError as e:
            assert e.status == S.Status.Device, (family, seed, str(e))


def _same_replicates(dev, ora):
    same = np.ones(dev.uncorrected.shape[0], dtype=bool)
    for f in FLAGS:
        a, b = getattr(dev, f), getattr(ora, f)
        if a is None or b is None:
            assert (a is None and b is None) or f == "factor_excluded", f
            continue
        same &= (a == b).reshape(a.shape[0], -1).all(axis=1)
    return same


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("family", FAMILIES)
def test_device_matches_oracle_on_random_specs(family, seed, engine, oracle):
    spec, ns, k, exact = random_spec(seed, family)
    dev = engine.run_batch(spec, ns, k, BASE_SEED)
    ora = oracle.run_batch(spec, ns, k, BASE_SEED, n_threads=16)
    same = _same_replicates(dev, ora)
    if exact:
        assert same.all(), (family, seed, np.flatnonzero(~same)[:8])
        assert np.array_equal(dev.counters, ora.counters)
    else:
        # scenario data path: one-ulp f32 differences (ocml vs glibc exp / log) can move a knife-edge decision; mixed
        # models: the optimiser's last probes can differ by rounding -- at most 1 replicate in 100
        assert (~same).sum() <= max(1, k // 100), (family, seed, np.flatnonzero(~same)[:8])
    red = engine.run_batch(spec, ns, k, BASE_SEED, per_sim=False)
    assert np.array_equal(red.counters, dev.counters)


@pytest.mark.gpu
def test_seed_1098_of_the_first_deep_sweep_stays_a_regression_case(engine, oracle):
    """The one device-vs-oracle mismatch of the 10 200-spec sweep: a random-intercept LMM under the `realistic` preset at
    n = 35 / 42 with 7 clusters and a SAMPLED 4-level factor (proportions down to 0.15: ~5 rows expected).  Root cause,
    asserted here instead of narrowing the generator around it: in the few replicates where the two sides differ a
    sampled level is (nearly) empty or confounded with the clusters, the fitted design is numerically singular
    (cond(X) > 1e7, i.e. cond(X'X) beyond double precision), both sides pass the reference's rank test EPS_RANK = 1e-12
    on min|L_ii| / max|L_ii| (batch.rs:40-41) on rounding noise, and what they then return is unrelated by
    construction -- the reference itself is not reproducible there.  Every well-conditioned replicate must agree."""
    spec, ns, k, exact = random_spec(1098, "lme_scen", legacy_sizes=True)
    assert ns == [35, 42] and spec.cluster.sizing == ("FixedClusters", 7) and spec.factor_n_levels == [4]
    dev = engine.run_batch(spec, ns, k, BASE_SEED)
    ora = oracle.run_batch(spec, ns, k, BASE_SEED, n_threads=16)
    same = _same_replicates(dev, ora)
    bad = np.flatnonzero(~same)
    assert bad.size <= 6, bad
    for sim in bad:
        x, y, ids = oracle.generate_sim_data(spec, BASE_SEED, int(sim), ns[-1])
        conds = [np.linalg.cond(np.asarray(x[:n], dtype=np.float64)) for n in ns]
        assert max(conds) > 1e7, (int(sim), conds)
    red = engine.run_batch(spec, ns, k, BASE_SEED, per_sim=False)
    assert np.array_equal(red.counters, dev.counters)


def test_oracle_is_thread_count_invariant_on_the_table_marginal(oracle):
    """The HighKurtosis marginal reads a lazily built t(3) table (marginals/t3.rs): replicate threads may ask for it
    at the same moment (the race this sweep found in the oracle's first version of the build)."""
    from helpers import base_spec
    spec = base_spec(2)
    spec.var_types = [S.Distribution.HighKurtosis, S.Distribution.Normal]
    spec.effect_sizes = [0.0, 0.2, 0.1]
    one = oracle.run_batch(spec, [100], 256, BASE_SEED, n_threads=1, stat_capture=True)
    many = oracle.run_batch(spec, [100], 256, BASE_SEED, n_threads=16, stat_capture=True)
    assert np.array_equal(one.stat_capture, many.stat_capture) and np.array_equal(one.counters, many.counters)
