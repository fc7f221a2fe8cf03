"""The reference bench's frozen 21-case grid (crates/engine-core/src/bin/throughput.rs:268-633): every case id, with
the reference's spec, n and seed, runs through the oracle (CPU) and -- replicate by replicate -- identically through
the C ABI on the device (GPU)."""
import numpy as np
import pytest

from mcpower_amd import workloads

CASES = workloads.throughput_cases()
# the reference sizes its effects for ~75 % power on the first target (throughput.rs:269-345); lmm_multislope's second
# slope and glm_rare's 5 % base rate sit lower, the wide designs higher
POWER_BAND = (0.30, 0.97)


def test_the_grid_is_the_reference_grid():
    assert list(CASES) == [
        "ols_multi", "ols_wide", "ols_large_n", "ols_factor_inter", "glm_multi", "glm_wide", "glm_large_n", "glm_rare",
        "lme_multi", "lme_many_clusters", "lme_matched_q1", "lmm_crossed", "lmm_nested", "lmm_crossed_nested",
        "lmm_slope", "lmm_multislope", "glmm_intercept", "glmm_slope", "glmm_crossed", "glmm_nested", "glmm_crossed_nested"]
    assert {cid: (n, k) for cid, (_, n, k) in CASES.items()}["glm_wide"] == (200, 5000)
    assert CASES["glm_wide"][0].n_predictors == 16 and CASES["ols_wide"][0].n_predictors == 16
    assert sum(k for _, _, k in CASES.values()) == 70_500          # throughput.rs:505-631


@pytest.mark.parametrize("cid", list(CASES))
def test_oracle_runs_every_frozen_case(cid, oracle):
    spec, n, _ = CASES[cid]
    k = 96 if cid.startswith("glmm") else 400
    r = oracle.run_batch(spec, [n], k, workloads.BASE_SEED, n_threads=8)
    assert r.converged.mean() >= 0.97, cid
    power = r.uncorrected[:, 0, 0].mean()
    assert POWER_BAND[0] < power < POWER_BAND[1], (cid, power)


@pytest.mark.gpu
@pytest.mark.parametrize("cid", list(CASES))
def test_device_matches_oracle_on_every_frozen_case(cid, engine, oracle):
    """OLS / GLM: flags and counters bit-identical.  Mixed models: the optimiser follows the oracle's path, so the
    flags agree replicate by replicate up to one knife-edge replicate per case (a p-value within rounding of alpha)."""
    spec, n, _ = CASES[cid]
    exact = cid.startswith(("ols", "glm_"))
    k = 512 if exact else (64 if cid.startswith("glmm") else 128)
    dev = engine.run_batch(spec, [n], k, workloads.BASE_SEED)
    ora = oracle.run_batch(spec, [n], k, workloads.BASE_SEED, n_threads=16)
    same = (dev.uncorrected == ora.uncorrected).all(axis=(1, 2)) & (dev.converged == ora.converged).all(axis=1)
    if exact:
        assert same.all(), (cid, np.flatnonzero(~same))
        assert np.array_equal(dev.counters, ora.counters)
    else:
        assert same.sum() >= k - 1, (cid, np.flatnonzero(~same))
        assert np.array_equal(dev.pinned_components, ora.pinned_components) or same.sum() >= k - 1
    red = engine.run_batch(spec, [n], k, workloads.BASE_SEED, per_sim=False)
    assert np.array_equal(red.counters, dev.counters)


# ---------------------------------------------------------------------------------------------
# wide IRLS designs, P = 9..16 (glm_wide is P = 16): every width, all three families, the reduced-fit arm
# ---------------------------------------------------------------------------------------------
def _wide_glm(k, family="logit", factor=False):
    from mcpower_amd import spec as S
    s = workloads._glm_spec(k, 0.3)
    s.effect_sizes[1] = s.effect_sizes[2] = 0.4
    s.target_indices = list(range(1, k + 1))
    s.report_overall = True
    s.correction_method = S.CorrectionMethod.Holm
    if family == "probit":
        s.link = S.LinkKind.Probit
    if family == "poisson":
        s.outcome_kind = S.OutcomeKind.Count
        s.intercept = s.effect_sizes[0] = 0.7
        s.effect_sizes[1] = s.effect_sizes[2] = 0.15
    if factor:                                    # a 3-level factor whose small level is sparse at the small n
        s.n_factor_dummies, s.factor_n_levels, s.factor_proportions = 2, [3], [0.62, 0.3, 0.08]
        s.factor_sampled = [None]
        s.effect_sizes = list(s.effect_sizes) + [0.3, 0.2]
        s.target_indices = list(range(1, k + 3))
        s.factor_min_level_count = 5
    return s


@pytest.mark.gpu
@pytest.mark.parametrize("p", range(9, 17))
def test_wide_logit_matches_oracle_at_every_width(p, engine, oracle):
    spec = _wide_glm(p - 1)
    g = engine.run_batch(spec, [120, 200], 600, workloads.BASE_SEED, stat_capture=True)
    o = oracle.run_batch(spec, [120, 200], 600, workloads.BASE_SEED, stat_capture=True, n_threads=16)
    a, b = g.stat_capture, o.stat_capture
    assert np.array_equal(np.isnan(a), np.isnan(b))
    m = ~np.isnan(a)
    assert np.all(np.abs(a[m] - b[m]) <= 1e-8 * np.abs(b[m]) + 1e-10)
    for f in ("uncorrected", "corrected", "converged", "overall"):
        assert np.array_equal(getattr(g, f), getattr(o, f)), f
    assert np.array_equal(g.counters, o.counters)
    assert g.converged.mean() > 0.99


@pytest.mark.gpu
@pytest.mark.parametrize("family,k", [("poisson", 15), ("poisson", 11), ("probit", 15), ("probit", 9)])
def test_wide_poisson_and_probit_match_oracle(family, k, engine, oracle):
    spec = _wide_glm(k, family)
    g = engine.run_batch(spec, [200], 600, workloads.BASE_SEED, stat_capture=True)
    o = oracle.run_batch(spec, [200], 600, workloads.BASE_SEED, stat_capture=True, n_threads=16)
    a, b = g.stat_capture, o.stat_capture
    m = ~np.isnan(a)
    assert np.array_equal(m, ~np.isnan(b)) and np.all(np.abs(a[m] - b[m]) <= 1e-7 * np.abs(b[m]) + 1e-9)
    for f in ("uncorrected", "corrected", "converged", "overall"):
        assert np.array_equal(getattr(g, f), getattr(o, f)), f
    assert g.converged.mean() > 0.99


@pytest.mark.gpu
@pytest.mark.parametrize("family,k", [("logit", 13), ("poisson", 10), ("logit", 8)])
def test_wide_designs_with_a_sparse_factor_take_the_reduced_fit(family, k, engine, oracle):
    """P = k + 3 (two dummies): the sparse level is excluded at n = 40 and kept at n = 200 (batch.rs:2573-2600)."""
    spec = _wide_glm(k, family, factor=True)
    g = engine.run_batch(spec, [40, 200], 400, workloads.BASE_SEED)
    o = oracle.run_batch(spec, [40, 200], 400, workloads.BASE_SEED, n_threads=16)
    assert np.array_equal(g.factor_excluded, o.factor_excluded) and g.factor_excluded[:, 0, 0].mean() > 0.5
    for f in ("uncorrected", "corrected", "converged", "overall"):
        assert np.array_equal(getattr(g, f), getattr(o, f)), f
    assert np.array_equal(g.counters, o.counters)


# ---------------------------------------------------------------------------------------------
# large-n logit designs: several waves share one replicate (glm_kernels.hip: glm_threads_per_replicate)
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("k,n,factor", [(3, 5000, False), (3, 1800, False), (7, 2600, False), (2, 3000, True), (5, 1200, True)])
def test_large_n_logit_on_several_waves_per_replicate_matches_oracle(k, n, factor, engine, oracle):
    """n x p f32 beyond 20 KB of LDS: 256 or 512 threads per replicate, sums crossing the waves through LDS; a grid whose
    smaller sizes leave most lanes without rows, and the factor walk on the wide block."""
    from mcpower_amd import spec as S
    spec = workloads._glm_spec(k, 0.3)
    spec.effect_sizes[1] = 0.09
    if k > 1:
        spec.effect_sizes[2] = -0.07
    spec.target_indices = list(range(1, k + 1))
    spec.report_overall = True
    spec.correction_method = S.CorrectionMethod.BenjaminiHochberg
    if factor:
        spec.n_factor_dummies, spec.factor_n_levels, spec.factor_proportions, spec.factor_sampled = 2, [3], [0.5, 0.3, 0.2], [None]
        spec.effect_sizes = list(spec.effect_sizes) + [0.2, -0.1]
        spec.target_indices = list(range(1, k + 3))
    grid = [90, n // 3, n]
    g = engine.run_batch(spec, grid, 256, workloads.BASE_SEED, stat_capture=True)
    o = oracle.run_batch(spec, grid, 256, workloads.BASE_SEED, stat_capture=True, n_threads=16)
    a, b = g.stat_capture, o.stat_capture
    m = ~np.isnan(a)
    assert np.array_equal(m, ~np.isnan(b)) and np.all(np.abs(a[m] - b[m]) <= 1e-8 * np.abs(b[m]) + 1e-10)
    for f in ("uncorrected", "corrected", "converged", "overall"):
        assert np.array_equal(getattr(g, f), getattr(o, f)), f
    assert np.array_equal(g.counters, o.counters) and g.converged[:, 2].all()


# ---------------------------------------------------------------------------------------------
# OLS designs of 9..12 columns run on the fused lane-per-replicate kernel (in-place Cholesky, <= 12 tests)
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("p", [9, 10, 11, 12])
@pytest.mark.parametrize("factor", [False, True])
def test_fused_ols_up_to_twelve_columns_matches_oracle(p, factor, engine, oracle):
    from mcpower_amd import spec as S
    k = p - 1 - (2 if factor else 0)
    spec = workloads.base_spec(k)
    spec.effect_sizes = [0.0] + [0.2, -0.15, 0.1][:k] + [0.0] * max(0, k - 3)
    if factor:
        spec.n_factor_dummies, spec.factor_n_levels, spec.factor_proportions, spec.factor_sampled = 2, [3], [0.5, 0.3, 0.2], [None]
        spec.effect_sizes = list(spec.effect_sizes) + [0.3, -0.2]
    spec.target_indices = list(range(1, p))
    spec.contrast_pairs = [(1, 2)]
    spec.report_overall = True
    spec.correction_method = S.CorrectionMethod.Holm
    for grid in ([120], [40, 90, 200]):
        g = engine.run_batch(spec, grid, 1024, workloads.BASE_SEED, stat_capture=True)
        o = oracle.run_batch(spec, grid, 1024, workloads.BASE_SEED, stat_capture=True, n_threads=16)
        a, b = g.stat_capture, o.stat_capture
        m = ~np.isnan(a)
        assert np.array_equal(m, ~np.isnan(b)) and np.all(np.abs(a[m] - b[m]) <= 1e-10 * np.abs(b[m]) + 1e-12)
        for f in ("uncorrected", "corrected", "converged", "overall"):
            assert np.array_equal(getattr(g, f), getattr(o, f)), (f, grid)
        assert np.array_equal(g.counters, o.counters)
        red = engine.run_batch(spec, grid, 1024, workloads.BASE_SEED, per_sim=False)
        assert np.array_equal(red.counters, g.counters)


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["probit", "poisson", "logit_exclusion"])
def test_large_n_reduced_fit_variants_on_several_waves_match_oracle(family, engine, oracle):
    """The superset kernel (run-time family, sparse-factor exclusion with its level counts, the separation fallback) on
    256 threads per replicate: counts and sums cross the waves through LDS."""
    from mcpower_amd import spec as S
    spec = workloads._glm_spec(3, 0.3)
    spec.effect_sizes[1], spec.effect_sizes[2] = 0.08, -0.06
    spec.n_factor_dummies, spec.factor_n_levels, spec.factor_proportions, spec.factor_sampled = 2, [3], [0.55, 0.43, 0.02], [None]
    spec.effect_sizes = list(spec.effect_sizes) + [0.15, 0.1]
    spec.target_indices = [1, 2, 3, 4, 5]
    spec.report_overall = True
    if family == "probit":
        spec.link = S.LinkKind.Probit
    elif family == "poisson":
        spec.outcome_kind = S.OutcomeKind.Count
        spec.intercept = spec.effect_sizes[0] = 0.5
    spec.factor_min_level_count = 12      # the 2 % level: ~6 rows at n = 300 (excluded), ~60 at n = 3000 (kept)
    grid = [300, 3000]
    g = engine.run_batch(spec, grid, 192, workloads.BASE_SEED, stat_capture=True)
    o = oracle.run_batch(spec, grid, 192, workloads.BASE_SEED, stat_capture=True, n_threads=16)
    assert np.array_equal(g.factor_excluded, o.factor_excluded) and g.factor_excluded[:, 0, 0].mean() > 0.8 and not g.factor_excluded[:, 1, 0].any()
    for f in ("uncorrected", "corrected", "converged", "overall"):
        assert np.array_equal(getattr(g, f), getattr(o, f)), f
    assert np.array_equal(g.counters, o.counters)
