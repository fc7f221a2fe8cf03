"""Device vs oracle on the clustered GLM (random-intercept GLMM, Laplace): same objective, same simplex, the
family kernel differs in the last bits (short exp / log1p on the device, glibc in the oracle), so decisions agree
replicate by replicate up to a rare borderline case and statistics to ~1e-6."""
import numpy as np
import pytest

from helpers import BASE_SEED, load_case
from mcpower_amd import spec as S

pytestmark = pytest.mark.gpu


def _compare(dev, ora, n_sims, stat_dev=None, stat_ora=None, slack=1):
    same = ((dev.uncorrected == ora.uncorrected).all(axis=(1, 2)) & (dev.converged == ora.converged).all(axis=1)
            & (dev.boundary_hit == ora.boundary_hit).all(axis=1) & (dev.joint_unc == ora.joint_unc).all(axis=1)
            & (dev.pinned_components == ora.pinned_components).all(axis=1))
    assert same.sum() >= n_sims - slack, np.flatnonzero(~same)
    both = (dev.converged == 1) & (ora.converged == 1)
    assert np.allclose(dev.tau_squared_hat[both], ora.tau_squared_hat[both], rtol=1e-4, atol=1e-7)
    assert np.isnan(dev.tau_squared_hat[dev.converged == 0]).all()
    if stat_dev is not None:
        ok = both[:, 0] & same
        assert np.allclose(stat_dev[ok], stat_ora[ok], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("truth", [True, False])
def test_glmm_logit_matches_oracle(truth, engine, oracle):
    spec, _, a, low = load_case("glmm_intercept")
    spec.scenario.truth_start = truth
    dev, stat = engine.run_batch_st_capture(spec, [300, 600], 64, BASE_SEED)
    ora = oracle.run_batch(spec, [300, 600], 64, BASE_SEED, n_threads=8, stat_capture=True)
    assert dev.converged.mean() > 0.95 and dev.shape.n_variance_components == 1
    _compare(dev, ora, 64, stat, ora.stat_capture)


@pytest.mark.parametrize("family", ["probit", "poisson"])
def test_glmm_probit_and_poisson_match_oracle(family, engine, oracle):
    spec, _, a, low = load_case("glmm_intercept")
    if family == "probit":
        spec.link = S.LinkKind.Probit
        spec.effect_sizes = [-0.52, 0.30]
        spec.cluster.tau_squared = 0.25
    else:
        spec.outcome_kind = S.OutcomeKind.Count
        spec.effect_sizes = [0.7, 0.15]
        spec.cluster.tau_squared = 0.10
    spec.scenario.truth_start = True
    dev = engine.run_batch(spec, [600], 48, BASE_SEED)
    ora = oracle.run_batch(spec, [600], 48, BASE_SEED, n_threads=8)
    assert dev.converged.mean() > 0.9
    _compare(dev, ora, 48)


def test_glmm_two_predictors_fixed_size_clusters_counters_and_shards(engine, oracle):
    spec, _, a, low = load_case("glmm_intercept")
    spec.n_non_factor = 2
    spec.correlation = [1.0, 0.3, 0.3, 1.0]
    spec.var_types, spec.var_params = [S.Distribution.Normal] * 2, [0.0, 0.0]
    spec.effect_sizes = [-0.85, 0.5, 0.3]
    spec.target_indices = [1, 2]
    spec.cluster = S.ClusterSpec(sizing=("FixedSize", 15), tau_squared=0.6)
    spec.correction_method = S.CorrectionMethod.Holm
    spec.scenario.truth_start = True
    dev = engine.run_batch(spec, [300, 450], 48, BASE_SEED)
    ora = oracle.run_batch(spec, [300, 450], 48, BASE_SEED, n_threads=8)
    _compare(dev, ora, 48)
    assert np.array_equal(dev.corrected[dev.converged.all(axis=1)].shape, ora.corrected[ora.converged.all(axis=1)].shape)
    red = engine.run_batch(spec, [300, 450], 48, BASE_SEED, per_sim=False)
    assert np.array_equal(red.counters, dev.counters)
    left = engine.run_batch(spec, [300, 450], 30, BASE_SEED, per_sim=False)
    right = engine.run_batch(spec, [300, 450], 18, BASE_SEED, per_sim=False, sim_offset=30)
    assert np.array_equal(left.counters + right.counters, dev.counters)


def test_glmm_refuses_what_it_cannot_do(engine):
    spec, _, a, low = load_case("glmm_intercept")
    spec.cluster.extra_groupings = [{"relation": {"Crossed": {"n_clusters": 8}}, "tau_squared": 0.2, "slopes": []}]
    spec.nagq = 5
    with pytest.raises(S.EngineError) as e:
        engine.run_batch(spec, [600], 4, BASE_SEED)
    assert e.value.status == S.Status.Unsupported and "extra groupings" in str(e.value)


# ---- extra groupings: glmm_general_kernels.hip vs oracle/fit_glmm_general.c (formulas.R:1130-1181) ----

@pytest.mark.parametrize("name,n_sims", [("glmm_slope_crossed", 24), ("glmm_slope_crossed_extra_slopes", 16),
                                         ("glmm_slope_nested_extra_slopes", 12)])
def test_glmm_with_extra_groupings_matches_oracle(name, n_sims, engine, oracle):
    spec, _, a, low = load_case(name)
    spec.scenario.truth_start = True
    dev, stat = engine.run_batch_st_capture(spec, [low["n"]], n_sims, BASE_SEED)
    ora = oracle.run_batch(spec, [low["n"]], n_sims, BASE_SEED, n_threads=8, stat_capture=True)
    assert dev.converged.mean() > 0.7
    _compare(dev, ora, n_sims, stat, ora.stat_capture, slack=2)


@pytest.mark.parametrize("link", ["logit", "poisson"])
def test_glmm_extra_groupings_cap_exit_of_the_fisher_loop(link, engine, oracle, monkeypatch):
    """Every evaluation is driven into the Fisher-step cap (lowered to 4 on both sides: MCP_DEBUG_GLMM_PIRLS_CAP for
    the kernel, mcpo_set_glmm_general_pirls_cap for the oracle).  On that exit the modes move once more AFTER the last
    factor of I + A'WA was formed, so the Wald tail's weights -- recomputed at the final modes, as
    oracle/fit_glmm_general.c:275-287 does -- differ from the last Fisher step's: the branch a converged loop never
    takes.  Statistics must still agree, and must differ from the uncapped fit (the knob took effect)."""
    spec, _, a, low = load_case("glmm_slope_crossed")
    if link == "poisson":
        spec.outcome_kind = S.OutcomeKind.Count
        spec.effect_sizes = [0.5, 0.15]
        spec.cluster.tau_squared = 0.10
    spec.scenario.truth_start = True
    free, stat_free = engine.run_batch_st_capture(spec, [low["n"]], 12, BASE_SEED)
    monkeypatch.setenv("MCP_DEBUG_GLMM_PIRLS_CAP", "4")
    oracle.set_glmm_general_pirls_cap(4)
    try:
        dev, stat = engine.run_batch_st_capture(spec, [low["n"]], 12, BASE_SEED)
        ora = oracle.run_batch(spec, [low["n"]], 12, BASE_SEED, n_threads=8, stat_capture=True)
    finally:
        oracle.set_glmm_general_pirls_cap(0)
    _compare(dev, ora, 12, stat, ora.stat_capture, slack=2)
    both = (dev.converged[:, 0] == 1) & (free.converged[:, 0] == 1)
    assert both.sum() >= 6 and not np.allclose(stat[both], stat_free[both], rtol=1e-9, atol=0)


@pytest.mark.parametrize("link", ["logit", "probit", "poisson"])
def test_glmm_extra_groupings_under_a_scenario_preset(link, engine, oracle):
    """General data path + heavy-tailed random effects + one ICC jitter per grouping; nested grouping on a two-point
    grid; statistics compared, not only decisions."""
    from helpers import preset
    spec, _, a, low = load_case("glmm_slope_nested_extra_slopes")
    if link == "probit":
        spec.link = S.LinkKind.Probit
        spec.effect_sizes = [-0.52, 0.30]
        spec.cluster.tau_squared = 0.25
    if link == "poisson":
        spec.outcome_kind = S.OutcomeKind.Count
        spec.effect_sizes = [0.5, 0.15]
        spec.cluster.tau_squared = 0.10
    spec.scenario = preset("realistic")
    spec.scenario.lme = S.LmeScenarioPerturbations(random_effect_dist=S.ResidualDist.HighKurtosis, random_effect_df=6.0,
                                                   icc_noise_sd=0.05)
    dev, stat = engine.run_batch_st_capture(spec, [360, 720], 12, BASE_SEED)
    ora = oracle.run_batch(spec, [360, 720], 12, BASE_SEED, n_threads=8, stat_capture=True)
    assert dev.converged.mean() > 0.6
    _compare(dev, ora, 12, stat, ora.stat_capture, slack=2)


def test_glmm_extra_groupings_two_predictors_test_formula_and_counts(engine, oracle):
    """P = 3 with the slope on the second predictor, a crossed grouping, a test_formula dropping the first predictor;
    then the same design with a Count outcome."""
    spec, _, a, low = load_case("glmm_slope_crossed")
    spec.n_non_factor = 2
    spec.correlation = [1.0, 0.2, 0.2, 1.0]
    spec.var_types, spec.var_params = [S.Distribution.Normal] * 2, [0.0, 0.0]
    spec.effect_sizes = [-0.85, 0.3, 0.5]
    spec.target_indices = [1, 2]
    spec.cluster.slopes = [{"column": 1, "variance": 0.10, "corr_with_intercept": 0.3, "corr_with": []}]
    spec.cluster_slope_design_cols = [2]
    spec.scenario.truth_start = True
    dev, stat = engine.run_batch_st_capture(spec, [480], 12, BASE_SEED)
    ora = oracle.run_batch(spec, [480], 12, BASE_SEED, n_threads=8, stat_capture=True)
    _compare(dev, ora, 12, stat, ora.stat_capture, slack=2)
    spec.fit_columns = [0, 2]
    spec.target_indices = [2]
    dev, stat = engine.run_batch_st_capture(spec, [480], 12, BASE_SEED)
    ora = oracle.run_batch(spec, [480], 12, BASE_SEED, n_threads=8, stat_capture=True)
    _compare(dev, ora, 12, stat, ora.stat_capture, slack=2)
    spec.outcome_kind = S.OutcomeKind.Count
    spec.effect_sizes = [0.5, 0.10, 0.15]
    spec.cluster.tau_squared = 0.10
    dev, stat = engine.run_batch_st_capture(spec, [480], 12, BASE_SEED)
    ora = oracle.run_batch(spec, [480], 12, BASE_SEED, n_threads=8, stat_capture=True)
    assert dev.converged.mean() > 0.7
    _compare(dev, ora, 12, stat, ora.stat_capture, slack=2)


def test_glmm_extra_groupings_hessian_wald_se(engine, oracle):
    """Wald SE mode "hessian" through the joint-mode kernel: interior fits get the larger, Hessian-based SE; singular
    fits difference over their free parameters (a block that does not factor keeps "rx"); statistics against the oracle at the stencil's amplification (2e-3)."""
    spec, _, a, low = load_case("glmm_slope_crossed")
    spec.cluster.tau_squared = 1.2
    spec.cluster.slopes = [{"column": 0, "variance": 0.6, "corr_with_intercept": 0.2, "corr_with": []}]
    spec.cluster.extra_groupings[0]["tau_squared"] = 0.5
    spec.scenario.truth_start = True
    rx, stat_rx = engine.run_batch_st_capture(spec, [480], 16, BASE_SEED)
    spec.wald_se = S.WaldSe.Hessian
    dev, stat = engine.run_batch_st_capture(spec, [480], 16, BASE_SEED)
    ora = oracle.run_batch(spec, [480], 16, BASE_SEED, n_threads=8, stat_capture=True)
    same = ((dev.uncorrected == ora.uncorrected).all(axis=(1, 2)) & (dev.converged == ora.converged).all(axis=1)
            & (dev.pinned_components == ora.pinned_components).all(axis=1))
    assert same.sum() >= 14
    ok = same & (dev.converged[:, 0] == 1)
    assert np.allclose(stat[ok], ora.stat_capture[ok], rtol=2e-3)
    assert np.array_equal(dev.converged, rx.converged) and np.array_equal(dev.pinned_components, rx.pinned_components)
    interior = ok & (dev.pinned_components[:, 0] == 0)
    assert interior.sum() >= 4 and np.all(stat[interior] < stat_rx[interior])
    singular = ok & (dev.pinned_components[:, 0] != 0)  # the stencil runs over the free parameters: never smaller SEs
    assert np.all(stat[singular] <= stat_rx[singular])


def test_glmm_extra_groupings_reference_capture_and_grid(engine, oracle):
    """sim 0 of the reference's glmm_slope_crossed seed equals a refit of the captured bytes; a two-point grid carries
    theta-hat; counters-only aggregation equals the per-replicate run."""
    spec, base_seed, a, low = load_case("glmm_slope_crossed")
    spec.scenario.truth_start = True
    dev, stat = engine.run_batch_st_capture(spec, [low["n"]], 1, base_seed)
    ids = [a["cluster_ids"]] + [a["extra_ids"][g] for g in range(a["extra_ids"].shape[0])]
    f = oracle.glmm_general_fit(a["design"], a["outcome"], ids, [2, 1], [[1], []], family="logit",
                                theta_start=oracle.truth_theta(spec), beta_start=spec.effect_sizes)
    assert dev.converged[0, 0] == 1 and f["converged"]
    assert np.isclose(stat[0, 0], (f["beta"][1] / f["se"][1]) ** 2, rtol=1e-5)
    assert dev.pinned_components[0, 0] == f["pinned"]
    dev = engine.run_batch(spec, [240, 480], 12, BASE_SEED)
    ora = oracle.run_batch(spec, [240, 480], 12, BASE_SEED, n_threads=8)
    _compare(dev, ora, 12, slack=2)
    red = engine.run_batch(spec, [240, 480], 12, BASE_SEED, per_sim=False)
    assert np.array_equal(red.counters, dev.counters)


# ---- random slopes on the primary grouping: glmm_slopes_kernels.hip vs oracle/fit_glmm_slopes.c ----

@pytest.mark.parametrize("name,truth", [("glmm_slope", True), ("glmm_slope", False), ("glmm_multislope", True)])
def test_glmm_slopes_match_oracle(name, truth, engine, oracle):
    spec, _, a, low = load_case(name)
    spec.scenario.truth_start = truth
    n_sims = 32 if name == "glmm_slope" else 16
    ns = [400, low["n"]]
    dev, stat = engine.run_batch_st_capture(spec, ns, n_sims, BASE_SEED)
    ora = oracle.run_batch(spec, ns, n_sims, BASE_SEED, n_threads=8, stat_capture=True)
    assert dev.converged.mean() > 0.85
    _compare(dev, ora, n_sims, stat, ora.stat_capture, slack=2)


def test_glmm_slopes_dataset_is_the_reference_capture(engine, oracle):
    """sim 0 of the reference's glmm_slope seed: the device's decision equals a refit of the captured bytes."""
    spec, base_seed, a, low = load_case("glmm_slope")
    spec.scenario.truth_start = True
    dev, stat = engine.run_batch_st_capture(spec, [low["n"]], 1, base_seed)
    f = oracle.glmm_slopes_fit(a["design"], a["outcome"], a["cluster_ids"], [1], family="logit",
                               theta_start=oracle.truth_theta(spec), beta_start=spec.effect_sizes)
    assert dev.converged[0, 0] == 1 and f["converged"]
    assert np.isclose(stat[0, 0], (f["beta"][1] / f["se"][1]) ** 2, rtol=1e-5)
    assert np.isclose(dev.tau_squared_hat[0, 0], f["theta"][0] ** 2, rtol=1e-4)
    assert dev.pinned_components[0, 0] == f["pinned"]


def test_glmm_slopes_fixed_size_poisson_second_column_and_test_formula(engine, oracle):
    """FixedSize clusters (the cluster count follows the grid: 30 clusters share lanes pairwise, 40 do not), a Poisson
    outcome, the slope on the SECOND predictor, and a test_formula that drops the first one."""
    spec, _, a, low = load_case("glmm_slope")
    spec.outcome_kind = S.OutcomeKind.Count
    spec.n_non_factor = 2
    spec.correlation = [1.0, 0.2, 0.2, 1.0]
    spec.var_types, spec.var_params = [S.Distribution.Normal] * 2, [0.0, 0.0]
    spec.effect_sizes = [0.6, 0.10, 0.20]
    spec.target_indices = [2]
    spec.cluster = S.ClusterSpec(sizing=("FixedSize", 15), tau_squared=0.10,
                                 slopes=[{"column": 1, "variance": 0.03, "corr_with_intercept": -0.2, "corr_with": []}])
    spec.cluster_slope_design_cols = [2]
    spec.scenario.truth_start = True
    dev = engine.run_batch(spec, [450, 600], 24, BASE_SEED)
    ora = oracle.run_batch(spec, [450, 600], 24, BASE_SEED, n_threads=8)
    assert dev.converged.mean() > 0.8
    _compare(dev, ora, 24, slack=2)
    spec.fit_columns = [0, 2]
    dev = engine.run_batch(spec, [450], 24, BASE_SEED)
    ora = oracle.run_batch(spec, [450], 24, BASE_SEED, n_threads=8)
    _compare(dev, ora, 24, slack=2)
    spec.fit_columns = [0, 1]
    spec.target_indices = [1]
    with pytest.raises(S.EngineError) as e:
        engine.run_batch(spec, [450], 4, BASE_SEED)
    assert e.value.status == S.Status.Unsupported and "random slope" in str(e.value)


# ---- Wald SE mode "hessian": finite-difference Hessian of the Laplace deviance (mcp_glmm_hessian.hpp) ----

@pytest.mark.parametrize("name", ["glmm_intercept", "glmm_slope", "glmm_multislope"])
def test_glmm_hessian_wald_se_matches_oracle(name, engine, oracle):
    """Same stencil on both sides; the second differences amplify the last-bit differences of the two family kernels
    (delta^-2 = 1e8), so the statistics agree to ~1e-4 instead of 1e-6.  Singular fits keep the "rx" covariance."""
    spec, _, a, low = load_case(name)
    spec.scenario.truth_start = True
    n_sims = 32 if name != "glmm_multislope" else 12
    rx, stat_rx = engine.run_batch_st_capture(spec, [low["n"]], n_sims, BASE_SEED)
    spec.wald_se = S.WaldSe.Hessian
    dev, stat = engine.run_batch_st_capture(spec, [low["n"]], n_sims, BASE_SEED)
    ora = oracle.run_batch(spec, [low["n"]], n_sims, BASE_SEED, n_threads=8, stat_capture=True)
    same = ((dev.uncorrected == ora.uncorrected).all(axis=(1, 2)) & (dev.converged == ora.converged).all(axis=1)
            & (dev.joint_unc == ora.joint_unc).all(axis=1) & (dev.pinned_components == ora.pinned_components).all(axis=1))
    assert same.sum() >= n_sims - 2
    ok = same & (dev.converged[:, 0] == 1)
    assert np.allclose(stat[ok], ora.stat_capture[ok], rtol=2e-3)
    # the mode moves the statistic only: estimates, convergence and pinned masks are the "rx" run's
    assert np.array_equal(dev.converged, rx.converged) and np.array_equal(dev.pinned_components, rx.pinned_components)
    interior = ok & (dev.pinned_components[:, 0] == 0)
    assert np.all(stat[interior] < stat_rx[interior])            # Hessian SEs are the larger ones
    singular = ok & (dev.pinned_components[:, 0] != 0)           # pinned components drop out of the stencil, the coupling
    assert np.all(stat[singular] <= stat_rx[singular])           # with the free ones stays ("rx" only if H does not factor)


# ---- adaptive Gauss-Hermite quadrature (spec.nagq) ----

@pytest.mark.parametrize("shape,nagq", [("intercept", 7), ("intercept", 25), ("laplace_bias", 7)])
def test_glmm_agq_matches_oracle(shape, nagq, engine, oracle):
    """lme4's glmerAGQ objective on both sides; `laplace_bias` is formulas.R:1141-1154 (150 clusters of 4, ICC 0.5)."""
    spec, _, a, low = load_case("glmm_intercept")
    if shape == "laplace_bias":
        spec.cluster = S.ClusterSpec(sizing=("FixedClusters", 150), tau_squared=np.pi ** 2 / 3.0)
    spec.scenario.truth_start = True
    lap = engine.run_batch(spec, [600], 32, BASE_SEED)
    spec.nagq = nagq
    dev, stat = engine.run_batch_st_capture(spec, [600], 32, BASE_SEED)
    ora = oracle.run_batch(spec, [600], 32, BASE_SEED, n_threads=8, stat_capture=True)
    assert dev.converged.mean() > 0.9
    _compare(dev, ora, 32, stat, ora.stat_capture)
    both = (dev.converged[:, 0] == 1) & (lap.converged[:, 0] == 1)
    assert not np.allclose(dev.tau_squared_hat[both], lap.tau_squared_hat[both], rtol=1e-4)   # the rule changes the fit
    if shape == "laplace_bias":   # Laplace shrinks tau-hat^2 on tiny clusters; the quadrature undoes it
        assert np.mean(dev.tau_squared_hat[both]) > np.mean(lap.tau_squared_hat[both])


@pytest.mark.parametrize("case,nagq", [("glmm_slope", 5), ("glmm_slope", 9), ("glmm_multislope", 3)])
def test_glmm_agq_with_random_slopes_matches_oracle(case, nagq, engine, oracle):
    """validate.rs:918-941: nagq > 1 for ONE grouping with <= 3 random effects -- the k^q tensor adaptive rule of
    glmm_slopes_kernel (25 / 81 nodes per cluster at q = 2, 27 at q = 3) against oracle/fit_glmm_slopes.c, which is pinned
    to the exact marginal likelihood (tests/test_oracle_glmm.py)."""
    spec, _, a, low = load_case(case)
    n = int(low["n"]) if "n" in low else a["design"].shape[0]
    spec.scenario.truth_start = True
    lap = engine.run_batch(spec, [n], 24, BASE_SEED)
    spec.nagq = nagq
    dev, stat = engine.run_batch_st_capture(spec, [n], 24, BASE_SEED)
    ora = oracle.run_batch(spec, [n], 24, BASE_SEED, n_threads=8, stat_capture=True)
    assert dev.converged.mean() > 0.85
    _compare(dev, ora, 24, stat, ora.stat_capture, slack=1)
    both = (dev.converged[:, 0] == 1) & (lap.converged[:, 0] == 1)
    assert not np.allclose(dev.tau_squared_hat[both], lap.tau_squared_hat[both], rtol=1e-5)   # the rule changes the fit
    red = engine.run_batch(spec, [n], 24, BASE_SEED, per_sim=False)
    assert np.array_equal(red.counters, dev.counters)


def test_glmm_agq_with_extra_groupings_is_refused(engine):
    spec, _, a, low = load_case("glmm_slope_crossed")
    spec.nagq = 5
    with pytest.raises(S.EngineError) as e:
        engine.run_batch(spec, [a["design"].shape[0]], 4, BASE_SEED)
    assert e.value.status == S.Status.Unsupported and "nagq" in str(e.value)
    spec, _, a, low = load_case("glmm_intercept")
    spec.nagq = 4
    with pytest.raises(S.EngineError) as e:
        engine.run_batch(spec, [600], 4, BASE_SEED)
    assert e.value.status == S.Status.InvalidSpec


def test_glmm_slopes_probit_scenario_and_counters(engine, oracle):
    from helpers import preset
    spec, _, a, low = load_case("glmm_slope")
    spec.link = S.LinkKind.Probit
    spec.effect_sizes = [-0.52, 0.30]
    spec.cluster.tau_squared = 0.25
    spec.scenario = preset("realistic")
    spec.scenario.lme = S.LmeScenarioPerturbations(random_effect_dist=S.ResidualDist.HighKurtosis, random_effect_df=6.0,
                                                   icc_noise_sd=0.15)
    dev = engine.run_batch(spec, [600], 32, BASE_SEED)
    ora = oracle.run_batch(spec, [600], 32, BASE_SEED, n_threads=8)
    assert dev.converged.mean() > 0.8
    _compare(dev, ora, 32, slack=2)
    red = engine.run_batch(spec, [600], 32, BASE_SEED, per_sim=False)
    assert np.array_equal(red.counters, dev.counters)


@pytest.mark.parametrize("preset_name", ["realistic", "doomer"])
def test_glmm_under_scenario_presets(preset_name, engine, oracle):
    """General data path for the clustered GLM: non-normal marginals, per-study effects on the link scale, heavy-tailed
    cluster effects and the per-replicate ICC jitter."""
    from helpers import preset
    spec, _, a, low = load_case("glmm_intercept")
    spec.scenario = preset(preset_name)
    spec.scenario.lme = S.LmeScenarioPerturbations(random_effect_dist=S.ResidualDist.HighKurtosis, random_effect_df=6.0,
                                                   icc_noise_sd=0.15)
    dev = engine.run_batch(spec, [600], 96, BASE_SEED)
    ora = oracle.run_batch(spec, [600], 96, BASE_SEED, n_threads=8)
    assert dev.converged.mean() > 0.9
    _compare(dev, ora, 96, slack=2)


def test_glmm_test_formula_fits_the_kept_columns(engine, oracle):
    """batch.rs `test_formula_reduced_fit_glmm_recovers_marginal_coefficient`: generate with x1 + x2, fit x1 only."""
    spec, _, a, low = load_case("glmm_intercept")
    spec.n_non_factor = 2
    spec.correlation = [1.0, 0.0, 0.0, 1.0]
    spec.var_types, spec.var_params = [S.Distribution.Normal] * 2, [0.0, 0.0]
    spec.effect_sizes = [-0.85, 0.5, 0.4]
    spec.fit_columns = [0, 1]
    spec.target_indices = [1]
    spec.scenario.truth_start = True
    dev, stat = engine.run_batch_st_capture(spec, [300, 600], 48, BASE_SEED)
    ora = oracle.run_batch(spec, [300, 600], 48, BASE_SEED, n_threads=8, stat_capture=True)
    assert dev.converged.mean() > 0.9
    _compare(dev, ora, 48, stat, ora.stat_capture)
