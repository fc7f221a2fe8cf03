"""GPU parity for the reduced-design rounds: sparse-factor exclusion (fixed and sampled allocation),
test_formula column subsets, post-hoc pairwise contrasts, Tukey HSD thresholds.  The device solves the
masked p x p system (identity rows at dropped columns); the oracle gathers the p_red x p_red system the way
the reference does (batch.rs:970-1013) -- the two must agree on every flag and on t^2 to 1e-10."""
import math

import numpy as np
import pytest

from helpers import BASE_SEED
from mcpower_amd import spec as S
from test_gpu_parity import _assert_batch_equal
from test_oracle_exclusion import factor_spec, three_level_posthoc_spec

pytestmark = pytest.mark.gpu


def _both(engine, oracle, spec, ns, n_sims, seed=BASE_SEED, **kw):
    g = engine.run_batch(spec, ns, n_sims, seed, stat_capture=True, **kw)
    o = oracle.run_batch(spec, ns, n_sims, seed, stat_capture=True, n_threads=8, sim_offset=kw.get("sim_offset", 0))
    return g, o


def _assert_reduced_equal(g, o, iterative=False):
    """`iterative`: IRLS statistics carry the stopping rule's slack (deviance change < 1e-8): z^2 is compared at
    1e-7 relative + 1e-9 absolute there, decisions and counters stay exact."""
    if iterative:
        a, b = g.stat_capture, o.stat_capture
        assert np.array_equal(np.isnan(a), np.isnan(b))
        m = ~np.isnan(a)
        assert np.all(np.abs(a[m] - b[m]) <= 1e-7 * np.abs(b[m]) + 1e-9)
        g.stat_capture = o.stat_capture = None
    _assert_batch_equal(g, o)
    for f in ("factor_excluded", "posthoc_unc", "posthoc_cor"):
        ga, oa = getattr(g, f), getattr(o, f)
        assert (ga is None) == (oa is None), f
        if ga is not None:
            assert np.array_equal(ga, oa), f"{f}: {int((ga != oa).sum())} of {ga.size} differ"
    assert g.shape.n_posthoc_contrasts == o.shape.n_posthoc_contrasts


def test_fixed_allocation_exclusion_on_a_grid(engine, oracle):
    """batch.rs `exclusion_is_per_n_on_a_grid`: [0.9, 0.1] excluded at N = 20, included at N = 100."""
    g, o = _both(engine, oracle, factor_spec(0.3, 0.8, [0.9, 0.1], 5), [20, 100], 3000, 42)
    _assert_reduced_equal(g, o)
    assert (g.factor_excluded[:, 0, 0] == 1).all() and not g.factor_excluded[:, 1, 0].any()
    assert not g.uncorrected[:, 0, 1].any() and g.uncorrected[:, 1, 1].sum() > 0
    assert np.isnan(g.stat_capture[:, 1]).all()
    assert g.counter("excl_sparse", 1)[:, 0].tolist() == [3000, 0]


def test_healthy_factor_takes_the_hot_kernel_and_matches_min_count_zero(engine, oracle):
    a = engine.run_batch(factor_spec(0.5, 0.8, [0.5, 0.5], 5), [40], 2000, 99)
    b = engine.run_batch(factor_spec(0.5, 0.8, [0.5, 0.5], 0), [40], 2000, 99)
    for f in ("uncorrected", "corrected", "converged", "overall", "counters"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    assert not a.factor_excluded.any()


def test_sampled_allocation_excludes_per_replicate(engine, oracle):
    spec = factor_spec(0.4, 0.6, [0.85, 0.15], 5, factor_sampled=[True])
    g, o = _both(engine, oracle, spec, [30, 60, 61, 200], 4000, 11)
    _assert_reduced_equal(g, o)
    fx = g.factor_excluded[:, :, 0]
    assert 0 < fx[:, 0].sum() < 4000 and fx[:, 3].sum() == 0
    assert not g.uncorrected[fx == 1][:, 1].any()
    # shard union == single run on this path too
    lo = engine.run_batch(spec, [30, 60, 61, 200], 1500, 11)
    hi = engine.run_batch(spec, [30, 60, 61, 200], 2500, 11, sim_offset=1500)
    assert np.array_equal(np.concatenate([lo.uncorrected, hi.uncorrected]), g.uncorrected)
    assert np.array_equal(lo.counters + hi.counters, g.counters)
    c = engine.run_batch(spec, [30, 60, 61, 200], 4000, 11, per_sim=False, chunk_sims=1024)
    assert np.array_equal(c.counters, g.counters) and c.uncorrected is None


def test_two_factors_interaction_and_contrast_under_exclusion(engine, oracle):
    spec = S.SimulationSpec(n_non_factor=1, n_factor_dummies=3, correlation=[1.0], var_types=[S.Distribution.Normal],
                            var_params=[0.0], factor_n_levels=[2, 3], factor_proportions=[0.9, 0.1, 0.55, 0.3, 0.15],
                            factor_sampled=[True, None], interactions=[[1, 2]],
                            effect_sizes=[0.0, 0.4, 0.5, 0.3, -0.3, 0.2], target_indices=[1, 2, 3, 5],
                            contrast_pairs=[(3, 4), (2, 3)], correction_method=S.CorrectionMethod.Holm,
                            factor_min_level_count=5, report_overall=True)
    g, o = _both(engine, oracle, spec, [24, 48, 150], 3000, 5)
    _assert_reduced_equal(g, o)
    fx = g.factor_excluded
    assert 0 < fx[:, 1, 0].sum() < 3000          # factor 1 sampled: mixed
    assert (fx[:, 0, 1] == 1).all() and not fx[:, 1:, 1].any()   # factor 2 fixed: 0.15 * 24 < 5 rows only at N = 24
    both = (fx[:, 0, 0] == 1)
    assert both.sum() > 0 and not g.uncorrected[both][:, 0, 1:].any()


@pytest.mark.parametrize("method", [S.CorrectionMethod.None_, S.CorrectionMethod.Bonferroni, S.CorrectionMethod.Holm,
                                    S.CorrectionMethod.BenjaminiHochberg, S.CorrectionMethod.TukeyHsd])
def test_posthoc_contrasts_match_oracle(engine, oracle, method):
    spec = three_level_posthoc_spec((0.4, 0.3, 0.3), 5, method)
    spec.effect_sizes = [0.0, 0.3, 0.5, -0.2]
    spec.target_indices = [1, 2, 3] if method != S.CorrectionMethod.TukeyHsd else [2, 3]
    spec.crit_values = S.CritValues(alpha=0.05, posthoc_alpha=0.1)
    g, o = _both(engine, oracle, spec, [45, 90, 300], 3000, 21)
    _assert_reduced_equal(g, o)
    assert g.posthoc_unc.shape == (3000, 3, 3) and g.posthoc_unc[:, 2].mean() > g.posthoc_unc[:, 0].mean()
    assert (g.posthoc_cor <= g.posthoc_unc).all()
    nt = len(spec.target_indices)
    assert g.counter("hist_unc", nt + 3 + 1).sum(axis=1).tolist() == [3000] * 3
    assert np.array_equal(g.counter("posthoc_cor", 3), g.posthoc_cor.sum(axis=0))
    crit = engine.build_crit_table(spec, [45, 90, 300])
    assert np.allclose(crit, oracle.build_crit_table(spec, [45, 90, 300]), rtol=1e-9, equal_nan=True)


def test_excluded_factor_posthoc_contrasts_report_zero(engine, oracle):
    g, o = _both(engine, oracle, three_level_posthoc_spec(), [40, 400], 1500, 42)
    _assert_reduced_equal(g, o)
    assert (g.factor_excluded[:, 0, 0] == 1).all() and not g.posthoc_unc[:, 0].any()
    assert g.posthoc_unc[:, 1].sum() > 0


def test_two_posthoc_blocks_and_sampled_levels(engine, oracle):
    spec = S.SimulationSpec(n_non_factor=1, n_factor_dummies=4, correlation=[1.0], var_types=[S.Distribution.Normal],
                            var_params=[0.0], factor_n_levels=[3, 3], factor_proportions=[0.5, 0.3, 0.2, 0.4, 0.4, 0.2],
                            effect_sizes=[0.0, 0.3, 0.5, -0.2, 0.4, 0.0], target_indices=[1],
                            posthoc=[S.PosthocSpec(0, [2, 3]), S.PosthocSpec(1, [4, 5])],
                            correction_method=S.CorrectionMethod.BenjaminiHochberg, factor_min_level_count=5,
                            scenario=S.ScenarioPerturbations(name="custom", sampled_factor_proportions=True))
    g, o = _both(engine, oracle, spec, [30, 80], 3000, 8)
    _assert_reduced_equal(g, o)
    assert g.shape.n_posthoc_contrasts == 6
    fx = g.factor_excluded[:, 0, :]
    assert 0 < fx[:, 0].sum() < 3000 and 0 < fx[:, 1].sum() < 3000
    assert not g.posthoc_unc[fx[:, 0] == 1][:, 0, :3].any() and not g.posthoc_unc[fx[:, 1] == 1][:, 0, 3:].any()


def test_fit_columns_subset_matches_oracle(engine, oracle):
    spec = S.SimulationSpec(n_non_factor=3, correlation=[1.0, 0.3, 0.0, 0.3, 1.0, 0.2, 0.0, 0.2, 1.0],
                            var_types=[S.Distribution.Normal] * 3, var_params=[0.0] * 3,
                            effect_sizes=[0.0, 0.3, 0.5, 0.2], target_indices=[1, 3], contrast_pairs=[(1, 3)],
                            fit_columns=[0, 1, 3], report_overall=True,
                            correction_method=S.CorrectionMethod.Bonferroni)
    g, o = _both(engine, oracle, spec, [60, 61, 200], 4000, 5)
    _assert_reduced_equal(g, o)
    full = engine.run_batch(S.SimulationSpec(**{**spec.__dict__, "fit_columns": []}), [60, 61, 200], 4000, 5)
    assert not np.array_equal(full.uncorrected, g.uncorrected)


def test_tukey_main_targets_on_the_hot_kernel(engine, oracle):
    spec = three_level_posthoc_spec((0.4, 0.3, 0.3), 0, S.CorrectionMethod.TukeyHsd)
    spec.posthoc = []
    spec.target_indices = [1, 2, 3]
    spec.contrast_pairs = [(3, 2)]
    g, o = _both(engine, oracle, spec, [50, 200], 5000, 9)
    _assert_batch_equal(g, o)
    assert not g.corrected[:, :, 0].any() and g.corrected[:, 1, 1:].sum() > 0


def test_exclusion_under_scenario_perturbations(engine, oracle):
    """realistic-like preset: sampled factor proportions + correlation noise + swaps + heterogeneity."""
    spec = S.SimulationSpec(n_non_factor=2, n_factor_dummies=2, correlation=[1.0, 0.3, 0.3, 1.0],
                            var_types=[S.Distribution.Normal] * 2, var_params=[0.0, 0.0], factor_n_levels=[3],
                            factor_proportions=[0.6, 0.25, 0.15], effect_sizes=[0.0, 0.3, 0.2, 0.5, 0.4],
                            target_indices=[1, 2, 3, 4], factor_min_level_count=5, report_overall=True,
                            posthoc=[S.PosthocSpec(0, [3, 4])], correction_method=S.CorrectionMethod.Holm,
                            scenario=S.ScenarioPerturbations(
                                name="realistic", heterogeneity=0.2, heteroskedasticity_ratio=2.0,
                                correlation_noise_sd=0.15, distribution_change_prob=0.5,
                                new_distributions=[S.Distribution.RightSkewed, S.Distribution.LeftSkewed,
                                                   S.Distribution.Uniform],
                                residual_change_prob=0.5,
                                residual_dists=[S.ResidualDist.HighKurtosis, S.ResidualDist.RightSkewed],
                                residual_df=8.0, sampled_factor_proportions=True))
    g = engine.run_batch(spec, [30, 60, 120], 3000, 77, stat_capture=True)
    o = oracle.run_batch(spec, [30, 60, 120], 3000, 77, stat_capture=True, n_threads=8)
    # the general data path is pinned to <= 1e-5 one-ulp marginal mismatches (test_gpu_parity scenario tests);
    # decisions can only differ where a statistic sits on a threshold
    assert np.array_equal(g.factor_excluded, o.factor_excluded)
    for f in ("uncorrected", "corrected", "posthoc_unc", "posthoc_cor", "overall", "converged"):
        assert (getattr(g, f) != getattr(o, f)).mean() < 2e-4, f
    assert 0 < g.factor_excluded[:, 0, 0].sum() < 3000


# ---------------------------------------------------------------------------------------------
# GLM / MLE arms
# ---------------------------------------------------------------------------------------------
def _glm_factor_spec(effects, props, min_count=5, **kw):
    s = S.SimulationSpec(n_non_factor=1, n_factor_dummies=len(props) - 1, correlation=[1.0],
                         var_types=[S.Distribution.Normal], var_params=[0.0], factor_n_levels=[len(props)],
                         factor_proportions=list(props), effect_sizes=list(effects),
                         target_indices=list(range(1, len(effects))), outcome_kind=S.OutcomeKind.Binary,
                         estimator=S.EstimatorSpec.Glm, report_overall=True, factor_min_level_count=min_count)
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def test_glm_sparse_exclusion_and_overall_df(engine, oracle):
    spec = _glm_factor_spec([math.log(0.3 / 0.7), 0.5, 0.8], [0.9, 0.1])
    g, o = _both(engine, oracle, spec, [30, 200], 2000, 13)
    _assert_reduced_equal(g, o, iterative=True)
    assert (g.factor_excluded[:, 0, 0] == 1).all() and not g.uncorrected[:, 0, 1].any()
    assert g.overall[:, 0].sum() > 0
    spec = _glm_factor_spec([math.log(0.3 / 0.7), 0.5, 0.8, -0.4], [0.5, 0.3, 0.2], factor_sampled=[True],
                            correction_method=S.CorrectionMethod.Holm)
    g, o = _both(engine, oracle, spec, [25, 50, 120], 2000, 14)
    _assert_reduced_equal(g, o, iterative=True)
    assert 0 < (g.factor_excluded[:, 0, 0] == 1).sum() < 2000


def test_glm_separation_fallback_codes_match_oracle(engine, oracle):
    spec = _glm_factor_spec([0.0, 2.0, 6.0], [0.6, 0.4])
    g, o = _both(engine, oracle, spec, [20, 40], 3000, 13)
    _assert_reduced_equal(g, o, iterative=True)
    sep = g.factor_excluded[:, 0, 0] == 2
    assert 100 < sep.sum() < 2500 and g.converged[sep, 0].mean() > 0.9
    assert g.counter("excl_sep", 1)[0, 0] == sep.sum() and g.counter("excl_sparse", 1)[0, 0] == 0
    off = engine.run_batch(S.SimulationSpec(**{**spec.__dict__, "factor_min_level_count": 0}), [20, 40], 3000, 13)
    assert off.converged.sum() < g.converged.sum() and not off.factor_excluded.any()
    assert np.array_equal(off.uncorrected[~sep, 0], g.uncorrected[~sep, 0])
    # two factors: the fallback drops the included factor with the smallest minimum level count
    spec = S.SimulationSpec(n_non_factor=1, n_factor_dummies=2, correlation=[1.0], var_types=[S.Distribution.Normal],
                            var_params=[0.0], factor_n_levels=[2, 2], factor_proportions=[0.5, 0.5, 0.7, 0.3],
                            effect_sizes=[0.0, 1.5, 1.0, 6.0], target_indices=[1, 2, 3],
                            outcome_kind=S.OutcomeKind.Binary, estimator=S.EstimatorSpec.Glm, report_overall=True,
                            factor_min_level_count=5)
    g, o = _both(engine, oracle, spec, [24], 3000, 15)
    _assert_reduced_equal(g, o, iterative=True)
    fx = g.factor_excluded[:, 0, :]
    assert (fx[:, 1] == 2).sum() > 0 and not (fx[:, 0] == 2).any()   # factor 2 has the smaller minimum count (7 vs 12)


def test_glm_fit_columns_subset(engine, oracle):
    spec = S.SimulationSpec(n_non_factor=3, correlation=np.eye(3).ravel().tolist(),
                            var_types=[S.Distribution.Normal] * 3, var_params=[0.0] * 3,
                            effect_sizes=[math.log(0.4 / 0.6), 0.5, 0.6, 0.3], target_indices=[1, 3],
                            fit_columns=[0, 1, 3], outcome_kind=S.OutcomeKind.Binary,
                            estimator=S.EstimatorSpec.Glm, report_overall=True)
    g, o = _both(engine, oracle, spec, [80, 160], 2000, 16)
    _assert_reduced_equal(g, o, iterative=True)


def test_lmm_exclusion_reduces_fit_and_joint_test(engine, oracle):
    spec = S.SimulationSpec(n_non_factor=1, n_factor_dummies=1, correlation=[1.0], var_types=[S.Distribution.Normal],
                            var_params=[0.0], factor_n_levels=[2], factor_proportions=[0.95, 0.05],
                            effect_sizes=[0.0, 0.4, 0.6], target_indices=[1, 2], estimator=S.EstimatorSpec.Mle,
                            cluster=S.ClusterSpec(sizing=("FixedClusters", 8), tau_squared=0.2),
                            factor_min_level_count=5,
                            scenario=S.ScenarioPerturbations(name="optimistic", truth_start=True))
    g = engine.run_batch(spec, [40, 80, 240], 1500, 17, stat_capture=True)
    o = oracle.run_batch(spec, [40, 80, 240], 1500, 17, stat_capture=True, n_threads=8)
    assert np.array_equal(g.factor_excluded, o.factor_excluded)
    assert (g.factor_excluded[:, 0, 0] == 1).all() and not g.factor_excluded[:, 2, 0].any()
    assert np.array_equal(np.isnan(g.stat_capture), np.isnan(o.stat_capture))
    m = ~np.isnan(o.stat_capture)
    assert np.max(np.abs(g.stat_capture[m] - o.stat_capture[m]) / np.maximum(o.stat_capture[m], 1e-12)) < 1e-4
    for f in ("uncorrected", "corrected", "joint_unc", "converged"):
        assert (getattr(g, f) != getattr(o, f)).mean() < 2e-3, f
    assert not g.uncorrected[:, 0, 1].any()
    # sampled levels: per-replicate reduced rounds must not feed the theta carry of later grid points
    spec.factor_proportions = [0.8, 0.2]
    spec.factor_sampled = [True]
    g = engine.run_batch(spec, [24, 48, 96], 1500, 18)
    o = oracle.run_batch(spec, [24, 48, 96], 1500, 18, n_threads=8)
    assert np.array_equal(g.factor_excluded, o.factor_excluded) and 0 < g.factor_excluded[:, 0, 0].sum() < 1500
    for f in ("uncorrected", "joint_unc", "converged", "boundary_hit"):
        assert (getattr(g, f) != getattr(o, f)).mean() < 1e-2, f


# ---------------------------------------------------------------------------------------------
# scenario perturbations under the GLM / MLE estimators (incl. the LME knobs)
# ---------------------------------------------------------------------------------------------
def _lme_preset(name):
    from helpers import preset
    sc = preset(name)
    if name == "realistic":
        sc.lme = S.LmeScenarioPerturbations(S.ResidualDist.HighKurtosis, 10.0, 0.15)
    elif name == "doomer":
        sc.lme = S.LmeScenarioPerturbations(S.ResidualDist.HighKurtosis, 5.0, 0.3)
    return sc


@pytest.mark.parametrize("name", ["realistic", "doomer"])
def test_glm_under_scenario_presets(engine, oracle, name):
    from helpers import preset
    from test_gpu_parity import _design_mismatch
    spec = S.SimulationSpec(n_non_factor=2, n_factor_dummies=2, correlation=[1.0, 0.4, 0.4, 1.0],
                            var_types=[S.Distribution.Normal] * 2, var_params=[0.0, 0.0], factor_n_levels=[3],
                            factor_proportions=[0.5, 0.3, 0.2], effect_sizes=[math.log(0.3 / 0.7), 0.5, 0.3, 0.6, -0.4],
                            target_indices=[1, 2, 3, 4], outcome_kind=S.OutcomeKind.Binary,
                            estimator=S.EstimatorSpec.Glm, report_overall=True, factor_min_level_count=5,
                            correction_method=S.CorrectionMethod.BenjaminiHochberg, scenario=preset(name))
    bx, by, worst, total = _design_mismatch(engine, oracle, spec, 150, 24)
    assert by == 0 and bx <= max(2, total * 1e-5), (bx, by, worst)          # binary outcomes agree exactly
    g = engine.run_batch(spec, [60, 150, 400], 2500, 31, stat_capture=True)
    o = oracle.run_batch(spec, [60, 150, 400], 2500, 31, stat_capture=True, n_threads=8)
    assert np.array_equal(g.factor_excluded, o.factor_excluded)
    for f in ("uncorrected", "corrected", "overall", "converged"):
        assert (getattr(g, f) != getattr(o, f)).mean() < 5e-4, f
    assert g.uncorrected[:, 2, 0].mean() > 0.8


@pytest.mark.parametrize("link", ["probit", "poisson"])
def test_other_glm_families_under_heterogeneity(engine, oracle, link):
    """validation scen_he_probit / scen_he_poisson shape: per-study effects on the link scale."""
    spec = S.SimulationSpec(n_non_factor=2, correlation=[1.0, 0.0, 0.0, 1.0], var_types=[S.Distribution.Normal] * 2,
                            var_params=[0.0, 0.0], target_indices=[1, 2], estimator=S.EstimatorSpec.Glm,
                            scenario=S.ScenarioPerturbations(name="he", heterogeneity=0.4))
    if link == "probit":
        spec.outcome_kind, spec.link, spec.effect_sizes = S.OutcomeKind.Binary, S.LinkKind.Probit, [-0.5, 0.5, 0.3]
    else:
        spec.outcome_kind, spec.effect_sizes = S.OutcomeKind.Count, [math.log(2.0), 0.5, 0.3]
    g = engine.run_batch(spec, [200], 3000, 41, stat_capture=True)
    o = oracle.run_batch(spec, [200], 3000, 41, stat_capture=True, n_threads=8)
    m = ~np.isnan(o.stat_capture)
    assert np.array_equal(np.isnan(g.stat_capture), ~m)
    assert np.all(np.abs(g.stat_capture[m] - o.stat_capture[m]) <= 1e-6 * np.abs(o.stat_capture[m]) + 1e-8)
    assert np.array_equal(g.uncorrected, o.uncorrected) and np.array_equal(g.converged, o.converged)


@pytest.mark.parametrize("name", ["realistic", "doomer"])
def test_lmm_under_scenario_presets_with_lme_knobs(engine, oracle, name):
    from test_gpu_parity import _design_mismatch
    spec = S.SimulationSpec(n_non_factor=2, n_factor_dummies=1, correlation=[1.0, 0.3, 0.3, 1.0],
                            var_types=[S.Distribution.Normal] * 2, var_params=[0.0, 0.0], factor_n_levels=[2],
                            factor_proportions=[0.7, 0.3], effect_sizes=[0.0, 0.3, 0.2, 0.4], target_indices=[1, 2, 3],
                            estimator=S.EstimatorSpec.Mle, factor_min_level_count=5,
                            cluster=S.ClusterSpec(sizing=("FixedClusters", 12), tau_squared=0.25),
                            scenario=_lme_preset(name))
    bx, by, worst, total = _design_mismatch(engine, oracle, spec, 120, 24)
    assert bx + by <= max(3, total * 2e-5) and worst < 2.5e-7, (bx, by, worst)
    g = engine.run_batch(spec, [36, 120, 240], 1500, 51, stat_capture=True)
    o = oracle.run_batch(spec, [36, 120, 240], 1500, 51, stat_capture=True, n_threads=8)
    assert np.array_equal(g.factor_excluded, o.factor_excluded) and g.factor_excluded[:, 0, 0].sum() > 0
    for f in ("uncorrected", "corrected", "joint_unc", "converged", "boundary_hit"):
        assert (getattr(g, f) != getattr(o, f)).mean() < 1e-2, f
    m = ~np.isnan(o.stat_capture) & ~np.isnan(g.stat_capture)
    rel = np.abs(g.stat_capture[m] - o.stat_capture[m]) / np.maximum(o.stat_capture[m], 1e-9)
    assert np.quantile(rel, 0.99) < 1e-4


# ---------------------------------------------------------------------------------------------
# wide OLS designs (9..12 columns, up to 12 tests): run-time sized solve on the reduced-design kernel
# ---------------------------------------------------------------------------------------------
def test_wide_ols_eleven_predictors(engine, oracle):
    from helpers import base_spec
    spec = base_spec(11)
    c = np.eye(11)
    c[0, 1] = c[1, 0] = 0.4
    c[2, 5] = c[5, 2] = -0.3
    c[9, 10] = c[10, 9] = 0.5
    spec.correlation = c.flatten().tolist()
    spec.effect_sizes = [0.0, 0.25, 0.2, 0.15, 0.1, 0.0, 0.3, -0.2, 0.0, 0.12, 0.18, -0.15]
    spec.target_indices = list(range(1, 12))
    spec.contrast_pairs = [(1, 2)]
    spec.correction_method = S.CorrectionMethod.Holm
    spec.report_overall = True
    g, o = _both(engine, oracle, spec, [40, 41, 150, 400], 3000, 61)
    _assert_reduced_equal(g, o)
    assert g.shape.n_targets == 12 and g.uncorrected[:, 3, 0].mean() > 0.9
    d, y = engine.generate_design(spec, 150, 3, 61, 0)
    for s in range(3):
        xo, yo, _ = oracle.generate_sim_data(spec, 61, s, 150)
        assert np.array_equal(d[s].view(np.uint32), xo.view(np.uint32)) and np.array_equal(y[s], yo)
    f = engine.fit_design(d, y)
    for s in range(3):
        fo = oracle.ols_fit(d[s], y[s])
        assert np.allclose(f["beta"][s], fo["beta"], rtol=1e-10, atol=1e-13)
        assert np.allclose(f["t_sq"][s], fo["t_sq"], rtol=1e-9)


def test_ols_wide_fifteen_predictors_like_the_reference_bench(engine, oracle):
    """`ols_wide` of the reference's throughput bench (k = 15, P = 16: throughput.rs): the widest design one u64
    of pending-tail bits and the reduced kernel's register file carry; 16 tests incl. a contrast pair."""
    from helpers import base_spec
    spec = base_spec(15)
    c = np.eye(15)
    c[0, 1] = c[1, 0] = 0.3
    c[7, 14] = c[14, 7] = -0.25
    spec.correlation = c.flatten().tolist()
    spec.effect_sizes = [0.0] + [0.18, 0.18, 0.0, 0.1, -0.1, 0.0, 0.05, 0.2, 0.0, -0.15, 0.12, 0.0, 0.08, 0.0, 0.22]
    spec.target_indices = list(range(1, 16))
    spec.contrast_pairs = [(1, 2)]
    spec.correction_method = S.CorrectionMethod.BenjaminiHochberg
    spec.report_overall = True
    g, o = _both(engine, oracle, spec, [40, 200, 500], 2000, 77)
    _assert_reduced_equal(g, o)
    assert g.shape.n_targets == 16
    d, y = engine.generate_design(spec, 200, 2, 77, 0)
    for s in range(2):
        xo, yo, _ = oracle.generate_sim_data(spec, 77, s, 200)
        assert np.array_equal(d[s].view(np.uint32), xo.view(np.uint32)) and np.array_equal(y[s], yo)
    f = engine.fit_design(d, y)
    for s in range(2):
        fo = oracle.ols_fit(d[s], y[s])
        assert np.allclose(f["beta"][s], fo["beta"], rtol=1e-10, atol=1e-13) and np.allclose(f["t_sq"][s], fo["t_sq"], rtol=1e-9)


def test_wide_ols_factors_interactions_posthoc_and_exclusion(engine, oracle):
    spec = S.SimulationSpec(
        n_non_factor=3, n_factor_dummies=5, correlation=[1.0, 0.3, 0.0, 0.3, 1.0, 0.2, 0.0, 0.2, 1.0],
        var_types=[S.Distribution.Normal] * 3, var_params=[0.0] * 3, factor_n_levels=[3, 4],
        factor_proportions=[0.5, 0.3, 0.2, 0.4, 0.3, 0.2, 0.1], factor_sampled=[None, True],
        interactions=[[1, 4], [1, 5], [2, 3]],
        effect_sizes=[0.0, 0.3, 0.2, 0.1, 0.4, -0.3, 0.2, 0.3, 0.5, 0.15, -0.1, 0.2],
        target_indices=[1, 2, 3, 4, 5, 9, 10, 11], contrast_pairs=[(6, 7), (8, 6)],
        posthoc=[S.PosthocSpec(1, [6, 7, 8])], correction_method=S.CorrectionMethod.BenjaminiHochberg,
        factor_min_level_count=5, report_overall=True)
    assert spec.n_predictors == 12
    g, o = _both(engine, oracle, spec, [40, 90, 300], 2500, 62)
    _assert_reduced_equal(g, o)
    assert g.shape.n_posthoc_contrasts == 6 and 0 < g.factor_excluded[:, 0, 1].sum() < 2500


def test_wide_ols_under_a_scenario_preset(engine, oracle):
    from helpers import preset
    spec = S.SimulationSpec(
        n_non_factor=5, n_factor_dummies=3, correlation=np.eye(5).ravel().tolist(),
        var_types=[S.Distribution.Normal] * 5, var_params=[0.0] * 5, factor_n_levels=[4],
        factor_proportions=[0.4, 0.3, 0.2, 0.1], interactions=[[1, 2]],
        effect_sizes=[0.0, 0.3, 0.2, 0.0, 0.1, 0.25, 0.4, -0.3, 0.2, 0.15], target_indices=[1, 2, 5, 6, 9],
        factor_min_level_count=5, scenario=preset("realistic"))
    assert spec.n_predictors == 10
    g = engine.run_batch(spec, [60, 200], 2500, 63, stat_capture=True)
    o = oracle.run_batch(spec, [60, 200], 2500, 63, stat_capture=True, n_threads=8)
    assert np.array_equal(g.factor_excluded, o.factor_excluded)
    for f in ("uncorrected", "corrected", "converged"):
        assert (getattr(g, f) != getattr(o, f)).mean() < 3e-4, f


def test_ols_on_clustered_data_ignoring_the_clustering(engine, oracle):
    """orchestrator tests/clustered_ols_smoke.rs: Continuous + cluster + Ols -- clustered generation, naive OLS fit."""
    from helpers import base_spec
    from test_gpu_parity import _design_mismatch
    spec = base_spec(2)
    spec.effect_sizes = [0.0, 0.5, 0.3]
    spec.cluster = S.ClusterSpec(sizing=("FixedClusters", 20), tau_squared=0.3)
    spec.report_overall = True
    bx, by, worst, total = _design_mismatch(engine, oracle, spec, 200, 16)
    assert bx == 0 and by == 0
    g, o = _both(engine, oracle, spec, [100, 200], 3000, 71)
    _assert_reduced_equal(g, o)
    assert g.uncorrected[:, 1, 0].mean() > 0.2
    spec.cluster = S.ClusterSpec(sizing=("FixedSize", 10), tau_squared=0.5)
    spec.scenario = S.ScenarioPerturbations(name="hsk", heteroskedasticity_ratio=2.0)
    g, o = _both(engine, oracle, spec, [50, 200], 2000, 72)
    for f in ("uncorrected", "corrected", "converged", "overall"):
        assert (getattr(g, f) != getattr(o, f)).mean() < 5e-4, f


def test_lmm_cluster_level_predictors_are_broadcast(engine, oracle):
    """data_gen.rs:606-636: marked columns take the value of the cluster's representative row; a cross-level
    interaction and the outcome use the broadcast values."""
    spec = S.SimulationSpec(n_non_factor=2, n_factor_dummies=1, correlation=[1.0, 0.2, 0.2, 1.0],
                            var_types=[S.Distribution.Normal] * 2, var_params=[0.0, 0.0], factor_n_levels=[2],
                            factor_proportions=[0.5, 0.5], factor_sampled=[True], interactions=[[1, 2]],
                            effect_sizes=[0.0, 0.3, 0.4, 0.5, 0.2], target_indices=[1, 2, 3, 4],
                            between_var_indices=[2, 3], estimator=S.EstimatorSpec.Mle,
                            cluster=S.ClusterSpec(sizing=("FixedClusters", 12), tau_squared=0.3),
                            scenario=S.ScenarioPerturbations(name="optimistic", truth_start=True))
    for sizing in (("FixedClusters", 12), ("FixedSize", 8)):
        spec.cluster = S.ClusterSpec(sizing=sizing, tau_squared=0.3)
        g = engine.run_batch(spec, [96, 240], 1500, 81, stat_capture=True)
        o = oracle.run_batch(spec, [96, 240], 1500, 81, stat_capture=True, n_threads=8)
        m = ~np.isnan(o.stat_capture)
        assert np.array_equal(np.isnan(g.stat_capture), ~m)
        # a replicate whose cluster-level dummy is constant over all clusters (p = 2 * 0.5^12) is a singular design
        # on both sides: compare all but such degenerate replicates
        bad = np.abs(g.stat_capture - o.stat_capture) > 1e-4 * np.abs(o.stat_capture) + 1e-6
        assert (bad & m).any(axis=1).sum() <= 2
        for f in ("uncorrected", "joint_unc", "converged", "boundary_hit"):
            assert (getattr(g, f) != getattr(o, f)).mean() < 5e-3, (sizing, f)
    x, _, ids = oracle.generate_sim_data(spec, 81, 0, 96)
    for c in np.unique(ids):      # the oracle's rows of one cluster share the marked columns
        rows = x[ids == c]
        assert (rows[:, 2] == rows[0, 2]).all() and (rows[:, 3] == rows[0, 3]).all()
        assert np.allclose(rows[:, 4], rows[:, 1] * rows[:, 2])


def test_ols_on_clustered_data_with_cluster_level_predictors(engine, oracle):
    """OLS ignoring the clustering (orchestrator tests/clustered_ols_smoke.rs) with cluster-level (between) predictors -- refused
    until round 3.  The lane-per-replicate kernel broadcasts on the fly (the representative never comes after the row):
    same data as the oracle, so the decisions follow; both sizings, plain and scenario data path, a cross-level interaction."""
    from helpers import preset
    spec = S.SimulationSpec(n_non_factor=2, n_factor_dummies=1, correlation=[1.0, 0.2, 0.2, 1.0],
                            var_types=[S.Distribution.Normal] * 2, var_params=[0.0, 0.0], factor_n_levels=[2],
                            factor_proportions=[0.5, 0.5], factor_sampled=[True], interactions=[[1, 2]],
                            effect_sizes=[0.0, 0.3, 0.4, 0.5, 0.2], target_indices=[1, 2, 3, 4],
                            between_var_indices=[2, 3], estimator=S.EstimatorSpec.Ols,
                            cluster=S.ClusterSpec(sizing=("FixedClusters", 12), tau_squared=0.3))
    for sizing in (("FixedClusters", 12), ("FixedSize", 8)):
        for scen in (None, "realistic"):
            spec.cluster = S.ClusterSpec(sizing=sizing, tau_squared=0.3)
            spec.scenario = preset(scen) if scen else S.ScenarioPerturbations(name="optimistic")
            g = engine.run_batch(spec, [96, 240], 1500, 83, stat_capture=True)
            o = oracle.run_batch(spec, [96, 240], 1500, 83, stat_capture=True, n_threads=8)
            m = ~np.isnan(o.stat_capture)
            assert np.array_equal(np.isnan(g.stat_capture), ~m), (sizing, scen)
            bad = np.abs(g.stat_capture - o.stat_capture) > 1e-6 * np.abs(o.stat_capture) + 1e-9
            assert (bad & m).any(axis=1).sum() <= (4 if scen else 0), (sizing, scen)
            for f in ("uncorrected", "corrected", "converged"):
                assert (getattr(g, f) != getattr(o, f)).sum() <= (3 if scen else 0), (sizing, scen, f)
            c = engine.run_batch(spec, [96, 240], 1500, 83, per_sim=False)
            assert np.array_equal(c.counters, g.counters)


def test_lmm_with_eight_design_columns(engine, oracle):
    spec = S.SimulationSpec(n_non_factor=3, n_factor_dummies=2, correlation=[1.0, 0.3, 0.0, 0.3, 1.0, 0.1, 0.0, 0.1, 1.0],
                            var_types=[S.Distribution.Normal] * 3, var_params=[0.0] * 3, factor_n_levels=[3],
                            factor_proportions=[0.4, 0.3, 0.3], interactions=[[1, 4], [1, 5]],
                            effect_sizes=[0.0, 0.3, 0.2, 0.1, 0.4, -0.3, 0.2, 0.15], target_indices=[1, 2, 3, 4, 5, 6, 7],
                            correction_method=S.CorrectionMethod.Holm, estimator=S.EstimatorSpec.Mle,
                            cluster=S.ClusterSpec(sizing=("FixedClusters", 15), tau_squared=0.25),
                            factor_min_level_count=5,
                            scenario=S.ScenarioPerturbations(name="optimistic", truth_start=True))
    assert spec.n_predictors == 8
    g = engine.run_batch(spec, [150, 450], 1500, 91, stat_capture=True)
    o = oracle.run_batch(spec, [150, 450], 1500, 91, stat_capture=True, n_threads=8)
    m = ~np.isnan(o.stat_capture)
    assert np.array_equal(np.isnan(g.stat_capture), ~m)
    assert np.all(np.abs(g.stat_capture[m] - o.stat_capture[m]) <= 1e-5 * o.stat_capture[m] + 1e-8)
    for f in ("uncorrected", "corrected", "joint_unc", "converged", "boundary_hit"):
        assert (getattr(g, f) != getattr(o, f)).mean() < 2e-3, f
