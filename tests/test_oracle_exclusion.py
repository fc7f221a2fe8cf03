"""Oracle: sparse-factor exclusion, test_formula reduced fits, post-hoc contrasts and Tukey thresholds.
Cases and expectations follow the reference's own tests (batch.rs:4100-4470 reduced-model tests,
critvals.rs:1017-1110 / :1862-1880 Tukey quantile properties and R goldens, posthoc.rs tests)."""
import math

import numpy as np
import pytest
from scipy import stats

from mcpower_amd import spec as S
from helpers import BASE_SEED


def factor_spec(x1_beta, g_beta, proportions, min_count, **kw):
    """batch.rs:4103-4160 `factor_spec`: y ~ x1 + g (2 levels), targets x1 and g[2], overall F on."""
    s = S.SimulationSpec(n_non_factor=1, n_factor_dummies=1, correlation=[1.0], var_types=[S.Distribution.Normal],
                         var_params=[0.0], factor_n_levels=[2], factor_proportions=list(proportions),
                         effect_sizes=[0.0, x1_beta, g_beta], target_indices=[1, 2], report_overall=True,
                         factor_min_level_count=min_count)
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def three_level_posthoc_spec(proportions=(0.90, 0.05, 0.05), min_count=5, correction=S.CorrectionMethod.None_):
    """batch.rs:4338-4400: x1 + 3-level factor, all pairwise post-hoc contrasts."""
    return S.SimulationSpec(n_non_factor=1, n_factor_dummies=2, correlation=[1.0], var_types=[S.Distribution.Normal],
                            var_params=[0.0], factor_n_levels=[3], factor_proportions=list(proportions),
                            effect_sizes=[0.0, 0.5, 0.3, 0.3], target_indices=[1],
                            posthoc=[S.PosthocSpec(factor_index=0, target_indices=[2, 3])],
                            correction_method=correction, factor_min_level_count=min_count)


def test_sparse_factor_is_excluded_and_targets_report_zero(oracle):
    r = oracle.run_batch(factor_spec(0.5, 0.8, [0.95, 0.05], 5), [40], 200, 42)
    assert (r.factor_excluded == 1).all() and (r.converged == 1).all()
    assert not r.uncorrected[:, 0, 1].any()          # excluded g[2] target never significant
    assert r.uncorrected[:, 0, 0].sum() > 0          # x1 still has power on the reduced model
    assert r.counter("excl_sparse", 1)[0, 0] == 200 and r.counter("excl_sep", 1)[0, 0] == 0


def test_healthy_factor_is_never_excluded_and_results_are_unchanged(oracle):
    a = oracle.run_batch(factor_spec(0.5, 0.8, [0.5, 0.5], 5), [40], 100, 99)
    b = oracle.run_batch(factor_spec(0.5, 0.8, [0.5, 0.5], 0), [40], 100, 99)
    assert not a.factor_excluded.any()
    for name in ("uncorrected", "corrected", "converged", "overall"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name


def test_exclusion_is_per_n_on_a_grid(oracle):
    r = oracle.run_batch(factor_spec(0.3, 0.8, [0.9, 0.1], 5), [20, 100], 200, 42)
    assert (r.factor_excluded[:, 0, 0] == 1).all() and (r.factor_excluded[:, 1, 0] == 0).all()
    assert r.uncorrected[:, 1, 1].sum() > 0 and not r.uncorrected[:, 0, 1].any()


def test_reduced_model_equals_the_model_without_the_factor(oracle):
    """An excluded 2-level factor leaves y ~ x1 fitted on the same rows: t^2(x1) and the overall F of the
    reduced round equal a direct fit of the kept columns, at df = n - p_red."""
    spec = factor_spec(0.5, 0.8, [0.95, 0.05], 5)
    r = oracle.run_batch(spec, [40], 64, 7, stat_capture=True)
    crit_t = stats.t.ppf(0.975, 40 - 2) ** 2
    crit_f = stats.f.ppf(0.95, 1, 38)
    for s in range(64):
        x, y, _ = oracle.generate_sim_data(spec, 7, s, 40)
        fit = oracle.ols_fit(np.ascontiguousarray(x[:, :2]), y)
        t2 = fit["t_sq"][1]
        assert r.stat_capture[s, 0] == pytest.approx(t2, rel=1e-12)
        assert math.isnan(r.stat_capture[s, 1])
        assert r.uncorrected[s, 0, 0] == (t2 > crit_t)
        f = ((fit["sst"] - fit["rss"]) / 1.0) / (fit["rss"] / 38.0)
        assert r.overall[s, 0] == (f > crit_f)


def test_sampled_proportions_exclude_per_replicate(oracle):
    """Under sampled allocation the level counts are stochastic: some replicates exclude, some do not, and the
    flag is exactly `min level count of the first N rows < 5` recomputed from the generated dummies."""
    spec = factor_spec(0.4, 0.6, [0.85, 0.15], 5, factor_sampled=[True])
    r = oracle.run_batch(spec, [30, 60], 300, 11)
    fx = r.factor_excluded[:, :, 0]
    assert 0 < fx[:, 0].sum() < 300 and fx[:, 1].sum() < fx[:, 0].sum()
    for s in range(40):
        x, _, _ = oracle.generate_sim_data(spec, 11, s, 60)
        for g, n in enumerate((30, 60)):
            ones = int((x[:n, 2] == 1.0).sum())
            assert fx[s, g] == (min(ones, n - ones) < 5)
    assert not r.uncorrected[fx == 1][:, 1].any()
    assert r.counter("excl_sparse", 1)[:, 0].tolist() == fx.sum(axis=0).tolist()


def test_interaction_with_an_excluded_dummy_is_dropped_too(oracle):
    spec = S.SimulationSpec(n_non_factor=1, n_factor_dummies=1, correlation=[1.0], var_types=[S.Distribution.Normal],
                            var_params=[0.0], factor_n_levels=[2], factor_proportions=[0.95, 0.05],
                            interactions=[[1, 2]], effect_sizes=[0.0, 0.5, 0.4, 0.3], target_indices=[1, 2, 3],
                            contrast_pairs=[(1, 3)], factor_min_level_count=5, report_overall=True)
    r = oracle.run_batch(spec, [40, 400], 100, 3, stat_capture=True)
    assert (r.factor_excluded[:, 0, 0] == 1).all() and (r.factor_excluded[:, 1, 0] == 0).all()
    assert not r.uncorrected[:, 0, 1:].any()            # g[2], x1:g[2] and the x1 - x1:g[2] contrast all untestable
    assert np.isnan(r.stat_capture[:, 1:]).all() and np.isfinite(r.stat_capture[:, 0]).all()
    assert r.uncorrected[:, 1, :].sum(axis=0).min() > 0  # all four testable at N = 400


def test_fit_columns_reduce_the_fitted_design(oracle):
    """test_formula drops x2 from the fitted model (contract_adapter.rs:157-172): the fit sees [1, x1, x3] only."""
    spec = S.SimulationSpec(n_non_factor=3, correlation=np.eye(3).ravel().tolist(),
                            var_types=[S.Distribution.Normal] * 3, var_params=[0.0] * 3,
                            effect_sizes=[0.0, 0.3, 0.5, 0.2], target_indices=[1, 3], fit_columns=[0, 1, 3],
                            report_overall=True)
    r = oracle.run_batch(spec, [60], 50, 5, stat_capture=True)
    crit = stats.t.ppf(0.975, 60 - 3) ** 2
    for s in range(50):
        x, y, _ = oracle.generate_sim_data(spec, 5, s, 60)
        fit = oracle.ols_fit(np.ascontiguousarray(x[:, [0, 1, 3]]), y)
        assert r.stat_capture[s, 0] == pytest.approx(fit["t_sq"][1], rel=1e-12)
        assert r.stat_capture[s, 1] == pytest.approx(fit["t_sq"][2], rel=1e-12)
        assert r.uncorrected[s, 0].tolist() == [fit["t_sq"][1] > crit, fit["t_sq"][2] > crit]
    full = oracle.run_batch(S.SimulationSpec(**{**spec.__dict__, "fit_columns": []}), [60], 50, 5, stat_capture=True)
    assert not np.allclose(full.stat_capture, r.stat_capture)            # the omitted-variable fit is a different fit


def test_excluded_factor_posthoc_contrasts_report_zero(oracle):
    r = oracle.run_batch(three_level_posthoc_spec(), [40], 150, 42)
    assert (r.converged == 1).all() and (r.factor_excluded == 1).all()
    assert r.posthoc_unc.shape == (150, 1, 3) and not r.posthoc_unc.any() and not r.posthoc_cor.any()
    assert r.uncorrected[:, 0, 0].sum() > 0


@pytest.mark.parametrize("method", [S.CorrectionMethod.None_, S.CorrectionMethod.Bonferroni, S.CorrectionMethod.Holm,
                                    S.CorrectionMethod.BenjaminiHochberg, S.CorrectionMethod.TukeyHsd])
def test_posthoc_pairwise_contrasts_match_direct_computation(oracle, method):
    """posthoc.rs:41-122 + batch.rs:2189-2223: contrasts in order (L1-L0, L2-L0, L2-L1); t^2 from the
    coefficient covariance of a direct f64 fit; the family correction runs over the three contrasts."""
    spec = three_level_posthoc_spec((0.4, 0.3, 0.3), 5, method)
    spec.effect_sizes = [0.0, 0.3, 0.5, -0.2]
    spec.crit_values = S.CritValues(alpha=0.05, posthoc_alpha=0.1)
    n, n_sims = 90, 120
    r = oracle.run_batch(spec, [n], n_sims, 21)
    df = n - 4
    crit_unc = stats.t.ppf(1 - 0.1 / 2, df) ** 2
    if method == S.CorrectionMethod.TukeyHsd:
        thr = [(stats.studentized_range.ppf(0.9, 3, df) / math.sqrt(2)) ** 2] * 3
    elif method == S.CorrectionMethod.Bonferroni:
        thr = [stats.t.ppf(1 - 0.1 / 6, df) ** 2] * 3
    elif method == S.CorrectionMethod.Holm:
        thr = [stats.t.ppf(1 - 0.1 / (2 * (3 - k)), df) ** 2 for k in range(3)]
    elif method == S.CorrectionMethod.BenjaminiHochberg:
        thr = [stats.t.ppf(1 - (k + 1) / 3 * 0.1 / 2, df) ** 2 for k in range(3)]
    else:
        thr = [crit_unc] * 3
    contrasts = np.array([[0, 0, 1, 0], [0, 0, 0, 1], [0, 0, -1, 1]], float)
    agree = 0
    for s in range(n_sims):
        x, y, _ = oracle.generate_sim_data(spec, 21, s, n)
        xd, yd = x.astype(np.float64), y.astype(np.float64)
        xtx_inv = np.linalg.inv(xd.T @ xd)
        beta = xtx_inv @ xd.T @ yd
        sigma2 = ((yd - xd @ beta) ** 2).sum() / df
        t2 = np.array([(c @ beta) ** 2 / (sigma2 * c @ xtx_inv @ c) for c in contrasts])
        want_unc = (t2 > crit_unc).astype(np.uint8)
        order = sorted(range(3), key=lambda i: (-t2[i], i))
        want_cor = np.zeros(3, np.uint8)
        if method in (S.CorrectionMethod.None_, S.CorrectionMethod.Bonferroni, S.CorrectionMethod.TukeyHsd):
            want_cor = (t2 > np.array(thr)).astype(np.uint8)
        elif method == S.CorrectionMethod.Holm:
            for k, i in enumerate(order):
                if t2[i] > thr[k]:
                    want_cor[i] = 1
                else:
                    break
        else:
            last = max([k for k, i in enumerate(order) if t2[i] > thr[k]], default=-1)
            for k in range(last + 1):
                want_cor[order[k]] = 1
        margin = np.abs(t2[:, None] / np.array([crit_unc] + list(thr))[None, :] - 1.0).min()
        if margin < 1e-4:        # knife edge between scipy's and the engine's quantile (Tukey: ~1e-6)
            continue
        assert r.posthoc_unc[s, 0].tolist() == want_unc.tolist(), s
        assert r.posthoc_cor[s, 0].tolist() == want_cor.tolist(), s
        agree += 1
    assert agree > n_sims * 0.9
    hist = r.counter("hist_unc", 1 + 3 + 1)[0]
    assert hist.sum() == n_sims                     # joint histogram spans targets + post-hoc contrasts
    assert (r.counter("posthoc_unc", 3)[0] == r.posthoc_unc[:, 0, :].sum(axis=0)).all()


def test_tukey_quantile_properties_and_r_goldens(oracle):
    q = oracle.q_tukey_ppf
    assert q(0.95, 3, 1e6) == pytest.approx(3.314493, abs=1e-4)      # R qtukey(0.95, 3, Inf)
    assert q(0.95, 4, 30) == pytest.approx(3.845401, abs=1e-4)       # R qtukey(0.95, 4, 30)
    prev = math.inf
    for df in (3.0, 5.0, 10.0, 20.0, 60.0, 1e6):                     # monotone decreasing in df
        v = q(0.95, 3, df)
        assert math.isfinite(v) and 0 < v <= prev + 1e-9
        prev = v
    for df in (1.0, 0.5, 0.0, -1.0):
        assert math.isnan(q(0.95, 3, df))
    assert q(0.95, 4, 30) > q(0.95, 3, 30) and q(0.99, 3, 30) > q(0.95, 3, 30)
    assert math.isnan(q(0.95, 1, 30)) and q(0.0, 3, 30) == 0.0 and math.isinf(q(1.0, 3, 30))
    for p, k, df in [(0.95, 3, 10), (0.99, 5, 20), (0.95, 2, 5), (0.9, 6, 14)]:
        assert q(p, k, df) == pytest.approx(stats.studentized_range.ppf(p, k, df), rel=2e-5)


def test_tukey_main_targets_use_their_factor_levels(oracle):
    """batch.rs:2346-2374: under TukeyHsd every marginal target's threshold is (q_{alpha,k,df}/sqrt2)^2 with k the
    level count of the factor owning the column; a continuous target gets NaN and can never pass."""
    spec = three_level_posthoc_spec((0.4, 0.3, 0.3), 0, S.CorrectionMethod.TukeyHsd)
    spec.posthoc = []
    spec.target_indices = [1, 2, 3]
    spec.contrast_pairs = [(3, 2)]
    crit = oracle.build_crit_table(spec, [50, 200])
    for row, n in zip(crit, (50, 200)):
        df = n - 4
        assert row[0] == pytest.approx(stats.t.ppf(0.975, df) ** 2, rel=1e-9)
        assert math.isnan(row[1])
        want = (stats.studentized_range.ppf(0.95, 3, df) / math.sqrt(2)) ** 2
        assert row[2] == pytest.approx(want, rel=5e-5) and row[3] == row[2] and row[4] == row[2]
    r = oracle.run_batch(spec, [200], 200, 9)
    assert not r.corrected[:, 0, 0].any() and r.corrected[:, 0, 1:].sum() > 0
    assert (r.corrected <= r.uncorrected).all()


def test_glm_exclusion_and_separation_fallback_codes(oracle):
    spec = S.SimulationSpec(n_non_factor=1, n_factor_dummies=1, correlation=[1.0], var_types=[S.Distribution.Normal],
                            var_params=[0.0], factor_n_levels=[2], factor_proportions=[0.9, 0.1],
                            effect_sizes=[math.log(0.3 / 0.7), 0.5, 0.8], target_indices=[1, 2],
                            outcome_kind=S.OutcomeKind.Binary, estimator=S.EstimatorSpec.Glm, report_overall=True,
                            factor_min_level_count=5)
    r = oracle.run_batch(spec, [30, 200], 300, 13)
    assert (r.factor_excluded[:, 0, 0] == 1).all()               # 3 rows < 5 at N = 30
    assert not r.uncorrected[:, 0, 1].any() and r.converged[:, 0].mean() > 0.95
    assert set(np.unique(r.factor_excluded[:, 1, 0])) <= {0, 2}  # included at N = 200 unless separation dropped it
    # small n with strong effects: IRLS overshoots past BETA_CAP in some replicates; the fallback drops the
    # (only) included factor with code 2 and the cold-start refit of y ~ x1 mostly converges
    spec.effect_sizes = [0.0, 2.0, 6.0]
    spec.factor_proportions = [0.6, 0.4]
    r = oracle.run_batch(spec, [20], 500, 13)
    sep = r.factor_excluded[:, 0, 0] == 2
    assert 20 < sep.sum() < 400 and (r.converged[sep, 0] == 1).mean() > 0.9
    assert not r.uncorrected[sep][:, 0, 1].any()
    assert r.counter("excl_sep", 1)[0, 0] == sep.sum() and r.counter("excl_sparse", 1)[0, 0] == 0
    off = oracle.run_batch(S.SimulationSpec(**{**spec.__dict__, "factor_min_level_count": 0}), [20], 500, 13)
    assert not off.factor_excluded.any() and off.converged.sum() < r.converged.sum()
    # replicates whose first attempt converged are untouched by the feature
    keep = ~sep
    assert np.array_equal(off.uncorrected[keep], r.uncorrected[keep])
    assert np.array_equal(off.converged[keep], r.converged[keep])


def test_lmm_exclusion_reduces_the_joint_test(oracle):
    spec = S.SimulationSpec(n_non_factor=1, n_factor_dummies=1, correlation=[1.0], var_types=[S.Distribution.Normal],
                            var_params=[0.0], factor_n_levels=[2], factor_proportions=[0.95, 0.05],
                            effect_sizes=[0.0, 0.4, 0.6], target_indices=[1, 2], estimator=S.EstimatorSpec.Mle,
                            cluster=S.ClusterSpec(sizing=("FixedClusters", 8), tau_squared=0.2),
                            factor_min_level_count=5,
                            scenario=S.ScenarioPerturbations(name="optimistic", truth_start=True))
    r = oracle.run_batch(spec, [40, 240], 120, 17, stat_capture=True)
    assert (r.factor_excluded[:, 0, 0] == 1).all() and (r.factor_excluded[:, 1, 0] == 0).all()
    assert not r.uncorrected[:, 0, 1].any()
    # joint Wald over the single kept target at N = 40: chi2(1) crit equals the marginal z^2 crit
    conv = r.converged[:, 0] == 1
    assert np.array_equal(r.joint_unc[conv, 0], r.uncorrected[conv, 0, 0])
    assert r.joint_unc[:, 1].sum() > 0
