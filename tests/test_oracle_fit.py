"""Pin the CPU oracle's OLS solver, critical values and corrections.

* OLS beta / se / t vs R `lm` on the same bytes (validation/data/ols_*.golden.rds), gate
  1e-11 relative, crit 1e-9 absolute (validation/tolerances.R:24-30).
* quantiles vs scipy (the reference's own KATs at critvals.rs:1807-1926 are scipy/R values).
* Holm / BH hand cases restated from correction.rs:125-275.
"""
import ctypes as C

import numpy as np
import pytest

from helpers import case_names, load_case, spec_c2b, BASE_SEED
from mcpower_amd import spec as S


@pytest.mark.parametrize("name", case_names("ols"))
def test_ols_fit_matches_r_goldens(oracle, name):
    spec, seed, arr, low = load_case(name)
    f = oracle.ols_fit(arr["design"], arr["outcome"])
    assert f["converged"]
    np.testing.assert_allclose(f["beta"], arr["golden_beta"], rtol=1e-11, atol=0)
    np.testing.assert_allclose(f["se"], arr["golden_se"], rtol=1e-11, atol=0)
    np.testing.assert_allclose(np.sign(f["beta"]) * np.sqrt(f["t_sq"]), arr["golden_stat"], rtol=1e-11, atol=0)
    n, p = arr["design"].shape
    assert abs(oracle.t_ppf(0.975, n - p) - arr["golden_crit"][0]) < 1e-9


@pytest.mark.parametrize("name", case_names("logit") + case_names("lme"))
def test_wald_z_crit_matches_r(oracle, name):
    _, _, arr, _ = load_case(name)
    # GLM/MLE: qnorm agreement ~1.6e-9, gate 1e-8 (tolerances.R crit_abs_iter)
    assert abs(oracle.norm_ppf(0.975) - arr["golden_crit"][0]) < 1e-8


def test_quantiles_vs_scipy(oracle):
    from scipy import stats
    for df in (1, 2, 3, 5, 10, 30, 95, 194, 398, 1000, 50000):
        for p in (0.9, 0.95, 0.975, 0.995, 0.9995):
            assert abs(oracle.t_ppf(p, df) - stats.t.ppf(p, df)) < 1e-9 * max(1.0, stats.t.ppf(p, df))
    for p in (0.001, 0.025, 0.5, 0.9, 0.975, 0.999):
        assert abs(oracle.norm_ppf(p) - stats.norm.ppf(p)) < 2e-9
    for k in (1, 2, 3, 5, 10, 40):
        for p in (0.9, 0.95, 0.99):
            assert abs(oracle.chi2_ppf(p, k) - stats.chi2.ppf(p, k)) < 1e-8 * stats.chi2.ppf(p, k)
    for dfn, dfd in ((1, 98), (2, 50), (5, 194), (15, 184)):
        for p in (0.9, 0.95, 0.99):
            assert abs(oracle.f_ppf(p, dfn, dfd) - stats.f.ppf(p, dfn, dfd)) < 1e-8 * stats.f.ppf(p, dfn, dfd)


def test_crit_table_layout(oracle):
    from scipy import stats
    spec = spec_c2b()
    spec.report_overall = True
    spec.correction_method = S.CorrectionMethod.Holm
    ns = [50, 200]
    tab = oracle.build_crit_table(spec, ns)
    for g, n in enumerate(ns):
        df = n - 6
        assert abs(tab[g, 0] - stats.t.ppf(0.975, df) ** 2) < 1e-9
        assert abs(tab[g, 1] - stats.t.ppf(1 - 0.05 / 4, df) ** 2) < 1e-9   # Holm rank 0: alpha/(2*m)
        assert abs(tab[g, 2] - stats.t.ppf(1 - 0.05 / 2, df) ** 2) < 1e-9   # rank 1: alpha/(2*1)
        assert abs(tab[g, 3] - tab[g, 0]) < 1e-12                          # posthoc alpha defaults to alpha
        assert np.isnan(tab[g, 4])                                         # joint chi2 only for Mle
        assert abs(tab[g, 5] - stats.f.ppf(0.95, 5, df)) < 1e-8


def _apply(oracle, method, t_sq, crit_unc, crit_sq):
    t = np.asarray(t_sq, dtype=np.float64)
    c = np.asarray(crit_sq, dtype=np.float64)
    out = np.zeros(t.size, dtype=np.uint8)
    pd = C.POINTER(C.c_double)
    oracle.lib().mcpo_apply_correction(C.c_int(int(method)), t.ctypes.data_as(pd), C.c_uint32(t.size),
                                       C.c_double(crit_unc), c.ctypes.data_as(pd),
                                       out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out.tolist()


def test_correction_hand_cases(oracle):
    M = S.CorrectionMethod
    assert _apply(oracle, M.None_, [1, 5, 3], 4.0, [10, 10, 10]) == [0, 1, 0]
    assert _apply(oracle, M.Bonferroni, [3, 9, 4.5], 4.0, [4, 4, 4]) == [0, 1, 1]
    assert _apply(oracle, M.Holm, [9, 4, 1], 3.84, [10, 5, 3]) == [0, 0, 0]
    assert _apply(oracle, M.Holm, [9, 4, 1], 3.84, [5, 5, 3]) == [1, 0, 0]
    assert _apply(oracle, M.Holm, [9, 4, 1], 0.5, [0.5, 0.5, 0.5]) == [1, 1, 1]
    assert _apply(oracle, M.BenjaminiHochberg, [9, 1, 4], 3.84, [10, 3, 0.5]) == [1, 1, 1]
    for m in (M.None_, M.Bonferroni, M.Holm, M.BenjaminiHochberg, M.TukeyHsd):
        assert _apply(oracle, m, [100, np.nan, 100], 1.0, [1, 1, 1]) == [1, 0, 1]
    t_sq = [x * x for x in (3.5, 0.7, 1.2, 2.3)]
    assert _apply(oracle, M.Holm, t_sq, 1.0, [x * x for x in (2.5, 2.0, 1.5, 1.0)]) == [1, 0, 0, 1]
    assert _apply(oracle, M.BenjaminiHochberg, t_sq, 1.0, [x * x for x in (3.0, 2.0, 1.5, 1.0)]) == [1, 0, 0, 1]
    assert _apply(oracle, M.BenjaminiHochberg, [4, 4, 0.25], 1.0, [9, 3.24, 0.16]) == [1, 1, 1]


def test_run_batch_validation_errors(oracle):
    spec = spec_c2b()
    with pytest.raises(S.EngineError) as e:
        oracle.run_batch(spec, [200, 100], 4, BASE_SEED)
    assert e.value.status == S.Status.InvalidSpec
    with pytest.raises(S.EngineError):
        oracle.run_batch(spec, [], 4, BASE_SEED)
    with pytest.raises(S.EngineError):
        oracle.run_batch(spec, [0, 10], 4, BASE_SEED)
    with pytest.raises(S.EngineError):  # df = N - P < 1
        oracle.run_batch(spec, [6], 4, BASE_SEED)
    bad = spec_c2b()
    bad.correlation = list(np.eye(5).flatten())
    bad.correlation[1] = bad.correlation[5] = 1.5
    with pytest.raises(S.EngineError) as e:
        oracle.run_batch(bad, [100], 4, BASE_SEED)
    assert e.value.status == S.Status.CorrelationNotPSD


def test_run_batch_shapes_thread_invariance_and_counters(oracle):
    spec = spec_c2b()
    spec.report_overall = True
    r1 = oracle.run_batch(spec, [50, 100, 200], 300, BASE_SEED, stat_capture=True, n_threads=1)
    r4 = oracle.run_batch(spec, [50, 100, 200], 300, BASE_SEED, stat_capture=True, n_threads=4)
    assert r1.uncorrected.shape == (300, 3, 2) and r1.converged.shape == (300, 3)
    for f in ("uncorrected", "corrected", "converged", "overall", "counters"):
        assert np.array_equal(getattr(r1, f), getattr(r4, f)), f      # batch.rs:2716 MT == ST
    assert np.array_equal(r1.stat_capture, r4.stat_capture, equal_nan=True)
    assert np.array_equal(r1.counter("k_unc", 2), r1.uncorrected.sum(axis=0))
    assert np.array_equal(r1.counter("converged")[:, 0], r1.converged.sum(axis=0))
    assert (r1.counter("hist_unc", 3).sum(axis=1) == 300).all()
    # power rises with N for a real effect, and the shard union equals the whole (sim_offset contract)
    assert r1.counter("k_unc", 2)[2, 0] > r1.counter("k_unc", 2)[0, 0]
    a = oracle.run_batch(spec, [50, 100, 200], 100, BASE_SEED, sim_offset=0)
    b = oracle.run_batch(spec, [50, 100, 200], 200, BASE_SEED, sim_offset=100)
    assert np.array_equal(a.counters + b.counters, r1.counters)
    assert np.array_equal(np.concatenate([a.uncorrected, b.uncorrected]), r1.uncorrected)


@pytest.mark.parametrize("name", case_names("logit") + case_names("probit") + case_names("poisson"))
def test_glm_irls_matches_r_goldens(oracle, name):
    """Oracle IRLS vs R `glm` (binomial logit / probit, poisson log) on the reference's own bytes: gate 1e-4
    relative on beta / se / z (validation/tolerances.R:29-31, estimate_rel_iter); observed <= 4e-5."""
    spec, seed, arr, low = load_case(name)
    fam = low["family"]
    f = oracle.glm_fit(arr["design"], arr["outcome"], spec.effect_sizes, family=fam)  # truth start (batch.rs:1945-1951)
    assert f["converged"] and 2 <= f["n_iter"] <= 25
    np.testing.assert_allclose(f["beta"], arr["golden_beta"], rtol=1e-4, atol=0)
    np.testing.assert_allclose(f["se"], arr["golden_se"], rtol=1e-4, atol=0)
    np.testing.assert_allclose(f["beta"] / f["se"], arr["golden_stat"], rtol=1e-4, atol=0)
    cold = oracle.glm_fit(arr["design"], arr["outcome"], None, family=fam)   # start-independent optimum
    np.testing.assert_allclose(cold["beta"], f["beta"], rtol=2e-5, atol=1e-8)  # the deviance rule stops both within ~1e-6


@pytest.mark.parametrize("name", case_names("lme"))
def test_lmm_reml_matches_lme4_goldens(oracle, name):
    """Oracle profiled-REML fit vs lme4 1.1.38 REML on the reference's own bytes: gate 1e-4 relative on
    beta / se / Wald z (tolerances.R:29-31); observed <= 6e-8."""
    spec, seed, arr, low = load_case(name)
    ids = arr["cluster_ids"].astype(np.int64)
    ids = ids - ids.min()
    f = oracle.lmm_fit(arr["design"], arr["outcome"], ids, np.sqrt(low["cluster"]["tau_squared"]))
    assert f["converged"] and f["boundary_hit"] == 0 and f["n_eval"] < 60
    np.testing.assert_allclose(f["beta"], arr["golden_beta"], rtol=1e-4, atol=0)
    np.testing.assert_allclose(f["se"], arr["golden_se"], rtol=1e-4, atol=0)
    np.testing.assert_allclose(f["beta"] / f["se"], arr["golden_stat"], rtol=1e-4, atol=0)
    blind = oracle.lmm_fit(arr["design"], arr["outcome"], ids, 0.0)      # blind start reaches the same optimum
    assert abs(blind["theta"] - f["theta"]) < 1e-6


def test_glm_and_lmm_run_batch_smoke(oracle):
    from helpers import spec_c3, spec_c4
    r = oracle.run_batch(spec_c3(), [500], 300, BASE_SEED, n_threads=4)
    assert r.converged.all() and r.counter("k_unc", 3)[0, 0] > 250
    r = oracle.run_batch(spec_c4(), [300, 600], 200, BASE_SEED, n_threads=4)
    assert r.converged.all() and (r.boundary_hit == 0).all()
    assert r.counter("joint_unc")[1, 0] >= r.counter("joint_unc")[0, 0]
    assert r.shape.n_variance_components == 1
