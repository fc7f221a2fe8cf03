"""Oracle: clustered GLM = GLMM with a random intercept (Laplace), the clustered branch of the Glm arm
(batch.rs:1575-1830).  Pinned to the reference's captured dataset `glmm_intercept.rds` (bit-exact data, with the
family-aware latent variance tau^2 = ICC / (1 - ICC) * pi^2 / 3 of the ports) and to lme4's glmer on those bytes
(`glmm_intercept.golden.rds`) inside the reference's own GLMM_TOL (tolerances.R:186-200)."""
import numpy as np
import pytest

from helpers import BASE_SEED, load_case
from mcpower_amd import spec as S


def test_glmm_dataset_is_reproduced_bit_exact(oracle):
    spec, base_seed, a, low = load_case("glmm_intercept")
    x, y, ids, _ = oracle.generate_sim_data_ids(spec, base_seed, 0, low["n"])
    assert np.array_equal(ids, a["cluster_ids"]) and np.array_equal(x, a["design"]) and np.array_equal(y, a["outcome"])


@pytest.mark.parametrize("start", ["blind", "truth"])
def test_laplace_fit_matches_glmer_within_the_reference_gate(start, oracle):
    spec, _, a, low = load_case("glmm_intercept")
    kw = {} if start == "blind" else dict(theta_start=spec.cluster.tau_squared ** 0.5, beta_start=spec.effect_sizes)
    r = oracle.glmm_fit(a["design"], a["outcome"], a["cluster_ids"], family="logit", **kw)
    assert r["converged"] and r["boundary_hit"] == 0
    # GLMM_TOL: beta_rel 1e-4 with an absolute floor of 1e-3 (glmer's own convergence tolerance)
    assert np.all(np.abs(r["beta"] - a["golden_beta"]) <= np.maximum(1e-4 * np.abs(a["golden_beta"]), 1e-3))
    assert abs(r["tau_sq"] - a["golden_vars"][0]) <= 2e-3 * a["golden_vars"][0]
    # glmer reports Hessian-based z, the engine's default SE convention is "rx" (conditional on theta-hat): few percent apart
    z = r["beta"] / r["se"]
    assert np.all(np.abs(z / a["golden_z"] - 1.0) < 0.03)


def test_this_optimum_is_below_glmers_on_an_independent_objective(oracle):
    """The Laplace deviance restated with scipy (bounded scalar search per cluster) is lower at the oracle's
    optimum than at glmer's reported one: the 4e-4 gap in beta is glmer stopping early, not a different objective."""
    from scipy.optimize import minimize_scalar
    from scipy.special import expit
    spec, _, a, low = load_case("glmm_intercept")
    x, y, ids = a["design"].astype(float), a["outcome"].astype(float), a["cluster_ids"]

    def laplace(theta, beta):
        eta0, tot = x @ beta, 0.0
        for c in range(ids.max() + 1):
            m = ids == c
            e0, yc = eta0[m], y[m]
            f = lambda u: -2.0 * np.sum(yc * np.log(expit(e0 + theta * u)) + (1 - yc) * np.log(1 - expit(e0 + theta * u))) + u * u
            r = minimize_scalar(f, bounds=(-10, 10), method="bounded", options={"xatol": 1e-12})
            mu = expit(e0 + theta * r.x)
            tot += r.fun + np.log(theta * theta * (mu * (1 - mu)).sum() + 1.0)
        return tot
    r = oracle.glmm_fit(a["design"], a["outcome"], a["cluster_ids"], family="logit")
    assert laplace(r["theta"], r["beta"]) < laplace(a["golden_vars"][0] ** 0.5, a["golden_beta"])


def test_glmm_arm_run_batch(oracle):
    """Decisions of run_batch equal refits of the same bytes; tau-hat^2 is reported on converged fits only; the grid
    carries theta-hat; a test_formula subset fits the kept columns."""
    spec, _, a, low = load_case("glmm_intercept")
    spec.target_indices = [1]
    spec.scenario.truth_start = True
    r = oracle.run_batch(spec, [300, 600], 24, BASE_SEED, n_threads=8)
    assert r.converged.mean() > 0.9 and np.array_equal(r.joint_unc, r.joint_cor)
    assert np.isnan(r.tau_squared_hat[r.converged == 0]).all() and np.isfinite(r.tau_squared_hat[r.converged == 1]).all()
    crit = oracle.build_crit_table(spec, [300, 600])
    for sim in range(4):
        x, y, ids, _ = oracle.generate_sim_data_ids(spec, BASE_SEED, sim, 600)
        f = oracle.glmm_fit(x[:300], y[:300], ids[:300], family="logit", theta_start=spec.cluster.tau_squared ** 0.5,
                            beta_start=spec.effect_sizes)
        if f["converged"] and r.converged[sim, 0]:
            assert bool(f["z_sq"][1] > crit[0, 0]) == bool(r.uncorrected[sim, 0, 0])
            assert abs(f["tau_sq"] - r.tau_squared_hat[sim, 0]) < 1e-9
    assert r.uncorrected[:, 1, 0].mean() > 0.9          # beta 0.5 at n = 600: se ~ 0.11


# ---- random slopes on the primary grouping (fit_glmm_slopes.c) ----

SLOPE_CASES = {"glmm_slope": [1], "glmm_multislope": [1, 2]}


@pytest.mark.parametrize("name", sorted(SLOPE_CASES))
def test_glmm_slope_datasets_are_reproduced_bit_exact(name, oracle):
    spec, base_seed, a, low = load_case(name)
    x, y, ids, _ = oracle.generate_sim_data_ids(spec, base_seed, 0, low["n"])
    assert np.array_equal(ids, a["cluster_ids"]) and np.array_equal(x, a["design"]) and np.array_equal(y, a["outcome"])


@pytest.mark.parametrize("start", ["blind", "truth"])
@pytest.mark.parametrize("name", sorted(SLOPE_CASES))
def test_laplace_slopes_fit_matches_glmer(name, start, oracle):
    """beta-hat inside GLMM_TOL (tolerances.R:186-200); variance components at the band glmer's own convergence allows
    on these (near-)singular fits: glmm_slope sits on the corr = -1 boundary, where one component is pinned."""
    spec, _, a, low = load_case(name)
    kw = {} if start == "blind" else dict(theta_start=oracle.truth_theta(spec), beta_start=spec.effect_sizes)
    r = oracle.glmm_slopes_fit(a["design"], a["outcome"], a["cluster_ids"], SLOPE_CASES[name], family="logit", **kw)
    assert r["converged"]
    assert np.all(np.abs(r["beta"] - a["golden_beta"]) <= np.maximum(1e-4 * np.abs(a["golden_beta"]), 1e-3))
    assert np.allclose(np.diag(r["cov"]), a["golden_vars"], rtol=1e-2)
    sd = np.sqrt(np.diag(r["cov"]))
    corr = (r["cov"] / np.outer(sd, sd))[np.tril_indices(len(sd), -1)]
    assert np.allclose(corr, a["golden_corr"], atol=5e-3)
    if name == "glmm_slope":
        assert r["boundary_hit"] == 1 and r["pinned"] == 2 and abs(corr[0] + 1.0) < 1e-6


def test_slopes_fit_with_q1_equals_the_scalar_fit(oracle):
    spec, _, a, low = load_case("glmm_intercept")
    s = oracle.glmm_fit(a["design"], a["outcome"], a["cluster_ids"], family="logit")
    v = oracle.glmm_slopes_fit(a["design"], a["outcome"], a["cluster_ids"], [], family="logit")
    assert np.allclose(v["beta"], s["beta"], atol=1e-6) and abs(v["theta"][0] - s["theta"]) < 1e-6
    assert np.allclose(v["se"], s["se"], rtol=1e-6)


def test_glmm_arm_with_slopes_run_batch(oracle):
    """run_batch routes a clustered GLM with primary slopes through the vector-mode fit: decisions equal refits of the
    same bytes from the same start, tau-hat^2 = theta[0]^2, the grid carries the whole theta vector."""
    spec, _, a, low = load_case("glmm_slope")
    spec.scenario.truth_start = True
    r = oracle.run_batch(spec, [400, 600], 12, BASE_SEED, n_threads=8)
    assert r.converged.mean() > 0.8
    crit = oracle.build_crit_table(spec, [400, 600])
    th0 = oracle.truth_theta(spec)
    for sim in range(3):
        x, y, ids, _ = oracle.generate_sim_data_ids(spec, BASE_SEED, sim, 600)
        f = oracle.glmm_slopes_fit(x[:400], y[:400], ids[:400], [1], family="logit", theta_start=th0, beta_start=spec.effect_sizes)
        assert f["converged"] == bool(r.converged[sim, 0])
        if f["converged"]:
            z_sq = (f["beta"][1] / f["se"][1]) ** 2
            assert bool(z_sq > crit[0, 0]) == bool(r.uncorrected[sim, 0, 0])
            assert abs(f["theta"][0] ** 2 - r.tau_squared_hat[sim, 0]) < 1e-9
            assert f["pinned"] == r.pinned_components[sim, 0]


# ---- Wald SE mode "hessian" (estimator.rs:39-56; lme4 use.hessian = TRUE) ----

@pytest.mark.parametrize("name", ["glmm_intercept", "glmm_multislope"])
def test_hessian_wald_se_tracks_glmers_reported_z(name, oracle):
    """glmer reports Hessian-based z: the "hessian" mode lands within 1 % of it on the interior goldens (the "rx" mode is
    1.4-4.7 % away); the rest is the noise of glmer's own finite differences (its deviance carries a 1e-7 PIRLS
    tolerance, this one 1e-13 -- the oracle's z does not move in the sixth digit between delta = 1e-4 and 1e-2)."""
    spec, _, a, low = load_case(name)
    fit = (lambda **kw: oracle.glmm_fit(a["design"], a["outcome"], a["cluster_ids"], family="logit", **kw)) \
        if name == "glmm_intercept" else \
        (lambda **kw: oracle.glmm_slopes_fit(a["design"], a["outcome"], a["cluster_ids"], SLOPE_CASES[name], family="logit", **kw))
    rx, he = fit(wald_se="rx"), fit(wald_se="hessian")
    assert np.array_equal(rx["beta"], he["beta"])
    z_rx, z_he = rx["beta"] / rx["se"], he["beta"] / he["se"]
    assert np.all(np.abs(z_he / a["golden_z"] - 1.0) < 0.01)
    assert np.all(np.abs(z_he / a["golden_z"] - 1.0) < np.abs(z_rx / a["golden_z"] - 1.0))
    assert np.all(he["se"] > rx["se"])   # "rx" ignores the beta-theta coupling: anticonservative (estimator.rs:51-54)


def test_hessian_mode_on_a_singular_fit_runs_over_the_free_parameters(oracle):
    """glmm_slope is singular (rho-hat = -1, one component pinned).  glmer's reported z there is the "rx" z to 1e-3 (its
    Hessian was not usable and vcov.merMod fell back) -- the "rx" mode reproduces it.  The "hessian" mode does NOT fall
    back on a pinned component: it differences over the free covariance parameters and beta (the pinned diagonal is an
    even point of the deviance and drops out), which keeps the beta-theta coupling "rx" ignores -- the reference ENGINE's
    own Hessian-mode power on cells where most fits are singular (tests/test_oracle2_power.py: few8_slope 72 %,
    few10_crossed 83 % singular) sits 5-7 pp below its "rx" power, and only this form reproduces it."""
    spec, _, a, low = load_case("glmm_slope")
    rx = oracle.glmm_slopes_fit(a["design"], a["outcome"], a["cluster_ids"], [1], family="logit", wald_se="rx")
    he = oracle.glmm_slopes_fit(a["design"], a["outcome"], a["cluster_ids"], [1], family="logit", wald_se="hessian")
    assert rx["pinned"] == 2 and he["pinned"] == 2 and np.array_equal(rx["beta"], he["beta"])
    assert np.all(np.abs(rx["beta"] / rx["se"] / a["golden_z"] - 1.0) < 1e-3)
    assert np.all(he["se"] > rx["se"]) and np.all(he["se"] < 1.15 * rx["se"])


def test_hessian_mode_through_run_batch(oracle):
    spec, _, a, low = load_case("glmm_intercept")
    spec.scenario.truth_start = True
    rx = oracle.run_batch(spec, [600], 16, BASE_SEED, n_threads=8, stat_capture=True)
    spec.wald_se = S.WaldSe.Hessian
    he = oracle.run_batch(spec, [600], 16, BASE_SEED, n_threads=8, stat_capture=True)
    assert np.array_equal(rx.converged, he.converged) and np.allclose(rx.tau_squared_hat, he.tau_squared_hat, equal_nan=True)
    ok = rx.converged[:, 0] == 1
    ratio = he.stat_capture[ok, 0] / rx.stat_capture[ok, 0]
    assert np.all(ratio < 1.0) and np.all(ratio > 0.9)   # larger SE, by a few percent
    x, y, ids, _ = oracle.generate_sim_data_ids(spec, BASE_SEED, 0, 600)
    f = oracle.glmm_fit(x, y, ids, family="logit", theta_start=spec.cluster.tau_squared ** 0.5, beta_start=spec.effect_sizes,
                        wald_se="hessian")
    assert np.isclose(f["z_sq"][1], he.stat_capture[0, 0], rtol=1e-12)


# ---- adaptive Gauss-Hermite quadrature (spec.nagq; validate.rs:918-945) ----

def test_gauss_hermite_rule_matches_numpy(oracle):
    for k in (3, 5, 7, 25):
        z, w = oracle.gh_rule(k)
        zz, ww = np.polynomial.hermite_e.hermegauss(k)
        o = np.argsort(z)
        assert np.allclose(z[o], zz, atol=2e-15) and np.allclose(w[o], ww / np.sqrt(2 * np.pi), atol=1e-15)
        assert z[k // 2] == 0.0


def _marginal_deviance(par, x, y, ids, nodes=200):
    """-2 log marginal likelihood of the random-intercept logistic model by a NON-adaptive 200-point rule (numpy's)."""
    from scipy.special import expit
    z, w = np.polynomial.hermite_e.hermegauss(nodes)
    w = w / np.sqrt(2 * np.pi)
    theta, beta = par[0], par[1:]
    eta = (x @ beta)[:, None] + theta * z[None, :]
    ll = np.where(y[:, None] > 0.5, np.log(expit(eta)), np.log(expit(-eta)))
    tot = 0.0
    for c in range(ids.max() + 1):
        tot += -2.0 * np.log(np.sum(w * np.exp(ll[ids == c].sum(axis=0))))
    return tot


def test_agq_lands_on_the_marginal_likelihood_optimum(oracle):
    """The reference freezes no AGQ estimates (glmm_agq.golden.rds holds a gap metric only): the mode is pinned by an
    independent restatement -- the exact marginal likelihood maximised with scipy.  nagq = 25 sits on that optimum, the
    Laplace fit does not; nagq = 1 through the quadrature switch is the Laplace fit itself."""
    from scipy.optimize import minimize
    spec, _, a, low = load_case("glmm_intercept")
    x, y, ids = a["design"].astype(float), a["outcome"].astype(float), a["cluster_ids"]
    lap = oracle.glmm_fit(a["design"], a["outcome"], ids, family="logit")
    one = oracle.glmm_fit(a["design"], a["outcome"], ids, family="logit", nagq=1)
    agq = oracle.glmm_fit(a["design"], a["outcome"], ids, family="logit", nagq=25)
    assert np.array_equal(lap["beta"], one["beta"]) and lap["theta"] == one["theta"]
    p_agq = np.concatenate([[agq["theta"]], agq["beta"]])
    p_lap = np.concatenate([[lap["theta"]], lap["beta"]])
    ref = minimize(_marginal_deviance, p_agq, args=(x, y, ids), method="Nelder-Mead",
                   options=dict(xatol=1e-9, fatol=1e-12, maxiter=4000))
    assert np.all(np.abs(ref.x - p_agq) < 5e-5)
    assert _marginal_deviance(p_agq, x, y, ids) <= ref.fun + 1e-7
    assert _marginal_deviance(p_lap, x, y, ids) > ref.fun + 1e-4      # the Laplace optimum is measurably elsewhere
    k7 = oracle.glmm_fit(a["design"], a["outcome"], ids, family="logit", nagq=7)
    assert np.all(np.abs(k7["beta"] - agq["beta"]) < 1e-5)


def test_agq_on_the_laplace_bias_shape_and_through_run_batch(oracle):
    """formulas.R:1141-1154 `glmm_laplace_bias`: ICC 0.5, 150 clusters of 4 -- the cell where Laplace is visibly biased.
    The reference gates Laplace vs nAGQ = 7 at 0.05 on beta (tolerances.R:193); the two fits differ, inside that gate."""
    spec, _, a, low = load_case("glmm_intercept")
    spec.cluster = S.ClusterSpec(sizing=("FixedClusters", 150), tau_squared=1.0 * np.pi ** 2 / 3.0)
    spec.scenario.truth_start = True
    x, y, ids, _ = oracle.generate_sim_data_ids(spec, 2404, 0, 600)
    kw = dict(family="logit", theta_start=spec.cluster.tau_squared ** 0.5, beta_start=spec.effect_sizes)
    lap, agq = oracle.glmm_fit(x, y, ids, **kw), oracle.glmm_fit(x, y, ids, nagq=7, **kw)
    gap = np.abs(lap["beta"] - agq["beta"]).max()
    assert 1e-3 < gap < 0.05 and agq["theta"] > lap["theta"]          # Laplace shrinks tau-hat on tiny clusters
    spec.nagq = 7
    r = oracle.run_batch(spec, [600], 8, 2404, n_threads=8, stat_capture=True)
    assert r.converged[0, 0] == 1 and np.isclose(r.tau_squared_hat[0, 0], agq["theta"] ** 2, rtol=1e-12)
    assert np.isclose(r.stat_capture[0, 0], agq["z_sq"][1], rtol=1e-12)


def _marginal_deviance_q2(par, x, y, ids, zcol, nodes=48):
    """-2 log marginal likelihood of the logistic model with a random intercept and one random slope on column `zcol`
    (theta = column-major vech of the lower template: t11, t21, t22) by a NON-adaptive nodes x nodes tensor rule."""
    from scipy.special import expit
    z, w = np.polynomial.hermite_e.hermegauss(nodes)
    w = w / np.sqrt(2 * np.pi)
    t11, t21, t22 = par[:3]
    beta = par[3:]
    u1, u2 = np.meshgrid(z, z, indexing="ij")
    ww = (w[:, None] * w[None, :]).ravel()
    b0 = (t11 * u1).ravel()                       # b = Lambda u
    b1 = (t21 * u1 + t22 * u2).ravel()
    eta0 = x @ beta
    tot = 0.0
    for c in range(ids.max() + 1):
        m = ids == c
        eta = eta0[m][:, None] + b0[None, :] + x[m, zcol][:, None] * b1[None, :]
        ll = np.where(y[m][:, None] > 0.5, np.log(expit(eta)), np.log(expit(-eta))).sum(axis=0)
        tot += -2.0 * np.log(np.sum(ww * np.exp(ll)))
    return tot


def test_agq_with_a_random_slope_lands_on_the_marginal_likelihood_optimum(oracle):
    """validate.rs:918-941 admits nagq > 1 for ONE grouping with <= 3 random effects; lme4 itself integrates a scalar only,
    and the reference freezes no estimates for the vector case, so the k^q tensor rule of fit_glmm_slopes.c is pinned by
    an independent restatement: the exact marginal likelihood (non-adaptive 48 x 48 rule) maximised with scipy.  The 9 x 9
    adaptive rule sits on that optimum, the Laplace fit does not; q = 1 through the vector code is the scalar AGQ fit."""
    from scipy.optimize import minimize
    spec, _, a, low = load_case("glmm_intercept")
    spec.cluster = S.ClusterSpec(sizing=("FixedClusters", 24), tau_squared=1.5,
                                 slopes=[{"column": 0, "variance": 0.6, "corr_with_intercept": 0.3, "corr_with": []}])
    spec.cluster_slope_design_cols = [1]
    spec.scenario.truth_start = True
    x, y, ids, _ = oracle.generate_sim_data_ids(spec, 777, 0, 288)   # 24 clusters of 12
    th0, b0 = oracle.truth_theta(spec, True), spec.effect_sizes
    lap = oracle.glmm_slopes_fit(x, y, ids, [1], theta_start=th0, beta_start=b0)
    agq = oracle.glmm_slopes_fit(x, y, ids, [1], theta_start=th0, beta_start=b0, nagq=9)
    assert lap["converged"] and agq["converged"] and agq["pinned"] == 0
    xf, yf = x.astype(float), y.astype(float)
    p_agq = np.concatenate([agq["theta"], agq["beta"]])
    p_lap = np.concatenate([lap["theta"], lap["beta"]])
    ref = minimize(_marginal_deviance_q2, p_agq, args=(xf, yf, ids, 1), method="Nelder-Mead",
                   options=dict(xatol=1e-8, fatol=1e-11, maxiter=6000))
    assert np.all(np.abs(ref.x - p_agq) < 2e-4), (ref.x, p_agq)
    assert _marginal_deviance_q2(p_agq, xf, yf, ids, 1) <= ref.fun + 1e-6
    assert _marginal_deviance_q2(p_lap, xf, yf, ids, 1) > ref.fun + 1e-3    # the Laplace optimum is measurably elsewhere
    k5 = oracle.glmm_slopes_fit(x, y, ids, [1], theta_start=th0, beta_start=b0, nagq=5)
    assert np.all(np.abs(k5["beta"] - agq["beta"]) < 5e-3)   # the 5 x 5 rule is already close (clusters of 12, tau^2 = 1.5)
    # q = 1 through the vector code == the scalar quadrature of fit_glmm.c
    s1 = oracle.glmm_slopes_fit(x, y, ids, [], theta_start=[1.0], beta_start=b0, nagq=7)
    s0 = oracle.glmm_fit(x, y, ids, theta_start=1.0, beta_start=b0, nagq=7)
    assert np.allclose(s1["beta"], s0["beta"], atol=2e-6) and abs(s1["theta"][0] - s0["theta"]) < 2e-6
    # and through run_batch (replicate 0 of the same seed is this dataset)
    spec.nagq = 9
    r = oracle.run_batch(spec, [288], 4, 777, n_threads=4, stat_capture=True)
    assert r.converged[0, 0] == 1 and np.isclose(r.tau_squared_hat[0, 0], agq["theta"][0] ** 2, rtol=1e-10)


# ---- extra groupings under a GLM family (fit_glmm_general.c; ORACLE ONLY -- the device path refuses them) ----

GENERAL_CASES = ["glmm_slope_crossed", "glmm_slope_crossed_extra_slopes", "glmm_slope_nested_extra_slopes"]


def _glmm_terms(spec, a):
    ids = [a["cluster_ids"]] + [a["extra_ids"][g] for g in range(a["extra_ids"].shape[0])]
    q = [1 + len(spec.cluster.slopes)] + [1 + len(g.get("slopes", [])) for g in spec.cluster.extra_groupings]
    cols = [spec.cluster_slope_design_cols] + list(spec.extra_slope_cols)
    return ids, q, cols


@pytest.mark.parametrize("name", GENERAL_CASES)
def test_general_glmm_datasets_are_reproduced_bit_exact(name, oracle):
    spec, base_seed, a, low = load_case(name)
    x, y, ids, extra = oracle.generate_sim_data_ids(spec, base_seed, 0, low["n"])
    assert np.array_equal(ids, a["cluster_ids"]) and np.array_equal(x, a["design"]) and np.array_equal(y, a["outcome"])
    assert all(np.array_equal(e, a["extra_ids"][g]) for g, e in enumerate(extra))


@pytest.mark.parametrize("start", ["blind", "truth"])
@pytest.mark.parametrize("name", GENERAL_CASES)
def test_general_laplace_fit_matches_glmer(name, start, oracle):
    """The remaining three of the reference's M4 GLMM cases (formulas.R:1130-1181): crossed / nested extra groupings,
    slopes on both.  beta-hat inside GLMM_TOL (1e-4 rel with a 1e-3 abs floor), variance components within 1 %; all
    three are singular fits with the same pinned structure as glmer's, whose reported z is then the "rx" z."""
    spec, _, a, low = load_case(name)
    ids, q, cols = _glmm_terms(spec, a)
    kw = {} if start == "blind" else dict(theta_start=oracle.truth_theta(spec), beta_start=spec.effect_sizes)
    r = oracle.glmm_general_fit(a["design"], a["outcome"], ids, q, cols, family="logit", **kw)
    assert r["converged"] and r["boundary_hit"] == 1
    assert np.all(np.abs(r["beta"] - a["golden_beta"]) <= np.maximum(1e-4 * np.abs(a["golden_beta"]), 1e-3))
    got = np.concatenate([np.diag(r["cov"][t])[:qq] for t, qq in enumerate(q)])
    assert np.allclose(got, a["golden_vars"], rtol=1e-2, atol=1e-6)
    assert np.all(np.abs(r["beta"] / r["se"] / a["golden_z"] - 1.0) < 2e-3)


@pytest.mark.parametrize("name", GENERAL_CASES)
def test_the_leading_grouping_of_the_joint_mode_system_is_a_cost_choice(name, oracle):
    """fit_glmm_general.c lays the random effects out with ONE grouping's levels first (block diagonal, so the device
    factors it one block per lane) and the others behind it; the batch takes mcpo_elim_term (most columns).  The Laplace
    deviance is the same function of (theta, beta) whichever grouping leads: every choice gives the glmer-pinned fit."""
    spec, _, a, low = load_case(name)
    ids, q, cols = _glmm_terms(spec, a)
    kw = dict(theta_start=oracle.truth_theta(spec), beta_start=spec.effect_sizes)
    base = oracle.glmm_general_fit(a["design"], a["outcome"], ids, q, cols, family="logit", **kw)
    for e in range(1, len(ids)):
        r = oracle.glmm_general_fit(a["design"], a["outcome"], ids, q, cols, family="logit", elim=e, **kw)
        assert r["converged"] == base["converged"] and r["boundary_hit"] == base["boundary_hit"] and r["pinned"] == base["pinned"]
        assert np.allclose(r["beta"], base["beta"], atol=5e-6), (name, e, r["beta"], base["beta"])
        assert np.allclose(r["se"], base["se"], rtol=1e-4), (name, e)
        assert np.allclose(r["theta"], base["theta"], atol=1e-3), (name, e, r["theta"], base["theta"])


def test_general_fit_on_a_single_grouping_equals_the_per_cluster_fit(oracle):
    spec, _, a, low = load_case("glmm_multislope")
    v = oracle.glmm_slopes_fit(a["design"], a["outcome"], a["cluster_ids"], [1, 2], family="logit")
    g = oracle.glmm_general_fit(a["design"], a["outcome"], [a["cluster_ids"]], [3], [[1, 2]], family="logit")
    assert np.allclose(g["beta"], v["beta"], atol=2e-6) and np.allclose(g["theta"], v["theta"], atol=2e-5)
    assert np.allclose(g["se"], v["se"], rtol=1e-5)


def test_glmm_arm_with_extra_groupings_run_batch(oracle):
    spec, _, a, low = load_case("glmm_slope_crossed")
    spec.scenario.truth_start = True
    r = oracle.run_batch(spec, [480], 6, BASE_SEED, n_threads=6, stat_capture=True)
    assert r.converged.mean() > 0.6
    ids, q, cols = _glmm_terms(spec, a)
    x, y, cid, extra = oracle.generate_sim_data_ids(spec, BASE_SEED, 0, 480)
    f = oracle.glmm_general_fit(x, y, [cid] + list(extra), q, cols, family="logit", theta_start=oracle.truth_theta(spec),
                                beta_start=spec.effect_sizes)
    assert f["converged"] == bool(r.converged[0, 0])
    if f["converged"]:
        assert np.isclose((f["beta"][1] / f["se"][1]) ** 2, r.stat_capture[0, 0], rtol=1e-10)
        assert f["pinned"] == r.pinned_components[0, 0] and np.isclose(f["theta"][0] ** 2, r.tau_squared_hat[0, 0])


def test_general_fit_hessian_mode_interior_and_singular(oracle):
    """Hessian SE through the joint-mode oracle: on a single grouping it equals the per-cluster fit's Hessian SE; on the
    singular reference fits it runs over the free parameters (larger than "rx" by the beta-theta coupling, as in
    test_hessian_mode_on_a_singular_fit_runs_over_the_free_parameters)."""
    spec, _, a, low = load_case("glmm_multislope")
    v = oracle.glmm_slopes_fit(a["design"], a["outcome"], a["cluster_ids"], [1, 2], family="logit", wald_se="hessian")
    g = oracle.glmm_general_fit(a["design"], a["outcome"], [a["cluster_ids"]], [3], [[1, 2]], family="logit", wald_se="hessian")
    assert np.allclose(g["se"], v["se"], rtol=2e-4) and np.all(np.abs(g["beta"] / g["se"] / a["golden_z"] - 1.0) < 0.01)
    spec, _, a, low = load_case("glmm_slope_crossed")
    ids, q, cols = _glmm_terms(spec, a)
    rx = oracle.glmm_general_fit(a["design"], a["outcome"], ids, q, cols, family="logit")
    he = oracle.glmm_general_fit(a["design"], a["outcome"], ids, q, cols, family="logit", wald_se="hessian")
    assert rx["pinned"] != 0 and np.array_equal(rx["beta"], he["beta"])
    assert np.all(he["se"] > rx["se"]) and np.all(he["se"] < 1.15 * rx["se"])
