"""Oracle: the general mixed-model path -- random slopes, crossed and nested extra groupings.

Data generation follows data_gen.rs:146-284 (draw order: primary intercepts, primary slopes, all extra
intercepts, all extra slopes) and :812-842 (latent-scale contribution); level ids follow
engine-contract/src/generation.rs:297-427.  Pinned bit-exact to the 11 multi-grouping / slope datasets the
reference captured at sim 0 (validation/data/lmm_*.rds, formulas.R:382-539)."""
import numpy as np
import pytest

from helpers import load_case, mixed_case_names


@pytest.mark.parametrize("name", mixed_case_names())
def test_mixed_model_datasets_are_reproduced_bit_exact(name, oracle):
    spec, base_seed, a, low = load_case(name)
    x, y, ids, extra = oracle.generate_sim_data_ids(spec, base_seed, 0, low["n"])
    assert np.array_equal(ids, a["cluster_ids"])
    assert len(extra) == a["extra_ids"].shape[0]
    for g, e in enumerate(extra):
        assert np.array_equal(e, a["extra_ids"][g]), g
    assert np.array_equal(x, a["design"])
    assert np.array_equal(y, a["outcome"])


def _terms(spec, a):
    ids = [a["cluster_ids"]] + [a["extra_ids"][g] for g in range(a["extra_ids"].shape[0])]
    q = [1 + len(spec.cluster.slopes)] + [1 + len(g.get("slopes", [])) for g in spec.cluster.extra_groupings]
    cols = [spec.cluster_slope_design_cols] + list(spec.extra_slope_cols)
    return ids, q, cols


@pytest.mark.parametrize("name", mixed_case_names())
def test_general_reml_fit_matches_lme4_goldens(name, oracle):
    """validation/tolerances.R:38-49: beta and Wald z at 1e-4 relative, variance components at 1e-3."""
    spec, _, a, low = load_case(name)
    ids, q, cols = _terms(spec, a)
    for start in (None, "truth"):
        th0 = None
        if start == "truth":
            th0 = oracle.truth_theta(spec)
        r = oracle.lmm_general_fit(a["design"], a["outcome"], ids, q, cols, theta_start=th0)
        assert r["converged"], (name, start, r)
        assert np.allclose(r["beta"], a["golden_beta"], rtol=1e-4, atol=1e-6), (name, start, r["beta"], a["golden_beta"])
        z = r["beta"] / r["se"]
        assert np.allclose(z, a["golden_z"], rtol=1e-4, atol=1e-5), (name, start, z, a["golden_z"])
        assert abs(r["sigma_sq"] - a["golden_sigma2"][0]) <= 1e-3 * a["golden_sigma2"][0]
        got_vars = sorted(v for t, qq in enumerate(q) for v in np.diag(r["cov"][t])[:qq])
        assert np.allclose(got_vars, sorted(a["golden_vars"]), rtol=2e-3, atol=2e-5), (name, start, got_vars, a["golden_vars"])


def test_run_batch_general_arm_power_and_flags(oracle):
    """The Mle arm over the general solver: per-replicate Wald decisions equal a refit of the same bytes, the grid
    carries theta-hat forward, pinned bits only appear on converged singular fits, and power sits where the
    standard errors of the golden fit put it."""
    from helpers import BASE_SEED
    spec, _, a, low = load_case("lmm_slope_a")
    spec.target_indices = [1]
    spec.scenario.truth_start = True
    r = oracle.run_batch(spec, [240, 480], 60, BASE_SEED, n_threads=4)
    assert r.converged.mean() > 0.95 and r.boundary_hit.max() <= 2
    crit = oracle.build_crit_table(spec, [240, 480])
    for sim in range(6):
        x, y, ids, _ = oracle.generate_sim_data_ids(spec, BASE_SEED, sim, 480)
        for gi, n in enumerate((240, 480)):
            f = oracle.lmm_general_fit(x[:n], y[:n], [ids[:n]], [2], [[1]], theta_start=oracle.truth_theta(spec))
            if gi == 0 and f["converged"] and r.converged[sim, gi]:
                assert bool(f["z_sq"][1] > crit[gi, 0]) == bool(r.uncorrected[sim, gi, 0])
    assert r.uncorrected[:, 1, 0].mean() > 0.9                 # beta 0.30, se ~ 0.076 at n = 480
    assert (r.pinned_components[r.converged == 0] == 0).all()
    assert np.array_equal(r.joint_unc, r.joint_cor)


def test_reduced_round_strips_slopes_but_keeps_extra_intercepts(oracle):
    """batch.rs:1359-1376: a test_formula / exclusion round fits intercept-only groupings on the kept columns."""
    from helpers import BASE_SEED
    spec, _, a, low = load_case("lmm_slope_crossed")
    spec.n_non_factor = 1
    spec.target_indices = [1]
    full = oracle.run_batch(spec, [480], 20, BASE_SEED, n_threads=4)
    assert full.converged.mean() > 0.9
    two, _, _, _ = load_case("lmm_crossed_a")                   # y ~ x1 + x2 with a crossed extra, no slopes
    two.fit_columns = [0, 1]
    two.target_indices = [1]
    red = oracle.run_batch(two, [480], 20, BASE_SEED, n_threads=4)
    x, y, ids, ex = oracle.generate_sim_data_ids(two, BASE_SEED, 0, 480)
    f = oracle.lmm_general_fit(x[:, :2], y, [ids, ex[0]], [1, 1], [[], []])
    crit = oracle.build_crit_table(two, [480])
    assert bool(f["z_sq"][1] > crit[0, 0]) == bool(red.uncorrected[0, 0, 0])


@pytest.mark.parametrize("name", mixed_case_names())
def test_the_eliminated_grouping_is_a_cost_choice_not_a_model_choice(name, oracle):
    """fit_lmm_general.c eliminates ONE grouping's levels block by block and factors the others densely; which one is
    `elim` (the batch takes mcpo_lmm_elim_term: a nested grouping with the most levels per parent).  The profiled
    criterion is the same function of theta whichever grouping is eliminated: every choice lands on the lme4 golden
    and on the elim = 0 fit."""
    spec, _, a, low = load_case(name)
    ids, q, cols = _terms(spec, a)
    base = oracle.lmm_general_fit(a["design"], a["outcome"], ids, q, cols, theta_start=oracle.truth_theta(spec))
    for e in range(1, len(ids)):
        r = oracle.lmm_general_fit(a["design"], a["outcome"], ids, q, cols, theta_start=oracle.truth_theta(spec), elim=e)
        assert r["converged"] == base["converged"]
        assert np.allclose(r["beta"], base["beta"], rtol=1e-6, atol=1e-8), (name, e)
        assert np.allclose(r["se"], base["se"], rtol=1e-5), (name, e)
        assert np.allclose(r["theta"], base["theta"], atol=2e-4), (name, e, r["theta"], base["theta"])
        assert np.allclose(r["beta"], a["golden_beta"], rtol=1e-4, atol=1e-6)
        assert np.allclose(r["beta"] / r["se"], a["golden_z"], rtol=1e-4, atol=1e-5)
