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
