"""End-to-end pin of the clustered-GLM path against the REFERENCE ENGINE's own numbers.

validation/oracle2_power.R:131-136 froze, for six harsh clustered-logistic cells (validation/oracle_cells.R:24-95),
what the reference's shipped `find_power(wald_se = "hessian", n_sims = 50 000, seed = cell seed)` returned for the
alternative model (`power_hessian_fp`) and, with seed + 2 000 000, for the null model (`typeI_hessian_fp`) --
tests/golden/oracle2.json, written by tests/golden/make_fixtures.py from validation/data/oracle2_<cell>.rds.  The same
call through `mcpower_amd._engine.find_power` sees the SAME data (identical seeds, bit-exact data plane), so a
difference can only come from the fit: the Laplace optimiser (bounded Nelder-Mead here, BOBYQA there), the
finite-difference Hessian, convergence / boundary classification.  Three kernels are exercised: glmm_kernel (random
intercept: few6_int, hiICC_int, smallclust, boundary), glmm_slopes_kernel (few8_slope) and glmm_general_kernel
(few10_crossed).  Gate: 0.3 percentage points on all twelve rates (Monte-Carlo noise is not in play: same draws)."""
import json
import math
import os

import msgpack
import pytest

from mcpower_amd import _engine

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE2 = json.load(open(os.path.join(HERE, "golden", "oracle2.json")))
CELLS = sorted(ORACLE2["cells"])


def cell_contract(cell: dict, beta: float) -> bytes:
    """The contract the ports build for `y ~ x1 + (1|grp)` (+ slopes / crossed grouping) under family logit:
    one synthetic normal predictor, intercept = logit(baseline), latent tau^2 = ICC / (1 - ICC) * pi^2 / 3
    (the logistic family's latent residual variance), FixedClusters sizing, optimistic scenario."""
    base = msgpack.unpackb(open(os.path.join(HERE, "golden", "contracts", "ols_contract.msgpack"), "rb").read(), raw=False)[0]
    base["generation"]["columns"] = [base["generation"]["columns"][0]]
    base["generation"]["correlations"] = "Identity"
    base["design_generation"]["terms"] = ["Const", {"Direct": {"column": 0}}]
    if base.get("design_test"):
        base["design_test"] = None
    icc = cell["icc"]
    extra = [{"relation": ({"Crossed": {"n_clusters": e["n_clusters"]}} if e["kind"] == "crossed"
                           else {"NestedWithin": {"n_per_parent": e["n_clusters"]}}),
              "tau_squared": e["tau_squared"], "slopes": []} for e in cell["extra"]]
    base["generation"]["cluster"] = {"sizing": {"FixedClusters": {"n_clusters": cell["n_clusters"]}},
                                     "tau_squared": icc / (1.0 - icc) * math.pi ** 2 / 3.0,
                                     "slopes": [dict(s) for s in cell["slopes"]], "extra_groupings": extra}
    base["estimator"] = "glm"
    base["wald_se"] = "hessian"
    base["outcome"]["kind"] = "binary"
    p0 = cell["baseline_probability"]
    base["outcome"]["intercept"] = math.log(p0 / (1.0 - p0))
    base["outcome"]["coefficients"] = [0.0, beta]
    base["test"]["targets"] = [{"kind": "marginal", "term": 1}]
    return msgpack.packb([base], use_bin_type=True)


def _power(res):
    sc = res["scenarios"]["optimistic"] if "scenarios" in res else res
    return sc["power_uncorrected"][0][0], sc["convergence_rate"][0]


def test_fixture_holds_the_twelve_frozen_rates():
    assert ORACLE2["k"] == 50000 and len(CELLS) == 6
    for name in CELLS:
        f = ORACLE2["cells"][name]["frozen"]
        assert 0.2 < f["power_hessian_fp"] < 0.9 and 0.03 < f["typeI_hessian_fp"] < 0.06
        assert abs(f["power_hessian_fp"] - f["power_hessian"]) < 0.006  # the reference's own consistency column


def test_random_intercept_cell_on_the_cpu_oracle_is_statistically_consistent(oracle):
    """A small slice of few6_int on the CPU oracle (the full 50 000 run on the device): same seed, so the first 1 500
    replicates are a subsample of the reference's 50 000 -- within 3.5 binomial standard errors of its rate."""
    from test_find_power import _oracle_runner
    c = ORACLE2["cells"]["few6_int"]
    n_sims = 1500
    res = _engine.find_power(cell_contract(c["cell"], c["cell"]["beta_star"]), c["cell"]["n"], n_sims, c["cell"]["seed"], None,
                             _runner=_oracle_runner(oracle))
    pw, conv = _power(res)
    p = c["frozen"]["power_hessian_fp"]
    assert abs(pw - p) < 3.5 * math.sqrt(p * (1 - p) / n_sims) and conv > 0.99


@pytest.mark.gpu
@pytest.mark.parametrize("name", CELLS)
def test_device_find_power_reproduces_the_reference_engine(name, engine):
    c = ORACLE2["cells"][name]
    cell, frozen, k = c["cell"], c["frozen"], ORACLE2["k"]
    alt, conv_alt = _power(_engine.find_power(cell_contract(cell, cell["beta_star"]), cell["n"], k, cell["seed"], None))
    null, conv_null = _power(_engine.find_power(cell_contract(cell, 0.0), cell["n"], k, cell["seed"] + ORACLE2["null_seed_offset"], None))
    print(f"{name}: power {alt:.5f} vs reference {frozen['power_hessian_fp']:.5f} ({100 * (alt - frozen['power_hessian_fp']):+.3f} pp, "
          f"converged {conv_alt:.5f}); type I {null:.5f} vs {frozen['typeI_hessian_fp']:.5f} "
          f"({100 * (null - frozen['typeI_hessian_fp']):+.3f} pp, converged {conv_null:.5f})")
    assert abs(alt - frozen["power_hessian_fp"]) <= 0.003, (alt, frozen["power_hessian_fp"])
    assert abs(null - frozen["typeI_hessian_fp"]) <= 0.003, (null, frozen["typeI_hessian_fp"])
