"""End-to-end pin of the clustered-GLM path against the REFERENCE ENGINE's own numbers.

validation/oracle2_power.R:131-136 froze, for six harsh clustered-logistic cells (validation/oracle_cells.R:24-95),
what the reference's shipped `find_power(wald_se = "hessian", n_sims = 50 000, seed = cell seed)` returned for the
alternative model (`power_hessian_fp`) and, with seed + 2 000 000, for the null model (`typeI_hessian_fp`) --
tests/golden/oracle2.json, written by tests/golden/make_fixtures.py from validation/data/oracle2_<cell>.rds.  The same
call through `mcpower_amd._engine.find_power` sees the SAME data (identical seeds, bit-exact data plane), so a
difference can only come from the fit: the Laplace optimiser (bounded Nelder-Mead here, BOBYQA there), the
finite-difference Hessian, convergence / boundary classification.  Three kernels are exercised: glmm_kernel (random
intercept: few6_int, hiICC_int, smallclust, boundary), glmm_slopes_kernel (few8_slope) and glmm_general_kernel
(few10_crossed).  Monte-Carlo noise is not in play (same draws), so the gate is tight: 0.3 percentage points on all six
type-I rates and on the power of the four cells where the reference's own finite-difference Hessian is clean; see
HESSIAN_STRAINED for the two cells where it is not.

What it took to get here (both found by this test, neither visible device-vs-oracle):
  * the conditional-mode solver needs lme4's step halving -- an undamped Fisher step 2-cycles on clusters far from
    their prior (15 events of 20 at a 10 % baseline), the garbage deviance then yields se ~ 1e-5 and a false rejection:
    type I read 9.8 % (boundary) / 9.2 % (hiICC_int) against the reference's 4.3 % / 4.4 %;
  * the Hessian mode must not fall back to "rx" on singular fits: it differences over the free covariance parameters
    (72 % / 83 % of the fits of few8_slope / few10_crossed are singular; with the fall-back their power read +4.7 / +5.3 pp)."""
import json
import math
import os

import msgpack
import pytest

from mcpower_amd import _engine

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE2 = json.load(open(os.path.join(HERE, "golden", "oracle2.json")))
CELLS = sorted(ORACLE2["cells"])
# Power of these two cells sits +1.0 / +1.3 pp above the reference's with type I inside 0.3 pp: its standard errors are
# ~2 % larger than the Hessian of the Laplace deviance gives.  The reference documents exactly that for these cells -- its
# FD Hessian against glmer's vcov(use.hessian = TRUE) on the same bytes: mean relative gap 2.7 % (hiICC_int) and 1.6 %
# (boundary), 0.28 % on smallclust where the two engines agree to 0.01 pp ("At the harshest cells the FD-Hessian may
# strain", web/documentation/validation/validation_MLE_solving.md section 8.3b).  The Hessian here is step-insensitive
# (delta 1e-4 .. 0.2: se to six digits) and equals a scipy restatement of the Laplace deviance differenced independently;
# the "rx" rates of the same cells, which involve no differencing, agree with the reference's within Monte-Carlo noise
# (test_device_rx_rates_agree_with_the_reference_per_draw_rates).  Reproducing the strain would need the glmm crate's
# stencil, which is not in the tree.
HESSIAN_STRAINED = {"hiICC_int": 0.015, "boundary": 0.015}
# singular-fit-dominated cells: the optimiser's endpoint on a flat boundary decides borderline replicates
POWER_GATE = {"few8_slope": 0.0035, "few10_crossed": 0.0035}


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
    gate = HESSIAN_STRAINED.get(name, POWER_GATE.get(name, 0.003))
    assert abs(alt - frozen["power_hessian_fp"]) <= gate, (alt, frozen["power_hessian_fp"])
    if name in HESSIAN_STRAINED:  # the documented direction: the reference's strained SEs are the larger ones
        assert alt > frozen["power_hessian_fp"]
    assert abs(null - frozen["typeI_hessian_fp"]) <= 0.003, (null, frozen["typeI_hessian_fp"])
    assert conv_alt > 0.9999 and conv_null > 0.9999   # the reference kept 49 996 .. 50 000 of 50 000 draws


@pytest.mark.gpu
@pytest.mark.parametrize("name", CELLS)
def test_device_rx_rates_agree_with_the_reference_per_draw_rates(name, engine):
    """The default "rx" mode involves no finite differences.  The reference froze its per-draw "rx" rejection rates
    (`power_rx`, `typeI_rx`: K = 50 000 single fits on OTHER seeds, oracle2_power.R:44-66), so the comparison carries the
    Monte-Carlo noise of two independent samples: 4 standard errors of the difference."""
    c = ORACLE2["cells"][name]
    cell, frozen, k = c["cell"], c["frozen"], ORACLE2["k"]

    def rx(beta, seed):
        blob = msgpack.unpackb(cell_contract(cell, beta), raw=False)
        blob[0]["wald_se"] = "rx"
        return _power(_engine.find_power(msgpack.packb(blob, use_bin_type=True), cell["n"], k, seed, None))[0]

    for got, ref in ((rx(cell["beta_star"], cell["seed"]), frozen["power_rx"]),
                     (rx(0.0, cell["seed"] + ORACLE2["null_seed_offset"]), frozen["typeI_rx"])):
        band = 4.0 * math.sqrt(2.0 * ref * (1.0 - ref) / k)
        print(f"{name}: rx rate {got:.5f} vs reference per-draw {ref:.5f} ({100 * (got - ref):+.3f} pp, band {100 * band:.3f} pp)")
        assert abs(got - ref) <= band, (name, got, ref)
