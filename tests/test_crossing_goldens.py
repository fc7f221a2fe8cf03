"""End-to-end power parity against outputs the reference itself froze.

`validation/data/cross_*.golden.rds` record, for four models, the required sample size per effect that the
reference's dense `find_sample_size` run produced (validation_crossing.rmd:118-122; models in
validation/formulas.R:661-677).  tests/golden/crossing.json carries those numbers.  Re-running the same call
(from 20 to 300 by 2, 100 000 replicates, seed 2137) through `_engine.find_sample_size` exercises the whole
path -- contract lowering, seeds, data generation, fit, decision, counters, isotonic crossing fit -- and has
to land on the SAME N: the reference's own gate for a cheaper run is 10 % (tolerances.R:99-103), an identical
run has no slack at all.
"""
import copy
import json
import math
import os

import msgpack
import pytest

from mcpower_amd import _engine as E

HERE = os.path.dirname(os.path.abspath(__file__))
CROSS = json.load(open(os.path.join(HERE, "golden", "crossing.json")))


def _contract_for(model):
    """The contract the R/Python ports build for `y ~ x1 [+ x2]` (shape of _golden/ols_contract.msgpack)."""
    c = msgpack.unpackb(open(os.path.join(HERE, "golden", "contracts", "ols_contract.msgpack"), "rb").read(),
                        raw=False)[0]
    c = copy.deepcopy(c)
    names = list(model["effects"])
    k = len(names)
    c["generation"]["columns"] = c["generation"]["columns"][:k]
    c["design_generation"]["terms"] = c["design_generation"]["terms"][:k + 1]
    c["outcome"]["coefficients"] = [0.0] + [model["effects"][n] for n in names]
    c["test"]["targets"] = [{"kind": "marginal", "term": i + 1} for i in range(k)]
    if model["family"] == "logit":
        bp = model["baseline_probability"]
        c["outcome"]["kind"] = "binary"
        c["outcome"]["intercept"] = math.log(bp / (1.0 - bp))
        c["estimator"] = "glm"
    elif model["family"] == "lme":
        icc = model["cluster"]["ICC"]
        c["generation"]["cluster"] = {"sizing": {"FixedClusters": {"n_clusters": model["cluster"]["n_clusters"]}},
                                      "tau_squared": icc / (1.0 - icc), "slopes": [], "extra_groupings": []}
        c["estimator"] = "mle"
    else:
        c["test"]["targets"].append({"kind": "joint", "terms": list(range(1, k + 1))})  # "all" carries the omnibus
    return c


def _dense_run(case, runner):
    g = CROSS[case]
    r = E.find_sample_size(msgpack.packb([_contract_for(g["model"])], use_bin_type=True), 80.0, g["from_size"],
                           g["to_size"], g["n_sims"], g["seed"], "grid", g["by"], "fixed", "linear", None, None,
                           _runner=runner)
    sc = r["scenarios"]["optimistic"] if "scenarios" in r else r
    return g, sc


def _check(g, sc):
    assert sc["n_sims"] == g["n_sims"]
    for pos, (label, want) in enumerate(g["dense"].items()):
        got = sc["fitted"][pos]
        assert got["status"] == want["status"], (label, got)
        if want["status"] == "fitted":
            assert got["n_achievable"] == want["n_achievable"], (label, got, want)


@pytest.mark.parametrize("case", ["cross_ols", "cross_partial", "cross_lme"])
def test_oracle_reproduces_the_frozen_required_n(case, oracle):
    def runner(spec, ns, n_sims, seed, sink):
        return oracle.run_batch(spec, ns, n_sims, seed, n_threads=min(8, os.cpu_count() or 1))
    _check(*_dense_run(case, runner))


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CROSS))
def test_device_reproduces_the_frozen_required_n(case, engine):
    _check(*_dense_run(case, None))     # the product's own runner: engine.run_batch on the MI355X
