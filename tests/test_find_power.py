"""`_engine.find_power` shim: contract lowering (contract_adapter.rs), orchestration (find_power.rs) and
the host dict (result_host.rs).  The CPU tests drive the shim with the oracle as the batch runner
(checker only); the gpu test calls the shipped path and compares it with that."""
import math
import os

import msgpack
import numpy as np
import pytest

from mcpower_amd import _engine, contract as K, spec as S

HERE = os.path.dirname(os.path.abspath(__file__))
FIELDS = ["n_sims", "n_sample_sizes", "n_targets", "target_indices", "contrast_pairs", "power_uncorrected",
          "power_corrected", "ci_uncorrected", "ci_corrected", "convergence_rate", "scenario", "sample_sizes",
          "estimator_extras", "boundary_hit", "overall_significant_rate", "overall_significant_ci",
          "success_count_histogram_uncorrected", "success_count_histogram_corrected", "grid_warnings", "posthoc",
          "factor_exclusion_counts", "factor_separation_counts"]


def _blob(name):
    return open(os.path.join(HERE, "golden", "contracts", f"{name}.msgpack"), "rb").read()


def _contract(**over):
    c = msgpack.unpackb(_blob("ols_contract"), raw=False)[0]
    for k, v in over.items():
        node = c
        *path, leaf = k.split("__")
        for p in path:
            node = node[p]
        node[leaf] = v
    return c


def _pack(*contracts):
    return msgpack.packb(list(contracts), use_bin_type=True)


def _oracle_runner(oracle):
    def run(spec, ns, n_sims, seed, sink):
        if sink is not None:
            assert sink(0, n_sims)
        r = oracle.run_batch(spec, ns, n_sims, seed, n_threads=4)
        if sink is not None and not sink(n_sims, n_sims):
            raise S.EngineError(S.Status.Cancelled, "cancelled")
        return r
    return run


# ---------------------------------------------------------------------------------------------
# lowering
# ---------------------------------------------------------------------------------------------
def test_reference_golden_contract_lowers_like_the_adapter():
    """ports/r/tests/testthat/_golden/ols_contract.msgpack: y ~ x1 + x2, x1=0.5, x2=0.3, overall joint."""
    (c,) = K.decode_contracts(_blob("ols_contract"))
    s = K.contract_to_simulation_spec(c)
    assert (s.n_non_factor, s.n_factor_dummies) == (2, 0)
    assert s.var_types == [S.Distribution.Normal, S.Distribution.Normal]
    assert s.correlation == [1.0, 0.0, 0.0, 1.0]                 # contract_adapter.rs example1 test
    assert s.effect_sizes == [0.0, 0.5, 0.3]
    assert s.target_indices == [1, 2] and s.contrast_pairs == []
    assert s.report_overall                                      # omnibus Joint -> report_overall
    assert s.fit_columns == [] and s.posthoc == []
    assert s.scenario.is_optimistic() and s.scenario.truth_start
    assert s.scenario.new_distributions == [S.Distribution.RightSkewed, S.Distribution.LeftSkewed,
                                            S.Distribution.Uniform]
    assert s.scenario.residual_dists == [S.ResidualDist.HighKurtosis, S.ResidualDist.RightSkewed]
    assert s.max_failed_fraction == 0.1 and s.factor_min_level_count == 0


def test_two_factor_contract_columns_and_interaction():
    (c,) = K.decode_contracts(_blob("two_factor_accumulation_contract"))
    s = K.contract_to_simulation_spec(c)
    assert (s.n_non_factor, s.n_factor_dummies) == (0, 2)
    assert s.factor_n_levels == [2, 2] and s.factor_proportions == [0.5, 0.5, 0.6, 0.4]
    assert s.factor_sampled == [None, None]
    assert s.interactions == [[1, 2]]
    assert s.effect_sizes == [0.0, 0.5, 0.4, 0.3]
    assert s.target_indices == [1, 2, 3] and s.report_overall


def test_effect_sizes_follow_kernel_column_order_not_term_order():
    """contract_adapter.rs `effect_sizes_layout_matches_design_matrix_columns`: factor first in the
    generation columns, continuous second -> kernel layout [intercept, x1, g[2], g[3]]."""
    c = _contract()
    c["generation"]["columns"] = [
        {"FactorSynthetic": {"n_levels": 3, "proportions": [1 / 3, 1 / 3, 1 / 3], "sampled_proportions": None}},
        {"Synthetic": {"kind": "Normal", "pinned": False}}]
    terms = ["Const", {"DummyOf": {"column": 0, "level_index": 1}}, {"DummyOf": {"column": 0, "level_index": 2}},
             {"Direct": {"column": 1}}]
    c["design_generation"] = {"terms": terms}
    c["design_test"] = {"terms": list(terms)}
    c["outcome"]["coefficients"] = [0.0, 0.3, 0.5, 0.4]
    c["test"]["targets"] = [{"kind": "marginal", "term": 3}]
    s = K.contract_to_simulation_spec(c)
    assert s.effect_sizes == [0.0, 0.4, 0.3, 0.5]
    assert s.target_indices == [1]
    assert K.report_targets_and_contrasts(c) == ([3], [])
    assert s.fit_columns == []      # design_test covers the whole generation design


def test_logit_intercept_folds_into_slot_zero_and_binary_params():
    c = _contract(outcome__kind="binary", outcome__intercept=math.log(0.3 / 0.7), estimator="glm")
    c["generation"]["columns"][1] = {"Synthetic": {"kind": {"Binary": {"p": 0.25}}, "pinned": True}}
    s = K.contract_to_simulation_spec(c)
    assert s.effect_sizes[0] == pytest.approx(-0.8472978603872037)
    assert s.var_types[1] == S.Distribution.Binary and s.var_params == [0.0, 0.25] and s.var_pinned == [False, True]
    assert s.estimator == S.EstimatorSpec.Glm and s.outcome_kind == S.OutcomeKind.Binary and s.link is None


def test_contrasts_collapse_on_reference_level_and_sort_by_kernel_column():
    c = _contract()
    c["generation"]["columns"] = [
        {"Synthetic": {"kind": "Normal", "pinned": False}},
        {"FactorSynthetic": {"n_levels": 3, "proportions": [0.4, 0.3, 0.3], "sampled_proportions": True}}]
    c["design_generation"] = {"terms": ["Const", {"DummyOf": {"column": 1, "level_index": 2}},
                                        {"DummyOf": {"column": 1, "level_index": 1}}, {"Direct": {"column": 0}}]}
    c["outcome"]["coefficients"] = [0.0, 0.2, 0.1, 0.4]
    c["test"]["targets"] = [{"kind": "marginal", "term": 1},                 # g[3] -> kernel col 3
                            {"kind": "contrast", "positive": 2, "negative": 0},  # g[2] vs reference -> marginal col 2
                            {"kind": "contrast", "positive": 1, "negative": 2},  # g[3] - g[2]
                            {"kind": "marginal", "term": 3}]                 # x1 -> kernel col 1
    s = K.contract_to_simulation_spec(c)
    assert s.effect_sizes == [0.0, 0.4, 0.1, 0.2]
    assert s.target_indices == [1, 2, 3] and s.contrast_pairs == [(3, 2)]
    assert K.report_targets_and_contrasts(c) == ([3, 2, 1], [(1, 2)])
    assert s.factor_sampled == [True] and not s.report_overall


def test_test_formula_subset_sets_fit_columns_and_posthoc_blocks():
    c = _contract()
    c["generation"]["columns"] = [
        {"Synthetic": {"kind": "Normal", "pinned": False}},
        {"Synthetic": {"kind": "Uniform", "pinned": False}},
        {"FactorSynthetic": {"n_levels": 3, "proportions": [0.4, 0.3, 0.3], "sampled_proportions": None}}]
    d1, d2 = {"DummyOf": {"column": 2, "level_index": 1}}, {"DummyOf": {"column": 2, "level_index": 2}}
    c["design_generation"] = {"terms": ["Const", {"Direct": {"column": 0}}, {"Direct": {"column": 1}}, d1, d2]}
    c["design_test"] = {"terms": ["Const", {"Direct": {"column": 0}}, d1, d2]}     # drops x2
    c["outcome"]["coefficients"] = [0.0, 0.3, 0.2, 0.1, 0.4]
    c["outcome"]["heteroskedasticity_driver"] = 1
    c["test"]["targets"] = [{"kind": "marginal", "term": 1}, {"kind": "joint", "terms": [3, 1, 2]}]
    c["posthoc"] = [{"factor_column": 2, "target_term_indices": [2, 3], "posthoc_alpha": 0.01}]
    s = K.contract_to_simulation_spec(c)
    assert s.fit_columns == [0, 1, 3, 4]
    assert s.heteroskedasticity_driver == 2
    assert s.report_overall
    assert [(b.factor_index, b.target_indices) for b in s.posthoc] == [(0, [3, 4])]
    assert s.crit_values.posthoc_alpha == 0.01
    assert _engine._posthoc_blocks(s) == [(3, 3)]


def test_cluster_block_and_lme_scenario():
    c = _contract(estimator="mle")
    c["generation"]["cluster"] = {"sizing": {"FixedSize": {"cluster_size": 8}}, "tau_squared": 0.25,
                                  "slopes": [], "extra_groupings": []}
    c["scenario"]["lme"] = {"random_effect_dist": "high_kurtosis", "random_effect_df": 5.0, "icc_noise_sd": 0.1}
    s = K.contract_to_simulation_spec(c)
    assert s.cluster.sizing == ("FixedSize", 8) and s.cluster.tau_squared == 0.25
    assert s.scenario.lme.random_effect_dist == S.ResidualDist.HighKurtosis and not s.scenario.is_optimistic()
    assert K.cluster_atom(c) == 8


@pytest.mark.parametrize("mutate, msg", [
    (dict(outcome__coefficients=[0.0, 0.5]), "coefficients.len (2) must equal design_generation.terms.len (3)"),
    (dict(test__alpha=0.0), "test.alpha must be in (0.0, 1.0); got 0.0"),
    (dict(max_failed_fraction=1.5), "max_failed_fraction must be in [0.0, 1.0]; got 1.5"),
    (dict(estimator="glm"), "estimator Glm requires outcome.kind == Binary or Count"),
    (dict(estimator="mle"), "estimator Mle requires generation.cluster to be Some(_)"),
    (dict(outcome__link="probit"), "outcome.link == Some(Probit) requires outcome.kind == Binary"),
    (dict(test__targets=[{"kind": "marginal", "term": 7}]),
     "test target term index 7 out of range (design_test has 3 terms)"),
])
def test_contract_validation_wording(mutate, msg):
    with pytest.raises(K.ContractError) as e:
        K.contract_to_simulation_spec(_contract(**mutate))
    assert msg in str(e.value)


def test_non_omnibus_joint_is_rejected():
    c = _contract(test__targets=[{"kind": "joint", "terms": [1]}])
    with pytest.raises(K.JointNotSupported):
        K.contract_to_simulation_spec(c)
    with pytest.raises(_engine.OrchestratorError) as e:
        _engine.lower_contracts([c], 1)
    assert 'contract 0 ("optimistic")' in str(e.value) and e.value.kind == "InvalidScenarios"


# ---------------------------------------------------------------------------------------------
# orchestration pieces
# ---------------------------------------------------------------------------------------------
def test_wilson_ci_known_values():
    assert _engine.wilson_ci(0, 0) == (0.0, 1.0)
    lo, hi = _engine.wilson_ci(50, 100)        # aggregation.rs tests: symmetric around 0.5
    assert lo == pytest.approx(0.4038, abs=5e-4) and hi == pytest.approx(0.5962, abs=5e-4)
    lo, hi = _engine.wilson_ci(0, 100)
    assert lo == 0.0 and 0.03 < hi < 0.04


def test_call_seed_is_the_mix13_finalizer(oracle):
    for s in (0, 1, 2137, (1 << 64) - 1):
        assert _engine.splitmix64_finalize(s) == oracle.splitmix64(s)
    sc = _engine.lower_contracts(K.decode_contracts(_blob("ols_contract")), 2137)
    assert sc[0]["base_seed"] == oracle.splitmix64(2137)
    assert sc[0]["spec"].factor_min_level_count == 5           # configs/config.json limits


def test_snap_single_n_floors_to_the_cluster_atom():
    c = _contract(estimator="mle")
    c["generation"]["cluster"] = {"sizing": {"FixedClusters": {"n_clusters": 12}}, "tau_squared": 0.2,
                                  "slopes": [], "extra_groupings": []}
    assert _engine.snap_single_n([c], 100) == (
        96, "sample_size 100 not a multiple of the cluster atom 12; using 96")
    assert _engine.snap_single_n([c], 96) == (96, None)
    assert _engine.snap_single_n([c], 5)[0] == 12              # never a zero-row fit
    c["generation"]["cluster"]["sizing"] = {"FixedSize": {"cluster_size": 10}}
    with pytest.raises(_engine.OrchestratorError) as e:
        _engine.snap_single_n([c], 45)
    assert e.value.kind == "ClusterTooFewAtN" and "yields only 4 cluster(s)" in str(e.value)
    assert _engine.snap_single_n([_contract()], 101) == (101, None)


def test_fixed_allocation_counts_and_preflight_warning():
    assert _engine.fixed_allocation_counts([0.5, 0.5], 7) == [4, 3]          # round-robin on ties
    assert _engine.fixed_allocation_counts([0.9, 0.1], 20) == [18, 2]
    assert sum(_engine.fixed_allocation_counts([0.2, 0.5, 0.3], 101)) == 101
    spec = S.SimulationSpec(factor_n_levels=[2], factor_proportions=[0.9, 0.1], factor_min_level_count=5)
    (w,) = _engine.factor_preflight_warnings(spec, 20)
    assert w.startswith("factor 1: level 2 receives 2 of 20 observations (minimum 5);")
    assert _engine.factor_preflight_warnings(spec, 200) == []
    spec.factor_sampled = [True]
    assert _engine.factor_preflight_warnings(spec, 20) == []                 # stochastic counts: silent


def _check_against_batch(res, batch, spec, n_sims):
    unc = batch.uncorrected[:, 0, :]
    assert res["power_uncorrected"] == [[float(v) / n_sims for v in unc.sum(axis=0)]]
    assert res["power_corrected"] == [[float(v) / n_sims for v in batch.corrected[:, 0, :].sum(axis=0)]]
    assert res["convergence_rate"] == [float(batch.converged[:, 0].sum()) / n_sims]
    hist = np.bincount(unc.sum(axis=1), minlength=unc.shape[1] + 1)
    assert res["success_count_histogram_uncorrected"] == [int(v) for v in hist]
    assert sum(res["success_count_histogram_corrected"]) == n_sims
    assert res["boundary_hit"] == [[int(v)] for v in batch.boundary_hit[:, 0]]
    if spec.report_overall:
        k = int(batch.overall[:, 0].sum())
        assert res["overall_significant_rate"] == k / n_sims
        assert res["overall_significant_ci"] == _engine.wilson_ci(k, n_sims)
    else:
        assert res["overall_significant_rate"] is None and res["overall_significant_ci"] is None
    for (lo, hi), p in zip(res["ci_uncorrected"][0], res["power_uncorrected"][0]):
        assert 0.0 <= lo <= p <= hi <= 1.0


def test_find_power_dict_shape_and_values_with_oracle_runner(oracle):
    n_sims = 600
    out = _engine.find_power(_blob("ols_contract"), 80, n_sims, 2137, _runner=_oracle_runner(oracle))
    assert list(out) == ["scenarios", "comparison"] and out["comparison"] == {}
    assert list(out["scenarios"]) == ["optimistic"]
    res = out["scenarios"]["optimistic"]
    assert list(res) == FIELDS                                   # result_host.rs:64-146 field order
    assert res["n_sims"] == n_sims and res["n_sample_sizes"] == 1 and res["n_targets"] == 2
    assert res["target_indices"] == [1, 2] and res["contrast_pairs"] == []
    assert res["sample_sizes"] == [80] and res["scenario"] == "optimistic"
    assert res["estimator_extras"] == {"estimator": "ols"}
    assert res["posthoc"] == [] and res["grid_warnings"] == []
    assert res["factor_exclusion_counts"] == [] and res["factor_separation_counts"] == []
    (sc,) = _engine.lower_contracts(K.decode_contracts(_blob("ols_contract")), 2137)
    batch = oracle.run_batch(sc["spec"], [80], n_sims, sc["base_seed"])
    _check_against_batch(res, batch, sc["spec"], n_sims)
    assert res["power_uncorrected"][0][0] > 0.95 and 0.6 < res["power_uncorrected"][0][1] < 0.9


def test_scenarios_are_paired_runs_and_progress_is_call_cumulative(oracle):
    base = _contract()
    other = _contract(scenario__name="noisy", scenario__residual_change_prob=1.0)
    seen = []
    out = _engine.find_power(_pack(base, other), 60, 200, 7, progress=lambda cur, tot: seen.append((cur, tot)) or True,
                             _runner=_oracle_runner(oracle))
    assert list(out["scenarios"]) == ["optimistic", "noisy"]
    assert seen == [(0, 400), (200, 400), (200, 400), (400, 400)]
    scs = _engine.lower_contracts([base, other], 7)
    assert scs[0]["base_seed"] == scs[1]["base_seed"]            # same call seed for every scenario
    # cancellation: progress returning False surfaces as the Cancelled orchestrator error
    with pytest.raises(_engine.OrchestratorError) as e:
        _engine.find_power(_pack(base, other), 60, 200, 7, progress=lambda cur, tot: cur < 200,
                           _runner=_oracle_runner(oracle))
    assert e.value.kind == "Cancelled"


def test_clustered_find_power_snaps_and_reports_mle_extras(oracle):
    c = _contract(estimator="mle")
    c["generation"]["cluster"] = {"sizing": {"FixedClusters": {"n_clusters": 10}}, "tau_squared": 0.25,
                                  "slopes": [], "extra_groupings": []}
    c["test"]["targets"] = [{"kind": "marginal", "term": 1}, {"kind": "marginal", "term": 2}]
    out = _engine.find_power(_pack(c), 125, 150, 3, _runner=_oracle_runner(oracle))
    res = out["scenarios"]["optimistic"]
    assert res["sample_sizes"] == [120]
    assert res["grid_warnings"] == ["sample_size 125 not a multiple of the cluster atom 10; using 120"]
    ex = res["estimator_extras"]
    assert list(ex) == ["estimator", "tau_estimate", "boundary_hits", "joint_uncorrected_rate",
                        "joint_corrected_rate", "singular_fit_rate", "boundary_rate_per_component"]
    assert ex["estimator"] == "mle" and math.isnan(ex["tau_estimate"])
    assert len(ex["boundary_rate_per_component"]) == 1 and 0.0 <= ex["singular_fit_rate"] <= 1.0
    assert ex["joint_uncorrected_rate"] > 0.5


# ---------------------------------------------------------------------------------------------
# the shipped path
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name, n", [("ols_contract", 80), ("two_factor_accumulation_contract", 120)])
def test_find_power_on_device_equals_oracle_runner(engine, oracle, name, n):
    n_sims = 4000
    dev = _engine.find_power(_blob(name), n, n_sims, 2137)
    ref = _engine.find_power(_blob(name), n, n_sims, 2137, _runner=_oracle_runner(oracle))
    assert dev == ref
    res = dev["scenarios"]["optimistic"]
    assert list(res) == FIELDS and res["n_sims"] == n_sims


@pytest.mark.gpu
def test_find_power_on_device_glm_and_lmm_extras(engine, oracle):
    c = _contract(outcome__kind="binary", outcome__intercept=math.log(0.3 / 0.7), estimator="glm")
    c["test"]["targets"] = [{"kind": "marginal", "term": 1}, {"kind": "marginal", "term": 2}]
    dev = _engine.find_power(_pack(c), 150, 2000, 11)
    ref = _engine.find_power(_pack(c), 150, 2000, 11, _runner=_oracle_runner(oracle))
    d, r = dev["scenarios"]["optimistic"], ref["scenarios"]["optimistic"]
    assert d["estimator_extras"]["estimator"] == "glm" and math.isnan(d["estimator_extras"]["baseline_prob_realized"])
    for k in FIELDS:
        if k != "estimator_extras":
            assert d[k] == r[k], k
    c = _contract(estimator="mle")
    c["generation"]["cluster"] = {"sizing": {"FixedClusters": {"n_clusters": 10}}, "tau_squared": 0.25,
                                  "slopes": [], "extra_groupings": []}
    c["test"]["targets"] = [{"kind": "marginal", "term": 1}, {"kind": "marginal", "term": 2}]
    dev = _engine.find_power(_pack(c), 125, 1500, 3)["scenarios"]["optimistic"]
    ref = _engine.find_power(_pack(c), 125, 1500, 3, _runner=_oracle_runner(oracle))["scenarios"]["optimistic"]
    assert dev["power_uncorrected"] == ref["power_uncorrected"] and dev["sample_sizes"] == [120]
    assert dev["estimator_extras"]["joint_uncorrected_rate"] == ref["estimator_extras"]["joint_uncorrected_rate"]


# ---------------------------------------------------------------------------------------------
# contracts with random slopes / extra groupings reach the general mixed-model solver
# ---------------------------------------------------------------------------------------------
def _mixed_contract():
    c = _contract()
    c["generation"]["cluster"] = {
        "sizing": {"FixedClusters": {"n_clusters": 12}}, "tau_squared": 0.25,
        "slopes": [{"column": 0, "variance": 0.10, "corr_with_intercept": 0.3, "corr_with": []}],
        "extra_groupings": [{"relation": {"Crossed": {"n_clusters": 4}}, "tau_squared": 0.16, "slopes": []}]}
    c["estimator"] = "mle"
    c["outcome"]["coefficients"] = [0.0, 0.30, 0.20]
    c["test"]["targets"] = [{"kind": "marginal", "term": 1}, {"kind": "marginal", "term": 2}]
    return c


def test_mixed_contract_lowers_slopes_and_groupings():
    s = K.contract_to_simulation_spec(_mixed_contract())
    assert s.estimator == S.EstimatorSpec.Mle and s.cluster.sizing == ("FixedClusters", 12)
    assert s.cluster_slope_design_cols == [1] and s.extra_slope_cols == [[]]
    assert K.cluster_atom(_mixed_contract()) == 48                      # generation.rs:291-295: sizing atom x block levels
    low = s.to_c()
    assert low.c.cluster.slopes.n_slopes == 1 and low.c.cluster.slopes.design_cols[0] == 1
    assert list(low.c.cluster.slopes.re_corr)[:4] == [1.0, 0.3, 0.3, 1.0]
    assert low.c.cluster.n_extra_groupings == 1 and low.c.cluster.extra[0].count == 4


def test_find_power_mixed_contract_on_the_oracle(oracle):
    r = _engine.find_power(_pack(_mixed_contract()), 240, 60, 2137, None, _runner=_oracle_runner(oracle))
    sc = r["scenarios"]["optimistic"] if "scenarios" in r else r
    assert sc["n_sims"] == 60 and len(sc["power_uncorrected"][0]) == 2
    assert 0.5 < sc["power_uncorrected"][0][0] <= 1.0 and sc["convergence_rate"][0] > 0.9


@pytest.mark.gpu
def test_find_power_mixed_contract_device_equals_oracle(engine, oracle):
    blob = _pack(_mixed_contract())
    dev = _engine.find_power(blob, 240, 200, 2137, None)
    ora = _engine.find_power(blob, 240, 200, 2137, None, _runner=_oracle_runner(oracle))
    d = dev["scenarios"]["optimistic"] if "scenarios" in dev else dev
    o = ora["scenarios"]["optimistic"] if "scenarios" in ora else ora
    for t in range(2):
        assert abs(d["power_uncorrected"][0][t] - o["power_uncorrected"][0][t]) <= 1.0 / 200 + 1e-12
    assert abs(d["convergence_rate"][0] - o["convergence_rate"][0]) <= 1.0 / 200 + 1e-12


# ---------------------------------------------------------------------------------------------
# clustered binary outcome: the Glm estimator with a ClusterSpec is a random-intercept GLMM
# ---------------------------------------------------------------------------------------------
def _glmm_contract():
    c = _contract()
    c["generation"]["cluster"] = {"sizing": {"FixedClusters": {"n_clusters": 20}},
                                  "tau_squared": 0.25 * math.pi ** 2 / 3.0, "slopes": [], "extra_groupings": []}
    c["estimator"] = "glm"
    c["outcome"]["kind"] = "binary"
    c["outcome"]["intercept"] = math.log(0.3 / 0.7)
    c["outcome"]["coefficients"] = [0.0, 0.50, 0.30]
    c["test"]["targets"] = [{"kind": "marginal", "term": 1}, {"kind": "marginal", "term": 2}]
    return c


def test_find_power_clustered_glm_on_the_oracle(oracle):
    r = _engine.find_power(_pack(_glmm_contract()), 400, 40, 2137, None, _runner=_oracle_runner(oracle))
    sc = r["scenarios"]["optimistic"] if "scenarios" in r else r
    assert sc["n_sims"] == 40 and 0.5 < sc["power_uncorrected"][0][0] <= 1.0 and sc["convergence_rate"][0] > 0.9
    ex = sc["estimator_extras"]
    assert ex["estimator"] == "glm" and 0.2 < ex["tau_squared_hat_mean"] < 2.0   # truth: 0.25 * pi^2 / 3 = 0.82


@pytest.mark.gpu
def test_find_power_clustered_glm_device_equals_oracle(engine, oracle):
    blob = _pack(_glmm_contract())
    dev = _engine.find_power(blob, 400, 200, 2137, None)
    ora = _engine.find_power(blob, 400, 200, 2137, None, _runner=_oracle_runner(oracle))
    d = dev["scenarios"]["optimistic"] if "scenarios" in dev else dev
    o = ora["scenarios"]["optimistic"] if "scenarios" in ora else ora
    for t in range(2):
        assert abs(d["power_uncorrected"][0][t] - o["power_uncorrected"][0][t]) <= 1.0 / 200 + 1e-12
    assert abs(d["convergence_rate"][0] - o["convergence_rate"][0]) <= 1.0 / 200 + 1e-12


def _glmm_slope_contract():
    c = _glmm_contract()
    c["generation"]["cluster"]["slopes"] = [{"column": 0, "variance": 0.10, "corr_with_intercept": 0.3, "corr_with": []}]
    return c


def test_find_power_clustered_glm_with_a_random_slope_on_the_oracle(oracle):
    """(1 + x1 | g) under a binary outcome: the contract's slope reaches the vector-mode Laplace fit."""
    spec = K.contract_to_simulation_spec(_glmm_slope_contract())
    assert spec.cluster.slopes and spec.cluster_slope_design_cols == [1]
    r = _engine.find_power(_pack(_glmm_slope_contract()), 400, 24, 2137, None, _runner=_oracle_runner(oracle))
    sc = r["scenarios"]["optimistic"] if "scenarios" in r else r
    assert sc["n_sims"] == 24 and sc["convergence_rate"][0] > 0.8 and 0.3 < sc["power_uncorrected"][0][0] <= 1.0


@pytest.mark.gpu
def test_find_power_clustered_glm_with_a_random_slope_device_equals_oracle(engine, oracle):
    blob = _pack(_glmm_slope_contract())
    dev = _engine.find_power(blob, 400, 96, 2137, None)
    ora = _engine.find_power(blob, 400, 96, 2137, None, _runner=_oracle_runner(oracle))
    d = dev["scenarios"]["optimistic"] if "scenarios" in dev else dev
    o = ora["scenarios"]["optimistic"] if "scenarios" in ora else ora
    for t in range(2):
        assert abs(d["power_uncorrected"][0][t] - o["power_uncorrected"][0][t]) <= 2.0 / 96 + 1e-12
    assert abs(d["convergence_rate"][0] - o["convergence_rate"][0]) <= 2.0 / 96 + 1e-12


def _glmm_crossed_contract():
    c = _glmm_slope_contract()
    c["generation"]["cluster"]["extra_groupings"] = [{"relation": {"Crossed": {"n_clusters": 5}}, "tau_squared": 0.16, "slopes": []}]
    return c


def test_find_power_clustered_glm_with_a_crossed_grouping_on_the_oracle(oracle):
    """(1 + x1 | g) + (1 | item) under a binary outcome: the contract reaches the joint-mode Laplace fit."""
    r = _engine.find_power(_pack(_glmm_crossed_contract()), 400, 12, 2137, None, _runner=_oracle_runner(oracle))
    sc = r["scenarios"]["optimistic"] if "scenarios" in r else r
    assert sc["n_sims"] == 12 and sc["convergence_rate"][0] > 0.7 and 0.2 < sc["power_uncorrected"][0][0] <= 1.0


@pytest.mark.gpu
def test_find_power_clustered_glm_with_a_crossed_grouping_device_equals_oracle(engine, oracle):
    blob = _pack(_glmm_crossed_contract())
    dev = _engine.find_power(blob, 400, 48, 2137, None)
    ora = _engine.find_power(blob, 400, 48, 2137, None, _runner=_oracle_runner(oracle))
    d = dev["scenarios"]["optimistic"] if "scenarios" in dev else dev
    o = ora["scenarios"]["optimistic"] if "scenarios" in ora else ora
    for t in range(2):
        assert abs(d["power_uncorrected"][0][t] - o["power_uncorrected"][0][t]) <= 2.0 / 48 + 1e-12
    assert abs(d["convergence_rate"][0] - o["convergence_rate"][0]) <= 2.0 / 48 + 1e-12


def test_clustered_glm_opt_in_modes_are_refused_not_approximated():
    c = _glmm_contract()
    c["nagq"] = 7              # quadrature over a scalar random intercept lowers to spec.nagq ...
    assert K.contract_to_simulation_spec(c).nagq == 7 and K.contract_to_simulation_spec(_glmm_contract()).nagq == 1
    c = _glmm_slope_contract()
    c["nagq"] = 7              # ... the tensor rule over slopes is refused, not approximated
    with pytest.raises(S.EngineError) as e:
        K.contract_to_simulation_spec(c)
    assert e.value.status == S.Status.Unsupported and "nagq" in str(e.value)
    c = _glmm_contract()
    c["wald_se"] = "hessian"   # the rigor opt-in lowers to spec.wald_se (contract_adapter.rs:227)
    assert K.contract_to_simulation_spec(c).wald_se == S.WaldSe.Hessian
    assert K.contract_to_simulation_spec(_glmm_contract()).wald_se == S.WaldSe.Rx
    c["wald_se"] = "sandwich"
    with pytest.raises(K.ContractError):
        K.contract_to_simulation_spec(c)
    c = _glmm_contract()
    c["nagq"] = 4
    with pytest.raises(K.ContractError) as e:
        K.contract_to_simulation_spec(c)
    assert "nagq must be odd" in str(e.value)
