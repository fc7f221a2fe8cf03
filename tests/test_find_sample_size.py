"""Sample-size search layer of the `_engine` shim: grid builder (grid.rs), isotonic
crossing fit (fit.rs), derivations and the host dict (result_host.rs:150-293).  Expected values of the
pure functions are the ones the reference's own unit tests assert (grid.rs / fit.rs / tests/test_grid.rs)."""
import math
import os

import msgpack
import numpy as np
import pytest

from mcpower_amd import _engine as E, contract as K, spec as S
from test_find_power import _blob, _contract, _oracle_runner, _pack

GRID = [50, 100, 150, 200]
SS_FIELDS = ["n_sims", "n_sample_sizes", "n_targets", "target_indices", "contrast_pairs", "power_uncorrected",
             "power_corrected", "ci_uncorrected", "ci_corrected", "convergence_rate", "sample_sizes", "first_achieved",
             "success_count_histogram_uncorrected", "success_count_histogram_corrected", "grid_warnings",
             "factor_exclusion_counts", "factor_separation_counts", "first_joint_achieved", "first_overall_achieved",
             "target_power", "scenario", "boundary_hit", "fitted", "fitted_joint", "fitted_overall", "cluster_atom"]


# ---------------------------------------------------------------------------------------------
# grid.rs
# ---------------------------------------------------------------------------------------------
def test_linear_grids_match_reference_cases():
    assert E.build_grid(30, 50, ("fixed", 5), "linear", 1, 1) == ([30, 35, 40, 45, 50], [])
    assert E.build_grid(30, 49, ("fixed", 5), "linear", 1, 1)[0] == [30, 35, 40, 45, 49]      # endpoint pinned
    assert E.build_grid(30, 52, ("fixed", 5), "linear", 1, 1)[0][-1] == 52
    g, w = E.build_grid(30, 200, ("auto", 12), "linear", 1, 1)
    assert g[0] == 30 and g[-1] == 200 and 11 <= len(g) <= 13 and w == []
    assert all(a < b for a, b in zip(g, g[1:]))


def test_clustered_grids_snap_to_the_atom_and_warn():
    g, w = E.build_grid(30, 200, ("auto", 12), "linear", 20, 40)
    assert all(n % 20 == 0 for n in g) and g[0] == 40 and g[-1] == 200
    assert any("raised" in m for m in w)
    g, w = E.build_grid(40, 205, ("auto", 12), "linear", 20, 40)
    assert g[-1] == 200 and any("lowered" in m for m in w)
    assert E.build_grid(40, 120, ("auto", 12), "linear", 20, 40)[0] == [40, 60, 80, 100, 120]
    g, w = E.build_grid(40, 80, ("auto", 12), "linear", 20, 40)
    assert len(g) < 4 and any("grid points" in m for m in w)
    with pytest.raises(E.OrchestratorError) as e:
        E.build_grid(190, 200, ("auto", 12), "linear", 60, 60)
    assert e.value.kind == "ClusterGridEmpty"
    with pytest.raises(E.OrchestratorError) as e:
        E.build_grid(40, 59, ("auto", 12), "linear", 20, 40)
    assert e.value.kind == "ClusterGridSinglePoint"
    with pytest.raises(E.OrchestratorError) as e:
        E.build_grid(0, 10, ("fixed", 1), "linear", 1, 1)
    assert e.value.kind == "InvalidGridBounds"


def test_log_grids_are_sorted_unique_atom_multiples():
    g, _ = E.build_grid(40, 400, ("auto", 8), "log", 20, 40)
    assert all(n % 20 == 0 for n in g) and g[0] == 40 and g[-1] == 400
    assert all(a < b for a, b in zip(g, g[1:]))
    g, _ = E.build_grid(10, 1000, ("fixed", 4), "log", 1, 1)
    assert g == [10, 46, 215, 1000]


def test_method_args_follow_from_host_args():
    assert E.sample_size_method_from_host_args("grid", 10, None, None) == (("fixed", 10), "linear")
    assert E.sample_size_method_from_host_args("grid", 12, "auto", "log") == (("auto", 12), "log")
    for bad in (("bisect", 1, None, None), ("grid", None, None, None), ("grid", 1, "step", None),
                ("grid", 1, None, "geometric")):
        with pytest.raises(ValueError):
            E.sample_size_method_from_host_args(*bad)


# ---------------------------------------------------------------------------------------------
# fit.rs goldens
# ---------------------------------------------------------------------------------------------
def test_pava_goldens():
    cases = [([0.1, 0.2, 0.3], [0.1, 0.2, 0.3]), ([0.3, 0.1], [0.2, 0.2]),
             ([0.1, 0.5, 0.3, 0.4, 0.9], [0.1, 0.4, 0.4, 0.4, 0.9]),
             ([0.2, 0.6, 0.5, 0.4, 0.8], [0.2, 0.5, 0.5, 0.5, 0.8]), ([0.9, 0.6, 0.3], [0.6, 0.6, 0.6])]
    for y, want in cases:
        assert E.pava(y) == pytest.approx(want, abs=1e-12)


def test_crossing_fit_goldens():
    f = E.fit_crossing(GRID, [60, 75, 85, 95], 100, 0.8, 1)
    assert f["status"] == "fitted" and f["n_star"] == pytest.approx(125.0, abs=1e-9) and f["n_achievable"] == 125
    assert f["ci_lo"] == pytest.approx(90.83640209948472, abs=1e-9)
    assert f["ci_hi"] == pytest.approx(163.5595794316606, abs=1e-9)
    f = E.fit_crossing(GRID, [60, 70, 80, 84], 100, 0.8, 1)                 # exact hit resolves leftmost
    assert f["n_star"] == pytest.approx(150.0, abs=1e-9) and f["n_achievable"] == 150
    assert f["ci_lo"] == pytest.approx(111.06986495758443, abs=1e-9) and f["ci_hi"] is None
    f = E.fit_crossing(GRID, [75, 85, 95, 98], 100, 0.8, 1)
    assert f["n_star"] == pytest.approx(75.0, abs=1e-9) and f["ci_lo"] is None
    assert f["ci_hi"] == pytest.approx(113.55957943166061, abs=1e-9)
    assert E.fit_crossing(GRID, [85, 90, 95, 98], 100, 0.8, 1) == {"status": "at_or_below_min", "n_min": 50}
    assert E.fit_crossing(GRID, [60, 75, 85, 95], 100, 0.8, 30)["n_achievable"] == 150
    assert E.fit_crossing(GRID, [60, 75, 85, 95], 100, 80.0, 1) == E.fit_crossing(GRID, [60, 75, 85, 95], 100, 0.8, 1)


def test_monotonicity_gate_goldens():
    assert E.fit_crossing([50, 100], [50, 40], 100, 0.9, 1) == {"status": "not_reached", "n_approx": None}
    f = E.fit_crossing([50, 100], [50, 35], 100, 0.9, 1)
    assert f["status"] == "non_monotone" and f["max_violation"] == pytest.approx(0.15, abs=1e-12)
    f = E.fit_crossing(GRID, [50, 20, 30, 95], 100, 0.8, 1)
    assert f["status"] == "non_monotone" and f["max_violation"] == pytest.approx(0.3, abs=1e-12)


def test_probit_hint_goldens():
    assert E.fit_crossing(GRID, [30, 45, 60, 72], 100, 0.9, 1) == {"status": "not_reached", "n_approx": 352}
    assert E.fit_crossing(GRID, [30, 45, 60, 72], 100, 0.9, 30) == {"status": "not_reached", "n_approx": 360}
    assert E.fit_crossing(GRID, [30, 40, 55, 65], 100, 0.9, 1) == {"status": "not_reached", "n_approx": None}
    assert E.fit_crossing(GRID, [20, 30, 40, 49], 100, 0.6, 1) == {"status": "not_reached", "n_approx": None}
    assert E.fit_crossing(GRID, [20, 30, 40, 51], 100, 0.6, 1) == {"status": "not_reached", "n_approx": 263}
    assert E.fit_crossing(GRID, [60, 60, 60, 60], 100, 0.7, 1) == {"status": "not_reached", "n_approx": None}
    assert E.fit_crossing(GRID, [0, 0, 0, 0], 0, 0.8, 1) == {"status": "not_reached", "n_approx": None}


def test_joint_first_n_from_histograms():
    hists = [[60, 30, 10], [40, 35, 25], [10, 30, 60]]       # exactly-j-significant buckets, n_sims = 100
    ss = [50, 100, 200]
    assert E.first_n_joint_at_target(hists, 100, ss, 0.50, 1) == 100
    assert E.first_n_joint_at_target(hists, 100, ss, 0.50, 2) == 200
    assert E.first_n_joint_at_target(hists, 100, ss, 0.90, 2) is None
    assert E.first_n_joint_at_target(hists, 100, ss, 50.0, 1) == 100
    assert E.first_n_joint_at_target(hists, 0, ss, 0.5, 1) is None


def test_cluster_grid_params():
    c = _contract(estimator="mle")
    c["generation"]["cluster"] = {"sizing": {"FixedClusters": {"n_clusters": 20}}, "tau_squared": 0.25, "slopes": [],
                                  "extra_groupings": [{"relation": {"Crossed": {"n_clusters": 12}},
                                                       "tau_squared": 0.1, "slopes": []}]}
    assert E.resolve_cluster_grid_params([c]) == (240, 40, [])          # find_sample_size.rs unit test
    c["generation"]["cluster"] = {"sizing": {"FixedClusters": {"n_clusters": 3}}, "tau_squared": 0.2, "slopes": [],
                                  "extra_groupings": []}
    atom, hard_min, w = E.resolve_cluster_grid_params([c])
    assert (atom, hard_min) == (3, 6) and "only 3 clusters" in w[0]
    c2 = _contract(estimator="mle")
    c2["generation"]["cluster"] = {"sizing": {"FixedSize": {"cluster_size": 8}}, "tau_squared": 0.2, "slopes": [],
                                   "extra_groupings": []}
    assert E.resolve_cluster_grid_params([c2]) == (8, 40, [])
    with pytest.raises(E.OrchestratorError) as e:
        E.resolve_cluster_grid_params([c, c2])
    assert e.value.kind == "MixedClusterAtoms"
    c2["generation"]["cluster"]["sizing"] = {"FixedSize": {"cluster_size": 1}}
    with pytest.raises(E.OrchestratorError) as e:
        E.resolve_cluster_grid_params([c2])
    assert e.value.kind == "ClusterSizeTooSmall"
    assert E.resolve_cluster_grid_params([_contract()]) == (1, 1, [])


def test_sparse_factor_sample_size_preflight():
    spec = S.SimulationSpec(factor_n_levels=[2], factor_proportions=[0.9, 0.1], factor_min_level_count=5)
    (w,) = E.factor_preflight_sample_size(spec, [20, 40, 60, 80])
    assert w.startswith("factor 1: levels reach the 5-observation minimum only at N >= ")
    n_inc = E.min_inclusion_n([0.9, 0.1], 5, 80)
    assert w.split(">= ")[1].startswith(str(n_inc)) and min(E.fixed_allocation_counts([0.9, 0.1], n_inc)) == 5
    assert min(E.fixed_allocation_counts([0.9, 0.1], n_inc - 1)) == 4
    (w,) = E.factor_preflight_sample_size(spec, [20, 30])
    assert "stays under the 5-observation minimum across the whole [20, 30] range" in w
    assert E.factor_preflight_sample_size(spec, [100, 200]) == []


# ---------------------------------------------------------------------------------------------
# end to end
# ---------------------------------------------------------------------------------------------
def _check_sample_size_result(res, oracle, blob, n_sims, seed, grid, target):
    (sc,) = E.lower_contracts(K.decode_contracts(blob), seed)
    batch = oracle.run_batch(sc["spec"], grid, n_sims, sc["base_seed"], n_threads=4)
    assert list(res) == SS_FIELDS
    assert res["sample_sizes"] == grid and res["n_sample_sizes"] == len(grid) and res["n_sims"] == n_sims
    want = (batch.corrected.sum(axis=0) / n_sims).tolist()
    assert res["power_corrected"] == want
    assert res["convergence_rate"] == (batch.converged.sum(axis=0) / n_sims).tolist()
    assert res["boundary_hit"] == batch.boundary_hit.tolist()
    for t in range(res["n_targets"]):
        first = next((n for n, row in zip(grid, want) if row[t] >= target), None)
        assert res["first_achieved"][t] == first
        counts = [int(v) for v in batch.corrected[:, :, t].sum(axis=0)]
        assert res["fitted"][t] == E.fit_crossing(grid, counts, n_sims, target, 1)
    per_sim = batch.corrected.sum(axis=2)
    for k in range(1, res["n_targets"] + 1):
        ge = [(per_sim[:, i] >= k).sum() / n_sims for i in range(len(grid))]
        assert res["first_joint_achieved"][k - 1] == next((n for n, p in zip(grid, ge) if p >= target), None)
    if sc["spec"].report_overall:
        counts = [int(v) for v in batch.overall.sum(axis=0)]
        assert res["fitted_overall"] == {0: E.fit_crossing(grid, counts, n_sims, target, 1)}
        assert res["first_overall_achieved"] == next((n for n, k in zip(grid, counts) if k / n_sims >= target), None)
    else:
        assert res["fitted_overall"] == {} and res["first_overall_achieved"] is None


def test_find_sample_size_with_oracle_runner(oracle):
    n_sims, seed = 500, 2137
    seen = []
    out = E.find_sample_size(_blob("ols_contract"), 0.8, 20, 140, n_sims, seed, "grid", by=20,
                             progress=lambda cur, tot: seen.append((cur, tot)) or True,
                             _runner=_oracle_runner(oracle))
    res = out["scenarios"]["optimistic"]
    grid = [20, 40, 60, 80, 100, 120, 140]
    _check_sample_size_result(res, oracle, _blob("ols_contract"), n_sims, seed, grid, 0.8)
    assert seen == [(0, n_sims * 7), (n_sims * 7, n_sims * 7)]        # progress counts fits, not draws
    assert res["cluster_atom"] == 1 and res["target_power"] == 0.8 and res["grid_warnings"] == []
    x1 = res["fitted"][0]
    assert x1["status"] in ("fitted", "at_or_below_min")               # x1 = 0.5 is detectable early
    x2 = res["fitted"][1]
    assert x2["status"] == "fitted" and 60 <= x2["n_star"] <= 120      # x2 = 0.3 needs ~90
    assert x2["ci_lo"] is None or x2["ci_lo"] <= x2["n_star"]
    assert res["first_joint_achieved"][0] <= res["first_joint_achieved"][1]


def test_long_grid_splits_into_device_sized_calls(oracle):
    calls = []
    inner = _oracle_runner(oracle)

    def runner(spec, ns, n_sims, seed, sink):
        calls.append(list(ns))
        return inner(spec, ns, n_sims, seed, sink)

    out = E.find_sample_size(_blob("ols_contract"), 0.8, 30, 329, 120, 5, "grid", by=1, _runner=runner)
    res = out["scenarios"]["optimistic"]
    assert [len(c) for c in calls] == [256, 44] and res["sample_sizes"] == list(range(30, 330))   # 256 sample sizes per device call
    (sc,) = E.lower_contracts(K.decode_contracts(_blob("ols_contract")), 5)
    # rows are prefix-stable, so a sub-grid reproduces the same flags as the full grid in one call
    part = oracle.run_batch(sc["spec"], [40, 100, 129], 120, sc["base_seed"])
    idx = [10, 70, 99]
    assert [res["power_uncorrected"][i] for i in idx] == (part.uncorrected.sum(axis=0) / 120).tolist()
    c = _contract(estimator="mle")
    c["generation"]["cluster"] = {"sizing": {"FixedClusters": {"n_clusters": 5}}, "tau_squared": 0.2, "slopes": [],
                                  "extra_groupings": []}
    c["test"]["targets"] = [{"kind": "marginal", "term": 1}]
    with pytest.raises(S.EngineError):
        E.find_sample_size(_pack(c), 0.8, 10, 2000, 50, 5, "grid", by=5, _runner=runner)   # 399 clustered sizes: not splittable


@pytest.mark.gpu
def test_find_sample_size_on_device_equals_oracle_runner(engine, oracle):
    args = (_blob("two_factor_accumulation_contract"), 0.8, 40, 400, 3000, 2137, "grid")
    dev = E.find_sample_size(*args, by=10, by_kind="auto", mode="log")
    ref = E.find_sample_size(*args, by=10, by_kind="auto", mode="log", _runner=_oracle_runner(oracle))
    assert dev == ref
    res = dev["scenarios"]["optimistic"]
    assert list(res) == SS_FIELDS and res["sample_sizes"][0] == 40 and res["sample_sizes"][-1] == 400
    assert res["fitted_overall"][0]["status"] in ("fitted", "at_or_below_min")


@pytest.mark.gpu
def test_find_sample_size_on_device_clustered_grid(engine, oracle):
    c = _contract(estimator="mle")
    c["generation"]["cluster"] = {"sizing": {"FixedClusters": {"n_clusters": 10}}, "tau_squared": 0.25, "slopes": [],
                                  "extra_groupings": []}
    c["test"]["targets"] = [{"kind": "marginal", "term": 1}, {"kind": "marginal", "term": 2}]
    dev = E.find_sample_size(_pack(c), 0.8, 30, 205, 800, 9, "grid", by=6, by_kind="auto")
    ref = E.find_sample_size(_pack(c), 0.8, 30, 205, 800, 9, "grid", by=6, by_kind="auto",
                             _runner=_oracle_runner(oracle))
    d, r = dev["scenarios"]["optimistic"], ref["scenarios"]["optimistic"]
    assert d["sample_sizes"] == r["sample_sizes"] and all(n % 10 == 0 for n in d["sample_sizes"])
    assert d["cluster_atom"] == 10 and any("lowered" in w for w in d["grid_warnings"])
    # REML fits agree to 1e-4; decisions may differ only at knife edges
    du, ru = np.array(d["power_uncorrected"]), np.array(r["power_uncorrected"])
    assert np.abs(du - ru).max() <= 2.0 / 800
