"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/mcpower_hip.h declares, and its host-only entry points (validation, critical values)
agree with the oracle.  No device compute here."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from helpers import BASE_SEED, spec_c1, spec_c2a, spec_c2b
from mcpower_amd import spec as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "mcpower_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mcp_[a-z0-9_]+)\s*\(", txt)) - {"mcp_progress_fn"})


def test_library_exports_every_declared_symbol(engine):
    L = engine.lib()
    syms = _declared_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/mcpower_hip.h but not exported"
    assert set(syms) == set(engine.EXPORTS)
    assert L.mcp_abi_version() == S.ABI_VERSION


def test_ctypes_mirror_matches_the_header_field_by_field(tmp_path):
    """sizeof / offsetof of every struct of include/mcpower_hip.h, from the C compiler, against the ctypes mirror
    (mcpower_amd/spec.py): a drift in either fails here, field by field."""
    import ctypes as C
    import subprocess
    mirror = {"mcp_spec": S.CSpec, "mcp_batch_result": S.CBatchResult, "mcp_run_options": S.CRunOptions,
              "mcp_counter_layout": S.CCounterLayout, "mcp_cluster": S.CCluster, "mcp_re_slopes": S.CReSlopes,
              "mcp_grouping": S.CGrouping, "mcp_scenario": S.CScenario}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "mcpower_hip.h"', 'int main(void) {']
    for cname, cls in mirror.items():
        lines.append(f'  printf("{cname} sizeof %zu\\n", sizeof({cname}));')
        for fname, *_ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  printf("mcp_comm_id sizeof %zu\\n", sizeof(mcp_comm_id));', '  return 0;', '}']
    src = tmp_path / "abi_table.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi_table"
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True, capture_output=True)
    table = {}
    for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n"):
        if ln:
            t, f, v = ln.split()
            table[(t, f)] = int(v)
    n_fields = 0
    for cname, cls in mirror.items():
        assert C.sizeof(cls) == table[(cname, "sizeof")], cname
        for fname, *_ in cls._fields_:
            assert getattr(cls, fname).offset == table[(cname, fname)], (cname, fname)
            n_fields += 1
    assert n_fields > 120 and table[("mcp_comm_id", "sizeof")] == 128
    # every member the header declares is mirrored (count the declarators of each struct body)
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mcpower_hip.h")).read(), flags=re.S)
    for cname, cls in mirror.items():
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), txt, flags=re.S).group(1)
        declared = sum(len(stmt.split(",")) for stmt in body.split(";") if stmt.strip())
        assert declared == len(cls._fields_), (cname, declared, len(cls._fields_))


def test_comm_entry_points_fail_loudly_without_a_device(engine):
    """The communicator needs its MI355X: on a GPU-less box every device-touching call reports MCP_ERR_DEVICE, the
    argument checks still run first, and the host-only sharding rule works."""
    with pytest.raises(S.EngineError) as e:
        engine.comm_init(2, 2, bytes(128))
    assert e.value.status == S.Status.InvalidSpec and "[0, world)" in str(e.value)
    if engine.device_count() == 0:
        for call in (engine.comm_unique_id, lambda: engine.comm_init(0, 1, bytes(128))):
            with pytest.raises(S.EngineError) as e:
                call()
            assert e.value.status == S.Status.Device and "no CPU fallback" in str(e.value)
        with pytest.raises(S.EngineError) as e:
            engine.run_batch(spec_c2a(), [200], 16, BASE_SEED, per_sim=False, allreduce=True)
        assert e.value.status == S.Status.InvalidSpec and "communicator" in str(e.value)
    engine.comm_destroy()  # no-op without a communicator
    assert engine.comm_rank() == -1 and engine.comm_world() == 0
    assert engine.shard_range(10, 0, 3) == (0, 3) and engine.shard_range(10, 2, 3) == (6, 4)
    with pytest.raises(S.EngineError):
        engine.shard_range(10, 3, 3)


def test_unloadable_rccl_is_a_clean_device_error():
    """A host where librccl cannot be loaded must get MCP_ERR_DEVICE and the loader's message from the communicator
    entry points, not a crash (the error string used to be built from a second, NULL-returning dlerror() call).
    Fresh process: the binding is cached per process."""
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['MCP_RCCL_PATH'] = '/nonexistent/librccl.so.1'\n"
            "from mcpower_amd import engine, spec as S\n"
            "try:\n    engine.comm_probe()\nexcept S.EngineError as e:\n"
            "    assert e.status == S.Status.Device, e.status\n    print('MSG', e)\nelse:\n    raise SystemExit('no error')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "RCCL is not loadable" in r.stdout and "/nonexistent/librccl.so.1" in r.stdout and "MCP_RCCL_PATH" in r.stdout


def test_crit_table_matches_oracle(engine, oracle):
    for spec in (spec_c1(), spec_c2a(), spec_c2b()):
        for corr in S.CorrectionMethod:
            if corr == S.CorrectionMethod.TukeyHsd:
                continue
            spec.correction_method = corr
            spec.report_overall = True
            ns = [50, 100, 1000]
            a, b = engine.build_crit_table(spec, ns), oracle.build_crit_table(spec, ns)
            assert np.allclose(a, b, rtol=1e-13, atol=0, equal_nan=True)


def test_quantiles_match_oracle_and_scipy(engine, oracle):
    from scipy import stats
    for df in (1, 2, 3, 7, 94, 194, 994, 200000):
        assert abs(engine.t_ppf(0.975, df) - oracle.t_ppf(0.975, df)) < 1e-13 * engine.t_ppf(0.975, df)
        if df < 1e5:
            assert abs(engine.t_ppf(0.975, df) - stats.t.ppf(0.975, df)) < 1e-9 * stats.t.ppf(0.975, df)
    assert abs(engine.norm_ppf(0.975) - oracle.norm_ppf(0.975)) < 1e-15
    assert abs(engine.chi2_ppf(0.95, 3) - stats.chi2.ppf(0.95, 3)) < 1e-8
    assert abs(engine.f_ppf(0.95, 5, 194) - stats.f.ppf(0.95, 5, 194)) < 1e-8


@pytest.mark.parametrize("ns,msg", [([200, 100], "strictly increasing"), ([], "non-empty"), ([0, 5], "must be > 0"),
                                    ([6], "df = N - P")])
def test_validation_errors_match_reference_wording(engine, ns, msg):
    with pytest.raises(S.EngineError) as e:
        engine.run_batch(spec_c2b(), ns, 4, BASE_SEED)
    assert e.value.status == S.Status.InvalidSpec and msg in str(e.value)


def test_not_psd_and_range_errors(engine):
    bad = spec_c2b()
    c = np.eye(5)
    c[0, 1] = c[1, 0] = 1.5
    bad.correlation = c.flatten().tolist()
    with pytest.raises(S.EngineError) as e:
        engine.run_batch(bad, [100], 4, BASE_SEED)
    assert e.value.status == S.Status.CorrelationNotPSD
    bad = spec_c2b()
    bad.target_indices = [1, 9]
    with pytest.raises(S.EngineError) as e:
        engine.run_batch(bad, [100], 4, BASE_SEED)
    assert e.value.status == S.Status.InvalidSpec and "out of range" in str(e.value)


def test_unsupported_fails_loudly_never_falls_back(engine):
    s = spec_c2b()
    s.estimator = S.EstimatorSpec.Mle  # Mle without cluster -> InvalidSpec like the reference (batch.rs:221)
    with pytest.raises(S.EngineError) as e:
        engine.run_batch(s, [100], 4, BASE_SEED)
    assert e.value.status == S.Status.InvalidSpec
    from helpers import base_spec
    s = base_spec(16)  # 17 design columns: one more than the widest kernel carries
    with pytest.raises(S.EngineError) as e:
        engine.run_batch(s, [100], 4, BASE_SEED)
    assert e.value.status == S.Status.Unsupported and "design columns" in str(e.value)
    s = spec_c2b()
    s.estimator = S.EstimatorSpec.Mle     # two random slopes run on the device up to 8 design columns (at most one slope: 12); this one has 9
    s.n_factor_dummies, s.factor_n_levels, s.factor_proportions = 3, [4], [0.25, 0.25, 0.25, 0.25]
    s.effect_sizes = s.effect_sizes + [0.1, 0.1, 0.1]
    s.cluster = S.ClusterSpec(sizing=("FixedClusters", 10), tau_squared=0.2,
                              slopes=[{"column": 0, "variance": 0.1, "corr_with_intercept": 0.0, "corr_with": [0.0]},
                                      {"column": 1, "variance": 0.1, "corr_with_intercept": 0.0, "corr_with": [0.0]}])
    s.cluster_slope_design_cols = [1, 2]
    with pytest.raises(S.EngineError) as e:
        engine.run_batch(s, [100], 4, BASE_SEED)
    assert e.value.status == S.Status.Unsupported and "more than 8 design columns" in str(e.value)


def test_product_does_not_reference_the_oracle():
    """The product path may never import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "mcpower_amd")
    for dirpath, _, files in os.walk(pkg):
        if "_build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "mcpo_" not in txt and "libmcp_oracle" not in txt, f
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, re.M), f


def test_design_capture_refuses_structures_it_does_not_draw(engine):
    """mcp_generate_design twins capture_sim0_data including random slopes and extra groupings (tests/test_gpu_goldens.py);
    what it would silently leave out of the outcome -- cluster-level (between) predictors -- is refused (host-side
    check, no device needed)."""
    from helpers import load_case
    spec = load_case("lmm_slope_a")[0]
    spec.between_var_indices = [1]
    with pytest.raises(S.EngineError) as e:
        engine.generate_design(spec, 100, 1, BASE_SEED)
    assert e.value.status == S.Status.Unsupported and "design capture" in str(e.value)


def _accepted(engine, spec, ns):
    """The host lowering takes the spec: on a box without a device the call then stops at the device check."""
    try:
        engine.run_batch(spec, ns, 4, BASE_SEED)
    except S.EngineError as e:
        assert e.status == S.Status.Device, str(e)


def _refused(engine, spec, ns, needle):
    with pytest.raises(S.EngineError) as e:
        engine.run_batch(spec, ns, 4, BASE_SEED)
    assert e.value.status == S.Status.Unsupported and needle in str(e.value), str(e.value)


def test_structural_limits_of_the_device_path(engine):
    """DESIGN.md section 8, "limits", as the host lowering enforces them (no device needed): what round 3 widened is accepted,
    the first size past each limit is refused with a message naming it."""
    import copy
    from helpers import base_spec, preset
    # general data path: 9 continuous predictors
    s = base_spec(9)
    s.scenario = preset("realistic")
    _accepted(engine, s, [120])
    s = base_spec(10)
    s.scenario = preset("realistic")
    _refused(engine, s, [120], "9 continuous predictors")
    # sample sizes per call
    _accepted(engine, spec_c2b(), list(range(30, 30 + 256)))
    _refused(engine, spec_c2b(), list(range(30, 30 + 257)), "256 sample sizes")
    # Mle: 12 design columns with at most one random slope on the primary grouping, 12 tests there; 8 otherwise
    m = base_spec(11)
    m.estimator = S.EstimatorSpec.Mle
    m.target_indices = list(range(1, 12))
    m.cluster = S.ClusterSpec(sizing=("FixedClusters", 10), tau_squared=0.2)
    _accepted(engine, m, [200])
    one = copy.deepcopy(m)
    one.cluster = S.ClusterSpec(sizing=("FixedClusters", 10), tau_squared=0.2,
                                slopes=[{"column": 0, "variance": 0.1, "corr_with_intercept": 0.0, "corr_with": []}])
    one.cluster_slope_design_cols = [1]
    _accepted(engine, one, [200])
    wide = base_spec(12)
    wide.estimator = S.EstimatorSpec.Mle
    wide.cluster = S.ClusterSpec(sizing=("FixedClusters", 10), tau_squared=0.2)
    _refused(engine, wide, [200], "design columns")
    # clustered GLM stays at 8 design columns
    g = base_spec(8)
    g.estimator, g.outcome_kind = S.EstimatorSpec.Glm, S.OutcomeKind.Binary
    g.cluster = S.ClusterSpec(sizing=("FixedClusters", 10), tau_squared=0.3)
    _refused(engine, g, [200], "more than 8 design columns")
    # OLS on clustered data with cluster-level predictors: accepted up to 96 cluster x column values under FixedClusters sizing
    o = base_spec(3)
    o.between_var_indices = [2]
    o.cluster = S.ClusterSpec(sizing=("FixedClusters", 40), tau_squared=0.2)
    _accepted(engine, o, [400])
    o.cluster = S.ClusterSpec(sizing=("FixedClusters", 120), tau_squared=0.2)
    _refused(engine, o, [480], "cluster-level")
    o.cluster = S.ClusterSpec(sizing=("FixedSize", 4), tau_squared=0.2)   # contiguous clusters: no limit
    _accepted(engine, o, [480])
