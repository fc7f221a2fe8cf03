"""GPU parity for the uploaded-data paths: NORTA empirical-quantile marginals and strict
bootstrap row copies, device vs oracle on the same seeds (the oracle carries the reference's own
test_upload_*.rs properties, tests/test_oracle_upload.py)."""
import numpy as np
import pytest

from mcpower_amd import spec as S
from test_gpu_parity import _assert_batch_equal, _design_mismatch
from test_oracle_upload import nonlinear_frame, norta_spec, std_normal_quantile_table, strict_spec

pytestmark = pytest.mark.gpu


def test_norta_marginals_match_oracle(engine, oracle):
    table = std_normal_quantile_table(200)
    up = [0.0] * 600
    up[0::3] = sorted(np.random.default_rng(1).gamma(2.0, 1.0, 200).tolist())
    up[2::3] = table
    spec = norta_spec(3, up, [S.Distribution.UploadedData, S.Distribution.UploadedBinary, S.Distribution.UploadedData],
                      [0.0, 0.3, 0.0], [1.0, 0.3, 0.5, 0.3, 1.0, 0.2, 0.5, 0.2, 1.0])
    spec.effect_sizes = [0.0, 0.2, 0.4, 0.3]
    spec.target_indices = [1, 2, 3]
    bx, by, worst, total = _design_mismatch(engine, oracle, spec, 150, 24)
    assert bx + by <= max(2, total * 1e-5) and worst < 2.5e-7, (bx, by, worst)
    d, _ = engine.generate_design(spec, 150, 4, 7, 0)
    assert d[..., 1].min() >= up[0] - 1e-6 and d[..., 1].max() <= max(up[0::3]) + 1e-6
    assert set(np.unique(d[..., 2])) <= {0.0, 1.0}
    g = engine.run_batch(spec, [60, 150], 3000, 7, stat_capture=True)
    o = oracle.run_batch(spec, [60, 150], 3000, 7, stat_capture=True, n_threads=8)
    for f in ("uncorrected", "corrected", "converged"):
        assert (getattr(g, f) != getattr(o, f)).mean() < 5e-4, f


def test_strict_bootstrap_rows_match_oracle_bit_for_bit(engine, oracle):
    frame = nonlinear_frame()
    spec = strict_spec(frame)
    spec.effect_sizes = [0.0, 0.5, 0.3]
    spec.target_indices = [1, 2]
    bx, by, worst, total = _design_mismatch(engine, oracle, spec, 200, 32)
    assert bx == 0 and by == 0                                   # copies + f64 sums only: exact
    d, _ = engine.generate_design(spec, 200, 8, 3, 0)
    assert np.allclose(d[..., 2], d[..., 1].astype(np.float64) ** 2, atol=1e-5)   # whole rows, never mixed
    g = engine.run_batch(spec, [50, 200], 4000, 3, stat_capture=True)
    o = oracle.run_batch(spec, [50, 200], 4000, 3, stat_capture=True, n_threads=8)
    _assert_batch_equal(g, o)


def test_strict_bootstrap_with_factor_binary_and_synthetic_columns(engine, oracle):
    p = 0.25
    frame = [[-1.5 + 3.0 * i / 59.0, (1.0 if i % 4 == 0 else 0.0) - p, float(i % 3)] for i in range(60)]
    spec = S.SimulationSpec(
        n_non_factor=3, n_factor_dummies=2, correlation=np.eye(3).ravel().tolist(),
        var_types=[S.Distribution.UploadedData, S.Distribution.UploadedBinary, S.Distribution.Normal],
        var_params=[0.0, p, 0.0], factor_n_levels=[3], factor_proportions=[0.4, 0.3, 0.3],
        upload_data=[v for r in frame for v in r], upload_data_shape=(60, 3), bootstrap_frame_map=[0, 1, None, 2],
        interactions=[[1, 4]], effect_sizes=[0.0, 0.4, 0.3, 0.2, 0.3, -0.2, 0.1], target_indices=[1, 2, 3, 4, 5, 6],
        factor_min_level_count=5, report_overall=True, correction_method=S.CorrectionMethod.Holm,
        max_failed_fraction=1.0)
    bx, by, worst, total = _design_mismatch(engine, oracle, spec, 120, 16)
    assert bx == 0 and by == 0
    g = engine.run_batch(spec, [24, 60, 120], 3000, 9, stat_capture=True)
    o = oracle.run_batch(spec, [24, 60, 120], 3000, 9, stat_capture=True, n_threads=8)
    _assert_batch_equal(g, o)
    assert np.array_equal(g.factor_excluded, o.factor_excluded)
    assert 0 < g.factor_excluded[:, 0, 0].sum() < 3000          # bootstrapped level counts are per replicate


def test_bootstrap_under_glm_and_scenario(engine, oracle):
    from helpers import preset
    frame = nonlinear_frame()
    spec = strict_spec(frame)
    spec.effect_sizes = [-0.5, 0.6, 0.2]
    spec.target_indices = [1, 2]
    spec.outcome_kind, spec.estimator = S.OutcomeKind.Binary, S.EstimatorSpec.Glm
    spec.scenario = preset("realistic")
    bx, by, worst, total = _design_mismatch(engine, oracle, spec, 150, 16)
    assert bx == 0 and by == 0
    g = engine.run_batch(spec, [150], 2000, 13)
    o = oracle.run_batch(spec, [150], 2000, 13, n_threads=8)
    assert np.array_equal(g.uncorrected, o.uncorrected) and np.array_equal(g.converged, o.converged)


def test_uploaded_frames_under_the_mixed_model_kernels(engine, oracle):
    """NORTA marginals and strict bootstrap rows feed the general LMM and the clustered GLM through the same row
    builder as every other estimator; the bootstrap uniform follows ALL random-effect draws on the sequential stream."""
    from helpers import load_case
    from test_oracle_upload import std_normal_quantile_table, nonlinear_frame
    # random slope + crossed grouping, x1 from an uploaded (NORTA) column
    spec = load_case("lmm_slope_crossed")[0]
    table = [v * 1.0 for v in std_normal_quantile_table(120)]
    spec.var_types = [S.Distribution.UploadedData]
    spec.upload_normal, spec.upload_normal_shape = table, (120, 1)
    spec.scenario.truth_start = True
    dev = engine.run_batch(spec, [480], 32, 2137)
    ora = oracle.run_batch(spec, [480], 32, 2137, n_threads=8)
    same = (dev.uncorrected == ora.uncorrected).all(axis=(1, 2)) & (dev.converged == ora.converged).all(axis=1)
    assert same.sum() >= 31 and dev.converged.mean() > 0.9
    # clustered logistic on strict-bootstrap rows of a two-column frame
    spec = load_case("glmm_intercept")[0]
    frame = nonlinear_frame()
    spec.n_non_factor = 2
    spec.correlation = [1.0, 0.0, 0.0, 1.0]
    spec.var_types, spec.var_params = [S.Distribution.UploadedData] * 2, [0.0, 0.0]
    spec.upload_data, spec.upload_data_shape = [v for r in frame for v in r], (len(frame), 2)
    spec.bootstrap_frame_map = [0, 1]
    spec.effect_sizes = [-0.85, 0.5, 0.1]
    spec.target_indices = [1, 2]
    spec.scenario.truth_start = True
    dev = engine.run_batch(spec, [600], 32, 2137)
    ora = oracle.run_batch(spec, [600], 32, 2137, n_threads=8)
    same = (dev.uncorrected == ora.uncorrected).all(axis=(1, 2)) & (dev.converged == ora.converged).all(axis=1)
    assert same.sum() >= 31 and dev.converged.mean() > 0.9
