"""Oracle: uploaded-data paths -- NORTA empirical-quantile marginals and strict bootstrap row copies.
Cases follow the reference's own tests (crates/engine-core/tests/test_upload_norta.rs B2 / B3,
tests/test_upload_bootstrap.rs) and contract_adapter.rs:2024-2200 for the frame lowering."""
import math

import msgpack
import numpy as np
import pytest
from scipy import stats

from mcpower_amd import contract as K, spec as S


def norta_spec(n_nf, upload_normal, var_types, var_params, corr, **kw):
    """test_upload_norta.rs:82-140 `uploaded_data_spec`."""
    u = len(upload_normal) // max(n_nf, 1)
    s = S.SimulationSpec(n_non_factor=n_nf, correlation=list(corr), var_types=list(var_types),
                         var_params=list(var_params), upload_normal=list(upload_normal),
                         upload_normal_shape=(u, n_nf), effect_sizes=[0.0] + [0.3] * n_nf,
                         target_indices=[1], max_failed_fraction=1.0)
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def std_normal_quantile_table(u):
    return [float(stats.norm.ppf((i + 0.5) / u)) for i in range(u)]


def nonlinear_frame():
    """test_upload_bootstrap.rs:93-100: x1 ramps over [-2, 2], x2 = x1^2."""
    x1 = [-2.0 + 4.0 * i / 39.0 for i in range(40)]
    return [[a, a * a] for a in x1]


def strict_spec(frame, var_types=None, var_params=None, **kw):
    ncols = len(frame[0])
    s = S.SimulationSpec(n_non_factor=ncols, correlation=np.eye(ncols).ravel().tolist(),
                         var_types=var_types or [S.Distribution.UploadedData] * ncols,
                         var_params=var_params or [0.0] * ncols,
                         upload_data=[v for row in frame for v in row], upload_data_shape=(len(frame), ncols),
                         bootstrap_frame_map=list(range(ncols)), effect_sizes=[0.0, 0.5, 0.3][:ncols + 1],
                         target_indices=[1], max_failed_fraction=1.0)
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def test_uploaded_data_moments_and_bounds(oracle):
    table = std_normal_quantile_table(200)
    spec = norta_spec(1, table, [S.Distribution.UploadedData], [0.0], [1.0])
    x, _, _ = oracle.generate_sim_data(spec, 2137, 0, 40000)
    col = x[:, 1].astype(np.float64)
    assert abs(col.mean()) < 0.05 and abs(col.var() - 1.0) < 0.05
    assert col.min() >= table[0] - 1e-6 and col.max() <= table[-1] + 1e-6


def test_uploaded_data_bounded_by_table_and_type7_interpolation(oracle):
    table = [1.0 + 4.0 * i / 49.0 for i in range(50)]
    spec = norta_spec(1, table, [S.Distribution.UploadedData], [0.0], [1.0])
    x, _, _ = oracle.generate_sim_data(spec, 2137, 0, 1000)
    col = x[:, 1].astype(np.float64)
    assert col.min() >= 1.0 - 1e-6 and col.max() <= 5.0 + 1e-6
    # the same latent normals through the identity marginal: value = quantile_7(table, Phi(z))
    z, _, _ = oracle.generate_sim_data(norta_spec(1, [], [S.Distribution.Normal], [0.0], [1.0],
                                                  upload_normal_shape=(0, 0)), 2137, 0, 1000)
    u = np.array([oracle.phi(float(v)) for v in z[:, 1]])
    want = np.quantile(np.array(table), np.clip(u, 0, 1), method="linear")
    assert np.allclose(col, want.astype(np.float32), rtol=2e-7, atol=0)


def test_uploaded_binary_proportion_and_attenuated_correlation(oracle):
    spec = norta_spec(1, [0.0], [S.Distribution.UploadedBinary], [0.3], [1.0])
    x, _, _ = oracle.generate_sim_data(spec, 2137, 0, 4000)
    col = x[:, 1]
    assert set(np.unique(col)) <= {0.0, 1.0} and abs(col.mean() - 0.3) < 0.03
    table = std_normal_quantile_table(200)
    up = [0.0] * 400
    up[1::2] = table
    spec = norta_spec(2, up, [S.Distribution.UploadedBinary, S.Distribution.UploadedData], [0.3, 0.0],
                      [1.0, 0.6, 0.6, 1.0])
    x, _, _ = oracle.generate_sim_data(spec, 2137, 0, 8000)
    r = np.corrcoef(x[:, 1], x[:, 2])[0, 1]
    assert 0.0 < r < 0.6                              # no polychoric correction: attenuated, still positive


def test_strict_bootstrap_copies_whole_rows_and_is_row_stable(oracle):
    frame = nonlinear_frame()
    spec = strict_spec(frame)
    x, _, _ = oracle.generate_sim_data(spec, 0, 2137, 500)
    f32 = np.array(frame, dtype=np.float32)
    for i in range(500):
        assert (np.abs(f32 - x[i, 1:3]).max(axis=1) < 1e-6).any(), i      # a whole uploaded row, never a mix
    assert np.allclose(x[:, 2], x[:, 1].astype(np.float64) ** 2, atol=1e-5)
    short, _, _ = oracle.generate_sim_data(spec, 0, 2137, 120)
    assert np.array_equal(short, x[:120])                                   # prefix-nested across max_n
    assert len({tuple(r) for r in x[:, 1:3].tolist()}) > 30               # and it does resample


def test_strict_bootstrap_factor_reexpands_level(oracle):
    """test_upload_bootstrap.rs:230-300: column 1 of the frame holds a level code 0..2 tied to x1's sign."""
    frame = [[-1.0 + 2.0 * i / 29.0, float(i % 3)] for i in range(30)]
    spec = S.SimulationSpec(n_non_factor=1, n_factor_dummies=2, correlation=[1.0],
                            var_types=[S.Distribution.UploadedData], var_params=[0.0], factor_n_levels=[3],
                            factor_proportions=[1 / 3, 1 / 3, 1 / 3], upload_data=[v for r in frame for v in r],
                            upload_data_shape=(30, 2), bootstrap_frame_map=[0, 1],
                            effect_sizes=[0.0, 0.4, 0.2, 0.3], target_indices=[1], max_failed_fraction=1.0)
    x, _, _ = oracle.generate_sim_data(spec, 5, 0, 400)
    lut = {np.float32(r[0]): int(r[1]) for r in frame}
    for i in range(400):
        lvl = lut[x[i, 1]]
        assert x[i, 2] == (1.0 if lvl == 1 else 0.0) and x[i, 3] == (1.0 if lvl == 2 else 0.0)


def test_strict_bootstrap_binary_recovers_zero_one_jointly(oracle):
    """stored binary = x - p (test_upload_bootstrap.rs:308-420): recovered as 1 when stored + p >= 0.5."""
    p = 0.25
    frame = [[-1.5 + 3.0 * i / 39.0, (1.0 if i >= 30 else 0.0) - p] for i in range(40)]
    spec = strict_spec(frame, var_types=[S.Distribution.UploadedData, S.Distribution.UploadedBinary],
                       var_params=[0.0, p])
    x, _, _ = oracle.generate_sim_data(spec, 9, 1, 600)
    assert set(np.unique(x[:, 2])) <= {0.0, 1.0}
    thr = np.float32(frame[30][0])
    assert np.array_equal(x[:, 2] == 1.0, x[:, 1] >= thr)     # the joint of the frame survives the copy


def test_partial_bootstrap_keeps_synthetic_columns_and_runs_a_batch(oracle):
    frame = nonlinear_frame()
    spec = S.SimulationSpec(n_non_factor=3, correlation=np.eye(3).ravel().tolist(),
                            var_types=[S.Distribution.UploadedData, S.Distribution.Normal, S.Distribution.UploadedData],
                            var_params=[0.0] * 3, upload_data=[v for r in frame for v in r],
                            upload_data_shape=(40, 2), bootstrap_frame_map=[0, None, 1],
                            effect_sizes=[0.0, 0.4, 0.3, 0.2], target_indices=[1, 2, 3], max_failed_fraction=1.0)
    x, _, _ = oracle.generate_sim_data(spec, 3, 2, 300)
    ref, _, _ = oracle.generate_sim_data(S.SimulationSpec(**{**spec.__dict__, "var_types": [S.Distribution.Normal] * 3,
                                                             "upload_data": [], "upload_data_shape": (0, 0),
                                                             "bootstrap_frame_map": []}), 3, 2, 300)
    assert np.array_equal(x[:, 2], ref[:, 2])                # the synthetic column keeps its planar draws
    assert np.allclose(x[:, 3], x[:, 1].astype(np.float64) ** 2, atol=1e-5)
    r = oracle.run_batch(spec, [60, 150], 300, 3)
    assert r.converged.all() and r.uncorrected[:, 1, 1].mean() > 0.5


def test_contract_frames_lower_to_norta_tables_and_bootstrap_maps():
    """contract_adapter.rs:2024-2200: NORTA sorts each referenced frame column ascending into upload_normal
    (Synthetic columns stay zero); strict mode leaves it empty and maps kernel columns to frame columns,
    non-factor first, then factors."""
    from test_find_power import _contract
    c = _contract()
    c["generation"]["columns"] = [{"Resampled": {"frame_column": 1}},
                                  {"Synthetic": {"kind": "Normal", "pinned": False}},
                                  {"ResampledBinary": {"frame_column": 0, "proportion": 0.4}}]
    c["design_generation"] = {"terms": ["Const", {"Direct": {"column": 0}}, {"Direct": {"column": 1}},
                                        {"Direct": {"column": 2}}]}
    c["outcome"]["coefficients"] = [0.0, 0.3, 0.2, 0.1]
    c["test"]["targets"] = [{"kind": "marginal", "term": 1}]
    frame = [[0.6, 2.0], [-0.4, -1.0], [0.6, 0.5], [-0.4, 1.0]]
    c["generation"]["uploaded_frame"] = {"data": [v for r in frame for v in r], "n_rows": 4, "n_cols": 2,
                                         "bootstrap": False}
    s = K.contract_to_simulation_spec(c)
    assert s.var_types == [S.Distribution.UploadedData, S.Distribution.Normal, S.Distribution.UploadedBinary]
    assert s.var_params == [0.0, 0.0, 0.4] and s.upload_normal_shape == (4, 3) and s.bootstrap_frame_map == []
    un = np.array(s.upload_normal).reshape(4, 3)
    assert un[:, 0].tolist() == [-1.0, 0.5, 1.0, 2.0] and not un[:, 1].any()
    assert un[:, 2].tolist() == [-0.4, -0.4, 0.6, 0.6]
    c["generation"]["uploaded_frame"]["bootstrap"] = True
    c["generation"]["columns"].append({"FactorFromFrame": {"frame_column": 1, "n_levels": 3,
                                                           "proportions": [0.5, 0.25, 0.25],
                                                           "sampled_proportions": None}})
    s = K.contract_to_simulation_spec(c)
    assert s.upload_normal == [] and s.upload_normal_shape == (0, 0)
    assert s.bootstrap_frame_map == [1, None, 0, 1] and s.upload_data_shape == (4, 2)
