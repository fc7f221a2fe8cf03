"""ols_wide_kernel (mcpower_amd/csrc/ols_wide_kernels.hip): OLS on 13 .. 16 design columns, one sample size per call --
four waves share 64 replicates, every accumulator in an architectural register.  The reference bench's `ols_wide`
(crates/engine-core/src/bin/throughput.rs:274-277: base_spec(15), P = 16) is its headline shape.

Device vs the CPU oracle on the same seeds: identical decisions / convergence / overall flags and counters, squared
statistics to 1e-10 (the factor is formed by the same operation sequence; beta and bxty cross the role boundaries in a
different summation order).  A grid of sample sizes, post-hoc blocks, exclusion or a scenario stay on the run-time-sized
reduced-design kernel: the routing is checked by feeding both paths the same spec."""
import numpy as np
import pytest

from helpers import BASE_SEED, base_spec, spec_ols_wide
from mcpower_amd import spec as S
from test_gpu_parity import _assert_batch_equal

pytestmark = pytest.mark.gpu


def _both(engine, oracle, spec, ns, n_sims, seed, **kw):
    """Device and oracle on the same seeds; the squared statistics are compared here -- 1e-10 relative with a 1e-12
    absolute floor: beta crosses the role boundaries in a different summation order than the oracle's single chain, so a
    statistic that is itself a cancellation (beta-hat ~ 1e-6 of its terms, t^2 ~ 1e-9) carries that in its last digits;
    thresholds are ~ 4 -- and dropped, so that _assert_batch_equal checks flags and counters exactly."""
    g = engine.run_batch(spec, ns, n_sims, seed, stat_capture=True, **kw)
    o = oracle.run_batch(spec, ns, n_sims, seed, stat_capture=True, n_threads=8, sim_offset=kw.get("sim_offset", 0))
    a, b = g.stat_capture, o.stat_capture
    assert np.array_equal(np.isnan(a), np.isnan(b))
    m = ~np.isnan(b)
    assert np.all(np.abs(a[m] - b[m]) <= kw.get("rtol", 1e-10) * np.abs(b[m]) + 1e-12)
    g.stat_capture = o.stat_capture = None
    return g, o


def _continuous(k, corr=True):
    spec = base_spec(k)
    c = np.eye(k)
    if corr:
        c[0, 1] = c[1, 0] = 0.3
        c[k - 8, k - 1] = c[k - 1, k - 8] = -0.25
        c[3, 4] = c[4, 3] = 0.45
    spec.correlation = c.flatten().tolist()
    eff = [0.18, 0.18, 0.0, 0.1, -0.1, 0.0, 0.05, 0.2, 0.0, -0.15, 0.12, 0.0, 0.08, 0.0, 0.22]
    spec.effect_sizes = [0.0] + eff[:k]
    return spec


@pytest.mark.parametrize("k", [12, 13, 14, 15])
def test_wide_continuous_designs_all_targets_contrast_bh_overall(k, engine, oracle):
    spec = _continuous(k)
    spec.target_indices = list(range(1, k + 1))[:15]
    spec.contrast_pairs = [(1, 2)] if k < 15 else [(2, 1)]
    spec.correction_method = S.CorrectionMethod.BenjaminiHochberg
    spec.report_overall = True
    assert spec.n_predictors == k + 1
    for n in (200, 41, 3 * k):   # ragged last quads
        g, o = _both(engine, oracle, spec, [n], 2500, 77 + n)
        _assert_batch_equal(g, o)
    assert g.shape.n_targets == min(k, 15) + 1
    # df = 2: the normal equations' conditioning shows in t^2 (3e-8 at df = 1 on the narrow kernels too, DESIGN.md
    # section 4); decisions, convergence and counters still agree
    g = engine.run_batch(spec, [k + 3], 2500, 5, stat_capture=True)
    o = oracle.run_batch(spec, [k + 3], 2500, 5, stat_capture=True, n_threads=8)
    m = ~np.isnan(o.stat_capture)
    assert np.array_equal(np.isnan(g.stat_capture), np.isnan(o.stat_capture))
    assert np.all(np.abs(g.stat_capture[m] - o.stat_capture[m]) <= 1e-6 * np.abs(o.stat_capture[m]) + 1e-12)
    g.stat_capture = o.stat_capture = None
    _assert_batch_equal(g, o)


@pytest.mark.parametrize("corr", [S.CorrectionMethod.None_, S.CorrectionMethod.Bonferroni, S.CorrectionMethod.Holm])
def test_ols_wide_of_the_reference_bench(corr, engine, oracle):
    spec = spec_ols_wide()
    spec.correction_method = corr
    g, o = _both(engine, oracle, spec, [200], 6000, BASE_SEED)
    _assert_batch_equal(g, o)
    # counters-only mode == fold of the per-sim tensors; a shard with an offset == the same slice of the whole
    c = engine.run_batch(spec, [200], 6000, BASE_SEED, per_sim=False)
    assert np.array_equal(c.counters, g.counters)
    part = engine.run_batch(spec, [200], 1000, BASE_SEED, sim_offset=2500)
    assert np.array_equal(part.uncorrected, g.uncorrected[2500:3500]) and np.array_equal(part.converged, g.converged[2500:3500])


def test_wide_design_with_factors_interactions_contrasts_holm(engine, oracle):
    """P = 16 = 1 + 6 continuous (correlated) + 2 + 3 dummies (a fixed-allocation and a sampled factor) + 4 interactions."""
    spec = S.SimulationSpec(
        n_non_factor=6, n_factor_dummies=5,
        correlation=(np.eye(6) + 0.2 * (np.ones((6, 6)) - np.eye(6))).ravel().tolist(),
        var_types=[S.Distribution.Normal] * 6, var_params=[0.0] * 6, factor_n_levels=[3, 4],
        factor_proportions=[0.5, 0.3, 0.2, 0.4, 0.3, 0.2, 0.1], factor_sampled=[None, True],
        interactions=[[1, 7], [1, 8], [2, 3], [4, 9]],
        effect_sizes=[0.0, 0.3, 0.2, 0.1, 0.0, -0.2, 0.15, 0.4, -0.3, 0.2, 0.3, 0.5, 0.15, -0.1, 0.2, 0.1],
        target_indices=[1, 2, 3, 7, 8, 9, 12, 13, 14, 15], contrast_pairs=[(7, 8), (10, 9), (11, 9)],
        correction_method=S.CorrectionMethod.Holm, report_overall=True)
    assert spec.n_predictors == 16
    for n in (300, 90):
        g, o = _both(engine, oracle, spec, [n], 2500, 62)
        _assert_batch_equal(g, o)
    assert g.shape.n_targets == 13 and 0.05 < g.uncorrected[:, 0, 0].mean() <= 1.0
    # P = 13 with one factor
    spec13 = S.SimulationSpec(
        n_non_factor=9, n_factor_dummies=3, correlation=np.eye(9).ravel().tolist(), var_types=[S.Distribution.Normal] * 9,
        var_params=[0.0] * 9, factor_n_levels=[4], factor_proportions=[0.4, 0.3, 0.2, 0.1],
        effect_sizes=[0.1, 0.3, 0.2, 0.0, 0.1, 0.25, 0.0, 0.0, -0.2, 0.15, 0.4, -0.3, 0.2],
        target_indices=[1, 2, 5, 10, 11, 12], contrast_pairs=[(10, 11)], correction_method=S.CorrectionMethod.Holm)
    assert spec13.n_predictors == 13
    g, o = _both(engine, oracle, spec13, [200], 3000, 63)
    _assert_batch_equal(g, o)


@pytest.mark.parametrize("k", [12, 13, 15])
def test_a_grid_of_sample_sizes_runs_as_the_split_wide_launch(k, engine, oracle):
    """Since round 3 a grid of sample sizes stays on ols_wide_kernel: the row kernel parks every role's band of the triangle
    in HBM at each grid point (ragged sizes: the boundary falls inside a row quad) and ols_wide_fit_rec_kernel runs the fits."""
    spec = _continuous(k)
    spec.target_indices = list(range(1, k + 1))[:15]
    spec.correction_method = S.CorrectionMethod.Holm
    spec.report_overall = True
    ns = [3 * k, 41 + k, 100, 203, 400]
    g, o = _both(engine, oracle, spec, ns, 1500, 311)
    _assert_batch_equal(g, o)
    c = engine.run_batch(spec, ns, 1500, 311, per_sim=False)
    assert np.array_equal(c.counters, g.counters)
    one = engine.run_batch(spec, [203], 1500, 311)          # the single-size launch of the same kernel family
    assert np.array_equal(one.uncorrected[:, 0], g.uncorrected[:, 3]) and np.array_equal(one.converged[:, 0], g.converged[:, 3])


def test_single_size_call_equals_the_same_point_of_a_grid_call(engine, oracle):
    """One sample size runs ols_wide_kernel fused, a grid its split launch: the shared point must agree with the
    oracle on both (and so with each other) -- the prefix property of the data plane (rng.rs:1-13)."""
    spec = _continuous(15)
    spec.target_indices = list(range(1, 9))
    single = engine.run_batch(spec, [200], 2000, 91, stat_capture=True)
    grid = engine.run_batch(spec, [200, 400], 2000, 91, stat_capture=True)
    o = oracle.run_batch(spec, [200], 2000, 91, stat_capture=True, n_threads=8)
    assert np.array_equal(single.uncorrected, o.uncorrected) and np.array_equal(grid.uncorrected[:, :1], o.uncorrected)
    m = ~np.isnan(o.stat_capture)
    assert np.all(np.abs(single.stat_capture[m] - o.stat_capture[m]) <= 1e-10 * np.abs(o.stat_capture[m]) + 1e-12)
    assert np.array_equal(np.isnan(single.stat_capture), np.isnan(o.stat_capture))


def test_rank_deficient_wide_design_is_flagged_not_converged(engine, oracle):
    """Two perfectly correlated predictors in the LAST role's rows and in the first's: min |L_ii| / max |L_ii| < 1e-12
    must reach the deciding role through the chain (batch.rs:40-41)."""
    for a, b in ((13, 14), (0, 1), (9, 12)):
        spec = _continuous(15, corr=False)
        c = np.eye(15)
        c[a, b] = c[b, a] = 1.0
        spec.correlation = c.flatten().tolist()
        try:
            g = engine.run_batch(spec, [200], 500, 5)
        except S.EngineError as e:   # the host's trial Cholesky may refuse the matrix first (batch.rs:320-330)
            assert e.status in (S.Status.CorrelationNotPSD, S.Status.InvalidSpec)
            continue
        o = oracle.run_batch(spec, [200], 500, 5, n_threads=4)
        assert np.array_equal(g.converged, o.converged) and np.array_equal(g.uncorrected, o.uncorrected)
