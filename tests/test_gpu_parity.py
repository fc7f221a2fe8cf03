"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs, against the committed golden fixtures, and -- at BASELINE.json's full sizes --
through size-independent properties.

Bars: f32 data plane bit-exact; squared statistics within 1e-10 relative
(summation order / fma only); significance flags identical except where |t^2 - crit^2| / crit^2
< 1e-9 (a knife-edge decision: expected count 0); counters identical whenever flags are.
"""
import math
import os

import numpy as np
import pytest

from helpers import BASE_SEED, base_spec, case_names, load_case, spec_c1, spec_c2a, spec_c2b
from mcpower_amd import spec as S

pytestmark = pytest.mark.gpu

TSQ_RTOL = 1e-10


def _assert_batch_equal(g, o, crit_unc=None):
    """Flags / counters identical; t^2 within TSQ_RTOL.  Knife-edge replicates (t^2 within 1e-9 of a
    threshold) are the only place a flag may legitimately differ -- none is expected, so assert none."""
    assert g.uncorrected.shape == o.uncorrected.shape
    if g.stat_capture is not None and o.stat_capture is not None:
        a, b = g.stat_capture, o.stat_capture
        assert np.array_equal(np.isnan(a), np.isnan(b))
        m = ~np.isnan(a)
        assert (not m.any()) or np.max(np.abs(a[m] - b[m]) / np.maximum(np.abs(b[m]), 1e-300)) < TSQ_RTOL
    for f in ("uncorrected", "corrected", "converged", "overall"):
        ga, oa = getattr(g, f), getattr(o, f)
        assert (ga is None) == (oa is None), f
        if ga is not None:
            assert np.array_equal(ga, oa), f"{f}: {int((ga != oa).sum())} of {ga.size} flags differ"
    assert np.array_equal(g.counters, o.counters)


# ---------------------------------------------------------------------------------------------
# data plane
# ---------------------------------------------------------------------------------------------
def test_icdf_exhaustive_all_reachable_uniforms(engine, oracle):
    """All 2^23 reachable uniforms (w >> 9 spans 23 bits): device icdf == oracle icdf, bit for bit
    (reference sweeps the same set, rng.rs:539)."""
    n = 1 << 23
    dev = engine.icdf_probe(0, 512, n)
    ref = oracle.icdf_probe(0, 512, n)
    assert np.array_equal(dev.view(np.uint32), ref.view(np.uint32))
    # the low 9 bits of the word must not matter
    assert np.array_equal(engine.icdf_probe(511, 512, 4096).view(np.uint32), ref[:4096].view(np.uint32))


@pytest.mark.parametrize("name", case_names())
def test_generated_design_matches_reference_bytes(engine, oracle, name):
    """mcp_generate_design (introspect::capture_sim0_data twin) == the reference's captured sim-0
    design/outcome bytes (validation/data/<case>.rds: OLS, logit, probit, Poisson and random-intercept LME
    cases), and == the oracle for later sims."""
    spec, seed, arr, low = load_case(name)
    n = low["n"]
    d, y = engine.generate_design(spec, n, 3, seed, 0)
    assert np.array_equal(d[0].view(np.uint32), arr["design"].view(np.uint32))
    assert np.array_equal(y[0].view(np.uint32), arr["outcome"].view(np.uint32))
    for s in (1, 2):
        xo, yo, _ = oracle.generate_sim_data(spec, seed, s, n)
        assert np.array_equal(d[s].view(np.uint32), xo.view(np.uint32))
        assert np.array_equal(y[s].view(np.uint32), yo.view(np.uint32))


def test_generated_rows_are_prefix_stable(engine):
    spec = spec_c2b()
    d200, y200 = engine.generate_design(spec, 200, 5, BASE_SEED, 10)
    d1000, y1000 = engine.generate_design(spec, 1000, 5, BASE_SEED, 10)
    assert np.array_equal(d200.view(np.uint32), d1000[:, :200].view(np.uint32))
    assert np.array_equal(y200.view(np.uint32), y1000[:, :200].view(np.uint32))
    # ragged max_n (not a multiple of the 4-row Philox block)
    d50, _ = engine.generate_design(spec, 50, 5, BASE_SEED, 10)
    assert np.array_equal(d50.view(np.uint32), d1000[:, :50].view(np.uint32))


@pytest.mark.parametrize("name", case_names("ols"))
def test_fit_provided_bytes_matches_r_goldens(engine, name):
    """mcp_fit_design (introspect::fit_provided_data twin) vs R lm on the same bytes: 1e-11 relative
    on beta / se / t (validation/tolerances.R:24-30)."""
    _, _, arr, _ = load_case(name)
    f = engine.fit_design(arr["design"], arr["outcome"])
    assert f["converged"][0]
    np.testing.assert_allclose(f["beta"][0], arr["golden_beta"], rtol=1e-11, atol=0)
    np.testing.assert_allclose(f["se"][0], arr["golden_se"], rtol=1e-11, atol=0)
    np.testing.assert_allclose(np.sign(f["beta"][0]) * np.sqrt(f["t_sq"][0]), arr["golden_stat"], rtol=1e-11, atol=0)


def test_fit_provided_rank_deficient_design_is_flagged(engine):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 40, 3)).astype(np.float32)
    x[0, :, 0] = 1.0
    x[0, :, 2] = x[0, :, 1]  # duplicate column -> Cholesky pivot ratio below EPS_RANK (batch.rs:41)
    y = rng.standard_normal((1, 40)).astype(np.float32)
    f = engine.fit_design(x, y)
    assert not f["converged"][0] and np.isnan(f["beta"][0]).all()


# ---------------------------------------------------------------------------------------------
# run_batch vs oracle
# ---------------------------------------------------------------------------------------------
def _both(engine, oracle, spec, ns, n_sims, **kw):
    g = engine.run_batch(spec, ns, n_sims, BASE_SEED, stat_capture=True, **kw)
    o = oracle.run_batch(spec, ns, n_sims, BASE_SEED, stat_capture=True, n_threads=8,
                         sim_offset=kw.get("sim_offset", 0))
    return g, o


def test_c1_two_group_ttest_reference_cpu_config(engine, oracle):
    """BASELINE config 0: n=100, p=2, 1000 sims (deterministic X, fixed allocation)."""
    g, o = _both(engine, oracle, spec_c1(), [100], 1000)
    _assert_batch_equal(g, o)
    assert g.converged.all()


@pytest.mark.parametrize("make", [spec_c2a, spec_c2b])
def test_c2_ols_flags_identical_to_oracle(engine, oracle, make):
    spec = make()
    spec.report_overall = True
    g, o = _both(engine, oracle, spec, [200], 20000)
    _assert_batch_equal(g, o)
    power = g.counter("k_unc", spec.n_targets)[0] / 20000.0
    assert 0.5 < power[0] < 0.9  # x1 = 0.18 at n = 200 is ~70% power


def test_grid_prefix_fits_match_oracle_including_ragged_sizes(engine, oracle):
    spec = spec_c2b()
    spec.report_overall = True
    g, o = _both(engine, oracle, spec, [50, 51, 99, 100, 250, 1000], 3000)
    _assert_batch_equal(g, o)
    ku = g.counter("k_unc", 2)[:, 0]
    assert ku[-1] > ku[0]  # power grows along the grid
    # df = 1 (n = 7, p = 6): X'X is nearly singular, so t^2 carries the conditioning of the normal
    # equations (observed ~3e-8 relative); decisions must still agree
    g, o = engine.run_batch(spec, [7, 9, 200], 3000, BASE_SEED), oracle.run_batch(spec, [7, 9, 200], 3000, BASE_SEED)
    _assert_batch_equal(g, o)


@pytest.mark.parametrize("method", [S.CorrectionMethod.Bonferroni, S.CorrectionMethod.Holm,
                                    S.CorrectionMethod.BenjaminiHochberg])
def test_multiplicity_corrections_match_oracle(engine, oracle, method):
    spec = base_spec(5)
    spec.effect_sizes = [0.0, 0.25, 0.18, 0.12, 0.0, 0.3]
    spec.target_indices = [1, 2, 3, 4, 5]
    spec.contrast_pairs = [(1, 2), (5, 3)]
    spec.correction_method = method
    g, o = _both(engine, oracle, spec, [80, 160], 6000)
    _assert_batch_equal(g, o)
    assert (g.corrected <= g.uncorrected).all()
    assert g.corrected.sum() < g.uncorrected.sum()


def test_correlated_predictors_factors_and_interactions_match_oracle(engine, oracle):
    spec, _, _, low = load_case("ols_cf_a")  # x1 * 3-level g: dummies + 2 interaction columns
    g, o = _both(engine, oracle, spec, [120, 800], 2000)
    _assert_batch_equal(g, o)
    spec, _, _, _ = load_case("ols_interaction_a")  # corr(x1,x2)=0.3 + x1:x2
    spec.report_overall = True
    g, o = _both(engine, oracle, spec, [60, 600], 2000)
    _assert_batch_equal(g, o)
    spec, _, _, _ = load_case("ols_ff_a")  # two factors + dummy:dummy interaction
    g, o = _both(engine, oracle, spec, [800], 1000)
    _assert_batch_equal(g, o)


def test_sampled_factor_proportions_use_the_sequential_stream(engine, oracle):
    spec, seed, _, low = load_case("ols_factor_a")
    spec.scenario = S.ScenarioPerturbations(name="custom", sampled_factor_proportions=True)
    d, y = engine.generate_design(spec, 120, 2, seed, 0)
    for s in range(2):
        xo, yo, _ = oracle.generate_sim_data(spec, seed, s, 120)
        assert np.array_equal(d[s].view(np.uint32), xo.view(np.uint32))
        assert np.array_equal(y[s].view(np.uint32), yo.view(np.uint32))
    g, o = _both(engine, oracle, spec, [60, 120], 1500)
    _assert_batch_equal(g, o)
    # per-factor override beats the scenario default (spec.rs:284-291)
    spec.factor_sampled = [False]
    g2, o2 = _both(engine, oracle, spec, [60, 120], 500)
    _assert_batch_equal(g2, o2)


def test_rank_deficient_replicates_report_not_converged(engine, oracle):
    """A 2-level factor at n=3 with skewed proportions leaves a level empty -> X'X singular ->
    converged = 0 and every flag 0 (spec.rs:560, batch.rs:1015)."""
    spec = S.SimulationSpec(n_non_factor=0, n_factor_dummies=1, correlation=[], factor_n_levels=[2],
                            factor_proportions=[0.9, 0.1], effect_sizes=[0.0, 0.5], target_indices=[1])
    g, o = _both(engine, oracle, spec, [4, 40], 64)
    _assert_batch_equal(g, o)
    assert not g.converged[:, 0].any() and g.converged[:, 1].all()
    assert not g.uncorrected[:, 0].any()


# ---------------------------------------------------------------------------------------------
# contract properties: sharding, chunking, pipelines, progress / cancel
# ---------------------------------------------------------------------------------------------
def test_shard_union_is_bit_identical_to_single_run(engine):
    """GPU g takes sim_id in [g*S/G, (g+1)*S/G) with the same base_seed."""
    spec = spec_c2b()
    whole = engine.run_batch(spec, [100, 200], 5000, BASE_SEED)
    parts = [engine.run_batch(spec, [100, 200], n, BASE_SEED, sim_offset=off)
             for off, n in ((0, 1250), (1250, 1250), (2500, 1251), (3751, 1249))]
    assert np.array_equal(np.concatenate([p.uncorrected for p in parts]), whole.uncorrected)
    assert np.array_equal(sum(p.counters for p in parts), whole.counters)


def test_chunked_launches_equal_single_launch(engine):
    spec = spec_c2a()
    a = engine.run_batch(spec, [200], 10000, BASE_SEED)
    b = engine.run_batch(spec, [200], 10000, BASE_SEED, chunk_sims=777)
    assert b.n_kernel_launches == 13 and a.n_kernel_launches == 1
    assert np.array_equal(a.uncorrected, b.uncorrected) and np.array_equal(a.counters, b.counters)


def test_staged_pipeline_equals_fused_kernel(engine):
    spec = spec_c2b()
    spec.report_overall = True
    f = engine.run_batch(spec, [50, 200], 8192 + 17, BASE_SEED, stat_capture=True)
    s = engine.run_batch(spec, [50, 200], 8192 + 17, BASE_SEED, stat_capture=True, staged=True, chunk_sims=4096)
    assert np.array_equal(f.uncorrected, s.uncorrected) and np.array_equal(f.corrected, s.corrected)
    assert np.array_equal(f.counters, s.counters)
    assert np.array_equal(f.stat_capture, s.stat_capture, equal_nan=True)  # same arithmetic, same order
    assert s.aux_kernel_ms > 0 and s.kernel_ms > 0


def test_counters_only_mode_matches_per_sim_fold(engine):
    spec = spec_c2b()
    full = engine.run_batch(spec, [200], 4000, BASE_SEED)
    red = engine.run_batch(spec, [200], 4000, BASE_SEED, per_sim=False)
    assert red.uncorrected is None and np.array_equal(red.counters, full.counters)
    assert np.array_equal(full.counter("k_unc", 2), full.uncorrected.sum(axis=0))
    assert full.counter("n_sims")[0, 0] == 4000
    assert full.counter("hist_unc", 3)[0].sum() == 4000


def test_batches_in_flight_equal_blocking_calls(engine):
    """mcp_run_batch_submit / _collect: four batches enqueued before the first is collected, across estimators
    and with a marshalled spec reused, give the counters of the blocking call; tickets are single-use."""
    from helpers import spec_c3, spec_c4
    jobs = [(spec_c2b(), [100, 200], 30000, 0), (spec_c2a(), [200], 50000, 7), (spec_c3(), [500], 3000, 11),
            (spec_c4(), [600], 1000, 5)]
    want = [engine.run_batch(sp, ns, n, BASE_SEED, per_sim=False, sim_offset=off).counters for sp, ns, n, off in jobs]
    pending = [engine.submit_batch(sp, ns, n, BASE_SEED, sim_offset=off) for sp, ns, n, off in jobs]
    with pytest.raises(S.EngineError) as e:                      # a fifth batch has no slot until one is collected
        engine.submit_batch(spec_c2b(), [200], 100, BASE_SEED)
    assert "in flight" in str(e.value)
    got = [p.collect() for p in reversed(pending)][::-1]          # collection order is free
    for w, g in zip(want, got):
        assert np.array_equal(w, g.counters) and g.uncorrected is None and g.kernel_ms > 0
    with pytest.raises(RuntimeError):
        pending[0].collect()
    low = spec_c2b().to_c()
    a = engine.submit_batch(low, [200], 20000, BASE_SEED, sim_offset=0)
    b = engine.submit_batch(low, [200], 20000, BASE_SEED, sim_offset=20000)
    whole = engine.run_batch(low, [200], 40000, BASE_SEED, per_sim=False)
    assert np.array_equal(a.collect().counters + b.collect().counters, whole.counters)
    staged = engine.submit_batch(low, [200], 20000, BASE_SEED, staged=True).collect()
    assert np.array_equal(staged.counters, engine.run_batch(low, [200], 20000, BASE_SEED, per_sim=False).counters)


def test_progress_reports_and_cancel(engine):
    spec = spec_c2b()
    seen = []
    engine.run_batch(spec, [200], 100000, BASE_SEED, lambda cur, tot: seen.append((cur, tot)) or True)
    assert seen and seen[-1] == (100000, 100000) and all(t == 100000 for _, t in seen)
    assert [c for c, _ in seen] == sorted(c for c, _ in seen)
    with pytest.raises(S.EngineError) as e:
        engine.run_batch(spec, [200], 100000, BASE_SEED, lambda cur, tot: False)
    assert e.value.status == S.Status.Cancelled


def test_zero_sims_returns_empty_shape(engine):
    r = engine.run_batch(spec_c2b(), [200], 0, BASE_SEED)
    assert r.shape.n_sims == 0 and r.shape.n_targets == 2 and r.counters.sum() == 0


# ---------------------------------------------------------------------------------------------
# BASELINE.json full size (10^6 replicates, n=200): size-independent properties
# ---------------------------------------------------------------------------------------------
def test_full_size_c2b_properties(engine, oracle):
    spec = spec_c2b()
    n = 1_000_000
    a = engine.run_batch(spec, [200], n, BASE_SEED, per_sim=False)
    b = engine.run_batch(spec, [200], n, BASE_SEED, per_sim=False)
    assert np.array_equal(a.counters, b.counters)  # deterministic (integer reductions only)
    halves = [engine.run_batch(spec, [200], n // 2, BASE_SEED, per_sim=False, sim_offset=o) for o in (0, n // 2)]
    assert np.array_equal(halves[0].counters + halves[1].counters, a.counters)  # 1-GPU == 2-shard union
    assert a.counter("n_sims")[0, 0] == n and a.counter("converged")[0, 0] == n
    assert a.counter("hist_unc", 3)[0].sum() == n
    ku = a.counter("k_unc", 2)[0].astype(np.float64)
    hist = a.counter("hist_unc", 3)[0].astype(np.float64)
    assert ku.sum() == hist[1] + 2 * hist[2]  # checksum of checksums
    # the first 20000 replicates of the big run are the oracle's 20000 (power delta exactly 0 there)
    head = engine.run_batch(spec, [200], 20000, BASE_SEED)
    o = oracle.run_batch(spec, [200], 20000, BASE_SEED, n_threads=8)
    assert np.array_equal(head.uncorrected, o.uncorrected)
    # analytic power with a random design: the noncentrality is beta * sqrt(W), W ~ chi2(n - p + 1), so
    # power = E_W[P(|nct(df, beta sqrt(W))| > t_crit)]  (0.7032 here; the fixed-design value would be 0.7167)
    from scipy import integrate, stats
    df = 194
    tc = stats.t.ppf(0.975, df)

    def integrand(w):
        ncp = 0.18 * np.sqrt(w)
        return (1 - stats.nct.cdf(tc, df, ncp) + stats.nct.cdf(-tc, df, ncp)) * stats.chi2.pdf(w, df + 1)

    power = integrate.quad(integrand, 100, 320)[0]
    se = np.sqrt(power * (1 - power) / n)
    assert abs(ku[0] / n - power) < 0.001 + 4 * se  # within the reference's 0.1 pp accuracy claim (README.md:77-83)


def test_type_one_error_under_the_null(engine):
    spec = spec_c2b()
    spec.effect_sizes = [0.0] * 6
    r = engine.run_batch(spec, [200], 400000, BASE_SEED, per_sim=False)
    rate = r.counter("k_unc", 2)[0] / 400000.0
    se = np.sqrt(0.05 * 0.95 / 400000)
    assert np.all(np.abs(rate - 0.05) < 5 * se)


# ---------------------------------------------------------------------------------------------
# logistic GLM (IRLS), BASELINE config C3
# ---------------------------------------------------------------------------------------------
def test_c3_logistic_flags_identical_to_oracle(engine, oracle):
    """C3: logit, n=500, p=4, truth start.  z^2 within 1e-9 relative (the IRLS sums are reduced in a
    different order on the device: lane-strided partials + butterfly), decisions identical."""
    from helpers import spec_c3
    spec = spec_c3()
    spec.report_overall = True
    g = engine.run_batch(spec, [500], 6000, BASE_SEED, stat_capture=True)
    o = oracle.run_batch(spec, [500], 6000, BASE_SEED, stat_capture=True, n_threads=8)
    a, b = g.stat_capture, o.stat_capture
    assert np.array_equal(np.isnan(a), np.isnan(b))
    m = ~np.isnan(a)
    assert np.max(np.abs(a[m] - b[m]) / np.abs(b[m])) < 1e-9
    for f in ("uncorrected", "corrected", "converged", "overall"):
        assert np.array_equal(getattr(g, f), getattr(o, f)), f
    assert np.array_equal(g.counters, o.counters)
    power = g.counter("k_unc", 3)[0] / 6000.0
    assert power[0] > 0.95 and power[1] > 0.95 and abs(power[2] - 0.05) < 0.02  # x3 is a true null


def test_glm_grid_factors_and_corrections_match_oracle(engine, oracle):
    spec, seed, _, low = load_case("glm_factor_a")          # x1 + 3-level factor, baseline 0.3
    spec.correction_method = S.CorrectionMethod.Holm
    g = engine.run_batch(spec, [100, 333, 1000], 1500, seed, stat_capture=True)
    o = oracle.run_batch(spec, [100, 333, 1000], 1500, seed, stat_capture=True, n_threads=8)
    for f in ("uncorrected", "corrected", "converged"):
        assert np.array_equal(getattr(g, f), getattr(o, f)), f
    assert np.array_equal(g.counters, o.counters)
    spec, seed, _, _ = load_case("glm_interaction_b")        # x1 * x2, baseline 0.5
    spec.correction_method = S.CorrectionMethod.BenjaminiHochberg
    g = engine.run_batch(spec, [250, 1000], 1500, seed)
    o = oracle.run_batch(spec, [250, 1000], 1500, seed, n_threads=8)
    assert np.array_equal(g.corrected, o.corrected) and np.array_equal(g.counters, o.counters)


def test_glm_separation_reports_not_converged_like_oracle(engine, oracle):
    """Tiny n with a strong effect: complete separation is common; IRLS must stop at BETA_CAP / the
    pass limit and report converged = 0 with all flags 0 (batch.rs:2033-2067), identically to the oracle."""
    from helpers import spec_c3
    spec = spec_c3()
    spec.effect_sizes = [0.0, 4.0, 0.0, 0.0]
    g = engine.run_batch(spec, [12, 40], 2000, BASE_SEED)
    o = oracle.run_batch(spec, [12, 40], 2000, BASE_SEED, n_threads=8)
    assert np.array_equal(g.converged, o.converged)
    assert 0.02 < 1.0 - g.converged[:, 0].mean() < 0.98          # some, not all, separate at n = 12
    assert not g.uncorrected[g.converged == 0].any()
    assert np.array_equal(g.uncorrected, o.uncorrected)


def test_glm_counters_only_shards_and_determinism(engine):
    from helpers import spec_c3
    spec = spec_c3()
    whole = engine.run_batch(spec, [500], 20000, BASE_SEED, per_sim=False)
    again = engine.run_batch(spec, [500], 20000, BASE_SEED, per_sim=False, chunk_sims=3001)
    assert np.array_equal(whole.counters, again.counters)
    parts = [engine.run_batch(spec, [500], 10000, BASE_SEED, per_sim=False, sim_offset=o) for o in (0, 10000)]
    assert np.array_equal(parts[0].counters + parts[1].counters, whole.counters)
    assert whole.counter("n_sims")[0, 0] == 20000


# ---------------------------------------------------------------------------------------------
# random-intercept LMM (REML), BASELINE config C4
# ---------------------------------------------------------------------------------------------
def _assert_lmm_equal(g, o, rtol=1e-6):
    """The 1-D minimiser stops within THETA_TOL = 1e-9 of the optimum on both sides, but its probe
    sequence may differ once reduction order perturbs deviance values in the last ulp, so z^2 agrees
    to ~1e-7 rather than 1e-10; decisions must still be identical (a flip needs |z^2 - crit| < 1e-6 crit)."""
    a, b = g.stat_capture, o.stat_capture
    assert np.array_equal(np.isnan(a), np.isnan(b))
    m = ~np.isnan(a)
    assert np.max(np.abs(a[m] - b[m]) / np.abs(b[m])) < rtol
    for f in ("uncorrected", "corrected", "converged", "boundary_hit", "joint_unc", "joint_cor", "pinned_components"):
        assert np.array_equal(getattr(g, f), getattr(o, f)), f
    assert np.array_equal(g.counters, o.counters)


def test_c4_random_intercept_lmm_matches_oracle(engine, oracle):
    """C4: 30 clusters x 20 rows, tau^2 = 0.25, truth start, Wald z + joint Wald chi2."""
    from helpers import spec_c4
    spec = spec_c4()
    g = engine.run_batch(spec, [600], 4000, BASE_SEED, stat_capture=True)
    o = oracle.run_batch(spec, [600], 4000, BASE_SEED, stat_capture=True, n_threads=8)
    _assert_lmm_equal(g, o)
    assert g.converged.all() and g.shape.n_variance_components == 1
    power = g.counter("k_unc", 2)[0] / 4000.0
    assert 0.9 < power[0] < 0.99
    assert g.counter("joint_unc")[0, 0] == g.joint_unc.sum()


def test_lmm_design_bytes_and_grid_with_warm_carry(engine, oracle):
    spec, seed, arr, low = load_case("lme_factor_b")   # x1 + 3-level factor, 30 clusters, ICC 0.3
    g = engine.run_batch(spec, [300, 600, 900], 800, seed, stat_capture=True)
    o = oracle.run_batch(spec, [300, 600, 900], 800, seed, stat_capture=True, n_threads=8)
    _assert_lmm_equal(g, o)
    spec.scenario = S.ScenarioPerturbations(name="blind", truth_start=False)  # blind start theta0
    g = engine.run_batch(spec, [300, 900], 500, seed, stat_capture=True)
    o = oracle.run_batch(spec, [300, 900], 500, seed, stat_capture=True, n_threads=8)
    _assert_lmm_equal(g, o)


def test_lmm_zero_variance_hits_the_boundary(engine, oracle):
    """tau^2 = 0: about half the replicates estimate tau-hat = 0 -> boundary_hit = 1, pinned bit 0, still
    converged.  Here the deviance is flat to the last ulp around theta = 0, so whether a replicate is
    *classified* as a boundary hit can differ between two correct evaluations of the same objective
    (the reference's own classification comes from BOBYQA and is unpinned); the fixed-effect
    decisions are insensitive to it and must still agree."""
    from helpers import spec_c4
    spec = spec_c4()
    spec.cluster.tau_squared = 0.0
    g = engine.run_batch(spec, [600], 2000, BASE_SEED, stat_capture=True)
    o = oracle.run_batch(spec, [600], 2000, BASE_SEED, stat_capture=True, n_threads=8)
    assert (g.boundary_hit != o.boundary_hit).mean() < 0.02
    a, b = g.stat_capture, o.stat_capture
    assert np.max(np.abs(a - b) / np.abs(b)) < 1e-5
    assert (g.uncorrected != o.uncorrected).sum() <= 2 and np.array_equal(g.converged, o.converged)
    frac = (g.boundary_hit == 1).mean()
    assert 0.3 < frac < 0.8 and g.converged.all()
    assert np.array_equal(g.pinned_components, (g.boundary_hit == 1).astype(np.uint64))
    assert g.counter("boundary_1")[0, 0] == (g.boundary_hit == 1).sum()


def test_lmm_fixed_size_clusters_and_shards(engine, oracle):
    from helpers import spec_c4
    spec = spec_c4()
    spec.cluster = S.ClusterSpec(sizing=("FixedSize", 20), tau_squared=0.4)
    g = engine.run_batch(spec, [200, 600], 1000, BASE_SEED, stat_capture=True)
    o = oracle.run_batch(spec, [200, 600], 1000, BASE_SEED, stat_capture=True, n_threads=8)
    _assert_lmm_equal(g, o)
    parts = [engine.run_batch(spec, [200, 600], 500, BASE_SEED, per_sim=False, sim_offset=off) for off in (0, 500)]
    assert np.array_equal(parts[0].counters + parts[1].counters, g.counters)
    with pytest.raises(S.EngineError) as e:
        engine.run_batch(spec, [210], 10, BASE_SEED)      # not a multiple of cluster_size
    assert e.value.status == S.Status.Unsupported


# ---------------------------------------------------------------------------------------------
# general data-generating path: scenario presets, non-normal shapes, BASELINE C5
# ---------------------------------------------------------------------------------------------
def _design_mismatch(engine, oracle, spec, n, n_sims, seed=BASE_SEED):
    d, y = engine.generate_design(spec, n, n_sims, seed, 0)
    bad_x = bad_y = 0
    worst = 0.0
    for s in range(n_sims):
        xo, yo, _ = oracle.generate_sim_data(spec, seed, s, n)
        mx = d[s].view(np.uint32) != xo.view(np.uint32)
        my = y[s].view(np.uint32) != yo.view(np.uint32)
        bad_x += int(mx.sum())
        bad_y += int(my.sum())
        if mx.any():
            worst = max(worst, float(np.max(np.abs(d[s][mx] - xo[mx]) / np.maximum(np.abs(xo[mx]), 1e-30))))
        if my.any():
            worst = max(worst, float(np.max(np.abs(y[s][my] - yo[my]) / np.maximum(np.abs(yo[my]), 1e-30))))
    return bad_x, bad_y, worst, d.size + y.size


def test_scenario_design_records_match_oracle_bytes(engine, oracle):
    """Per-replicate scenario design (perturbed + repaired correlation, marginal / residual swaps, beta jitter,
    heteroskedastic scaling).  Everything up to the f64 transform plane is the same arithmetic as the oracle;
    the transcendental kernels differ (ocml vs glibc, both <= 1 ULP), so an f32 value may differ in its last
    bit when the f64 result sits on a rounding boundary: allow <= 1e-5 of the values to move by one f32 ulp."""
    from helpers import preset
    for name in ("realistic", "doomer"):
        spec = spec_c2b()
        spec.scenario = preset(name)
        bx, by, worst, total = _design_mismatch(engine, oracle, spec, 120, 48)
        assert bx + by <= max(2, total * 1e-5) and worst < 2.5e-7, (name, bx, by, worst)


@pytest.mark.parametrize("dist", [S.Distribution.Binary, S.Distribution.RightSkewed, S.Distribution.LeftSkewed,
                                  S.Distribution.HighKurtosis, S.Distribution.Uniform])
def test_user_specified_marginals_match_oracle(engine, oracle, dist):
    spec = base_spec(3)
    c = np.eye(3)
    c[0, 1] = c[1, 0] = 0.4
    spec.correlation = c.flatten().tolist()
    spec.var_types = [S.Distribution.Normal, dist, dist]
    spec.var_params = [0.0, 0.3, 0.5]
    spec.effect_sizes = [0.0, 0.2, 0.3, 0.1]
    bx, by, worst, total = _design_mismatch(engine, oracle, spec, 101, 16)
    assert bx + by <= max(2, total * 1e-5) and worst < 2.5e-7, (dist, bx, by, worst)


@pytest.mark.parametrize("rd,df", [(S.ResidualDist.HighKurtosis, 5.0), (S.ResidualDist.RightSkewed, 8.0),
                                   (S.ResidualDist.LeftSkewed, 3.0), (S.ResidualDist.Uniform, 0.0)])
def test_residual_shapes_match_oracle(engine, oracle, rd, df):
    spec = spec_c2a()
    spec.residual_dist = rd
    spec.residual_pinned = True
    spec.scenario = S.ScenarioPerturbations(name="custom", residual_df=df)
    bx, by, worst, total = _design_mismatch(engine, oracle, spec, 64, 16)
    assert bx == 0 and by <= 2 and worst < 2.5e-7, (rd, bx, by, worst)  # sqrt / div only: f64-exact on both sides


def test_scenario_presets_run_batch_matches_oracle(engine, oracle):
    from helpers import preset
    for name in ("realistic", "doomer"):
        spec = spec_c2b()
        spec.scenario = preset(name)
        spec.report_overall = True
        g = engine.run_batch(spec, [50, 200, 500], 3000, BASE_SEED, stat_capture=True)
        o = oracle.run_batch(spec, [50, 200, 500], 3000, BASE_SEED, stat_capture=True, n_threads=8)
        a, b = g.stat_capture, o.stat_capture
        m = ~np.isnan(b)
        # one-ulp f32 differences in single design values move t^2 by ~1e-7 relative at most
        assert np.quantile(np.abs(a[m] - b[m]) / np.abs(b[m]), 0.999) < 1e-6
        assert (g.uncorrected != o.uncorrected).sum() <= 2 and (g.overall != o.overall).sum() <= 2
        assert np.abs(g.counters.astype(np.int64) - o.counters.astype(np.int64)).max() <= 2
    # power drops from optimistic to doomer (what the scenarios are for)
    opt = spec_c2b()
    p_opt = engine.run_batch(opt, [200], 20000, BASE_SEED, per_sim=False).counter("k_unc", 2)[0, 0]
    p_doom = g.counter("k_unc", 2)[1, 0] / 3000 * 20000
    assert p_doom < p_opt


def test_c5_grid_sweep_optimistic_matches_oracle(engine, oracle):
    """BASELINE C5 shape: C2b design on the grid N = 50..1000 (20 points) -- prefix fits on shared draws."""
    from helpers import C5_GRID
    spec = spec_c2b()
    g = engine.run_batch(spec, C5_GRID, 1000, BASE_SEED, stat_capture=True)
    o = oracle.run_batch(spec, C5_GRID, 1000, BASE_SEED, stat_capture=True, n_threads=8)
    _assert_batch_equal(g, o)
    ku = g.counter("k_unc", 2)[:, 0]
    assert (np.diff(ku.astype(np.int64)) >= -25).all() and ku[-1] > 0.99 * 1000


@pytest.mark.parametrize("name", ["realistic", "doomer"])
def test_c5_grid_sweep_under_the_robustness_presets_matches_oracle(name, engine, oracle):
    """BASELINE configs[4] on the device, preset by preset: the C2b design on the FULL grid N = 50..1000 (20 prefix fits
    per replicate on shared draws) under the reference bench's frozen realistic / doomer presets
    (throughput.rs:76-132) -- per-replicate design perturbation, marginal / residual swaps, per-study effects,
    heteroskedastic scaling.  Decisions equal the oracle's up to the general path's one-ulp band (<= 2 flags of
    1 200 x 20 x 2), counters within 2, and the counters-only form equals the per-sim one bit for bit."""
    from helpers import C5_GRID, preset
    spec = spec_c2b()
    spec.scenario = preset(name)
    n_sims = 1200
    g = engine.run_batch(spec, C5_GRID, n_sims, BASE_SEED)
    o = oracle.run_batch(spec, C5_GRID, n_sims, BASE_SEED, n_threads=8)
    assert g.uncorrected.shape == (n_sims, 20, 2)
    assert (g.uncorrected != o.uncorrected).sum() <= 2 and (g.corrected != o.corrected).sum() <= 2
    assert np.array_equal(g.converged, o.converged)
    assert np.abs(g.counters.astype(np.int64) - o.counters.astype(np.int64)).max() <= 2
    red = engine.run_batch(spec, C5_GRID, n_sims, BASE_SEED, per_sim=False)
    assert np.array_equal(red.counters, g.counters)
    ku = g.counter("k_unc", 2)[:, 0].astype(np.int64)
    assert ku[-1] > ku[0] and (np.diff(ku) >= -40).all()      # power rises along the grid (shared draws: nearly monotone)


def test_grids_longer_than_64_sample_sizes_in_one_call(engine, oracle):
    """One call carries up to 256 sample sizes (64 until round 3): a 150-point OLS grid and a 90-point clustered LMM grid --
    the fit state of a clustered grid (theta carried from N to N) cannot be split over calls."""
    spec = spec_c2b()
    grid = list(range(30, 30 + 150 * 6, 6))
    g, o = _both(engine, oracle, spec, grid, 300)
    _assert_batch_equal(g, o)
    from helpers import spec_c4
    lmm = spec_c4()
    sizes = [30 * v for v in range(4, 94)]
    g = engine.run_batch(lmm, sizes, 64, BASE_SEED)
    o = oracle.run_batch(lmm, sizes, 64, BASE_SEED, n_threads=16)
    assert g.uncorrected.shape == (64, 90, lmm.n_targets)
    assert (g.uncorrected != o.uncorrected).sum() <= 3 and (g.converged != o.converged).sum() <= 1
    with pytest.raises(S.EngineError) as e:
        engine.run_batch(spec, list(range(30, 30 + 257)), 8, BASE_SEED)
    assert e.value.status == S.Status.Unsupported and "256 sample sizes" in str(e.value)


@pytest.mark.parametrize("k", [7, 8, 9])
def test_seven_to_nine_continuous_predictors_on_the_general_path(k, engine, oracle):
    """The per-replicate scenario record packs one 3-bit marginal code per continuous predictor: up to 9 of them (6 in round 2).
    P = 8 runs the split fused launch, P = 9 the fused kernel, P = 10 the reduced-design kernel; the design bytes match the
    oracle inside the general path's one-ulp band and the decisions of a grid call follow."""
    from helpers import preset
    spec = base_spec(k)
    c = np.eye(k)
    c[0, 1] = c[1, 0] = 0.35
    c[2, k - 1] = c[k - 1, 2] = -0.3
    spec.correlation = c.flatten().tolist()
    spec.effect_sizes = [0.0] + [0.2, 0.15, 0.0, 0.1, -0.15, 0.0, 0.12, 0.1, -0.1][:k]
    spec.target_indices = [1, 2, k]
    for name in ("realistic", "doomer"):
        spec.scenario = preset(name)
        bx, by, worst, total = _design_mismatch(engine, oracle, spec, 90, 32)
        assert bx + by <= max(2, total * 1e-5) and worst < 2.5e-7, (k, name, bx, by, worst)
        g = engine.run_batch(spec, [80, 240], 1500, BASE_SEED)
        o = oracle.run_batch(spec, [80, 240], 1500, BASE_SEED, n_threads=8)
        assert (g.uncorrected != o.uncorrected).sum() <= 2 and np.array_equal(g.converged, o.converged), (k, name)
        assert np.abs(g.counters.astype(np.int64) - o.counters.astype(np.int64)).max() <= 2
    with pytest.raises(S.EngineError) as e:   # the tenth one is refused, loudly
        wide = base_spec(10)
        wide.scenario = preset("realistic")
        engine.run_batch(wide, [100], 8, BASE_SEED)
    assert e.value.status == S.Status.Unsupported and "9 continuous predictors" in str(e.value)


def test_seven_continuous_predictors_under_the_presets_glm_and_lmm(engine, oracle):
    from helpers import preset
    k = 7
    spec = base_spec(k)
    spec.effect_sizes = [0.0, 0.3, 0.2, 0.0, 0.1, -0.2, 0.0, 0.15]
    spec.target_indices = [1, 2, 7]
    spec.scenario = preset("realistic")
    import copy
    glm = copy.deepcopy(spec)
    glm.estimator, glm.outcome_kind = S.EstimatorSpec.Glm, S.OutcomeKind.Binary
    glm.intercept = glm.effect_sizes[0] = -0.4
    g = engine.run_batch(glm, [300], 600, BASE_SEED)
    o = oracle.run_batch(glm, [300], 600, BASE_SEED, n_threads=8)
    assert (g.uncorrected != o.uncorrected).sum() <= 3 and (g.converged != o.converged).sum() <= 1
    lmm = copy.deepcopy(spec)
    lmm.estimator = S.EstimatorSpec.Mle
    lmm.cluster = S.ClusterSpec(sizing=("FixedClusters", 12), tau_squared=0.25)
    g = engine.run_batch(lmm, [240], 600, BASE_SEED)
    o = oracle.run_batch(lmm, [240], 600, BASE_SEED, n_threads=8)
    assert (g.uncorrected != o.uncorrected).sum() <= 3 and (g.converged != o.converged).sum() <= 1


# ---------------------------------------------------------------------------------------------
# probit / Poisson GLM families
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", case_names("probit") + case_names("poisson"))
def test_probit_and_poisson_run_batch_match_oracle(engine, oracle, name):
    spec, seed, arr, low = load_case(name)
    spec.report_overall = True
    spec.correction_method = S.CorrectionMethod.Holm
    n = low["n"]
    g = engine.run_batch(spec, [n // 4, n], 1500, seed, stat_capture=True)
    o = oracle.run_batch(spec, [n // 4, n], 1500, seed, stat_capture=True, n_threads=8)
    a, b = g.stat_capture, o.stat_capture
    assert np.array_equal(np.isnan(a), np.isnan(b))
    m = ~np.isnan(a)
    assert np.all(np.abs(a[m] - b[m]) <= 1e-7 * np.abs(b[m]) + 1e-9)
    for f in ("uncorrected", "corrected", "converged", "overall"):
        assert np.array_equal(getattr(g, f), getattr(o, f)), f
    assert np.array_equal(g.counters, o.counters)
    assert g.converged.mean() > 0.99 and g.uncorrected[:, 1, 0].mean() > 0.9


def test_poisson_large_rate_and_factor_exclusion(engine, oracle):
    """lambda > 20 takes the rounded-normal arm of the Poisson quantile (rng.rs:370-419); sparse levels are
    excluded under the Count family as under the others."""
    spec, seed, arr, low = load_case("poisson_factor_a")
    spec.effect_sizes = [math.log(30.0)] + list(spec.effect_sizes[1:])
    spec.factor_proportions = [0.7, 0.2, 0.1]
    spec.factor_min_level_count = 5
    spec.report_overall = True
    g = engine.run_batch(spec, [30, 200], 1500, seed, stat_capture=True)
    o = oracle.run_batch(spec, [30, 200], 1500, seed, stat_capture=True, n_threads=8)
    assert np.array_equal(g.factor_excluded, o.factor_excluded) and (g.factor_excluded[:, 0, 0] == 1).all()
    for f in ("uncorrected", "corrected", "converged", "overall"):
        assert np.array_equal(getattr(g, f), getattr(o, f)), f
    d, y = engine.generate_design(spec, 200, 2, seed, 5)
    xo, yo, _ = oracle.generate_sim_data(spec, seed, 6, 200)
    assert np.array_equal(y[1], yo) and yo.mean() > 25


# ---------------------------------------------------------------------------------------------
# BASELINE.json full sizes of the iterative estimators: size-independent properties
# ---------------------------------------------------------------------------------------------
def test_full_size_c3_logistic_properties(engine, oracle):
    """configs[2]: logistic IRLS n = 500, p = 4, 10^5 replicates -- determinism, shard union, counter checksums, the
    oracle's first replicates, and the null-effect target (x3 = 0) rejecting at alpha."""
    from helpers import spec_c3
    spec, n = spec_c3(), 100_000
    a = engine.run_batch(spec, [500], n, BASE_SEED, per_sim=False)
    b = engine.run_batch(spec, [500], n, BASE_SEED, per_sim=False)
    assert np.array_equal(a.counters, b.counters)
    parts = [engine.run_batch(spec, [500], n // 4, BASE_SEED, per_sim=False, sim_offset=o * (n // 4)) for o in range(4)]
    assert np.array_equal(sum(p.counters for p in parts), a.counters)
    nt = spec.n_targets
    assert a.counter("n_sims")[0, 0] == n and a.counter("hist_unc", nt + 1)[0].sum() == n
    hist = a.counter("hist_unc", nt + 1)[0].astype(np.float64)
    assert a.counter("k_unc", nt)[0].sum() == (hist * np.arange(nt + 1)).sum()
    assert a.counter("converged")[0, 0] >= 0.999 * n
    head = engine.run_batch(spec, [500], 2000, BASE_SEED)
    o = oracle.run_batch(spec, [500], 2000, BASE_SEED, n_threads=8)
    assert np.array_equal(head.uncorrected, o.uncorrected) and np.array_equal(head.converged, o.converged)
    null_rate = a.counter("k_unc", nt)[0, 2] / n                       # x3 carries no effect
    assert abs(null_rate - 0.05) < 5 * np.sqrt(0.05 * 0.95 / n) + 0.002  # Wald z on a logit fit: within 0.2 pp of alpha


def test_full_size_c4_lmm_properties(engine, oracle):
    """configs[3]: random-intercept REML, 30 clusters x 20, 10^4 replicates -- same properties plus joint <= marginal logic."""
    from helpers import spec_c4
    spec, n = spec_c4(), 10_000
    a = engine.run_batch(spec, [600], n, BASE_SEED)
    b = engine.run_batch(spec, [600], n, BASE_SEED, per_sim=False)
    assert np.array_equal(a.counters, b.counters)
    parts = [engine.run_batch(spec, [600], n // 2, BASE_SEED, per_sim=False, sim_offset=o * (n // 2)) for o in range(2)]
    assert np.array_equal(parts[0].counters + parts[1].counters, a.counters)
    assert a.counter("n_sims")[0, 0] == n and a.converged.mean() > 0.999
    assert np.array_equal(a.joint_unc, a.joint_cor)                                  # batch.rs:1453-1477
    assert (a.pinned_components == (a.boundary_hit == 1)).all()
    head = oracle.run_batch(spec, [600], 1000, BASE_SEED, n_threads=8)
    assert np.array_equal(a.uncorrected[:1000], head.uncorrected) and np.array_equal(a.boundary_hit[:1000], head.boundary_hit)
    # Wald power of x1 = 0.15 with se = 1 / sqrt(n) on standardised within-cluster variation: ~0.95
    assert 0.93 < a.uncorrected[:, 0, 0].mean() < 0.97


def test_transform_class_lane_order_changes_no_result(engine):
    """The general-path OLS kernel runs the replicates of a launch grouped by transform class (scenario_kernels.hip);
    the identity order (MCP_DEBUG_NO_CLASS_ORDER=1, read once per process) must give the same counters and flags."""
    import json
    import subprocess
    import sys
    code = ("import json, sys; sys.path.insert(0, 'tests');"
            "from helpers import spec_c2b, preset, BASE_SEED, C5_GRID; from mcpower_amd import engine;"
            "s = spec_c2b(); s.scenario = preset('realistic');"
            "r = engine.run_batch(s, C5_GRID, 20000, BASE_SEED, chunk_sims=7000);"
            "print(json.dumps([r.counters.tolist(), int(r.uncorrected.astype('u8').sum()), r.uncorrected[:64].tolist()]))")
    outs = []
    for env_extra in ({}, {"MCP_DEBUG_NO_CLASS_ORDER": "1"}):
        env = dict(os.environ, **env_extra)
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=os.path.dirname(os.path.dirname(__file__)))
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(json.loads(p.stdout.strip().splitlines()[-1]))
    assert outs[0] == outs[1]
