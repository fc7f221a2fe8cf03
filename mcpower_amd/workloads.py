"""Frozen workload specs: the *inputs* of the reference's throughput cases
(crates/engine-core/src/bin/throughput.rs:167-247, 268-321), BASELINE.md's C1..C5 configurations and the reference's
captured mixed-model cases (validation/formulas.R).  `bench.py`, `__graft_entry__.smoke()`, the tools and the tests
all take their specs from here."""
from __future__ import annotations

import math

import numpy as np

from . import spec as S

MASK64 = (1 << 64) - 1


def splitmix64(z: int) -> int:
    """Mix13 finalizer: base_seed = splitmix64(user_seed) (reference find_power.rs:259); pinned in the tests against
    the reference's known answers (tests/golden/rng_kat.json)."""
    z &= MASK64
    z ^= z >> 30
    z = (z * 0xBF58476D1CE4E5B9) & MASK64
    z ^= z >> 27
    z = (z * 0x94D049BB133111EB) & MASK64
    z ^= z >> 31
    return z


# ---- frozen bench / parity specs (throughput.rs:167-321) ----
def base_spec(k: int) -> S.SimulationSpec:
    corr = np.eye(k).flatten().tolist()
    return S.SimulationSpec(
        n_non_factor=k, correlation=corr, var_types=[S.Distribution.Normal] * k, var_params=[0.0] * k,
        effect_sizes=[0.0] * (1 + k), target_indices=[1, 2], max_failed_fraction=0.1,
    )


def spec_c1() -> S.SimulationSpec:
    """C1: two-group t-test, n=100, p=2 (deterministic X, fixed allocation)."""
    return S.SimulationSpec(
        n_non_factor=0, n_factor_dummies=1, correlation=[], factor_n_levels=[2], factor_proportions=[0.5, 0.5],
        effect_sizes=[0.0, 0.5], target_indices=[1],
    )


def spec_c2a() -> S.SimulationSpec:
    """C2a: x1,x2,x3 + x1:x2 -> p=5, beta=(0,.18,.18,0,.10), targets [1,2,4]."""
    s = base_spec(3)
    s.interactions = [[1, 2]]
    s.effect_sizes = [0.0, 0.18, 0.18, 0.0, 0.10]
    s.target_indices = [1, 2, 4]
    return s


def spec_c2b() -> S.SimulationSpec:
    """C2b: the reference's frozen `ols_multi` (throughput.rs:269-272): base_spec(5), x1=x2=0.18."""
    s = base_spec(5)
    s.effect_sizes[1] = 0.18
    s.effect_sizes[2] = 0.18
    return s


def spec_k4() -> S.SimulationSpec:
    """Tuning aid: 4 independent normal predictors, p = 5, no interaction."""
    s = base_spec(4)
    s.effect_sizes[1] = 0.18
    s.effect_sizes[2] = 0.18
    return s


def spec_c3() -> S.SimulationSpec:
    """C3: glm_spec(3, 0.3): logit, beta0=ln(.3/.7), x1=x2=0.4, targets [1,2,3]."""
    s = base_spec(3)
    s.outcome_kind = S.OutcomeKind.Binary
    s.estimator = S.EstimatorSpec.Glm
    s.intercept = math.log(0.3 / 0.7)
    s.effect_sizes = [s.intercept, 0.4, 0.4, 0.0]
    s.target_indices = [1, 2, 3]
    s.max_failed_fraction = 0.25
    return s


def spec_c4() -> S.SimulationSpec:
    """C4: lme_spec(2, 30 clusters) at n=600, tau^2=0.25, x1=x2=0.15, truth_start."""
    s = base_spec(2)
    s.estimator = S.EstimatorSpec.Mle
    s.cluster = S.ClusterSpec(sizing=("FixedClusters", 30), tau_squared=0.25)
    s.effect_sizes = [0.0, 0.15, 0.15]
    s.scenario = S.ScenarioPerturbations(name="optimistic", truth_start=True)
    s.max_failed_fraction = 0.25
    return s


USER_SEED = 2137
BASE_SEED = splitmix64(USER_SEED)  # find_power.rs:259


def preset(name: str) -> S.ScenarioPerturbations:
    """Frozen scenario presets of the reference bench (crates/engine-core/src/bin/throughput.rs:76-132)."""
    menus = dict(new_distributions=[S.Distribution.RightSkewed, S.Distribution.LeftSkewed, S.Distribution.Uniform],
                 residual_dists=[S.ResidualDist.HighKurtosis, S.ResidualDist.RightSkewed])
    if name == "optimistic":
        return S.ScenarioPerturbations(name=name, residual_df=10.0, truth_start=True, **menus)
    if name == "realistic":
        return S.ScenarioPerturbations(name=name, heterogeneity=0.2, heteroskedasticity_ratio=2.0,
                                       correlation_noise_sd=0.15, distribution_change_prob=0.5,
                                       residual_change_prob=0.5, residual_df=8.0, sampled_factor_proportions=True,
                                       **menus)
    if name == "doomer":
        return S.ScenarioPerturbations(name=name, heterogeneity=0.4, heteroskedasticity_ratio=4.0,
                                       correlation_noise_sd=0.3, distribution_change_prob=0.8,
                                       residual_change_prob=0.8, residual_df=5.0, sampled_factor_proportions=True,
                                       **menus)
    raise KeyError(name)


def spec_ols_wide() -> S.SimulationSpec:
    """`ols_wide` of the reference bench (throughput.rs:274-277): base_spec(15), x1 = x2 = 0.2."""
    s = base_spec(15)
    s.effect_sizes[1] = s.effect_sizes[2] = 0.2
    return s


def _mixed(family, k, effects, n_clusters, tau_squared, slopes=(), extra=(), correlation=None) -> S.SimulationSpec:
    """A clustered continuous-predictor design as `tests/helpers.load_case` builds it from the reference's captured
    spec (validation/formulas.R); `slopes` / `extra` use the wire shape of ClusterSpec (generation.rs:171-252)."""
    s = S.SimulationSpec(
        n_non_factor=k, correlation=correlation or np.eye(k).flatten().tolist(), var_types=[S.Distribution.Normal] * k,
        var_params=[0.0] * k, effect_sizes=list(effects), target_indices=list(range(1, 1 + k)),
        outcome_kind=S.OutcomeKind.Binary if family == "logit" else S.OutcomeKind.Continuous,
        estimator=S.EstimatorSpec.Glm if family == "logit" else S.EstimatorSpec.Mle, intercept=effects[0],
    )
    s.cluster = S.ClusterSpec(sizing=("FixedClusters", n_clusters), tau_squared=tau_squared, slopes=list(slopes),
                              extra_groupings=list(extra))
    s.cluster_slope_design_cols = [1 + sl["column"] for sl in s.cluster.slopes]
    s.extra_slope_cols = [[1 + sl["column"] for sl in g.get("slopes", [])] for g in s.cluster.extra_groupings]
    s.scenario.truth_start = True
    return s


_LOGIT_03 = math.log(0.3 / 0.7)                 # baseline probability 0.3
_TAU2_LOGIT_ICC02 = 0.2 / 0.8 * math.pi ** 2 / 3.0  # latent-scale tau^2 of ICC 0.2 under the logistic family
_SLOPE_X1 = {"column": 0, "variance": 0.1, "corr_with_intercept": 0.3, "corr_with": []}


def spec_glmm() -> S.SimulationSpec:
    """Reference case glmm_intercept (validation/formulas.R:1092-1100): logit, y ~ x1 + (1 | g), 30 clusters x 20."""
    return _mixed("logit", 1, [_LOGIT_03, 0.5], 30, _TAU2_LOGIT_ICC02)


def spec_glmm_slope() -> S.SimulationSpec:
    """Reference case glmm_slope (validation/formulas.R:1103-1115): logit, y ~ x1 + (1 + x1 | g), 30 clusters x 20."""
    return _mixed("logit", 1, [_LOGIT_03, 0.5], 30, _TAU2_LOGIT_ICC02, slopes=[dict(_SLOPE_X1)])


def spec_glmm_crossed() -> S.SimulationSpec:
    """Reference case glmm_slope_crossed (validation/formulas.R:1130-1140): logit, (1 + x1 | g) + (1 | item), 24 x 6."""
    return _mixed("logit", 1, [_LOGIT_03, 0.5], 24, _TAU2_LOGIT_ICC02, slopes=[dict(_SLOPE_X1)],
                  extra=[{"relation": {"Crossed": {"n_clusters": 6}}, "tau_squared": 0.16, "slopes": []}])


def spec_lmm_slope() -> S.SimulationSpec:
    """Reference case lmm_slope_a (validation/formulas.R:461-469): y ~ x1 + (1 + x1 | g), 24 clusters x 20."""
    return _mixed("lme", 1, [0.0, 0.3], 24, 0.25, slopes=[dict(_SLOPE_X1)])


def spec_lmm_crossed() -> S.SimulationSpec:
    """Reference case lmm_crossed_a (formulas.R:385-395): y ~ x1 + x2 + (1 | subj) + (1 | item), 20 x 12 levels."""
    return _mixed("lme", 2, [0.0, 0.3, 0.2], 20, 0.25,
                  extra=[{"relation": {"Crossed": {"n_clusters": 12}}, "tau_squared": 0.15, "slopes": []}])


C5_GRID = list(range(50, 1001, 50))  # BASELINE C5: N = 50, 100, ..., 1000


# ---- the 21 frozen cases of the reference bench (crates/engine-core/src/bin/throughput.rs:268-633) ----
def _glm_spec(k: int, baseline_p: float) -> S.SimulationSpec:
    """glm_spec (throughput.rs:224-232): logit outcome, effect_sizes[0] = logit(baseline_p)."""
    s = base_spec(k)
    s.outcome_kind = S.OutcomeKind.Binary
    s.estimator = S.EstimatorSpec.Glm
    s.intercept = math.log(baseline_p / (1.0 - baseline_p))
    s.effect_sizes[0] = s.intercept
    s.max_failed_fraction = 0.25
    return s


def _lme_spec(k: int, n_clusters: int) -> S.SimulationSpec:
    """lme_spec (throughput.rs:236-247): Mle estimator, fixed cluster count, tau^2 = 0.25 (ICC 0.2)."""
    s = base_spec(k)
    s.estimator = S.EstimatorSpec.Mle
    s.cluster = S.ClusterSpec(sizing=("FixedClusters", n_clusters), tau_squared=0.25)
    s.max_failed_fraction = 0.25
    return s


def _glmm_spec(k: int, baseline_p: float, n_clusters: int, tau_squared: float) -> S.SimulationSpec:
    """glmm_spec (throughput.rs:256-266): glm_spec + a random intercept; the Wald SE stays WaldSe::default() = Rx
    (engine-contract estimator.rs:54-55; the bench comment at throughput.rs:264 predates that default)."""
    s = _glm_spec(k, baseline_p)
    s.cluster = S.ClusterSpec(sizing=("FixedClusters", n_clusters), tau_squared=tau_squared)
    return s


def _with(s: S.SimulationSpec, effects: dict, targets=None, slopes=(), extra=()) -> S.SimulationSpec:
    for i, v in effects.items():
        s.effect_sizes[i] = v
    if targets is not None:
        s.target_indices = list(targets)
    if s.cluster is not None:
        s.cluster.slopes = [dict(sl) for sl in slopes]
        s.cluster.extra_groupings = [dict(g) for g in extra]
        s.cluster_slope_design_cols = [1 + sl["column"] for sl in s.cluster.slopes]
    return s


def _crossed(n_clusters: int, tau_squared: float) -> dict:
    return {"relation": {"Crossed": {"n_clusters": n_clusters}}, "tau_squared": tau_squared, "slopes": []}


def _nested(n_per_parent: int, tau_squared: float) -> dict:
    return {"relation": {"NestedWithin": {"n_per_parent": n_per_parent}}, "tau_squared": tau_squared, "slopes": []}


_SLOPE_X2 = {"column": 1, "variance": 0.08, "corr_with_intercept": 0.1, "corr_with": [0.2]}


def throughput_cases() -> dict:
    """id -> (spec, n, n_sims): the frozen case grid of the reference's fits/sec bench, same ids, same inputs, in the
    reference's order (throughput.rs:268-633); its per-case timings are the `off` rows (optimistic scenario)."""
    def ols_factor_inter():
        s = base_spec(1)
        s.n_factor_dummies = 1
        s.factor_n_levels = [2]
        s.factor_proportions = [0.5, 0.5]
        s.interactions = [[1, 2]]
        s.effect_sizes = [0.0, 0.27, 0.5, 0.3]
        s.target_indices = [1]
        return s

    x12 = lambda v: {1: v, 2: v}
    return {
        "ols_multi": (_with(base_spec(5), x12(0.18)), 200, 10_000),
        "ols_wide": (_with(base_spec(15), x12(0.2)), 200, 10_000),
        "ols_large_n": (_with(base_spec(3), x12(0.037)), 5000, 2_000),
        "ols_factor_inter": (ols_factor_inter(), 200, 10_000),
        "glm_multi": (_with(_glm_spec(5, 0.3), x12(0.4)), 200, 10_000),
        "glm_wide": (_with(_glm_spec(15, 0.3), x12(0.4)), 200, 5_000),
        "glm_large_n": (_with(_glm_spec(3, 0.3), x12(0.09)), 5000, 1_000),
        "glm_rare": (_with(_glm_spec(2, 0.05), x12(0.45)), 500, 10_000),
        "lme_multi": (_with(_lme_spec(5, 20), x12(0.15)), 500, 2_000),
        "lme_many_clusters": (_with(_lme_spec(3, 100), x12(0.07)), 1000, 1_000),
        "lme_matched_q1": (_with(_lme_spec(4, 20), x12(0.15)), 480, 1_000),
        "lmm_crossed": (_with(_lme_spec(4, 20), x12(0.15), extra=[_crossed(12, 0.15)]), 480, 1_000),
        "lmm_nested": (_with(_lme_spec(4, 20), x12(0.15), extra=[_nested(4, 0.10)]), 480, 1_000),
        "lmm_crossed_nested": (_with(_lme_spec(4, 10), x12(0.15), extra=[_crossed(6, 0.15), _nested(2, 0.08)]), 480, 1_000),
        "lmm_slope": (_with(_lme_spec(4, 20), x12(0.15), slopes=[_SLOPE_X1]), 480, 1_000),
        "lmm_multislope": (_with(_lme_spec(4, 20), x12(0.15), slopes=[_SLOPE_X1, _SLOPE_X2]), 480, 1_000),
        "glmm_intercept": (_with(_glmm_spec(1, 0.3, 8, 0.822), {1: 0.28}, targets=[1]), 480, 1_000),
        "glmm_slope": (_with(_glmm_spec(1, 0.3, 8, 0.822), {1: 0.5}, targets=[1], slopes=[_SLOPE_X1]), 480, 1_000),
        "glmm_crossed": (_with(_glmm_spec(1, 0.3, 8, 0.822), {1: 0.30}, targets=[1], extra=[_crossed(6, 0.49)]), 480, 500),
        "glmm_nested": (_with(_glmm_spec(1, 0.3, 8, 0.822), {1: 0.30}, targets=[1], extra=[_nested(2, 0.33)]), 480, 500),
        "glmm_crossed_nested": (_with(_glmm_spec(1, 0.3, 8, 0.822), {1: 0.30}, targets=[1],
                                      extra=[_crossed(6, 0.49), _nested(2, 0.26)]), 480, 500),
    }


def spec_glm_wide() -> S.SimulationSpec:
    """`glm_wide` of the reference bench (throughput.rs:300-303): glm_spec(15, 0.3), x1 = x2 = 0.4, P = 16."""
    return throughput_cases()["glm_wide"][0]
