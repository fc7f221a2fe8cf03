"""Shared test helpers: golden-case loading and frozen bench specs.

The frozen specs restate the *inputs* of the reference's throughput cases
(crates/engine-core/src/bin/throughput.rs:167-247, 268-321) and BASELINE.md's
C1..C5 configurations.
"""
from __future__ import annotations

import json
import math
import os

import numpy as np

from mcpower_amd import spec as S

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MASK64 = (1 << 64) - 1


def splitmix64(z: int) -> int:
    """Mix13 finalizer (host-side restatement used to derive base_seed = splitmix64(user_seed),
    reference find_power.rs:259); pinned in tests against rng_kat.json."""
    z &= MASK64
    z ^= z >> 30
    z = (z * 0xBF58476D1CE4E5B9) & MASK64
    z ^= z >> 27
    z = (z * 0x94D049BB133111EB) & MASK64
    z ^= z >> 31
    return z


def load_kat():
    with open(os.path.join(GOLDEN, "rng_kat.json")) as fh:
        return json.load(fh)


def case_names(family=None):
    with open(os.path.join(GOLDEN, "cases.json")) as fh:
        idx = json.load(fh)
    return [k for k, v in idx.items() if family is None or v["family"] == family]


def mixed_case_names():
    return sorted(n for n in json.load(open(os.path.join(GOLDEN, "mixed", "cases.json"))) if n.startswith("lmm_"))


def load_case(name):
    """-> (SimulationSpec, base_seed, arrays dict, lowered-json dict)."""
    z = np.load(os.path.join(GOLDEN, "mixed" if name.startswith(("lmm_", "glmm_")) else "", f"{name}.npz"))
    low = json.loads(str(z["spec_json"]))
    k = low["n_non_factor"]
    p = 1 + k + low["n_factor_dummies"] + len(low["interactions"])
    fam = low["family"]
    spec = S.SimulationSpec(
        n_non_factor=k,
        n_factor_dummies=low["n_factor_dummies"],
        correlation=low["correlation"],
        var_types=[S.Distribution.Normal] * k,
        var_params=[0.0] * k,
        factor_n_levels=low["factor_n_levels"],
        factor_proportions=low["factor_proportions"],
        effect_sizes=low["effect_sizes"],
        target_indices=list(range(1, p)),
        interactions=low["interactions"],
        outcome_kind={"logit": S.OutcomeKind.Binary, "probit": S.OutcomeKind.Binary,
                      "poisson": S.OutcomeKind.Count}.get(fam, S.OutcomeKind.Continuous),
        link=S.LinkKind.Probit if fam == "probit" else None,
        estimator={"ols": S.EstimatorSpec.Ols, "logit": S.EstimatorSpec.Glm, "probit": S.EstimatorSpec.Glm,
                   "poisson": S.EstimatorSpec.Glm, "lme": S.EstimatorSpec.Mle}[fam],
        intercept=low["effect_sizes"][0],
    )
    if "cluster" in low:
        cl = low["cluster"]
        spec.cluster = S.ClusterSpec(sizing=("FixedClusters", cl["n_clusters"]), tau_squared=cl["tau_squared"],
                                     slopes=cl.get("slopes", []), extra_groupings=cl.get("extra_groupings", []))
        # slope columns name generation columns; these cases are continuous-only, so design column = 1 + column
        spec.cluster_slope_design_cols = [1 + s["column"] for s in spec.cluster.slopes]
        spec.extra_slope_cols = [[1 + s["column"] for s in g.get("slopes", [])] for g in spec.cluster.extra_groupings]
    base_seed = splitmix64(low["user_seed"])
    arrays = {k2: z[k2] for k2 in z.files if k2 != "spec_json"}
    return spec, base_seed, arrays, low


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


def spec_glmm() -> S.SimulationSpec:
    """Reference case glmm_intercept (validation/formulas.R:1092-1100): logit, y ~ x1 + (1 | g), 30 clusters x 20."""
    spec = load_case("glmm_intercept")[0]
    spec.scenario.truth_start = True
    return spec


def spec_glmm_slope() -> S.SimulationSpec:
    """Reference case glmm_slope (validation/formulas.R:1103-1115): logit, y ~ x1 + (1 + x1 | g), 30 clusters x 20."""
    spec = load_case("glmm_slope")[0]
    spec.scenario.truth_start = True
    return spec


def spec_glmm_crossed() -> S.SimulationSpec:
    """Reference case glmm_slope_crossed (validation/formulas.R:1130-1140): logit, (1 + x1 | g) + (1 | item), 24 x 6."""
    spec = load_case("glmm_slope_crossed")[0]
    spec.scenario.truth_start = True
    return spec


def spec_lmm_slope() -> S.SimulationSpec:
    """Reference case lmm_slope_a (validation/formulas.R:461-469): y ~ x1 + (1 + x1 | g), 24 clusters x 20."""
    spec = load_case("lmm_slope_a")[0]
    spec.scenario.truth_start = True
    return spec


def spec_lmm_crossed() -> S.SimulationSpec:
    """Reference case lmm_crossed_a (formulas.R:385-395): y ~ x1 + x2 + (1 | subj) + (1 | item), 20 x 12 levels."""
    spec = load_case("lmm_crossed_a")[0]
    spec.scenario.truth_start = True
    return spec


C5_GRID = list(range(50, 1001, 50))  # BASELINE C5: N = 50, 100, ..., 1000
