"""Shared test helpers: golden-case loading; the frozen bench specs live in mcpower_amd/workloads.py and are
re-exported here."""
from __future__ import annotations

import json
import math
import os

import numpy as np

from mcpower_amd import spec as S

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
from mcpower_amd.workloads import (BASE_SEED, C5_GRID, MASK64, USER_SEED, base_spec, preset, spec_c1, spec_c2a,  # noqa: F401
                                   spec_c2b, spec_c3, spec_c4, spec_glmm, spec_glmm_crossed, spec_glmm_slope, spec_k4,
                                   spec_lmm_crossed, spec_lmm_slope, spec_ols_wide, splitmix64)


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
