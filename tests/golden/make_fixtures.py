#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ from the DATA the
reference's own tests hold.  Run in the build container (needs /root/reference);
the GPU box never sees the reference, only the small files this writes.

Sources (all data, no code):
  * crates/engine-core/tests/rng_byte_identity.rs:18-50   Philox / splitmix64 / uniform KATs
  * crates/engine-core/tests/golden_rng.rs:37-235          64-draw stream goldens, phi, marginal_uniform
  * crates/engine-core/src/critvals.rs:1807-1926           quantile known answers (hand-copied table below)
  * validation/data/<case>.rds + <case>.golden.rds         sim-0 design/outcome bytes (introspect::capture_sim0_data)
                                                           and R lm/glm/lme4 fits of the same bytes
  * validation/oracle_cells.R:24-95 + data/oracle2_*.rds   six clustered-logistic cells and the reference engine's own
                                                           find_power power / type-I rates on them (K = 50 000)
Each per-case fixture stores the *kernel-level* spec the case lowers to
(contract_adapter.rs column layout: [intercept | continuous | factor dummies |
interactions]) so tests can rebuild a SimulationSpec
without parsing formulas on the GPU box.
"""
from __future__ import annotations

import json
import math
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from rds_reader import read_rds, to_py  # noqa: E402

REF = "/root/reference"


# ---------------------------------------------------------------------------
# 1. RNG known answers: pull the hex tables out of the Rust test sources
# ---------------------------------------------------------------------------
def _rust_array(src: str, name: str):
    m = re.search(r"const\s+" + name + r"\s*:\s*\[[^\]]*\]\s*=\s*\[(.*?)\];", src, re.S)
    if not m:
        raise KeyError(name)
    body = re.sub(r"//[^\n]*", "", m.group(1))
    vals = []
    for tok in body.replace("\n", " ").split(","):
        tok = tok.strip().replace("_", "")
        if not tok:
            continue
        if tok.lower().startswith("0x"):
            vals.append(int(tok, 16))
        else:
            vals.append(float(tok) if ("." in tok or "e" in tok.lower()) else int(tok))
    return vals


def make_rng_kat():
    g = open(f"{REF}/crates/engine-core/tests/golden_rng.rs").read()
    kat = {
        "source": "crates/engine-core/tests/{rng_byte_identity.rs,golden_rng.rs}",
        "philox": [
            {"ctr": [0, 0, 0, 0], "key": [0, 0], "out": [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]},
            {"ctr": [0xFFFFFFFF] * 4, "key": [0xFFFFFFFF] * 2, "out": [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]},
            {"ctr": [0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], "key": [0xA4093822, 0x299F31D0],
             "out": [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]},
        ],
        "splitmix64": [[0, 0], [1, 0x5692161D100B05E5], [0x0123456789ABCDEF, 0xB2C058E4EBB5112C],
                       [0xFFFFFFFFFFFFFFFF, 0xB4D055FCF2CBBD7B]],
        "u32_to_unit_f32_bits": [[0, 0x33800000], [0xFFFFFFFF, 0x3F7FFFFF], [0x9E3779B9, 0x3F1E3779]],
        # golden_rng.rs: normals seed (42, sim 0), uniforms (42, 1), categoricals (42, 2) probs [.2,.3,.5],
        # scenario uniforms (42, 0)
        "normals_seed42_sim0_bits": _rust_array(g, "GOLDEN_NORMALS"),
        "uniforms_seed42_sim1_bits": _rust_array(g, "GOLDEN_UNIFORMS"),
        "categoricals_seed42_sim2": _rust_array(g, "GOLDEN_CATEGORICALS"),
        "categorical_probs": [0.2, 0.3, 0.5],
        "scenario_uniforms_seed42_sim0_bits": _rust_array(g, "GOLDEN_SCENARIO_U32S"),
        "z_grid": _rust_array(g, "Z_GRID"),
        "phi_bits": _rust_array(g, "GOLDEN_PHI"),
        "marginal_uniform_bits": _rust_array(g, "GOLDEN_MARGINAL_UNIFORM"),
    }
    # confirm the seeds used by the golden tests (data, read from the test bodies)
    assert "SimRng::new(42, 0)" in g and "SimRng::new(42, 1)" in g and "SimRng::new(42, 2)" in g, "golden seeds moved"
    with open(os.path.join(HERE, "rng_kat.json"), "w") as fh:
        json.dump(kat, fh, indent=1)
    print("rng_kat.json", {k: (len(v) if isinstance(v, list) else v) for k, v in kat.items()})


# ---------------------------------------------------------------------------
# 2. validation cases: formula -> kernel spec
# ---------------------------------------------------------------------------
def _parse_kv(s: str):
    out = {}
    for part in re.split(r",\s*(?![^()]*\))", s):
        part = part.strip()
        if not part:
            continue
        k, v = part.split("=", 1)
        out[k.strip()] = v.strip()
    return out


def lower_case(spec: dict, columns):
    """Lower the readable case description to the kernel layout."""
    one = lambda v: v[0] if isinstance(v, (list, np.ndarray)) and len(v) == 1 else v
    formula = one(spec["formula"])
    family = one(spec["family"])
    rhs = formula.split("~", 1)[1]
    rhs = re.sub(r"\+\s*\([^)]*\)", "", rhs).strip()  # drop (1|grp)
    # variables + factor declarations
    vt = spec.get("variable_types")
    vts = [vt] if isinstance(vt, str) else (list(vt) if vt is not None else [])
    factors = {}
    for entry in vts:
        for name, val in _parse_kv(entry).items():
            m = re.match(r"\(factor,\s*(.*)\)", val)
            if m:
                factors[name] = [float(x) for x in m.group(1).split(",")]
    terms = [t.strip() for t in rhs.split("+")]
    mains, inters = [], []
    for t in terms:
        if "*" in t:
            a, b = [x.strip() for x in t.split("*")]
            for v in (a, b):
                if v not in mains:
                    mains.append(v)
            inters.append((a, b))
        elif ":" in t:
            a, b = [x.strip() for x in t.split(":")]
            inters.append((a, b))
        elif t not in mains:
            mains.append(t)
    cont = [v for v in mains if v not in factors]
    facs = [v for v in mains if v in factors]
    k = len(cont)
    col_of = {v: 1 + i for i, v in enumerate(cont)}
    dummy_cols = {}
    c = 1 + k
    factor_n_levels, factor_props = [], []
    for f in facs:
        nl = len(factors[f])
        factor_n_levels.append(nl)
        factor_props.extend(factors[f])
        for lvl in range(2, nl + 1):
            dummy_cols[(f, lvl)] = c
            c += 1
    n_fd = c - 1 - k
    interactions, inter_names = [], []
    for a, b in inters:
        a_cols = [(a, col_of[a])] if a in col_of else [(f"{a}[{l}]", dummy_cols[(a, l)]) for l in range(2, len(factors[a]) + 1)]
        b_cols = [(b, col_of[b])] if b in col_of else [(f"{b}[{l}]", dummy_cols[(b, l)]) for l in range(2, len(factors[b]) + 1)]
        for an, ac in a_cols:
            for bn, bc in b_cols:
                interactions.append([ac, bc])
                inter_names.append(f"{an}:{bn}")
    p = 1 + k + n_fd + len(interactions)
    assert p == len(columns), (formula, p, columns)
    eff = [0.0] * p
    for name, val in _parse_kv(one(spec["effects"])).items():
        v = float(val)
        if ":" in name:
            idx = 1 + k + n_fd + inter_names.index(name) if name in inter_names else None
            if idx is None:  # 2-level factor named bare in the interaction
                a, b = name.split(":")
                cand = [i for i, nm in enumerate(inter_names) if nm.split(":")[0].split("[")[0] == a.split("[")[0]
                        and nm.split(":")[1].split("[")[0] == b.split("[")[0]]
                assert len(cand) == 1, name
                idx = 1 + k + n_fd + cand[0]
            eff[idx] = v
        else:
            m = re.match(r"(\w+)\[(\d+)\]", name)
            if m:
                eff[dummy_cols[(m.group(1), int(m.group(2)))]] = v
            elif name in col_of:
                eff[col_of[name]] = v
            else:  # bare 2-level factor
                eff[dummy_cols[(name, 2)]] = v
    corr = np.eye(k)
    cs = spec.get("correlations")
    if cs is not None:
        for m in re.finditer(r"corr\((\w+),\s*(\w+)\)\s*=\s*([-\d.]+)", one(cs)):
            i, j = col_of[m.group(1)] - 1, col_of[m.group(2)] - 1
            corr[i, j] = corr[j, i] = float(m.group(3))
    out = dict(
        family=family, n=int(one(spec["n"])), user_seed=int(one(spec["seed"])),
        n_non_factor=k, n_factor_dummies=n_fd, correlation=corr.flatten(order="F").tolist(),
        factor_n_levels=factor_n_levels, factor_proportions=factor_props, interactions=interactions,
        effect_sizes=eff, formula=formula,
    )
    if family in ("logit", "probit"):
        bp = float(one(spec["baseline_probability"]))
        out["baseline_probability"] = bp
        if family == "logit":
            out["effect_sizes"][0] = math.log(bp / (1.0 - bp))
        else:  # probit: the intercept is the standard normal quantile of the baseline probability
            from scipy.stats import norm
            out["effect_sizes"][0] = float(norm.ppf(bp))
    if family == "poisson":
        br = float(one(spec["baseline_rate"]))
        out["baseline_rate"] = br
        out["effect_sizes"][0] = math.log(br)
    if "cluster" in spec and spec["cluster"] is not None:
        cl = spec["cluster"]
        icc = float(one(cl["ICC"]))
        # family-aware latent-scale variance (ports/py/mcpower/model.py:1737-1758): logit carries pi^2 / 3
        tau_sq = icc / (1.0 - icc) * (math.pi ** 2 / 3.0 if family == "logit" else 1.0)
        out["cluster"] = dict(n_clusters=int(one(cl["n_clusters"])), cluster_size=int(one(cl["cluster_size"])),
                              icc=icc, tau_squared=tau_sq)
    return out


CASES = (
    [f"ols_{n}" for n in ("simple_a", "simple_b", "two_a", "two_b", "zero_a", "corr_a", "corr_b", "factor_a",
                          "factor_b", "ff_a", "ff_b", "interaction_a", "interaction_b", "cf_a", "cf_b")]
    + [f"glm_{n}" for n in ("simple_a", "simple_b", "two_a", "two_b", "factor_a", "factor_b", "interaction_a",
                            "interaction_b")]
    + [f"lme_{n}" for n in ("simple_a", "simple_b", "two_a", "two_b", "factor_a", "factor_b", "interaction_a",
                            "interaction_b")]
    + [f"probit_{n}" for n in ("simple_a", "simple_b", "two_a", "factor_a")]
    + [f"poisson_{n}" for n in ("simple_a", "simple_b", "two_a", "factor_a")]
)


def make_cases():
    index = {}
    for name in CASES:
        d = to_py(read_rds(f"{REF}/validation/data/{name}.rds"))
        g = to_py(read_rds(f"{REF}/validation/data/{name}.golden.rds"))
        low = lower_case(d["spec"], d["columns"])
        design = np.asarray(d["design"], dtype=np.float64)
        outcome = np.asarray(d["outcome"], dtype=np.float64)
        # the reference's data plane is f32: the stored doubles must be exact f32 values
        assert np.array_equal(design.astype(np.float32).astype(np.float64), design), name
        assert np.array_equal(outcome.astype(np.float32).astype(np.float64), outcome), name
        arrays = dict(
            design=design.astype(np.float32), outcome=outcome.astype(np.float32),
            golden_beta=np.asarray(g["beta"], dtype=np.float64), golden_se=np.asarray(g["se"], dtype=np.float64),
            golden_stat=np.asarray(g["stat"], dtype=np.float64), golden_crit=np.asarray(g["crit"], dtype=np.float64),
        )
        if d.get("cluster_ids") is not None:
            arrays["cluster_ids"] = np.asarray(d["cluster_ids"], dtype=np.int64)
        for extra in ("vc", "theta", "sigma", "tau_squared"):
            if extra in g and g[extra] is not None:
                try:
                    arrays["golden_" + extra] = np.asarray(g[extra], dtype=np.float64)
                except Exception:
                    pass
        low["golden_keys"] = sorted(k for k in g.keys())
        low["hash"] = d["hash"][0] if isinstance(d["hash"], list) else d["hash"]
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), spec_json=np.array(json.dumps(low)), **arrays)
        index[name] = dict(n=low["n"], p=len(d["columns"]), family=low["family"])
        print(name, design.shape, low["family"], "golden:", low["golden_keys"])
    with open(os.path.join(HERE, "cases.json"), "w") as fh:
        json.dump(index, fh, indent=1)


MIXED_CASES = ("lmm_crossed_a", "lmm_crossed_b", "lmm_nested_a", "lmm_nested_b", "lmm_crossed_nested_a", "lmm_slope_a",
               "lmm_slope_b", "lmm_multislope", "lmm_slope_crossed", "lmm_slope_crossed_extra_slopes",
               "lmm_slope_nested_extra_slopes", "glmm_intercept", "glmm_slope", "glmm_multislope", "glmm_slope_crossed",
               "glmm_slope_crossed_extra_slopes", "glmm_slope_nested_extra_slopes")


def _one(v):
    return v[0] if isinstance(v, (list, np.ndarray)) and len(v) == 1 else v


def _slope_list(slopes):
    return [dict(column=int(_one(s["column"])), variance=float(_one(s["variance"])),
                 corr_with_intercept=float(_one(s["corr_with_intercept"])),
                 corr_with=[float(v) for v in np.atleast_1d(s["corr_with"])]) for s in (slopes or [])]


def make_mixed_cases():
    """The general mixed-model datasets of validation/formulas.R:382-539 (M2 crossed / nested intercepts, M3 random
    slopes and their compositions): sim-0 design / outcome / level ids captured by the reference, with lme4's
    REML fit (beta, Wald z, variance components, RE correlations, sigma^2) as goldens.  Extra groupings are stored
    in the contract's shape ({relation: {Crossed: {n_clusters}} | {NestedWithin: {n_per_parent}}, tau_squared,
    slopes}); common.R:84-110 shows the M3 `extra` list being translated the same way."""
    os.makedirs(f"{HERE}/mixed", exist_ok=True)
    index = {}
    for name in MIXED_CASES:
        d = to_py(read_rds(f"{REF}/validation/data/{name}.rds"))
        g = to_py(read_rds(f"{REF}/validation/data/{name}.golden.rds"))
        sp = d["spec"]
        low = lower_case(sp, d["columns"])
        low["cluster"]["slopes"] = _slope_list(sp.get("slopes"))
        extras = []
        for e in (sp.get("extra_groupings") or []):                       # M2: already the contract shape
            (rel, body), = e["relation"].items()
            extras.append(dict(relation={rel: {k: int(_one(v)) for k, v in body.items()}},
                               tau_squared=float(_one(e["tau_squared"])), slopes=_slope_list(e.get("slopes"))))
        for e in (sp.get("extra") or []):                                 # M3: var / kind / n_clusters
            cnt = int(_one(e["n_clusters"]))
            rel = {"Crossed": {"n_clusters": cnt}} if _one(e["kind"]) == "crossed" else {"NestedWithin": {"n_per_parent": cnt}}
            extras.append(dict(relation=rel, tau_squared=float(_one(e["tau_squared"])), slopes=_slope_list(e.get("slopes"))))
        low["cluster"]["extra_groupings"] = extras
        low["lmer_re"] = _one(sp.get("lmer_re") or sp.get("glmer_re"))
        design = np.asarray(d["design"], dtype=np.float64)
        outcome = np.asarray(d["outcome"], dtype=np.float64)
        assert np.array_equal(design.astype(np.float32).astype(np.float64), design), name
        assert np.array_equal(outcome.astype(np.float32).astype(np.float64), outcome), name
        ex = d.get("extra_grouping_ids") or []
        gv = g["vars"]
        arrays = dict(design=design.astype(np.float32), outcome=outcome.astype(np.float32),
                      cluster_ids=np.asarray(d["cluster_ids"], dtype=np.int64),
                      extra_ids=np.asarray(ex, dtype=np.int64).reshape(len(ex), design.shape[0]),
                      golden_beta=np.asarray(g["beta"], dtype=np.float64), golden_z=np.asarray(g["z"], dtype=np.float64),
                      golden_vars=np.asarray(list(gv.values()) if isinstance(gv, dict) else gv, dtype=np.float64),
                      golden_corr=np.asarray(g["corr"], dtype=np.float64),
                      golden_sigma2=np.asarray(g["sigma2"], dtype=np.float64))
        low["hash"] = _one(d["hash"])
        np.savez_compressed(os.path.join(HERE, "mixed", f"{name}.npz"), spec_json=np.array(json.dumps(low)), **arrays)
        index[name] = dict(n=low["n"], p=len(d["columns"]), lmer_re=low["lmer_re"])
        print(name, design.shape, low["lmer_re"], arrays["golden_vars"], arrays["golden_corr"])
    with open(os.path.join(HERE, "mixed", "cases.json"), "w") as fh:
        json.dump(index, fh, indent=1)


# quantile known answers restated from crates/engine-core/src/critvals.rs tests (:1807-1926) are checked
# against scipy in tests/test_oracle_critvals.py; the per-case `crit` goldens above pin t_ppf/norm_ppf to R.

def make_contracts():
    """The msgpack `Vec<SimulationContract>` blobs the reference's own port tests hold
    (ports/r/tests/testthat/_golden, written by its gen_golden.py / gen_two_factor_golden.py):
    copied byte for byte as the input fixtures of the `_engine.find_power` shim tests."""
    os.makedirs(f"{HERE}/contracts", exist_ok=True)
    for name in ("ols_contract", "two_factor_accumulation_contract"):
        blob = open(f"{REF}/ports/r/tests/testthat/_golden/{name}.msgpack", "rb").read()
        with open(f"{HERE}/contracts/{name}.msgpack", "wb") as f:
            f.write(blob)


def make_crossings():
    """End-to-end power goldens: validation/data/cross_*.golden.rds hold the required N per effect that the
    reference's own dense `find_sample_size` run froze (validation_crossing.rmd:118-122: from 20 to 300 by 2,
    100 000 sims, seed 2137), with the model of each case from validation/formulas.R:661-677 (CROSS_CASES)."""
    cases = {
        "cross_ols": dict(family="ols", effects={"x1": 0.40, "x2": 0.25}),
        "cross_glm": dict(family="logit", effects={"x1": 0.50, "x2": 0.45}, baseline_probability=0.3),
        "cross_lme": dict(family="lme", effects={"x1": 0.45}, cluster=dict(ICC=0.2, n_clusters=10)),
        "cross_partial": dict(family="ols", effects={"x1": 0.40, "x2": 0.05}),
    }
    out = {}
    for name, model in cases.items():
        g = to_py(read_rds(f"{REF}/validation/data/{name}.golden.rds"))
        out[name] = dict(model=model, n_sims=int(g["n_sims"][0]), by=int(g["by"][0]), from_size=int(g["from_size"][0]),
                         to_size=int(g["to_size"][0]), seed=int(g["seed"][0]),
                         dense={k: dict(status=v["status"][0],
                                        n_achievable=None if v["n_achievable"] is None else int(v["n_achievable"][0]))
                                for k, v in g["dense"].items()})
    with open(f"{HERE}/crossing.json", "w") as f:
        json.dump(out, f, indent=1)


def make_scenario_goldens():
    """validation/data/scenario_golden.rds: the headline statistics the reference froze for its scenario
    probes (validation_scenarios.rmd:549-575 -- per case, the `est` of every gate plus `golden_only` extras),
    each a deterministic function of generated data at committed seeds.  Values only; the probes themselves
    are restated in tests/test_scenario_goldens.py."""
    g = to_py(read_rds(f"{REF}/validation/data/scenario_golden.rds"))
    out = {case: {k: float(v) for k, v in vals.items()} for case, vals in g.items()}
    with open(f"{HERE}/scenario_golden.json", "w") as f:
        json.dump(out, f, indent=1, ensure_ascii=False)


# ---------------------------------------------------------------------------
# 7. Oracle-2: the reference ENGINE's own find_power numbers on six harsh clustered-logistic cells
#    (validation/oracle_cells.R:24-95 = the cell definitions, data; validation/data/oracle2_<cell>.rds = one frozen row
#    per cell written by validation/oracle2_power.R:131-136 at K = 50 000: power_hessian_fp / typeI_hessian_fp are
#    find_power(wald_se = "hessian", n_sims = K, seed = cell seed [+ 2 000 000 for the null model]); power_rx, typeI_rx,
#    power_hessian, typeI_hessian are per-draw Monte-Carlo rates of the same model over K other seeds)
# ---------------------------------------------------------------------------
def make_oracle2():
    src = open(f"{REF}/validation/oracle_cells.R").read()
    src = re.sub(r"#[^\n]*", "", src)  # the comments carry no data
    cells = {}
    for m in re.finditer(r"(\w+)\s*=\s*list\(\s*label\s*=\s*\"(\w+)\"(.*?)\n  \)", src, re.S):
        body = m.group(3)

        def num(key, text=body):
            mm = re.search(key + r"\s*=\s*(-?[0-9.]+)L?", text)
            return float(mm.group(1)) if mm else None

        eff = re.search(r'effects\s*=\s*"(\w+)=([0-9.]+)"', body)
        cell = {
            "formula": re.search(r'formula\s*=\s*"([^"]+)"', body).group(1),
            "target_term": eff.group(1), "beta_star": float(eff.group(2)),
            "baseline_probability": num("baseline_probability"), "icc": num("ICC"),
            "n_clusters": int(num("n_clusters", body[body.index("cluster = list"):])),
            "cluster_size": int(num("cluster_size")), "n": int(num(r"\bn")), "seed": int(num("seed")),
            "slopes": [], "extra": [],
        }
        sl = re.search(r"slopes\s*=\s*list\(list\((.*?)\)\)", body, re.S)
        if sl:
            cell["slopes"].append({"column": int(num("column", sl.group(1))), "variance": num("variance", sl.group(1)),
                                   "corr_with_intercept": num("corr_with_intercept", sl.group(1)), "corr_with": []})
        ex = re.search(r"extra\s*=\s*list\(list\((.*?)\)\)", body, re.S)
        if ex:
            kind = re.search(r'kind\s*=\s*"(\w+)"', ex.group(1)).group(1)
            cell["extra"].append({"kind": kind, "n_clusters": int(num("n_clusters", ex.group(1))),
                                  "tau_squared": num("tau_squared", ex.group(1))})
        cells[m.group(2)] = cell
    assert sorted(cells) == sorted(["few6_int", "few8_slope", "few10_crossed", "hiICC_int", "smallclust", "boundary"]), cells.keys()
    out = {"source": "validation/oracle_cells.R:24-95 (cells), validation/data/oracle2_<cell>.rds (frozen rows of "
                     "validation/oracle2_power.R at K = 50 000); null-model seed offset 2 000 000 (oracle2_power.R:135-136)",
           "k": 50000, "null_seed_offset": 2000000, "cells": {}}
    for name, cell in cells.items():
        row = to_py(read_rds(f"{REF}/validation/data/oracle2_{name}.rds"))
        frozen = {}
        for k, v in row.items():
            v0 = np.asarray(v).reshape(-1)[0]
            frozen[k] = str(v0) if isinstance(v0, (str, np.str_)) else float(v0)
        assert int(frozen["n"]) == cell["n"] and int(frozen["n_clusters"]) == cell["n_clusters"] and \
            abs(frozen["ICC"] - cell["icc"]) < 1e-12 and int(frozen["cluster_size"]) == cell["cluster_size"], name
        out["cells"][name] = {"cell": cell, "frozen": frozen}
    with open(os.path.join(HERE, "oracle2.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("oracle2.json:", {k: (v["frozen"]["power_hessian_fp"], v["frozen"]["typeI_hessian_fp"]) for k, v in out["cells"].items()})


if __name__ == "__main__":
    make_rng_kat()
    make_cases()
    make_contracts()
    make_crossings()
    make_scenario_goldens()
    make_mixed_cases()
    make_oracle2()
