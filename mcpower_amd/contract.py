"""`SimulationContract` (the msgpack envelope every reference host passes in) -> `SimulationSpec`.

Follows `crates/engine-core/src/contract_adapter.rs:74-230` (`contract_to_simulation_spec`) and its
`translate_*` helpers, over the wire shape of `crates/engine-contract/src/{contract,generation,design,
outcome,test_spec,scenarios}.rs` as `rmp_serde::to_vec_named` writes it: structs are string-keyed
maps, data-less enum variants are bare strings, data-carrying variants are one-key maps
(`{"Synthetic": {...}}`), `#[serde(tag = "kind")]` targets are flat maps with a `kind` key, and the
`rename_all = "snake_case"` enums (outcome kind, estimator, correction, residual) are lower-case.

Kernel column order (contract_adapter.rs:440-524): `[intercept | non-factor columns in generation
order | factor dummies in generation order | interactions in design_generation order]`.
"""
from __future__ import annotations

import math
from typing import Any, Dict, List, Optional, Sequence, Tuple

from . import spec as S


class ContractError(ValueError):
    """`AdapterError::ContractInvalid` / `ContractError` (engine-contract/src/error.rs) wording."""


class JointNotSupported(ContractError):
    """`AdapterError::JointNotSupported` (contract_adapter.rs:90-94)."""


# ---------------------------------------------------------------------------------------------
# wire helpers
# ---------------------------------------------------------------------------------------------
def decode_contracts(contracts_bytes: bytes) -> List[dict]:
    """`decode_contracts` (engine-py/src/lib.rs): a msgpack array of named-map contracts."""
    import msgpack

    try:
        out = msgpack.unpackb(bytes(contracts_bytes), raw=False, strict_map_key=False)
    except Exception as e:  # msgpack raises several unrelated exception types
        raise ValueError(f"contracts payload is not valid msgpack: {e}") from e
    if not isinstance(out, list) or not all(isinstance(c, dict) for c in out):
        raise ValueError("contracts payload must be a msgpack array of SimulationContract maps")
    return out


def _variant(v: Any) -> Tuple[str, dict]:
    """externally tagged serde enum -> (variant name, fields)."""
    if isinstance(v, str):
        return v, {}
    if isinstance(v, dict) and len(v) == 1:
        (name, body), = v.items()
        return name, (body if isinstance(body, dict) else {"_": body})
    raise ContractError(f"malformed enum value {v!r}")


def _col_id(v: Any) -> int:
    """`ColumnId` is a newtype struct: rmp_serde writes it transparently (an int); tolerate [int]."""
    if isinstance(v, (list, tuple)) and len(v) == 1:
        v = v[0]
    return int(v)


_SYNTH = {"Normal": S.Distribution.Normal, "Binary": S.Distribution.Binary,
          "RightSkewed": S.Distribution.RightSkewed, "LeftSkewed": S.Distribution.LeftSkewed,
          "HighKurtosis": S.Distribution.HighKurtosis, "Uniform": S.Distribution.Uniform}
_RESID = {"normal": S.ResidualDist.Normal, "right_skewed": S.ResidualDist.RightSkewed,
          "left_skewed": S.ResidualDist.LeftSkewed, "high_kurtosis": S.ResidualDist.HighKurtosis,
          "uniform": S.ResidualDist.Uniform}
_OUTCOME = {"continuous": S.OutcomeKind.Continuous, "binary": S.OutcomeKind.Binary, "count": S.OutcomeKind.Count}
_ESTIMATOR = {"ols": S.EstimatorSpec.Ols, "glm": S.EstimatorSpec.Glm, "mle": S.EstimatorSpec.Mle}
_CORRECTION = {"none": S.CorrectionMethod.None_, "bonferroni": S.CorrectionMethod.Bonferroni,
               "holm": S.CorrectionMethod.Holm, "benjamini_hochberg": S.CorrectionMethod.BenjaminiHochberg,
               "tukey_hsd": S.CorrectionMethod.TukeyHsd}
_NON_FACTOR = ("Synthetic", "Resampled", "ResampledBinary")
_FACTOR = ("FactorSynthetic", "FactorFromFrame")


def _lookup(table: dict, key: Any, what: str):
    try:
        return table[key]
    except (KeyError, TypeError):
        raise ContractError(f"unknown {what} {key!r}") from None


class _Columns:
    """Generation columns with the kernel positions contract_adapter.rs:735-775 derives."""

    def __init__(self, contract: dict):
        self.cols = [_variant(c) for c in contract["generation"]["columns"]]
        self.n_non_factor = sum(1 for k, _ in self.cols if k in _NON_FACTOR)
        self.n_factor_dummies = 0
        self.continuous_pos: Dict[int, int] = {}   # generation index -> x_full column (1-based)
        self.dummy_base: Dict[int, int] = {}       # generation index -> x_full column of level 1
        self.factor_pos: Dict[int, int] = {}       # generation index -> factor ordinal
        nf_cursor, fd_cursor, k = 1, 1 + self.n_non_factor, 0
        for i, (kind, body) in enumerate(self.cols):
            if kind in _NON_FACTOR:
                self.continuous_pos[i] = nf_cursor
                nf_cursor += 1
            elif kind in _FACTOR:
                nl = int(body["n_levels"])
                self.dummy_base[i] = fd_cursor
                self.factor_pos[i] = k
                fd_cursor += max(nl - 1, 0)
                self.n_factor_dummies += max(nl - 1, 0)
                k += 1
            else:
                raise ContractError(f"unknown ColumnSpec variant {kind!r}")

    def continuous(self, gen_idx: int) -> int:   # column_position_for_continuous
        return self.continuous_pos[gen_idx]

    def dummy(self, gen_idx: int, level_index: int) -> int:   # column_position_for_dummy
        return self.dummy_base[gen_idx] + (level_index - 1)


def _terms(design: dict) -> List[Tuple[str, dict]]:
    return [_variant(t) for t in design["terms"]]


def _component_key(term: Tuple[str, dict]):
    kind, body = term
    if kind == "Direct":
        return ("Direct", _col_id(body["column"]))
    if kind == "DummyOf":
        return ("DummyOf", _col_id(body["column"]), int(body["level_index"]))
    if kind == "Interaction":
        return ("Interaction", tuple(_component_key(_variant(c)) for c in body["components"]))
    return (kind,)


# ---------------------------------------------------------------------------------------------
# validate() -- the invariants of engine-contract/src/validate.rs that guard the lowering below
# ---------------------------------------------------------------------------------------------
def validate_contract(c: dict) -> None:
    gen, outcome, test = c["generation"], c["outcome"], c["test"]
    cols = [_variant(x) for x in gen["columns"]]
    n_columns = len(cols)
    gen_terms = _terms(c["design_generation"])
    test_design = c.get("design_test") or c["design_generation"]
    test_terms = _terms(test_design)

    # 25 nagq backstop (engine-contract/src/error.rs:67-73): odd, 1..=25
    nagq = int(c.get("nagq", 1))
    if nagq < 1 or nagq > 25 or nagq % 2 == 0:
        raise ContractError(f"nagq must be odd and in 1..=25; got {nagq}")
    # 01 coefficient length
    if len(outcome["coefficients"]) != len(gen_terms):
        raise ContractError(f"coefficients.len ({len(outcome['coefficients'])}) must equal "
                            f"design_generation.terms.len ({len(gen_terms)})")
    # 02/03 targets
    n_terms = len(test_terms)
    for t in test["targets"]:
        kind = t.get("kind")
        refs = {"marginal": [t.get("term")], "joint": list(t.get("terms", [])),
                "contrast": [t.get("positive"), t.get("negative")]}.get(kind)
        if refs is None:
            raise ContractError(f"invalid TestSpec: unknown target kind {kind!r}")
        for r in refs:
            if r is None or not (0 <= int(r) < n_terms):
                raise ContractError(f"test target term index {r} out of range (design_test has {n_terms} terms)")
        if kind == "joint" and len(refs) == 0:
            raise ContractError("invalid TestSpec: Joint target needs at least one term")
        if kind == "contrast" and int(t["positive"]) == int(t["negative"]):
            raise ContractError("invalid TestSpec: Contrast positive and negative must differ")

    # 04/08/09/18 column ids, dummy levels, Direct on continuous, interaction shape
    def check_term(term, top=True):
        kind, body = term
        if kind == "Const":
            if not top:
                raise ContractError("interaction component must be Direct or DummyOf (no Const or nested Interaction)")
            return
        if kind == "Interaction":
            if not top:
                raise ContractError("interaction component must be Direct or DummyOf (no Const or nested Interaction)")
            comps = body["components"]
            if len(comps) < 2:
                raise ContractError(f"interaction term has {len(comps)} component(s); needs at least 2")
            for comp in comps:
                check_term(_variant(comp), top=False)
            return
        cid = _col_id(body["column"])
        if not (0 <= cid < n_columns):
            raise ContractError(f"ColumnId({cid}) out of range (generation has {n_columns} columns)")
        ckind, cbody = cols[cid]
        if kind == "Direct":
            if ckind in _FACTOR:
                raise ContractError(f"Direct term references factor column ColumnId({cid}); use DummyOf instead")
        elif kind == "DummyOf":
            nl = int(cbody.get("n_levels", 0)) if ckind in _FACTOR else 0
            li = int(body["level_index"])
            if not (1 <= li < nl):
                raise ContractError(f"dummy level_index {li} out of range (column ColumnId({cid}) has {nl} levels)")
        else:
            raise ContractError(f"unknown DesignTerm variant {kind!r}")

    for term in gen_terms + (test_terms if c.get("design_test") else []):
        check_term(term)

    # 05/06 correlations
    ckind, cbody = _variant(gen["correlations"])
    n_nf = sum(1 for k, _ in cols if k in _NON_FACTOR)
    if ckind == "Matrix":
        ids = [_col_id(v) for v in cbody["continuous_columns"]]
        n = len(ids)
        if len(cbody["values"]) != n * n:
            raise ContractError(f"correlation matrix has {len(cbody['values'])} entries, expected {n * n} (n={n})")
        for cid in ids:
            if not (0 <= cid < n_columns):
                raise ContractError(f"ColumnId({cid}) out of range (generation has {n_columns} columns)")
            if cols[cid][0] not in _NON_FACTOR:
                raise ContractError(f"correlation references non-continuous column (ColumnId({cid}))")
        if n != n_nf:
            raise ContractError(f"correlation matrix has {len(cbody['values'])} entries, expected {n_nf * n_nf} (n={n_nf})")
    elif ckind != "Identity":
        raise ContractError(f"unknown Correlations variant {ckind!r}")

    # 07 factor proportions, 23 binary probabilities
    for kind, body in cols:
        if kind in _FACTOR:
            props = body["proportions"]
            if int(body["n_levels"]) < 2:
                raise ContractError("factor column has invalid proportions: n_levels must be >= 2")
            if len(props) != int(body["n_levels"]):
                raise ContractError("factor column has invalid proportions: length must equal n_levels")
            if any((not math.isfinite(p)) or p < 0.0 for p in props):
                raise ContractError("factor column has invalid proportions: entries must be finite and >= 0")
            if abs(sum(props) - 1.0) > 1e-6:
                raise ContractError("factor column has invalid proportions: must sum to 1")
        elif kind == "Synthetic":
            sk, sb = _variant(body["kind"])
            if sk == "Binary" and not (math.isfinite(sb["p"]) and 0.0 <= sb["p"] <= 1.0):
                raise ContractError(f"SyntheticKind::Binary {{ p }} must be finite and in [0.0, 1.0]; got {sb['p']}")
        elif kind == "ResampledBinary":
            p = body["proportion"]
            if not (math.isfinite(p) and 0.0 <= p <= 1.0):
                raise ContractError(f"ResampledBinary {{ proportion }} must be finite and in [0.0, 1.0]; got {p}")

    # 10 heteroskedasticity driver
    drv = outcome.get("heteroskedasticity_driver")
    if drv is not None:
        cid = _col_id(drv)
        if not (0 <= cid < n_columns):
            raise ContractError(f"ColumnId({cid}) out of range (generation has {n_columns} columns)")
        if cols[cid][0] not in _NON_FACTOR:
            raise ContractError(f"heteroskedasticity references non-continuous column ColumnId({cid})")

    # 11 uploaded frame
    frame = gen.get("uploaded_frame")
    for kind, body in cols:
        if "frame_column" in body:
            fc = int(body["frame_column"])
            if frame is None or fc >= int(frame["n_cols"]):
                raise ContractError(f"frame_column {fc} references uploaded_frame but it is None or has fewer columns")

    # 12/13/24 estimator x outcome
    est = _lookup(_ESTIMATOR, c["estimator"], "estimator")
    okind = _lookup(_OUTCOME, outcome["kind"], "outcome kind")
    if est == S.EstimatorSpec.Mle and gen.get("cluster") is None:
        raise ContractError("estimator Mle requires generation.cluster to be Some(_)")
    if est == S.EstimatorSpec.Glm and okind == S.OutcomeKind.Continuous:
        raise ContractError("estimator Glm requires outcome.kind == Binary or Count")
    if outcome.get("link") is not None and okind != S.OutcomeKind.Binary:
        raise ContractError("outcome.link == Some(Probit) requires outcome.kind == Binary")
    # invariant 13 (validate.rs:458-474): the random-effect knobs need an estimator that draws random effects --
    # Mle, or Glm on clustered data (the GLMM); the message keeps the variant's wording
    if (c["scenario"].get("lme") is not None and est != S.EstimatorSpec.Mle
            and not (est == S.EstimatorSpec.Glm and gen.get("cluster") is not None)):
        raise ContractError("scenario.lme requires estimator == Mle")

    # 14/15
    mff = c["max_failed_fraction"]
    if not (0.0 <= mff <= 1.0):
        raise ContractError(f"max_failed_fraction must be in [0.0, 1.0]; got {mff}")
    alpha = test["alpha"]
    if not (0.0 < alpha < 1.0):
        raise ContractError(f"test.alpha must be in (0.0, 1.0); got {alpha}")

    # 17 posthoc
    if c.get("posthoc"):
        if est != S.EstimatorSpec.Ols:
            raise ContractError("posthoc tests require estimator == Ols")
        for ph in c["posthoc"]:
            fc = _col_id(ph["factor_column"])
            if not (0 <= fc < n_columns) or cols[fc][0] not in _FACTOR:
                raise ContractError("invalid PosthocSpec: factor_column must reference a factor column")
            for t in ph["target_term_indices"]:
                if not (0 <= int(t) < n_terms):
                    raise ContractError("invalid PosthocSpec: target term index out of range")
                tk, tb = test_terms[int(t)]
                if tk != "DummyOf" or _col_id(tb["column"]) != fc:
                    raise ContractError("invalid PosthocSpec: targets must be DummyOf terms of factor_column")

    # 22 scenario ranges
    sc = c["scenario"]
    for fld, lo_ok in (("heterogeneity", 0.0), ("correlation_noise_sd", 0.0)):
        v = sc[fld]
        if not (math.isfinite(v) and v >= lo_ok):
            raise ContractError(f"scenario perturbation field out of range: {fld} got {v}")
    for fld in ("distribution_change_prob", "residual_change_prob"):
        v = sc[fld]
        if not (math.isfinite(v) and 0.0 <= v <= 1.0):
            raise ContractError(f"scenario perturbation field out of range: {fld} got {v}")
    v = sc["heteroskedasticity_ratio"]
    if not (math.isfinite(v) and v > 0.0):
        raise ContractError(f"scenario perturbation field out of range: heteroskedasticity_ratio got {v}")


# ---------------------------------------------------------------------------------------------
# lowering
# ---------------------------------------------------------------------------------------------
def _term_to_kernel_col(cols: _Columns, gen_terms, term: Tuple[str, dict]) -> int:
    """`term_to_kernel_col` (contract_adapter.rs:528-545)."""
    kind, body = term
    if kind == "Const":
        return 0
    if kind == "Direct":
        return cols.continuous(_col_id(body["column"]))
    if kind == "DummyOf":
        return cols.dummy(_col_id(body["column"]), int(body["level_index"]))
    # interaction_kernel_col (contract_adapter.rs:709-727): match by components on the generation design
    want = _component_key(term)
    base, j = 1 + cols.n_non_factor + cols.n_factor_dummies, 0
    for g in gen_terms:
        if g[0] == "Interaction":
            if _component_key(g) == want:
                return base + j
            j += 1
    raise ContractError("test interaction has no matching generation interaction")


def _joint_is_omnibus(terms: Sequence[int], test_terms) -> bool:
    expected = sorted(i for i, t in enumerate(test_terms) if t[0] != "Const")
    return sorted(set(int(t) for t in terms)) == expected


def targets_and_contrasts(c: dict):
    """`translate_targets_and_contrasts` (contract_adapter.rs:568-632): kernel-space and report-space
    twins, in lockstep.  -> (target_indices, report_indices, contrast_pairs, report_pairs)."""
    cols = _Columns(c)
    gen_terms = _terms(c["design_generation"])
    test_terms = _terms(c.get("design_test") or c["design_generation"])
    marginal: Dict[int, int] = {}
    contrasts, contrasts_report = [], []
    for t in c["test"]["targets"]:
        kind = t["kind"]
        if kind == "marginal":
            marginal[_term_to_kernel_col(cols, gen_terms, test_terms[int(t["term"])])] = int(t["term"])
        elif kind == "contrast":
            pos, neg = int(t["positive"]), int(t["negative"])
            p_col = _term_to_kernel_col(cols, gen_terms, test_terms[pos])
            if test_terms[neg][0] == "Const":          # reference-level collapse
                marginal[p_col] = pos
            else:
                contrasts.append((p_col, _term_to_kernel_col(cols, gen_terms, test_terms[neg])))
                contrasts_report.append((pos, neg))
    keys = sorted(marginal)
    return keys, [marginal[k] for k in keys], contrasts, contrasts_report


def report_targets_and_contrasts(c: dict):
    """`report_targets_and_contrasts` (contract_adapter.rs:646-668)."""
    _, rep, _, rep_pairs = targets_and_contrasts(c)
    return rep, rep_pairs


def contract_to_simulation_spec(c: dict) -> S.SimulationSpec:
    """`contract_to_simulation_spec` (contract_adapter.rs:74-230)."""
    validate_contract(c)
    # wald_se / nagq only steer the clustered-GLM estimator (estimator.rs:58-66).  The device serves both Wald SE modes
    # and quadrature over a scalar random intercept; quadrature over slopes is refused, not approximated.
    wald_se = S.WaldSe.Rx
    if c.get("estimator") == "glm" and c["generation"].get("cluster") is not None:
        if int(c.get("nagq", 1)) != 1 and (c["generation"]["cluster"].get("slopes") or []):
            raise S.EngineError(S.Status.Unsupported, "unsupported on the device path: adaptive Gauss-Hermite quadrature (nagq > 1) with random slopes (random intercept only)")
        wald_se = _lookup({"rx": S.WaldSe.Rx, "hessian": S.WaldSe.Hessian}, str(c.get("wald_se", "rx")), "wald_se")
    cols = _Columns(c)
    gen = c["generation"]
    gen_terms = _terms(c["design_generation"])
    has_test_design = c.get("design_test") is not None
    test_terms = _terms(c["design_test"]) if has_test_design else gen_terms

    report_overall = False
    for t in c["test"]["targets"]:
        if t["kind"] == "joint":
            if _joint_is_omnibus(t["terms"], test_terms):
                report_overall = True
            else:
                got = sorted(int(v) for v in t["terms"])
                raise JointNotSupported(
                    f"Joint target over terms {got} is not supported: only the omnibus (all non-intercept terms) "
                    f"joint test is routed to the kernel")

    # translate_generation (contract_adapter.rs:246-318)
    var_types, var_pinned, var_params = [], [], []
    factor_n_levels, factor_proportions, factor_sampled = [], [], []
    for kind, body in cols.cols:
        if kind == "Synthetic":
            sk, sb = _variant(body["kind"])
            var_types.append(_lookup(_SYNTH, sk, "SyntheticKind"))
            var_pinned.append(bool(body.get("pinned", False)))
            var_params.append(float(sb.get("p", 0.0)) if sk == "Binary" else 0.0)
        elif kind == "Resampled":
            var_types.append(S.Distribution.UploadedData)
            var_pinned.append(False)
            var_params.append(0.0)
        elif kind == "ResampledBinary":
            var_types.append(S.Distribution.UploadedBinary)
            var_pinned.append(False)
            var_params.append(float(body["proportion"]))
        else:
            factor_n_levels.append(int(body["n_levels"]))
            factor_proportions.extend(float(p) for p in body["proportions"])
            sp = body.get("sampled_proportions")
            factor_sampled.append(None if sp is None else bool(sp))
    n_nf, n_fd = cols.n_non_factor, cols.n_factor_dummies

    # translate_correlation (contract_adapter.rs:320-333)
    ckind, cbody = _variant(gen["correlations"])
    if ckind == "Identity":
        correlation = [1.0 if i == j else 0.0 for i in range(n_nf) for j in range(n_nf)]
    else:
        correlation = [float(v) for v in cbody["values"]]

    # translate_uploaded_frame (contract_adapter.rs:356-469)
    upload_normal, upload_normal_shape = [], (0, 0)
    upload_data, upload_data_shape, bootstrap_frame_map = [], (0, 0), []
    frame = gen.get("uploaded_frame")
    if frame is not None:
        n_rows, n_cols = int(frame["n_rows"]), int(frame["n_cols"])
        upload_data, upload_data_shape = [float(v) for v in frame["data"]], (n_rows, n_cols)
        if frame.get("bootstrap", False):
            for kind, body in cols.cols:
                if kind in ("Resampled", "ResampledBinary"):
                    bootstrap_frame_map.append(int(body["frame_column"]))
                elif kind == "Synthetic":
                    bootstrap_frame_map.append(None)
            for kind, body in cols.cols:
                if kind == "FactorFromFrame":
                    bootstrap_frame_map.append(int(body["frame_column"]))
                elif kind == "FactorSynthetic":
                    bootstrap_frame_map.append(None)
        else:
            upload_normal = [0.0] * (n_rows * n_nf)
            j = 0
            for kind, body in cols.cols:
                if kind in ("Resampled", "ResampledBinary"):
                    fc = int(body["frame_column"])
                    vals = sorted(upload_data[r * n_cols + fc] for r in range(n_rows))
                    for r in range(n_rows):
                        upload_normal[r * n_nf + j] = vals[r]
                    j += 1
                elif kind == "Synthetic":
                    j += 1
            upload_normal_shape = (n_rows, n_nf)

    # translate_interactions (contract_adapter.rs:674-703)
    interactions = []
    for kind, body in gen_terms:
        if kind == "Interaction":
            interactions.append([_term_to_kernel_col(cols, gen_terms, _variant(comp)) for comp in body["components"]])

    # translate_effect_sizes (contract_adapter.rs:471-525)
    n_total = 1 + n_nf + n_fd + len(interactions)
    effects = [0.0] * n_total
    k_int = 0
    for term, beta in zip(gen_terms, c["outcome"]["coefficients"]):
        if term[0] == "Interaction":
            col = 1 + n_nf + n_fd + k_int
            k_int += 1
        else:
            col = _term_to_kernel_col(cols, gen_terms, term)
        effects[col] = float(beta)
    effects[0] = float(c["outcome"]["intercept"])

    # expand_cluster_level_columns (contract_adapter.rs:781-815)
    between = []
    for cid in gen.get("cluster_level_columns", []) or []:
        idx = _col_id(cid)
        kind, body = cols.cols[idx]
        if kind in _NON_FACTOR:
            between.append(cols.continuous(idx))
        else:
            between.extend(cols.dummy(idx, li) for li in range(1, int(body["n_levels"])))

    target_indices, _, contrast_pairs, _ = targets_and_contrasts(c)

    # translate_posthoc (contract_adapter.rs:834-877)
    posthoc, posthoc_alpha = [], None
    if c.get("posthoc"):
        for ph in c["posthoc"]:
            fcol = _col_id(ph["factor_column"])
            tix = []
            for t in ph["target_term_indices"]:
                tk, tb = test_terms[int(t)]
                tix.append(cols.dummy(_col_id(tb["column"]), int(tb["level_index"])))
            posthoc.append(S.PosthocSpec(factor_index=cols.factor_pos[fcol], target_indices=tix))
        posthoc_alpha = c["posthoc"][0].get("posthoc_alpha")

    drv = c["outcome"].get("heteroskedasticity_driver")
    het_driver = None if drv is None else cols.continuous(_col_id(drv))

    cluster = None
    slope_cols: List[int] = []
    extra_slope_cols: List[List[int]] = []
    cl = gen.get("cluster")
    if cl is not None:
        skind, sbody = _variant(cl["sizing"])
        size = sbody["n_clusters"] if skind == "FixedClusters" else sbody["cluster_size"]
        cluster = S.ClusterSpec(sizing=(skind, int(size)), tau_squared=float(cl["tau_squared"]),
                                slopes=list(cl.get("slopes", []) or []),
                                extra_groupings=list(cl.get("extra_groupings", []) or []))
        slope_cols = [cols.continuous(_col_id(s["column"])) for s in cluster.slopes]
        extra_slope_cols = [[cols.continuous(_col_id(s["column"])) for s in (g.get("slopes", []) or [])]
                            for g in cluster.extra_groupings]

    # fit_columns: only when an explicit design_test drops a generation column (contract_adapter.rs:157-172)
    fit_columns: List[int] = []
    if has_test_design:
        kept = sorted({_term_to_kernel_col(cols, gen_terms, t) for t in test_terms})
        if len(kept) != n_total:
            fit_columns = kept

    sc = c["scenario"]
    lme = None
    if sc.get("lme") is not None:
        lme = S.LmeScenarioPerturbations(
            random_effect_dist=_lookup(_RESID, sc["lme"]["random_effect_dist"], "residual distribution"),
            random_effect_df=float(sc["lme"]["random_effect_df"]), icc_noise_sd=float(sc["lme"]["icc_noise_sd"]))
    scenario = S.ScenarioPerturbations(
        name=str(sc["name"]), heterogeneity=float(sc["heterogeneity"]),
        heteroskedasticity_ratio=float(sc["heteroskedasticity_ratio"]),
        correlation_noise_sd=float(sc["correlation_noise_sd"]),
        distribution_change_prob=float(sc["distribution_change_prob"]),
        new_distributions=[_lookup(_SYNTH, _variant(k)[0], "SyntheticKind") for k in sc["new_distributions"]],
        residual_change_prob=float(sc["residual_change_prob"]),
        residual_dists=[_lookup(_RESID, r, "residual distribution") for r in sc["residual_dists"]],
        residual_df=float(sc["residual_df"]),
        sampled_factor_proportions=bool(sc.get("sampled_factor_proportions", False)),
        truth_start=bool(sc.get("truth_start", False)), lme=lme)

    link = c["outcome"].get("link")
    if link is not None and link != "probit":
        raise ContractError(f"unknown link {link!r}")

    return S.SimulationSpec(
        n_non_factor=n_nf, n_factor_dummies=n_fd, correlation=correlation,
        var_types=var_types, var_pinned=var_pinned, var_params=var_params,
        upload_normal=upload_normal, upload_normal_shape=upload_normal_shape,
        upload_data=upload_data, upload_data_shape=upload_data_shape, bootstrap_frame_map=bootstrap_frame_map,
        between_var_indices=between, factor_n_levels=factor_n_levels, factor_proportions=factor_proportions,
        factor_sampled=factor_sampled, effect_sizes=effects, target_indices=target_indices,
        fit_columns=fit_columns, contrast_pairs=contrast_pairs, interactions=interactions,
        correction_method=_lookup(_CORRECTION, c["test"]["correction"], "correction"),
        crit_values=S.CritValues(alpha=float(c["test"]["alpha"]), posthoc_alpha=posthoc_alpha),
        heteroskedasticity_driver=het_driver, cluster_slope_design_cols=slope_cols,
        extra_slope_cols=extra_slope_cols,
        residual_dist=_lookup(_RESID, c["outcome"]["residual"]["distribution"], "residual distribution"),
        residual_pinned=bool(c["outcome"]["residual"].get("pinned", False)),
        outcome_kind=_lookup(_OUTCOME, c["outcome"]["kind"], "outcome kind"),
        link=None if link is None else S.LinkKind.Probit,
        estimator=_lookup(_ESTIMATOR, c["estimator"], "estimator"),
        wald_se=wald_se,
        nagq=int(c.get("nagq", 1)) if (c.get("estimator") == "glm" and c["generation"].get("cluster") is not None) else 1,
        intercept=float(c["outcome"]["intercept"]), posthoc=posthoc,
        max_failed_fraction=float(c["max_failed_fraction"]), cluster=cluster, scenario=scenario,
        report_overall=report_overall, factor_min_level_count=0)


def cluster_atom(c: dict) -> int:
    """`ClusterSpec::atom` (generation.rs:188-192): sizing atom x every extra grouping's block levels."""
    cl = c["generation"].get("cluster")
    if cl is None:
        return 1
    _, sbody = _variant(cl["sizing"])
    a = max(int(next(iter(sbody.values()))), 1)
    for g in cl.get("extra_groupings", []) or []:
        _, rbody = _variant(g["relation"])
        a *= max(int(next(iter(rbody.values()))), 1)
    return a
