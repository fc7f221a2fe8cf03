"""Host-side mirror of the reference's kernel-facing types.

`SimulationSpec`, `ScenarioPerturbations`, `ClusterSpec`, `BatchResult` follow
`/root/reference/crates/engine-core/src/spec.rs:118-237, 251-417, 535-600` and
`crates/engine-contract/src/generation.rs:121-171` field for field (same names,
same defaults), so tests written against the reference's `run_batch` read the
same here.  `to_c()` lowers a spec to the flat `mcp_spec` POD of
`include/mcpower_hip.h` (plain pointers and sizes) for the C ABI.
"""
from __future__ import annotations

import ctypes as C
import enum
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

ABI_VERSION = 3


class Distribution(enum.IntEnum):  # engine_contract::Distribution
    Normal = 0
    Binary = 1
    RightSkewed = 2
    LeftSkewed = 3
    HighKurtosis = 4
    Uniform = 5
    UploadedData = 6
    UploadedBinary = 7
    UploadedFactor = 8


class ResidualDist(enum.IntEnum):
    Normal = 0
    HighKurtosis = 1
    RightSkewed = 2
    LeftSkewed = 3
    Uniform = 4


class OutcomeKind(enum.IntEnum):
    Continuous = 0
    Binary = 1
    Count = 2


class LinkKind(enum.IntEnum):
    Canonical = 0  # spec.link == None
    Probit = 1


class WaldSe(enum.IntEnum):
    """engine-contract estimator.rs:39-56: fixed-effect Wald SE of the clustered GLM."""
    Rx = 0
    Hessian = 1


class EstimatorSpec(enum.IntEnum):
    Ols = 0
    Glm = 1
    Mle = 2


class CorrectionMethod(enum.IntEnum):
    None_ = 0
    Bonferroni = 1
    Holm = 2
    BenjaminiHochberg = 3
    TukeyHsd = 4


class Status(enum.IntEnum):  # EngineError (spec.rs:20-29) + device-path codes
    OK = 0
    InvalidSpec = 1
    CorrelationNotPSD = 2
    RankDeficient = 3
    Cancelled = 4
    Unsupported = 5
    Device = 6


class EngineError(Exception):
    """`EngineError` of spec.rs:20-29; `.status` carries the variant."""

    def __init__(self, status: int, message: str):
        super().__init__(message)
        self.status = Status(status)


# ---------------------------------------------------------------------------
# ctypes PODs -- must match include/mcpower_hip.h exactly
# ---------------------------------------------------------------------------
_pd = C.POINTER(C.c_double)
_pi32 = C.POINTER(C.c_int32)
_pu32 = C.POINTER(C.c_uint32)
_pu8 = C.POINTER(C.c_uint8)
_pi8 = C.POINTER(C.c_int8)
_pu64 = C.POINTER(C.c_uint64)
_pf = C.POINTER(C.c_float)


class CScenario(C.Structure):
    _fields_ = [
        ("heterogeneity", C.c_double),
        ("heteroskedasticity_ratio", C.c_double),
        ("correlation_noise_sd", C.c_double),
        ("distribution_change_prob", C.c_double),
        ("new_distributions", _pi32),
        ("n_new_distributions", C.c_uint32),
        ("residual_change_prob", C.c_double),
        ("residual_dists", _pi32),
        ("n_residual_dists", C.c_uint32),
        ("residual_df", C.c_double),
        ("sampled_factor_proportions", C.c_uint8),
        ("truth_start", C.c_uint8),
        ("has_lme", C.c_uint8),
        ("_pad0", C.c_uint8),
        ("lme_random_effect_dist", C.c_int32),
        ("lme_random_effect_df", C.c_double),
        ("lme_icc_noise_sd", C.c_double),
    ]


MAX_RE_SLOPES = 2
MAX_EXTRA_GROUPINGS = 2


class CReSlopes(C.Structure):
    _fields_ = [
        ("n_slopes", C.c_uint32),
        ("design_cols", C.c_uint32 * MAX_RE_SLOPES),
        ("_pad", C.c_uint32),
        ("variance", C.c_double * MAX_RE_SLOPES),
        ("re_corr", C.c_double * ((1 + MAX_RE_SLOPES) ** 2)),
    ]


class CGrouping(C.Structure):
    _fields_ = [
        ("relation", C.c_int32),
        ("count", C.c_uint32),
        ("tau_squared", C.c_double),
        ("slopes", CReSlopes),
    ]


class CCluster(C.Structure):
    _fields_ = [
        ("present", C.c_uint8),
        ("_pad", C.c_uint8 * 3),
        ("sizing", C.c_int32),
        ("sizing_value", C.c_uint32),
        ("n_extra_groupings", C.c_uint32),
        ("tau_squared", C.c_double),
        ("slopes", CReSlopes),
        ("extra", CGrouping * MAX_EXTRA_GROUPINGS),
    ]


def re_correlation_matrix(slopes) -> List[float]:
    """`re_correlation_from_slopes` (generation.rs:226-241): row-major q x q over [intercept, slope_0, ...]."""
    q = 1 + len(slopes)
    r = [0.0] * (q * q)
    for d in range(q):
        r[d * q + d] = 1.0
    for k, s in enumerate(slopes):
        r[k + 1] = r[(k + 1) * q] = float(s.get("corr_with_intercept", 0.0))
        for i, cik in enumerate(s.get("corr_with", []) or []):
            r[(i + 1) * q + (k + 1)] = r[(k + 1) * q + (i + 1)] = float(cik)
    return r


def _fill_re_slopes(dst: CReSlopes, slopes, design_cols, what: str):
    if len(slopes) > MAX_RE_SLOPES:
        raise EngineError(Status.Unsupported, f"unsupported on the device path: more than {MAX_RE_SLOPES} random slopes on {what}")
    if len(design_cols) != len(slopes):
        raise EngineError(Status.InvalidSpec, f"invalid spec: {what} has {len(slopes)} random slopes but {len(design_cols)} resolved design columns")
    dst.n_slopes = len(slopes)
    for k, (s, col) in enumerate(zip(slopes, design_cols)):
        dst.design_cols[k] = int(col)
        dst.variance[k] = float(s["variance"])
    for i, v in enumerate(re_correlation_matrix(slopes)):
        dst.re_corr[i] = v


class CSpec(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("n_non_factor", C.c_uint32),
        ("n_factor_dummies", C.c_uint32),
        ("correlation", _pd),
        ("var_types", _pi32),
        ("var_pinned", _pu8),
        ("n_var_pinned", C.c_uint32),
        ("var_params", _pd),
        ("n_var_params", C.c_uint32),
        ("upload_normal", _pd),
        ("upload_normal_rows", C.c_uint32),
        ("upload_normal_cols", C.c_uint32),
        ("upload_data", _pd),
        ("upload_data_rows", C.c_uint32),
        ("upload_data_cols", C.c_uint32),
        ("bootstrap_frame_map", _pi32),
        ("n_bootstrap_frame_map", C.c_uint32),
        ("between_var_indices", _pu32),
        ("n_between_var_indices", C.c_uint32),
        ("factor_n_levels", _pi32),
        ("n_factors", C.c_uint32),
        ("factor_proportions", _pd),
        ("n_factor_proportions", C.c_uint32),
        ("factor_sampled", _pi8),
        ("n_factor_sampled", C.c_uint32),
        ("effect_sizes", _pd),
        ("n_effect_sizes", C.c_uint32),
        ("target_indices", _pu32),
        ("n_target_indices", C.c_uint32),
        ("fit_columns", _pu32),
        ("n_fit_columns", C.c_uint32),
        ("contrast_pairs", _pu32),
        ("n_contrast_pairs", C.c_uint32),
        ("interaction_components", _pu32),
        ("interaction_offsets", _pu32),
        ("n_interactions", C.c_uint32),
        ("correction_method", C.c_int32),
        ("alpha", C.c_double),
        ("posthoc_alpha", C.c_double),
        ("heteroskedasticity_driver", C.c_int32),
        ("residual_dist", C.c_int32),
        ("residual_pinned", C.c_uint8),
        ("report_overall", C.c_uint8),
        ("wald_se", C.c_uint8),
        ("nagq", C.c_uint8),
        ("outcome_kind", C.c_int32),
        ("link", C.c_int32),
        ("estimator", C.c_int32),
        ("intercept", C.c_double),
        ("posthoc_factor_index", _pu32),
        ("posthoc_target_offsets", _pu32),
        ("posthoc_target_indices", _pu32),
        ("n_posthoc", C.c_uint32),
        ("max_failed_fraction", C.c_double),
        ("factor_min_level_count", C.c_uint32),
        ("_pad2", C.c_uint32),
        ("cluster", CCluster),
        ("scenario", CScenario),
    ]


class CCounterLayout(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "stride", "k_unc", "k_cor", "converged", "overall", "hist_unc", "hist_cor", "posthoc_unc",
        "posthoc_cor", "excl_sparse", "excl_sep", "joint_unc", "joint_cor", "boundary_1", "boundary_2", "n_sims")]


class CBatchResult(C.Structure):
    _fields_ = [
        ("n_sims", C.c_uint32), ("n_sample_sizes", C.c_uint32), ("n_targets", C.c_uint32), ("n_factors", C.c_uint32),
        ("n_posthoc_contrasts", C.c_uint32), ("n_variance_components", C.c_uint32),
        ("uncorrected", _pu8), ("corrected", _pu8), ("posthoc_unc", _pu8), ("posthoc_cor", _pu8),
        ("converged", _pu8), ("boundary_hit", _pu8), ("pinned_components", _pu64),
        ("joint_unc", _pu8), ("joint_cor", _pu8), ("overall", _pu8), ("factor_excluded", _pu8),
        ("tau_squared_hat", _pd), ("stat_capture", _pd),
        ("counters", _pu64), ("layout", CCounterLayout),
        ("device_ms", C.c_double), ("kernel_ms", C.c_double), ("aux_kernel_ms", C.c_double),
        ("n_kernel_launches", C.c_uint32), ("_pad", C.c_uint32),
        ("kernel_begin_ms", C.c_double), ("kernel_end_ms", C.c_double),
    ]


class CRunOptions(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("chunk_sims", C.c_uint32), ("sim_offset", C.c_uint64), ("hip_stream", C.c_void_p)]


WANT_PER_SIM = 0x1
WANT_STAT_CAPTURE = 0x2
PIPELINE_STAGED = 0x4
SEED_PER_REPLICATE = 0x10  # validation aid (clustered GLM): replicate i = replicate 0 of call seed splitmix64(base_seed + i)
ALLREDUCE_COUNTERS = 0x8  # pool the counters over the process's RCCL communicator inside the call

PROGRESS_FN = C.CFUNCTYPE(C.c_int, C.c_uint64, C.c_uint64, C.c_void_p)


# ---------------------------------------------------------------------------
# dataclasses (reference field names)
# ---------------------------------------------------------------------------
@dataclass
class LmeScenarioPerturbations:
    random_effect_dist: ResidualDist = ResidualDist.Normal
    random_effect_df: float = 0.0
    icc_noise_sd: float = 0.0


@dataclass
class ScenarioPerturbations:  # spec.rs:118-199
    name: str = ""
    heterogeneity: float = 0.0
    heteroskedasticity_ratio: float = 1.0
    correlation_noise_sd: float = 0.0
    distribution_change_prob: float = 0.0
    new_distributions: List[Distribution] = field(default_factory=list)
    residual_change_prob: float = 0.0
    residual_dists: List[ResidualDist] = field(default_factory=list)
    residual_df: float = 0.0
    sampled_factor_proportions: bool = False
    truth_start: bool = False
    lme: Optional[LmeScenarioPerturbations] = None

    @staticmethod
    def optimistic() -> "ScenarioPerturbations":
        return ScenarioPerturbations(name="optimistic")

    def is_optimistic(self) -> bool:  # spec.rs:229-236
        return (self.heterogeneity == 0.0 and self.heteroskedasticity_ratio == 1.0
                and self.correlation_noise_sd == 0.0 and self.distribution_change_prob == 0.0
                and self.residual_change_prob == 0.0 and self.lme is None)


@dataclass
class ClusterSpec:  # generation.rs:171 (sizing: ("FixedClusters", n) | ("FixedSize", n))
    sizing: Tuple[str, int] = ("FixedClusters", 1)
    tau_squared: float = 0.0
    slopes: list = field(default_factory=list)
    extra_groupings: list = field(default_factory=list)

    def n_clusters_at(self, n: int) -> int:
        kind, v = self.sizing
        v = max(int(v), 1)
        return v if kind == "FixedClusters" else n // v


@dataclass
class PosthocSpec:
    factor_index: int
    target_indices: List[int]


@dataclass
class CritValues:
    alpha: float = 0.05
    posthoc_alpha: Optional[float] = None


@dataclass
class SimulationSpec:  # spec.rs:251-417
    n_non_factor: int = 0
    n_factor_dummies: int = 0
    correlation: Sequence[float] = field(default_factory=list)
    var_types: List[Distribution] = field(default_factory=list)
    var_pinned: List[bool] = field(default_factory=list)
    var_params: List[float] = field(default_factory=list)
    upload_normal: Sequence[float] = field(default_factory=list)
    upload_normal_shape: Tuple[int, int] = (0, 0)
    upload_data: Sequence[float] = field(default_factory=list)
    upload_data_shape: Tuple[int, int] = (0, 0)
    bootstrap_frame_map: List[Optional[int]] = field(default_factory=list)
    between_var_indices: List[int] = field(default_factory=list)
    factor_n_levels: List[int] = field(default_factory=list)
    factor_proportions: List[float] = field(default_factory=list)
    factor_sampled: List[Optional[bool]] = field(default_factory=list)
    effect_sizes: List[float] = field(default_factory=list)
    target_indices: List[int] = field(default_factory=list)
    fit_columns: List[int] = field(default_factory=list)
    contrast_pairs: List[Tuple[int, int]] = field(default_factory=list)
    interactions: List[List[int]] = field(default_factory=list)
    correction_method: CorrectionMethod = CorrectionMethod.None_
    crit_values: CritValues = field(default_factory=CritValues)
    heteroskedasticity_driver: Optional[int] = None
    cluster_slope_design_cols: List[int] = field(default_factory=list)
    extra_slope_cols: List[List[int]] = field(default_factory=list)
    residual_dist: ResidualDist = ResidualDist.Normal
    residual_pinned: bool = False
    outcome_kind: OutcomeKind = OutcomeKind.Continuous
    link: Optional[LinkKind] = None
    estimator: EstimatorSpec = EstimatorSpec.Ols
    intercept: float = 0.0
    posthoc: List[PosthocSpec] = field(default_factory=list)
    max_failed_fraction: float = 0.1
    cluster: Optional[ClusterSpec] = None
    scenario: ScenarioPerturbations = field(default_factory=ScenarioPerturbations.optimistic)
    report_overall: bool = False
    factor_min_level_count: int = 0
    wald_se: WaldSe = WaldSe.Rx  # spec.rs:359-362: clustered GLM only
    nagq: int = 1                # spec.rs:363-366: 1 = Laplace

    @property
    def n_predictors(self) -> int:
        return 1 + self.n_non_factor + self.n_factor_dummies + len(self.interactions)

    @property
    def n_targets(self) -> int:
        return len(self.target_indices) + len(self.contrast_pairs)

    # -- lowering to the C ABI ------------------------------------------------
    def to_c(self) -> "LoweredSpec":
        return LoweredSpec(self)


def _arr(values, dtype):
    a = np.ascontiguousarray(np.asarray(list(values) if not isinstance(values, np.ndarray) else values, dtype=dtype))
    return a


class LoweredSpec:
    """Owns the numpy buffers a `CSpec` points into (keep alive during the call)."""

    def __init__(self, s: SimulationSpec):
        self._keep = []
        c = CSpec()
        c.abi_version = ABI_VERSION
        c.n_non_factor = s.n_non_factor
        c.n_factor_dummies = s.n_factor_dummies

        def ptr(values, dtype, ctype):
            a = _arr(values, dtype)
            self._keep.append(a)
            return a.ctypes.data_as(C.POINTER(ctype)), a.size

        c.correlation, _ = ptr(s.correlation, np.float64, C.c_double)
        c.var_types, _ = ptr([int(v) for v in s.var_types], np.int32, C.c_int32)
        c.var_pinned, c.n_var_pinned = ptr([1 if v else 0 for v in s.var_pinned], np.uint8, C.c_uint8)
        c.var_params, c.n_var_params = ptr(s.var_params, np.float64, C.c_double)
        c.upload_normal, _ = ptr(s.upload_normal, np.float64, C.c_double)
        c.upload_normal_rows, c.upload_normal_cols = s.upload_normal_shape
        c.upload_data, _ = ptr(s.upload_data, np.float64, C.c_double)
        c.upload_data_rows, c.upload_data_cols = s.upload_data_shape
        c.bootstrap_frame_map, c.n_bootstrap_frame_map = ptr(
            [-1 if v is None else int(v) for v in s.bootstrap_frame_map], np.int32, C.c_int32)
        c.between_var_indices, c.n_between_var_indices = ptr(s.between_var_indices, np.uint32, C.c_uint32)
        c.factor_n_levels, c.n_factors = ptr(s.factor_n_levels, np.int32, C.c_int32)
        c.factor_proportions, c.n_factor_proportions = ptr(s.factor_proportions, np.float64, C.c_double)
        c.factor_sampled, c.n_factor_sampled = ptr(
            [-1 if v is None else (1 if v else 0) for v in s.factor_sampled], np.int8, C.c_int8)
        c.effect_sizes, c.n_effect_sizes = ptr(s.effect_sizes, np.float64, C.c_double)
        c.target_indices, c.n_target_indices = ptr(s.target_indices, np.uint32, C.c_uint32)
        c.fit_columns, c.n_fit_columns = ptr(s.fit_columns, np.uint32, C.c_uint32)
        flat_pairs = [int(v) for pr in s.contrast_pairs for v in pr]
        c.contrast_pairs, _ = ptr(flat_pairs, np.uint32, C.c_uint32)
        c.n_contrast_pairs = len(s.contrast_pairs)
        comps, offs = [], [0]
        for it in s.interactions:
            comps.extend(int(v) for v in it)
            offs.append(len(comps))
        c.interaction_components, _ = ptr(comps, np.uint32, C.c_uint32)
        c.interaction_offsets, _ = ptr(offs, np.uint32, C.c_uint32)
        c.n_interactions = len(s.interactions)
        c.correction_method = int(s.correction_method)
        c.alpha = float(s.crit_values.alpha)
        c.posthoc_alpha = math.nan if s.crit_values.posthoc_alpha is None else float(s.crit_values.posthoc_alpha)
        c.heteroskedasticity_driver = -1 if s.heteroskedasticity_driver is None else int(s.heteroskedasticity_driver)
        c.residual_dist = int(s.residual_dist)
        c.residual_pinned = 1 if s.residual_pinned else 0
        c.report_overall = 1 if s.report_overall else 0
        c.wald_se = int(s.wald_se)
        c.nagq = int(s.nagq)
        c.outcome_kind = int(s.outcome_kind)
        c.link = int(LinkKind.Canonical if s.link is None else s.link)
        c.estimator = int(s.estimator)
        c.intercept = float(s.intercept)
        pf, po, pt = [], [0], []
        for blk in s.posthoc:
            pf.append(int(blk.factor_index))
            pt.extend(int(v) for v in blk.target_indices)
            po.append(len(pt))
        c.posthoc_factor_index, _ = ptr(pf, np.uint32, C.c_uint32)
        c.posthoc_target_offsets, _ = ptr(po, np.uint32, C.c_uint32)
        c.posthoc_target_indices, _ = ptr(pt, np.uint32, C.c_uint32)
        c.n_posthoc = len(s.posthoc)
        c.max_failed_fraction = float(s.max_failed_fraction)
        c.factor_min_level_count = int(s.factor_min_level_count)
        if s.cluster is not None:
            c.cluster.present = 1
            kind, v = s.cluster.sizing
            c.cluster.sizing = 0 if kind == "FixedClusters" else 1
            c.cluster.sizing_value = int(v)
            c.cluster.tau_squared = float(s.cluster.tau_squared)
            _fill_re_slopes(c.cluster.slopes, s.cluster.slopes, s.cluster_slope_design_cols, "the primary grouping")
            if len(s.cluster.extra_groupings) > MAX_EXTRA_GROUPINGS:
                raise EngineError(Status.Unsupported, f"unsupported on the device path: more than {MAX_EXTRA_GROUPINGS} extra groupings")
            c.cluster.n_extra_groupings = len(s.cluster.extra_groupings)
            for g, grp in enumerate(s.cluster.extra_groupings):
                (rel, body), = grp["relation"].items() if isinstance(grp["relation"], dict) else ((grp["relation"], {}),)
                dst = c.cluster.extra[g]
                dst.relation = 0 if rel == "Crossed" else 1
                dst.count = int(body["n_clusters"] if rel == "Crossed" else body["n_per_parent"])
                dst.tau_squared = float(grp["tau_squared"])
                cols = s.extra_slope_cols[g] if g < len(s.extra_slope_cols) else []
                _fill_re_slopes(dst.slopes, grp.get("slopes", []) or [], cols, f"extra grouping {g}")
        sc = s.scenario
        c.scenario.heterogeneity = sc.heterogeneity
        c.scenario.heteroskedasticity_ratio = sc.heteroskedasticity_ratio
        c.scenario.correlation_noise_sd = sc.correlation_noise_sd
        c.scenario.distribution_change_prob = sc.distribution_change_prob
        c.scenario.new_distributions, c.scenario.n_new_distributions = ptr(
            [int(v) for v in sc.new_distributions], np.int32, C.c_int32)
        c.scenario.residual_change_prob = sc.residual_change_prob
        c.scenario.residual_dists, c.scenario.n_residual_dists = ptr(
            [int(v) for v in sc.residual_dists], np.int32, C.c_int32)
        c.scenario.residual_df = sc.residual_df
        c.scenario.sampled_factor_proportions = 1 if sc.sampled_factor_proportions else 0
        c.scenario.truth_start = 1 if sc.truth_start else 0
        if sc.lme is not None:
            c.scenario.has_lme = 1
            c.scenario.lme_random_effect_dist = int(sc.lme.random_effect_dist)
            c.scenario.lme_random_effect_df = sc.lme.random_effect_df
            c.scenario.lme_icc_noise_sd = sc.lme.icc_noise_sd
        self.c = c

    def byref(self):
        return C.byref(self.c)


# ---------------------------------------------------------------------------
# BatchResult (spec.rs:553-600) -- numpy views copied out of the C result
# ---------------------------------------------------------------------------
@dataclass
class ResultShape:
    n_sims: int
    n_sample_sizes: int
    n_targets: int
    n_factors: int
    n_posthoc_contrasts: int
    n_variance_components: int


@dataclass
class BatchResult:
    shape: ResultShape
    uncorrected: Optional[np.ndarray]      # (n_sims, n_ss, n_targets) u8
    corrected: Optional[np.ndarray]
    posthoc_unc: Optional[np.ndarray]
    posthoc_cor: Optional[np.ndarray]
    converged: Optional[np.ndarray]        # (n_sims, n_ss) u8
    boundary_hit: Optional[np.ndarray]
    pinned_components: Optional[np.ndarray]
    joint_unc: Optional[np.ndarray]
    joint_cor: Optional[np.ndarray]
    overall: Optional[np.ndarray]
    factor_excluded: Optional[np.ndarray]
    tau_squared_hat: Optional[np.ndarray]
    stat_capture: Optional[np.ndarray]     # (n_sims, n_targets) f64
    counters: np.ndarray                   # (n_ss, stride) u64
    layout: dict
    device_ms: float = 0.0
    kernel_ms: float = 0.0
    aux_kernel_ms: float = 0.0
    n_kernel_launches: int = 0
    kernel_begin_ms: float = 0.0           # first / last dominant-kernel launch of the call, ms since the host thread's
    kernel_end_ms: float = 0.0             # first call: overlapping batches -> take the union of these intervals

    def counter(self, name: str, width: int = 1) -> np.ndarray:
        """(n_ss, width) block of the additive counters, e.g. counter('k_unc', n_targets)."""
        off = self.layout[name]
        return self.counters[:, off:off + width]


def result_from_c(r: CBatchResult) -> BatchResult:
    ns, nss, nt, nf, npc = r.n_sims, r.n_sample_sizes, r.n_targets, r.n_factors, r.n_posthoc_contrasts

    def grab(p, shape, dtype):
        if not p:
            return None
        n = int(np.prod(shape))
        if n == 0:
            return np.zeros(shape, dtype=dtype)
        return np.ctypeslib.as_array(p, shape=(n,)).astype(dtype, copy=True).reshape(shape)

    layout = {n: getattr(r.layout, n) for n, _ in CCounterLayout._fields_}
    return BatchResult(
        shape=ResultShape(ns, nss, nt, nf, npc, r.n_variance_components),
        uncorrected=grab(r.uncorrected, (ns, nss, nt), np.uint8),
        corrected=grab(r.corrected, (ns, nss, nt), np.uint8),
        posthoc_unc=grab(r.posthoc_unc, (ns, nss, npc), np.uint8),
        posthoc_cor=grab(r.posthoc_cor, (ns, nss, npc), np.uint8),
        converged=grab(r.converged, (ns, nss), np.uint8),
        boundary_hit=grab(r.boundary_hit, (ns, nss), np.uint8),
        pinned_components=grab(r.pinned_components, (ns, nss), np.uint64),
        joint_unc=grab(r.joint_unc, (ns, nss), np.uint8),
        joint_cor=grab(r.joint_cor, (ns, nss), np.uint8),
        overall=grab(r.overall, (ns, nss), np.uint8),
        factor_excluded=grab(r.factor_excluded, (ns, nss, nf), np.uint8),
        tau_squared_hat=grab(r.tau_squared_hat, (ns, nss), np.float64),
        stat_capture=grab(r.stat_capture, (ns, nt), np.float64),
        counters=grab(r.counters, (nss, r.layout.stride), np.uint64),
        layout=layout,
        device_ms=r.device_ms,
        kernel_ms=r.kernel_ms,
        aux_kernel_ms=r.aux_kernel_ms,
        n_kernel_launches=r.n_kernel_launches,
        kernel_begin_ms=r.kernel_begin_ms,
        kernel_end_ms=r.kernel_end_ms,
    )
