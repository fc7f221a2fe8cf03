"""Drop-in for the compiled `mcpower._engine` extension's power entry points, backed by the HIP engine.

Mirrors `crates/engine-py/src/lib.rs:46-81` (`find_power`) over
`crates/engine-orchestrator/src/find_power.rs` (cluster-atom snap, contract lowering with the call
seed `splitmix64(base_seed)` shared by every scenario, one `run_batch` per scenario, sparse-factor
preflight warnings), `aggregation.rs:62-291` (`aggregate_batch`) and `result_host.rs:64-146`
(`power_result_to_host`: field names, field order, leaf shapes).  A host that does
`from mcpower import _engine` can bind this module instead and get the same dict.

The replicates run through `mcp_run_batch` of `libmcpower_hip.so`; there is no CPU path here.
"""
from __future__ import annotations

import math
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import contract as K
from . import spec as S

# configs/config.json `limits` (embedded in the reference at build time; find_power.rs:268, :43)
FACTOR_MIN_LEVEL_COUNT = 5
MIN_CLUSTERS = 5

_MASK64 = (1 << 64) - 1


def splitmix64_finalize(x: int) -> int:
    """`rng::splitmix64_finalize` (rng.rs:30, = `rand_philox::splitmix64`): the bare Mix13 finalizer,
    no golden-gamma increment (pinned by tests/golden/rng_kat.json)."""
    x &= _MASK64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _MASK64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _MASK64
    return x ^ (x >> 31)


class OrchestratorError(RuntimeError):
    """`OrchestratorError` (engine-orchestrator/src/result.rs); `.kind` carries the variant name."""

    def __init__(self, kind: str, message: str):
        super().__init__(message)
        self.kind = kind


def set_n_threads(n: int) -> None:
    """`set_n_threads` sizes the reference's rayon pool; replicates here are device lanes, so this only
    validates its argument."""
    if int(n) < 0:
        raise ValueError("n must be >= 0")


# ---------------------------------------------------------------------------------------------
# aggregation.rs
# ---------------------------------------------------------------------------------------------
def wilson_ci(k: int, n: int) -> Tuple[float, float]:
    """`wilson_ci` (aggregation.rs:19-35), z = 1.96."""
    z = 1.96
    if n == 0:
        return (0.0, 1.0)
    n_f = float(n)
    p = k / n_f
    z2 = z * z
    denom = 1.0 + z2 / n_f
    center = (p + z2 / (2.0 * n_f)) / denom
    margin = z * math.sqrt(p * (1.0 - p) / n_f + z2 / (4.0 * n_f * n_f)) / denom
    return (max(center - margin, 0.0), min(center + margin, 1.0))


def _rate(k: int, n: int) -> float:
    return 0.0 if n == 0 else k / n


def fixed_allocation_counts(probs: Sequence[float], n: int) -> List[int]:
    """`fixed_allocation_counts` (data_gen.rs:966-972) over `fixed_level_next` (data_gen.rs:943-961)."""
    total = 0.0
    for p in probs:
        total += p
    counts = [0] * len(probs)
    for row in range(n):
        t = float(row + 1)
        best, best_deficit = 0, -math.inf
        for g, p in enumerate(probs):
            deficit = (p / total) * t - counts[g]
            if deficit > best_deficit:
                best_deficit, best = deficit, g
        counts[best] += 1
    return counts


def factor_preflight_warnings(spec: S.SimulationSpec, n: int) -> List[str]:
    """`factor_preflight_warnings` (find_power.rs:69-121)."""
    k_min = spec.factor_min_level_count
    if k_min == 0 or len(spec.upload_data) > 0:
        return []
    out, off = [], 0
    for f, nl in enumerate(spec.factor_n_levels):
        l = max(int(nl), 0)
        sampled = spec.factor_sampled[f] if f < len(spec.factor_sampled) else None
        if sampled is None:
            sampled = spec.scenario.sampled_factor_proportions
        if not sampled:
            counts = fixed_allocation_counts(spec.factor_proportions[off:off + l], n)
            if counts:
                lvl = min(range(len(counts)), key=lambda i: counts[i])
                c = counts[lvl]
                if c < k_min:
                    out.append(
                        f"factor {f + 1}: level {lvl + 1} receives {c} of {n} observations (minimum {k_min}); "
                        f"the factor is excluded from every simulation at N={n} and its "
                        f"effects report power 0 — increase N or this level's proportion")
        off += l
    return out


def aggregate_batch(batch: S.BatchResult, target_indices: Sequence[int], contrast_pairs: Sequence[Tuple[int, int]],
                    estimator: S.EstimatorSpec, posthoc_blocks: Sequence[Tuple[int, int]] = ()) -> List[dict]:
    """`aggregate_batch` (aggregation.rs:62-291): one PowerResult-shaped dict per sample size.
    `posthoc_blocks` = [(n_levels, n_contrasts), ...] (`ResultShape::posthoc_blocks`)."""
    sh = batch.shape
    n_sims, n_ss, n_targets, n_factors = sh.n_sims, sh.n_sample_sizes, sh.n_targets, sh.n_factors
    total_contrasts = sum(w for _, w in posthoc_blocks)
    out = []
    for ss in range(n_ss):
        unc = batch.uncorrected[:, ss, :] if n_sims else np.zeros((0, n_targets), np.uint8)
        cor = batch.corrected[:, ss, :] if n_sims else np.zeros((0, n_targets), np.uint8)
        k_unc = [int(v) for v in unc.sum(axis=0, dtype=np.uint64)]
        k_cor = [int(v) for v in cor.sum(axis=0, dtype=np.uint64)]
        conv = batch.converged[:, ss] if n_sims else np.zeros(0, np.uint8)
        bh = batch.boundary_hit[:, ss] if n_sims else np.zeros(0, np.uint8)
        conv_count = int((conv != 0).sum()) if n_sims else 0
        # Rust sums the raw u8 values; the engine only ever writes 0/1 there
        conv_rate = 1.0 if n_sims == 0 else int(conv.sum(dtype=np.uint64)) / n_sims

        extras = _estimator_extras(estimator, batch, ss)

        posthoc = []
        ph_unc = ph_cor = None
        if posthoc_blocks:
            ph_unc = batch.posthoc_unc[:, ss, :]
            ph_cor = batch.posthoc_cor[:, ss, :]
            offset = 0
            for n_levels, w in posthoc_blocks:
                ku = [int(v) for v in ph_unc[:, offset:offset + w].sum(axis=0, dtype=np.uint64)]
                kc = [int(v) for v in ph_cor[:, offset:offset + w].sum(axis=0, dtype=np.uint64)]
                posthoc.append(dict(
                    n_levels=int(n_levels),
                    power_uncorrected=[_rate(k, n_sims) for k in ku],
                    power_corrected=[_rate(k, n_sims) for k in kc],
                    ci_uncorrected=[wilson_ci(k, n_sims) for k in ku],
                    ci_corrected=[wilson_ci(k, n_sims) for k in kc],
                    success_counts_uncorrected=ku, success_counts_corrected=kc))
                offset += w

        overall_rate = overall_ci = None
        overall_count = 0
        if batch.overall is not None and batch.overall.size > 0:
            overall_count = int(batch.overall[:, ss].sum(dtype=np.uint64))
            overall_rate = _rate(overall_count, n_sims)
            overall_ci = wilson_ci(overall_count, n_sims)

        n_joint = n_targets + total_contrasts
        per_sim_unc = (unc != 0).sum(axis=1)
        per_sim_cor = (cor != 0).sum(axis=1)
        if total_contrasts:
            per_sim_unc = per_sim_unc + (ph_unc != 0).sum(axis=1)
            per_sim_cor = per_sim_cor + (ph_cor != 0).sum(axis=1)
        hist_unc = [int(v) for v in np.bincount(per_sim_unc, minlength=n_joint + 1)]
        hist_cor = [int(v) for v in np.bincount(per_sim_cor, minlength=n_joint + 1)]

        excl_sparse, excl_sep = [0] * n_factors, [0] * n_factors
        if n_factors > 0 and batch.factor_excluded is not None and batch.factor_excluded.size:
            fx = batch.factor_excluded[:, ss, :]
            excl_sparse = [int(v) for v in (fx == 1).sum(axis=0)]
            excl_sep = [int(v) for v in (fx == 2).sum(axis=0)]

        out.append(dict(
            n=0, n_sims=n_sims, target_indices=[int(i) for i in target_indices],
            contrast_pairs=[(int(a), int(b)) for a, b in contrast_pairs],
            power_uncorrected=[_rate(k, n_sims) for k in k_unc],
            power_corrected=[_rate(k, n_sims) for k in k_cor],
            ci_uncorrected=[wilson_ci(k, n_sims) for k in k_unc],
            ci_corrected=[wilson_ci(k, n_sims) for k in k_cor],
            convergence_rate=conv_rate, boundary_hit=[int(v) for v in bh], estimator_extras=extras,
            overall_significant_rate=overall_rate, success_counts_uncorrected=k_unc,
            success_counts_corrected=k_cor, convergence_count=conv_count,
            overall_significant_count=overall_count, overall_significant_ci=overall_ci,
            success_count_histogram_uncorrected=hist_unc, success_count_histogram_corrected=hist_cor,
            grid_warnings=[], posthoc=posthoc, factor_exclusion_counts=excl_sparse,
            factor_separation_counts=excl_sep))
    return out


def _estimator_extras(estimator: S.EstimatorSpec, batch: S.BatchResult, ss: int) -> dict:
    """`EstimatorExtras::from_batch` (result.rs:278-403) projected by `estimator_extras_to_host`
    (result_host.rs:366-424)."""
    n_sims = batch.shape.n_sims
    if estimator == S.EstimatorSpec.Ols:
        return {"estimator": "ols"}
    conv = batch.converged[:, ss] != 0 if n_sims else np.zeros(0, bool)
    bh = batch.boundary_hit[:, ss] if n_sims else np.zeros(0, np.uint8)
    singular_n = int(conv.sum())
    singular_count = int((conv & (bh == 1)).sum())
    singular_rate = singular_count / singular_n if singular_n > 0 else 0.0
    if estimator == S.EstimatorSpec.Glm:
        tau_mean = math.nan
        if batch.tau_squared_hat is not None and batch.tau_squared_hat.size:
            th = batch.tau_squared_hat[:, ss][conv]
            th = th[np.isfinite(th)]
            if th.size:
                tau_mean = float(th.sum() / th.size)
        return {"estimator": "glm", "baseline_prob_realized": math.nan, "singular_fit_rate": singular_rate,
                "tau_squared_hat_mean": tau_mean}
    k_ju = int(batch.joint_unc[:, ss].sum(dtype=np.uint64)) if n_sims else 0
    k_jc = int(batch.joint_cor[:, ss].sum(dtype=np.uint64)) if n_sims else 0
    n_comp = batch.shape.n_variance_components
    comp_rates = []
    for k in range(n_comp):
        bits = batch.pinned_components[:, ss][conv] if n_sims else np.zeros(0, np.uint64)
        c = int(((bits >> np.uint64(k)) & np.uint64(1)).sum())
        comp_rates.append(c / singular_n if singular_n > 0 else 0.0)
    return {"estimator": "mle", "tau_estimate": math.nan, "boundary_hits": int((bh != 0).sum()),
            "joint_uncorrected_rate": _rate(k_ju, n_sims), "joint_corrected_rate": _rate(k_jc, n_sims),
            "singular_fit_rate": singular_rate, "boundary_rate_per_component": comp_rates}


def power_result_to_host(pr: dict, label: str) -> dict:
    """`power_result_to_host` (result_host.rs:64-146) rendered as `host_value_to_py` does
    (engine-py/src/orchestrator_bridge.rs:129-182): Ci -> (lo, hi) tuple, BoundaryHit -> list of rows."""
    return {
        "n_sims": pr["n_sims"],
        "n_sample_sizes": 1,
        "n_targets": len(pr["target_indices"]),
        "target_indices": list(pr["target_indices"]),
        "contrast_pairs": [[p, n] for p, n in pr["contrast_pairs"]],
        "power_uncorrected": [list(pr["power_uncorrected"])],
        "power_corrected": [list(pr["power_corrected"])],
        "ci_uncorrected": [list(pr["ci_uncorrected"])],
        "ci_corrected": [list(pr["ci_corrected"])],
        "convergence_rate": [pr["convergence_rate"]],
        "scenario": label,
        "sample_sizes": [pr["n"]],
        "estimator_extras": pr["estimator_extras"],
        "boundary_hit": [[v] for v in pr["boundary_hit"]],
        "overall_significant_rate": pr["overall_significant_rate"],
        "overall_significant_ci": pr["overall_significant_ci"],
        "success_count_histogram_uncorrected": list(pr["success_count_histogram_uncorrected"]),
        "success_count_histogram_corrected": list(pr["success_count_histogram_corrected"]),
        "grid_warnings": list(pr["grid_warnings"]),
        "posthoc": [dict(b) for b in pr["posthoc"]],
        "factor_exclusion_counts": list(pr["factor_exclusion_counts"]),
        "factor_separation_counts": list(pr["factor_separation_counts"]),
    }


# ---------------------------------------------------------------------------------------------
# find_power.rs
# ---------------------------------------------------------------------------------------------
def snap_single_n(contracts: Sequence[dict], requested: int) -> Tuple[int, Optional[str]]:
    """`snap_single_n` (find_power.rs:25-58)."""
    cl = next((c["generation"].get("cluster") for c in contracts if c["generation"].get("cluster") is not None), None)
    if cl is None:
        return requested, None
    a = K.cluster_atom({"generation": {"cluster": cl}})
    if a <= 1:
        return requested, None
    snapped = (requested // a) * a
    if snapped == 0:
        snapped = a
    skind, sbody = K._variant(cl["sizing"])
    if skind == "FixedSize":
        cs = max(int(sbody["cluster_size"]), 1)
        got = snapped // cs
        if got < MIN_CLUSTERS:
            raise OrchestratorError(
                "ClusterTooFewAtN",
                f"at N={snapped} the FixedSize regime yields only {got} cluster(s) (cluster_size={cs}); need >= "
                f"{MIN_CLUSTERS}. Raise sample_size to at least {MIN_CLUSTERS}*{cs}.")
    warning = None
    if snapped != requested:
        warning = f"sample_size {requested} not a multiple of the cluster atom {a}; using {snapped}"
    return snapped, warning


def lower_contracts(contracts: Sequence[dict], base_seed: int) -> List[dict]:
    """`lower_contracts` (find_power.rs:243-283): every scenario gets the same call-level seed."""
    if not contracts:
        raise OrchestratorError("InvalidScenarios", "invalid scenarios input: contracts slice is empty")
    call_seed = splitmix64_finalize(int(base_seed) & _MASK64)
    out = []
    for idx, c in enumerate(contracts):
        label = c["scenario"]["name"]
        try:
            spec = K.contract_to_simulation_spec(c)
        except K.ContractError as e:
            raise OrchestratorError("InvalidScenarios",
                                    f'invalid scenarios input: contract {idx} ("{label}"): {e}') from e
        spec.factor_min_level_count = FACTOR_MIN_LEVEL_COUNT
        rep_t, rep_c = K.report_targets_and_contrasts(c)
        out.append(dict(label=label, spec=spec, base_seed=call_seed, report_target_indices=rep_t,
                        report_contrast_pairs=rep_c))
    return out


def _posthoc_blocks(spec: S.SimulationSpec) -> List[Tuple[int, int]]:
    """`PosthocBlockShape` per block: all pairwise contrasts of the factor's levels (posthoc.rs)."""
    blocks = []
    for blk in spec.posthoc:
        nl = int(spec.factor_n_levels[blk.factor_index])
        blocks.append((nl, nl * (nl - 1) // 2))
    return blocks


def _default_runner(spec, sample_sizes, n_sims, base_seed, sink):
    from . import engine

    return engine.run_batch(spec, sample_sizes, n_sims, base_seed, sink, per_sim=True)


def find_power(contracts_bytes: bytes, sample_size: int, n_sims: int, base_seed: int,
               progress: Optional[Callable[[int, int], bool]] = None, *, _runner=None) -> Dict[str, Any]:
    """`_engine.find_power(contracts_bytes, sample_size, n_sims, base_seed, progress=None)`
    (engine-py/src/lib.rs:46-81 over find_power.rs:297-381).

    `progress(current, total) -> bool`: cumulative fits over the whole call; return False to cancel.
    Returns `{"scenarios": {label: <power_result_to_host dict>}, "comparison": {}}`.
    """
    contracts = K.decode_contracts(contracts_bytes)
    if not contracts:
        raise OrchestratorError("InvalidScenarios", "invalid scenarios input: contracts slice is empty")
    sample_size, snap_warning = snap_single_n(contracts, int(sample_size))
    scenarios = lower_contracts(contracts, base_seed)
    runner = _runner or _default_runner
    total_sims = int(n_sims) * len(scenarios)
    cancelled = [False]

    out: Dict[str, Any] = {}
    for idx, sc in enumerate(scenarios):
        if cancelled[0]:
            raise OrchestratorError("Cancelled", f"cancelled at scenario {idx}, n=Some({sample_size})")
        offset = idx * int(n_sims)

        def sink(current: int, _batch_total: int, _offset=offset) -> bool:
            # EngineSinkAdapter::report (find_power.rs:211-229): rescale to call-level cumulative fits
            if cancelled[0]:
                return False
            if progress is not None:
                try:
                    keep = progress(_offset + int(current), total_sims)
                    keep = True if keep is None else bool(keep)
                except Exception:
                    keep = False
                if not keep:
                    cancelled[0] = True
            return not cancelled[0]

        spec: S.SimulationSpec = sc["spec"]
        try:
            batch = runner(spec, [sample_size], int(n_sims), sc["base_seed"], sink if progress is not None else None)
        except S.EngineError as e:
            if e.status == S.Status.Cancelled:
                raise OrchestratorError("Cancelled", f"cancelled at scenario {idx}, n=Some({sample_size})") from e
            raise
        aggs = aggregate_batch(batch, sc["report_target_indices"], sc["report_contrast_pairs"], spec.estimator,
                               _posthoc_blocks(spec))
        assert len(aggs) == 1
        pr = aggs[0]
        pr["n"] = sample_size
        if snap_warning is not None:
            pr["grid_warnings"].append(snap_warning)
        pr["grid_warnings"].extend(factor_preflight_warnings(spec, sample_size))
        out[sc["label"]] = power_result_to_host(pr, sc["label"])
    return {"scenarios": out, "comparison": {}}
