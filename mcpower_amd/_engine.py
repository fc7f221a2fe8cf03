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
                    estimator: S.EstimatorSpec, posthoc_blocks: Sequence[Tuple[int, int]] = (),
                    report_overall: bool = False) -> List[dict]:
    """`aggregate_batch` (aggregation.rs:62-291): one PowerResult-shaped dict per sample size.
    `posthoc_blocks` = [(n_levels, n_contrasts), ...] (`ResultShape::posthoc_blocks`); `report_overall` is only
    consulted for counters-only results (a per-sim result carries the `overall` tensor itself)."""
    if batch.uncorrected is None:  # counters-only result (10^7-replicate or sharded runs): same numbers from the pooled counts
        return _aggregate_counters(batch, target_indices, contrast_pairs, estimator, posthoc_blocks, report_overall)
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


def _aggregate_counters(batch: S.BatchResult, target_indices, contrast_pairs, estimator: S.EstimatorSpec,
                        posthoc_blocks, report_overall: bool) -> List[dict]:
    """`aggregate_batch` from the additive counter block alone -- what `merge_power_results` (merge.rs:26-70) pools
    across workers: every count, rate, Wilson interval and histogram of the per-sim path.  Not recoverable from
    counts: the per-sim `boundary_hit` list (left empty), per-component pin rates beyond the first component and the
    mean tau-hat^2 of a clustered GLM (NaN)."""
    sh, lay, ctr = batch.shape, batch.layout, batch.counters
    n_ss, n_targets, n_factors = sh.n_sample_sizes, sh.n_targets, sh.n_factors
    total_contrasts = sum(w for _, w in posthoc_blocks)
    out = []
    for ss in range(n_ss):
        row = ctr[ss]
        n_sims = int(row[lay["n_sims"]])

        def block(name, width):
            return [int(v) for v in row[lay[name]:lay[name] + width]]

        k_unc, k_cor = block("k_unc", n_targets), block("k_cor", n_targets)
        conv_count = int(row[lay["converged"]])
        b1, b2 = int(row[lay["boundary_1"]]), int(row[lay["boundary_2"]])
        singular_rate = b1 / conv_count if conv_count > 0 else 0.0
        if estimator == S.EstimatorSpec.Ols:
            extras = {"estimator": "ols"}
        elif estimator == S.EstimatorSpec.Glm:
            extras = {"estimator": "glm", "baseline_prob_realized": math.nan, "singular_fit_rate": singular_rate,
                      "tau_squared_hat_mean": math.nan}
        else:
            comp = [singular_rate] + [math.nan] * max(sh.n_variance_components - 1, 0)
            extras = {"estimator": "mle", "tau_estimate": math.nan, "boundary_hits": b1 + b2,
                      "joint_uncorrected_rate": _rate(int(row[lay["joint_unc"]]), n_sims),
                      "joint_corrected_rate": _rate(int(row[lay["joint_cor"]]), n_sims),
                      "singular_fit_rate": singular_rate, "boundary_rate_per_component": comp[:sh.n_variance_components]}
        posthoc, offset = [], 0
        for n_levels, w in posthoc_blocks:
            ku = [int(v) for v in row[lay["posthoc_unc"] + offset:lay["posthoc_unc"] + offset + w]]
            kc = [int(v) for v in row[lay["posthoc_cor"] + offset:lay["posthoc_cor"] + offset + w]]
            posthoc.append(dict(n_levels=int(n_levels), power_uncorrected=[_rate(k, n_sims) for k in ku],
                                power_corrected=[_rate(k, n_sims) for k in kc],
                                ci_uncorrected=[wilson_ci(k, n_sims) for k in ku],
                                ci_corrected=[wilson_ci(k, n_sims) for k in kc],
                                success_counts_uncorrected=ku, success_counts_corrected=kc))
            offset += w
        overall_rate = overall_ci = None
        overall_count = 0
        if report_overall:
            overall_count = int(row[lay["overall"]])
            overall_rate, overall_ci = _rate(overall_count, n_sims), wilson_ci(overall_count, n_sims)
        n_joint = n_targets + total_contrasts
        out.append(dict(
            n=0, n_sims=n_sims, target_indices=[int(i) for i in target_indices],
            contrast_pairs=[(int(a), int(b)) for a, b in contrast_pairs],
            power_uncorrected=[_rate(k, n_sims) for k in k_unc], power_corrected=[_rate(k, n_sims) for k in k_cor],
            ci_uncorrected=[wilson_ci(k, n_sims) for k in k_unc], ci_corrected=[wilson_ci(k, n_sims) for k in k_cor],
            convergence_rate=1.0 if n_sims == 0 else conv_count / n_sims, boundary_hit=[], estimator_extras=extras,
            overall_significant_rate=overall_rate, success_counts_uncorrected=k_unc, success_counts_corrected=k_cor,
            convergence_count=conv_count, overall_significant_count=overall_count, overall_significant_ci=overall_ci,
            success_count_histogram_uncorrected=block("hist_unc", n_joint + 1),
            success_count_histogram_corrected=block("hist_cor", n_joint + 1), grid_warnings=[], posthoc=posthoc,
            factor_exclusion_counts=block("excl_sparse", n_factors), factor_separation_counts=block("excl_sep", n_factors)))
    return out


def sharded_runner(rank: int, world: int, group=None, run_batch=None, allow_host_pool: bool = False):
    """Runner for `find_power` / `find_sample_size` when one process drives one GPU: this rank simulates its
    contiguous slice of the replicates (same base seed, so the union is the single-device run), one all-reduce pools
    the counters, and every rank aggregates the same pooled block -- identical results on all ranks.
    Usage: `find_power(blob, n, n_sims, seed, _runner=sharded_runner(rank, world))` on every rank.

    With a C-ABI communicator of this world (`mcpower_amd.dist.init_comm`) the collective runs inside `run_batch`
    on the device-resident counters, and a cancelling progress callback on any rank cancels every rank (the verdict
    rides in the reduced payload); an empty shard (fewer replicates than ranks) still joins it.  Without a communicator
    the device path REFUSES to run sharded: only rehearsals -- a `run_batch=` stand-in (the CPU oracle under gloo in the
    tests) or `allow_host_pool=True` -- pool the host blocks through the torch process group; there only rank 0 drives
    the progress sink and its verdict is agreed on with a one-word all-reduce ahead of the block's, so no rank is left
    waiting in a collective."""
    def run(spec, sample_sizes, n_sims, base_seed, sink):
        from . import dist as mdist
        from . import engine
        off, cnt = mdist.shard_range(int(n_sims), rank, world)
        if run_batch is None and engine.comm_world() == world and world > 1:
            local = engine.run_batch(spec, sample_sizes, cnt, base_seed, sink, per_sim=False, sim_offset=off, allreduce=True)
        else:
            if run_batch is None and world > 1 and not allow_host_pool:
                raise OrchestratorError("InvalidScenarios", f"sharded_runner: no C-ABI communicator of world {world} in this process "
                                        "(mcpower_amd.dist.init_comm first); allow_host_pool=True pools host counter blocks through "
                                        "the torch process group instead (rehearsals)")
            my_sink = sink if rank == 0 else None
            cancelled = 0
            try:
                if run_batch is None:
                    local = engine.run_batch(spec, sample_sizes, cnt, base_seed, my_sink, per_sim=False, sim_offset=off)
                else:
                    local = run_batch(spec, sample_sizes, cnt, base_seed, off)
                block = local.counters
            except S.EngineError as e:
                if e.status != S.Status.Cancelled:
                    raise
                cancelled, local, block = 1, None, None
            # agree on the verdict first (rank 0 may hold no block to contribute if its sink cancelled), then pool
            verdict = mdist.allreduce_counters(np.array([cancelled], dtype=np.uint64), group)
            if verdict[0] != 0:
                raise S.EngineError(S.Status.Cancelled, "cancelled by host")
            local.counters = mdist.allreduce_counters(block, group)
        local.uncorrected = None  # a runner handing back per-sim tensors of its shard must not be mistaken for the whole
        local.shape.n_sims = int(n_sims)
        return local
    return run


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
                               _posthoc_blocks(spec), report_overall=bool(spec.report_overall))
        assert len(aggs) == 1
        pr = aggs[0]
        pr["n"] = sample_size
        if snap_warning is not None:
            pr["grid_warnings"].append(snap_warning)
        pr["grid_warnings"].extend(factor_preflight_warnings(spec, sample_size))
        out[sc["label"]] = power_result_to_host(pr, sc["label"])
    return {"scenarios": out, "comparison": {}}


# ---------------------------------------------------------------------------------------------
# grid.rs / fit.rs / find_sample_size.rs
# ---------------------------------------------------------------------------------------------
MIN_ROWS_PER_CLUSTER = 2          # configs/config.json limits.min_rows_per_cluster
Z_GATE = 2.0                      # fit.rs:27
MIN_HINT_POWER = 0.5              # fit.rs:33
EXTRAPOLATION_CAP = 2.0           # fit.rs:38
MAX_GRID_PER_CALL = 256           # device limit of one run_batch call (mcp_params.hpp MAX_GRID)


def _ceil_to(x: int, a: int) -> int:
    a = max(a, 1)
    return -(-x // a) * a


def _floor_to(x: int, a: int) -> int:
    a = max(a, 1)
    return (x // a) * a


def _round_to(x: int, a: int) -> int:
    a = max(a, 1)
    return ((x + a // 2) // a) * a


def _rust_round(v: float) -> int:
    """f64::round: half away from zero (Python's round() is half-to-even)."""
    return int(math.floor(v + 0.5)) if v >= 0 else -int(math.floor(-v + 0.5))


def build_grid(from_req: int, to_req: int, by: Tuple[str, int], mode: str, atom: int, hard_min: int):
    """`build_grid` (grid.rs:37-156). `by` = ("fixed", step) | ("auto", count); mode "linear" | "log".
    -> (grid, warnings)."""
    if atom == 0:
        raise OrchestratorError("InvalidClusterAtom", "invalid cluster atom: atom must be >= 1")
    if from_req < 1 or to_req < from_req:
        raise OrchestratorError("InvalidGridBounds", f"invalid grid bounds: from={from_req}, to={to_req}")
    warnings: List[str] = []
    snapped_from = max(_ceil_to(from_req, atom), _ceil_to(hard_min, atom))
    snapped_to = _floor_to(to_req, atom)
    if snapped_from > from_req:
        warnings.append(f"raised `from` from {from_req} to {snapped_from} so each cluster keeps enough rows (atom={atom})")
    if snapped_to < to_req:
        warnings.append(f"lowered `to` from {to_req} to {snapped_to} (nearest multiple of the cluster atom {atom})")
    if snapped_to < snapped_from:
        raise OrchestratorError(
            "ClusterGridEmpty",
            f"after cluster snapping the [{snapped_from},{snapped_to}] range is empty (atom={atom}); "
            f"widen bounds or reduce n_clusters/cluster_size")
    max_feasible = (snapped_to - snapped_from) // atom + 1
    if max_feasible < 2:
        raise OrchestratorError(
            "ClusterGridSinglePoint",
            f"range admits one sample size (snapped [{snapped_from},{snapped_to}], atom={atom}); nothing to search")
    if max_feasible < 4:
        warnings.append(f"only {max_feasible} grid points after cluster snapping; required-N (and model-based "
                        f"crossing) will be coarse")
    kind, value = by
    if mode == "linear":
        if kind == "fixed":
            step = max(_ceil_to(value, atom), atom)
        else:
            c = min(max(value, 2), max_feasible)
            span = snapped_to - snapped_from
            denom = atom * (c - 1)
            step = atom * max((span + denom // 2) // denom, 1)
        grid = list(range(snapped_from, snapped_to + 1, step))
        if grid[-1] != snapped_to:
            grid.append(snapped_to)
    else:
        c = min(max(value, 2), max_feasible)
        lo, hi = math.log10(snapped_from), math.log10(snapped_to)
        raw = []
        for i in range(c):
            t = i / (c - 1)
            v = _rust_round(10.0 ** (lo + t * (hi - lo)))
            raw.append(min(max(_round_to(v, atom), snapped_from), snapped_to))
        raw[0], raw[-1] = snapped_from, snapped_to
        grid = sorted(set(raw))
    return grid, warnings


def as_proportion(p: float) -> float:
    return p / 100.0 if p > 1.0 else p


def pava(y: Sequence[float]) -> List[float]:
    """Pool-adjacent-violators, unweighted; only strict decreases pool (fit.rs:135-155)."""
    means: List[float] = []
    lens: List[int] = []
    for v in y:
        means.append(float(v))
        lens.append(1)
        while len(means) >= 2 and means[-2] > means[-1]:
            m1, l1 = means.pop(), lens.pop()
            l0 = lens[-1]
            means[-1] = (means[-1] * l0 + m1 * l1) / (l0 + l1)
            lens[-1] = l0 + l1
    out: List[float] = []
    for m, l in zip(means, lens):
        out.extend([m] * l)
    return out


def _crossing(grid, fitted, target) -> Optional[float]:
    if fitted[0] >= target:
        return float(grid[0])
    for i in range(1, len(fitted)):
        if fitted[i] >= target:
            x0, x1 = float(grid[i - 1]), float(grid[i])
            y0, y1 = fitted[i - 1], fitted[i]
            return x0 + (target - y0) * (x1 - x0) / (y1 - y0)
    return None


def _ceil_to_atom(n: float, atom: int) -> int:
    a = max(atom, 1)
    n_int = int(max(math.ceil(n - 1e-9), 1.0))
    return -(-n_int // a) * a


def _norm_ppf(p: float) -> float:
    from . import engine

    return engine.norm_ppf(p)


def _probit_hint(grid, p_hat, fitted, n_sims: float, target: float, atom: int) -> Optional[int]:
    """`probit_hint` (fit.rs:196-247): WLS of probit(p) on sqrt(N), delta-method weights."""
    if fitted[-1] < MIN_HINT_POWER:
        return None
    clamp_lo, clamp_hi = 0.5 / n_sims, 1.0 - 0.5 / n_sims
    sw = swx = swy = swxx = swxy = 0.0
    for n, ph in zip(grid, p_hat):
        p = min(max(ph, clamp_lo), clamp_hi)
        z = _norm_ppf(p)
        pdf = math.exp(-0.5 * z * z) / math.sqrt(2.0 * math.pi)
        w = n_sims * pdf * pdf / (p * (1.0 - p))
        x = math.sqrt(float(n))
        sw += w
        swx += w * x
        swy += w * z
        swxx += w * x * x
        swxy += w * x * z
    denom = sw * swxx - swx * swx
    if not math.isfinite(denom) or denom <= 0.0:
        return None
    slope = (sw * swxy - swx * swy) / denom
    if not math.isfinite(slope) or slope <= 0.0:
        return None
    intercept = (swy - slope * swx) / sw
    sqrt_n = (_norm_ppf(target) - intercept) / slope
    if not math.isfinite(sqrt_n) or sqrt_n <= 0.0:
        return None
    n_cross = sqrt_n * sqrt_n
    if n_cross > EXTRAPOLATION_CAP * float(grid[-1]):
        return None
    return _ceil_to_atom(n_cross, atom)


def fit_crossing(grid: Sequence[int], counts: Sequence[int], n_sims: int, target_power: float, atom: int) -> dict:
    """`fit_crossing` (fit.rs:46-111) -> the status-tagged map of `crossing_fit_to_host`
    (result_host.rs:296-322)."""
    if n_sims == 0 or len(grid) == 0:
        return {"status": "not_reached", "n_approx": None}
    target = as_proportion(target_power)
    n_f = float(n_sims)
    p_hat = [k / n_f for k in counts]
    # monotonicity gate (fit.rs:116-131)
    max_drop, pair = 0.0, None
    for i in range(len(p_hat)):
        for j in range(i + 1, len(p_hat)):
            d = p_hat[i] - p_hat[j]
            if d > max_drop:
                max_drop, pair = d, (p_hat[i], p_hat[j])
    if pair is not None:
        pi, pj = pair
        se = math.sqrt((pi * (1.0 - pi) + pj * (1.0 - pj)) / n_f)
        if max_drop > Z_GATE * se:
            return {"status": "non_monotone", "max_violation": max_drop}
    fitted = pava(p_hat)
    if fitted[0] >= target:
        return {"status": "at_or_below_min", "n_min": int(grid[0])}
    if fitted[-1] < target:
        return {"status": "not_reached", "n_approx": _probit_hint(grid, p_hat, fitted, n_f, target, atom)}
    n_star = _crossing(grid, fitted, target)
    hi_band = pava([wilson_ci(k, n_sims)[1] for k in counts])
    lo_band = pava([wilson_ci(k, n_sims)[0] for k in counts])
    ci_lo = None if hi_band[0] >= target else _crossing(grid, hi_band, target)
    ci_hi = _crossing(grid, lo_band, target)
    return {"status": "fitted", "n_star": n_star, "n_achievable": _ceil_to_atom(n_star, atom), "ci_lo": ci_lo,
            "ci_hi": ci_hi}


def first_n_at_target(powers_corrected, sample_sizes, target_power: float, target_index: int) -> Optional[int]:
    p_target = as_proportion(target_power)
    for row, n in zip(powers_corrected, sample_sizes):
        if row[target_index] >= p_target:
            return int(n)
    return None


def first_n_joint_at_target(histograms, n_sims: int, sample_sizes, target_power: float, k: int) -> Optional[int]:
    if n_sims == 0:
        return None
    p_target = as_proportion(target_power)
    for hist, n in zip(histograms, sample_sizes):
        if sum(hist[k:]) / n_sims >= p_target:
            return int(n)
    return None


def derive_sample_size_outputs(per_n: Sequence[dict], sample_sizes: Sequence[int], target_power: float, atom: int) -> dict:
    """`derive_sample_size_outputs` (grid.rs:232-330)."""
    powers = [pr["power_corrected"] for pr in per_n]
    n_targets = len(per_n[0]["power_corrected"]) if per_n else 0
    n_sims = per_n[0]["n_sims"] if per_n else 0
    first_achieved = [first_n_at_target(powers, sample_sizes, target_power, t) for t in range(n_targets)]
    fitted = [fit_crossing(sample_sizes, [pr["success_counts_corrected"][t] for pr in per_n], n_sims, target_power, atom)
              for t in range(n_targets)]
    hists = [pr["success_count_histogram_corrected"] for pr in per_n]
    joint_n_sims = sum(hists[0]) if hists else 0
    n_joint = max(len(hists[0]) - 1, 0) if hists else 0
    first_joint = [first_n_joint_at_target(hists, joint_n_sims, sample_sizes, target_power, k)
                   for k in range(1, n_joint + 1)]
    fitted_joint = [fit_crossing(sample_sizes, [sum(h[k:]) for h in hists], joint_n_sims, target_power, atom)
                    for k in range(1, n_joint + 1)]
    first_overall, fitted_overall = None, None
    if per_n and per_n[0]["overall_significant_rate"] is not None:
        p_target = as_proportion(target_power)
        first_overall = next((int(n) for n, pr in zip(sample_sizes, per_n)
                              if (pr["overall_significant_rate"] or 0.0) >= p_target), None)
        fitted_overall = fit_crossing(sample_sizes, [pr["overall_significant_count"] for pr in per_n], n_sims,
                                      target_power, atom)
    return dict(first_achieved=first_achieved, first_joint_achieved=first_joint, fitted=fitted,
                fitted_joint=fitted_joint, first_overall_achieved=first_overall, fitted_overall=fitted_overall)


def min_inclusion_n(probs: Sequence[float], k_min: int, n_max: int) -> Optional[int]:
    """`min_inclusion_n` (data_gen.rs:977-986)."""
    total = 0.0
    for p in probs:
        total += p
    counts = [0] * len(probs)
    for n in range(1, n_max + 1):
        t = float(n)
        best, best_deficit = 0, -math.inf
        for g, p in enumerate(probs):
            deficit = (p / total) * t - counts[g]
            if deficit > best_deficit:
                best_deficit, best = deficit, g
        counts[best] += 1
        if all(c >= k_min for c in counts):
            return n
    return None


def factor_preflight_sample_size(spec: S.SimulationSpec, grid: Sequence[int]) -> List[str]:
    """`factor_preflight_sample_size` (find_power.rs:127-178)."""
    k_min = spec.factor_min_level_count
    if not grid or k_min == 0 or len(spec.upload_data) > 0:
        return []
    lo, hi = grid[0], grid[-1]
    out, off = [], 0
    for f, nl in enumerate(spec.factor_n_levels):
        l = max(int(nl), 0)
        sampled = spec.factor_sampled[f] if f < len(spec.factor_sampled) else None
        if sampled is None:
            sampled = spec.scenario.sampled_factor_proportions
        if not sampled:
            probs = spec.factor_proportions[off:off + l]
            if any(c < k_min for c in fixed_allocation_counts(probs, lo)):
                n_inc = min_inclusion_n(probs, k_min, hi)
                if n_inc is not None:
                    out.append(f"factor {f + 1}: levels reach the {k_min}-observation minimum only at N >= {n_inc}; "
                               f"grid points below that exclude the factor and report power 0 for its effects")
                else:
                    out.append(f"factor {f + 1}: a level stays under the {k_min}-observation minimum across the whole "
                               f"[{lo}, {hi}] range; the factor is excluded everywhere and its effects report power 0")
        off += l
    return out


def resolve_cluster_grid_params(contracts: Sequence[dict]) -> Tuple[int, int, List[str]]:
    """`resolve_cluster_grid_params` (find_sample_size.rs:18-69) -> (atom, hard_min, warnings)."""
    cluster = None
    for c in contracts:
        cl = c["generation"].get("cluster")
        if cl is not None:
            if cluster is None:
                cluster = cl
            else:
                a, b = K.cluster_atom({"generation": {"cluster": cluster}}), K.cluster_atom({"generation": {"cluster": cl}})
                if a != b:
                    raise OrchestratorError(
                        "MixedClusterAtoms",
                        f"scenarios in this call disagree on the cluster atom ({a} vs {b}); a single shared grid "
                        f"cannot be a multiple of every atom")
    if cluster is None:
        return 1, 1, []
    atom = K.cluster_atom({"generation": {"cluster": cluster}})
    skind, sbody = K._variant(cluster["sizing"])
    warnings = []
    if skind == "FixedClusters":
        n = max(int(sbody["n_clusters"]), 1)
        if n < MIN_CLUSTERS:
            warnings.append(f"only {n} clusters (< {MIN_CLUSTERS}); τ² estimates may be unstable")
        return atom, n * MIN_ROWS_PER_CLUSTER, warnings
    cs = int(sbody["cluster_size"])
    if cs < MIN_ROWS_PER_CLUSTER:
        raise OrchestratorError(
            "ClusterSizeTooSmall",
            f"FixedSize cluster regime needs cluster_size >= {MIN_ROWS_PER_CLUSTER} (no within-cluster information "
            f"below that); got {cs}")
    return atom, MIN_CLUSTERS * cs, warnings


def sample_size_method_from_host_args(method: str, by: Optional[int], by_kind: Optional[str], mode: Optional[str]):
    """`SampleSizeMethod::from_host_args` (result.rs:552-575); errors are ValueError like the PyO3 shell."""
    if method != "grid":
        raise ValueError(f'unknown method "{method}"')
    if by is None:
        raise ValueError("grid method requires `by`")
    kind = "fixed" if by_kind is None else by_kind
    if kind not in ("fixed", "auto"):
        raise ValueError(f'unknown by_kind "{kind}"')
    m = "linear" if mode is None else mode
    if m not in ("linear", "log"):
        raise ValueError(f'unknown mode "{m}"')
    return (kind, int(by)), m


def sample_size_result_to_host(ssr: dict, label: str) -> dict:
    """`sample_size_result_to_host` (result_host.rs:150-293), Python rendering of each leaf."""
    grid = ssr["grid_or_trace"]
    first = grid[0] if grid else None
    rows = len(first["boundary_hit"]) if first else 0
    bh = [[pr["boundary_hit"][sim] for pr in grid] for sim in range(rows)]
    fitted_overall = [] if ssr["fitted_overall"] is None else [ssr["fitted_overall"]]
    return {
        "n_sims": first["n_sims"] if first else 0,
        "n_sample_sizes": len(grid),
        "n_targets": len(first["target_indices"]) if first else 0,
        "target_indices": list(first["target_indices"]) if first else [],
        "contrast_pairs": [[p, n] for p, n in first["contrast_pairs"]] if first else [],
        "power_uncorrected": [list(pr["power_uncorrected"]) for pr in grid],
        "power_corrected": [list(pr["power_corrected"]) for pr in grid],
        "ci_uncorrected": [list(pr["ci_uncorrected"]) for pr in grid],
        "ci_corrected": [list(pr["ci_corrected"]) for pr in grid],
        "convergence_rate": [pr["convergence_rate"] for pr in grid],
        "sample_sizes": [pr["n"] for pr in grid],
        "first_achieved": dict(enumerate(ssr["first_achieved"])),
        "success_count_histogram_uncorrected": [list(pr["success_count_histogram_uncorrected"]) for pr in grid],
        "success_count_histogram_corrected": [list(pr["success_count_histogram_corrected"]) for pr in grid],
        "grid_warnings": list(ssr["grid_warnings"]),
        "factor_exclusion_counts": [list(pr["factor_exclusion_counts"]) for pr in grid],
        "factor_separation_counts": [list(pr["factor_separation_counts"]) for pr in grid],
        "first_joint_achieved": dict(enumerate(ssr["first_joint_achieved"])),
        "first_overall_achieved": ssr["first_overall_achieved"],
        "target_power": ssr["target_power"],
        "scenario": label,
        "boundary_hit": bh,
        "fitted": dict(enumerate(ssr["fitted"])),
        "fitted_joint": dict(enumerate(ssr["fitted_joint"])),
        "fitted_overall": dict(enumerate(fitted_overall)),
        "cluster_atom": ssr["cluster_atom"],
    }


def _run_grid(runner, spec: S.SimulationSpec, grid: Sequence[int], n_sims: int, seed: int, sink) -> S.BatchResult:
    """One `run_batch` over the whole grid (find_sample_size.rs:196-209).  A grid longer than one device
    call carries is split into consecutive calls; only allowed where the rows of a replicate do not depend
    on the largest N of the call (unclustered designs: prefix-stable rows, no cross-N fit state)."""
    if len(grid) <= MAX_GRID_PER_CALL:
        return runner(spec, list(grid), n_sims, seed, sink)
    if spec.cluster is not None:
        raise S.EngineError(S.Status.Unsupported,
                            f"a clustered grid of {len(grid)} sample sizes exceeds the {MAX_GRID_PER_CALL} one "
                            f"run_batch call carries; use a coarser step")
    parts = [runner(spec, list(grid[i:i + MAX_GRID_PER_CALL]), n_sims, seed, sink)
             for i in range(0, len(grid), MAX_GRID_PER_CALL)]

    def cat(name):
        arrs = [getattr(p, name) for p in parts]
        return None if any(a is None for a in arrs) else np.concatenate(arrs, axis=1)

    first = parts[0]
    shape = S.ResultShape(first.shape.n_sims, len(grid), first.shape.n_targets, first.shape.n_factors,
                          first.shape.n_posthoc_contrasts, first.shape.n_variance_components)
    return S.BatchResult(
        shape=shape, uncorrected=cat("uncorrected"), corrected=cat("corrected"), posthoc_unc=cat("posthoc_unc"),
        posthoc_cor=cat("posthoc_cor"), converged=cat("converged"), boundary_hit=cat("boundary_hit"),
        pinned_components=cat("pinned_components"), joint_unc=cat("joint_unc"), joint_cor=cat("joint_cor"),
        overall=cat("overall"), factor_excluded=cat("factor_excluded"), tau_squared_hat=cat("tau_squared_hat"),
        stat_capture=first.stat_capture, counters=np.concatenate([p.counters for p in parts], axis=0),
        layout=first.layout, device_ms=sum(p.device_ms for p in parts), kernel_ms=sum(p.kernel_ms for p in parts),
        aux_kernel_ms=sum(p.aux_kernel_ms for p in parts),
        n_kernel_launches=sum(p.n_kernel_launches for p in parts))


def find_sample_size(contracts_bytes: bytes, target_power: float, lo: int, hi: int, n_sims: int, base_seed: int,
                     method: str, by: Optional[int] = None, by_kind: Optional[str] = None, mode: Optional[str] = None,
                     tol_n: Optional[int] = None, progress: Optional[Callable[[int, int], bool]] = None, *,
                     _runner=None) -> Dict[str, Any]:
    """`_engine.find_sample_size(...)` (engine-py/src/lib.rs:88-135 over find_sample_size.rs:72-152):
    every grid N is fitted on the same nested-prefix replicates in one device call per scenario."""
    del tol_n  # bisection-only in the reference; kept for signature stability
    contracts = K.decode_contracts(contracts_bytes)
    by_value, grid_mode = sample_size_method_from_host_args(method, by, by_kind, mode)
    scenarios = lower_contracts(contracts, base_seed)
    atom, hard_min, grid_warnings = resolve_cluster_grid_params(contracts)
    grid, build_warnings = build_grid(int(lo), int(hi), by_value, grid_mode, atom, hard_min)
    grid_warnings = grid_warnings + build_warnings
    runner = _runner or _default_runner
    fits_per_sim = len(grid)
    total = int(n_sims) * len(scenarios) * fits_per_sim
    cancelled = [False]

    out: Dict[str, Any] = {}
    for idx, sc in enumerate(scenarios):
        if cancelled[0]:
            raise OrchestratorError("Cancelled", f"cancelled at scenario {idx}, n=None")
        offset = idx * int(n_sims) * fits_per_sim

        def sink(current: int, _batch_total: int, _offset=offset) -> bool:
            if cancelled[0]:
                return False
            if progress is not None:
                try:
                    keep = progress(_offset + int(current) * fits_per_sim, total)
                    keep = True if keep is None else bool(keep)
                except Exception:
                    keep = False
                if not keep:
                    cancelled[0] = True
            return not cancelled[0]

        spec: S.SimulationSpec = sc["spec"]
        try:
            batch = _run_grid(runner, spec, grid, int(n_sims), sc["base_seed"], sink if progress is not None else None)
        except S.EngineError as e:
            if e.status == S.Status.Cancelled:
                raise OrchestratorError("Cancelled", f"cancelled at scenario {idx}, n=None") from e
            raise
        aggs = aggregate_batch(batch, sc["report_target_indices"], sc["report_contrast_pairs"], spec.estimator,
                               _posthoc_blocks(spec), report_overall=bool(spec.report_overall))
        for pr, n in zip(aggs, grid):
            pr["n"] = int(n)
        d = derive_sample_size_outputs(aggs, grid, float(target_power), atom)
        ssr = dict(grid_or_trace=aggs, cluster_atom=atom, target_power=float(target_power),
                   grid_warnings=grid_warnings + factor_preflight_sample_size(spec, grid), **d)
        out[sc["label"]] = sample_size_result_to_host(ssr, sc["label"])
    return {"scenarios": out, "comparison": {}}
