"""ctypes binding over the C ABI (`include/mcpower_hip.h`), mirroring the reference's
`engine_core` entry points: `run_batch`, `run_batch_st`, `run_batch_st_capture`,
`set_n_threads` (-> `set_device`) -- `/root/reference/crates/engine-core/src/batch.rs:93-197`.

The library is the product: if `libmcpower_hip.so` is missing or no MI355X is
visible the calls raise; nothing here falls back to a CPU path.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Callable, Optional, Sequence

import numpy as np

from . import spec as S

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "_build", "libmcpower_hip.so")
EXPORTS = (
    "mcp_abi_version", "mcp_last_error", "mcp_set_device", "mcp_device_count", "mcp_run_batch", "mcp_free_result",
    "mcp_build_crit_table", "mcp_t_ppf", "mcp_norm_ppf", "mcp_chi2_ppf", "mcp_f_ppf", "mcp_generate_design",
    "mcp_fit_design", "mcp_icdf_probe", "mcp_run_batch_submit", "mcp_run_batch_collect",
    "mcp_comm_unique_id", "mcp_comm_init", "mcp_comm_destroy", "mcp_comm_rank", "mcp_comm_world", "mcp_shard_range",
    "mcp_comm_probe",
)
COMM_ID_BYTES = 128
_lib = None


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "clean"], check=True, capture_output=True)
    r = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j8"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc build failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with mcpower_amd.engine.build() "
                               "(make -C mcpower_amd/csrc); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.mcp_abi_version.restype = C.c_uint32
        L.mcp_last_error.restype = C.c_char_p
        L.mcp_set_device.argtypes = [C.c_int]
        L.mcp_run_batch.restype = C.c_int
        L.mcp_run_batch.argtypes = [C.POINTER(S.CSpec), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_uint64,
                                    C.POINTER(S.CRunOptions), S.PROGRESS_FN, C.c_void_p, C.POINTER(S.CBatchResult)]
        L.mcp_free_result.argtypes = [C.POINTER(S.CBatchResult)]
        L.mcp_run_batch_submit.restype = C.c_int
        L.mcp_run_batch_submit.argtypes = [C.POINTER(S.CSpec), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32,
                                           C.c_uint64, C.POINTER(S.CRunOptions), C.POINTER(C.c_uint64)]
        L.mcp_run_batch_collect.restype = C.c_int
        L.mcp_run_batch_collect.argtypes = [C.c_uint64, C.POINTER(S.CBatchResult)]
        L.mcp_comm_unique_id.argtypes = [C.c_void_p]
        L.mcp_comm_init.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.mcp_shard_range.argtypes = [C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        for name, n in (("mcp_t_ppf", 2), ("mcp_norm_ppf", 1), ("mcp_chi2_ppf", 2), ("mcp_f_ppf", 3)):
            getattr(L, name).restype = C.c_double
            getattr(L, name).argtypes = [C.c_double] * n
        if L.mcp_abi_version() != S.ABI_VERSION:
            raise RuntimeError("libmcpower_hip.so ABI version mismatch")
        _lib = L
    return _lib


def _check(rc: int):
    if rc != 0:
        raise S.EngineError(rc, (lib().mcp_last_error() or b"").decode() or f"status {rc}")


def device_count() -> int:
    return int(lib().mcp_device_count())


def set_device(index: int) -> None:
    """One process per GPU: replaces `set_n_threads` (batch.rs:93)."""
    _check(lib().mcp_set_device(int(index)))


def comm_unique_id() -> bytes:
    """`mcp_comm_unique_id`: the 128-byte rendezvous id of a new communicator (rank 0 creates it, the host's own
    channel -- a file, a TCP store, torch.distributed's object broadcast -- carries it to the other ranks)."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    _check(lib().mcp_comm_unique_id(C.cast(buf, C.c_void_p)))
    return bytes(buf)


def comm_init(rank: int, world: int, comm_id: bytes) -> None:
    """`mcp_comm_init` (collective): join the RCCL communicator on the device selected with `set_device`."""
    if len(comm_id) != COMM_ID_BYTES:
        raise ValueError(f"communicator id must be {COMM_ID_BYTES} bytes")
    buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(comm_id)
    _check(lib().mcp_comm_init(int(rank), int(world), C.cast(buf, C.c_void_p)))


def comm_destroy() -> None:
    _check(lib().mcp_comm_destroy())


def comm_probe() -> None:
    """`mcp_comm_probe` (host-only): bind RCCL now; raises `EngineError(Device)` with the loader's message when it is not
    loadable (what `comm_init` would fail with on a box with a GPU)."""
    _check(lib().mcp_comm_probe())


def comm_rank() -> int:
    return int(lib().mcp_comm_rank())


def comm_world() -> int:
    return int(lib().mcp_comm_world())


def shard_range(n_sims_total: int, rank: int, world: int):
    """`mcp_shard_range`: (offset, count) of rank's contiguous replicate slice [r*S/W, (r+1)*S/W)."""
    off, cnt = C.c_uint64(0), C.c_uint64(0)
    _check(lib().mcp_shard_range(C.c_uint64(n_sims_total), int(rank), int(world), C.byref(off), C.byref(cnt)))
    return int(off.value), int(cnt.value)


def run_batch(spec: "S.SimulationSpec | S.LoweredSpec", sample_sizes: Sequence[int], n_sims: int, base_seed: int,
              sink: Optional[Callable[[int, int], bool]] = None, *, per_sim: bool = True, stat_capture: bool = False,
              staged: bool = False, sim_offset: int = 0, chunk_sims: int = 0, hip_stream: int = 0,
              allreduce: bool = False, seed_per_replicate: bool = False) -> S.BatchResult:
    """`run_batch(spec, sample_sizes, n_sims, base_seed, sink)` (batch.rs:130).

    `sink(current, total) -> bool` is `ProgressSink::report` (return False to cancel -> EngineError Cancelled).
    Keyword extras: `per_sim=False` returns only the additive counters (reduced form for 1e7-replicate runs),
    `sim_offset` makes this call the shard [sim_offset, sim_offset + n_sims) of a larger run,
    `staged=True` selects the generator -> HBM design tile -> fit pipeline instead of the fused kernel,
    `allreduce=True` pools the counters over the process's RCCL communicator inside the call (`comm_init`).
    """
    low = spec if isinstance(spec, S.LoweredSpec) else spec.to_c()  # a marshalled spec may be reused across calls
    ns = np.ascontiguousarray(sample_sizes, dtype=np.uint32)
    opts = S.CRunOptions()
    opts.flags = (S.WANT_PER_SIM if per_sim else 0) | (S.WANT_STAT_CAPTURE if stat_capture else 0) | (
        S.PIPELINE_STAGED if staged else 0) | (S.ALLREDUCE_COUNTERS if allreduce else 0) | (
        S.SEED_PER_REPLICATE if seed_per_replicate else 0)
    opts.chunk_sims = int(chunk_sims)
    opts.sim_offset = int(sim_offset)
    opts.hip_stream = hip_stream or None
    res = S.CBatchResult()
    cb = S.PROGRESS_FN(lambda cur, tot, _u: 1 if sink(int(cur), int(tot)) else 0) if sink else S.PROGRESS_FN()
    rc = lib().mcp_run_batch(low.byref(), ns.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint32(ns.size),
                             C.c_uint32(n_sims), C.c_uint64(base_seed), C.byref(opts), cb, None, C.byref(res))
    _check(rc)
    try:
        return S.result_from_c(res)
    finally:
        lib().mcp_free_result(C.byref(res))


class PendingBatch:
    """A counters-only batch in flight on the device (`mcp_run_batch_submit`); `collect()` waits for it."""

    def __init__(self, ticket: int, keep):
        self._ticket, self._keep = ticket, keep

    def collect(self) -> S.BatchResult:
        if self._ticket is None:
            raise RuntimeError("batch already collected")
        res = S.CBatchResult()
        ticket, self._ticket = self._ticket, None
        _check(lib().mcp_run_batch_collect(C.c_uint64(ticket), C.byref(res)))
        self._keep = None
        try:
            return S.result_from_c(res)
        finally:
            lib().mcp_free_result(C.byref(res))


def submit_batch(spec: "S.SimulationSpec | S.LoweredSpec", sample_sizes: Sequence[int], n_sims: int, base_seed: int, *,
                 staged: bool = False, sim_offset: int = 0, chunk_sims: int = 0, hip_stream: int = 0,
                 allreduce: bool = False) -> PendingBatch:
    """Enqueue a counters-only `run_batch` and return without waiting: up to four batches may be in flight per
    host thread, executing in order on one stream, so the host work of the next batch hides behind the kernels
    of the previous one.  Collect on the same thread."""
    low = spec if isinstance(spec, S.LoweredSpec) else spec.to_c()
    ns = np.ascontiguousarray(sample_sizes, dtype=np.uint32)
    opts = S.CRunOptions()
    opts.flags = (S.PIPELINE_STAGED if staged else 0) | (S.ALLREDUCE_COUNTERS if allreduce else 0)
    opts.chunk_sims = int(chunk_sims)
    opts.sim_offset = int(sim_offset)
    opts.hip_stream = hip_stream or None
    ticket = C.c_uint64(0)
    _check(lib().mcp_run_batch_submit(low.byref(), ns.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint32(ns.size),
                                      C.c_uint32(n_sims), C.c_uint64(base_seed), C.byref(opts), C.byref(ticket)))
    return PendingBatch(ticket.value, (low, ns))


def run_batch_st(spec, sample_sizes, n_sims, base_seed, sink=None, **kw) -> S.BatchResult:
    """`run_batch_st` (batch.rs:157): bit-identical to `run_batch`; there is no thread order on the device."""
    return run_batch(spec, sample_sizes, n_sims, base_seed, sink, **kw)


def run_batch_st_capture(spec, sample_sizes, n_sims, base_seed, **kw):
    """`run_batch_st_capture` (batch.rs:180): also returns the (n_sims, n_targets) squared statistics of grid point 0."""
    r = run_batch(spec, sample_sizes, n_sims, base_seed, None, stat_capture=True, **kw)
    return r, r.stat_capture


def build_crit_table(spec: S.SimulationSpec, sample_sizes) -> np.ndarray:
    low = spec.to_c()
    ns = np.ascontiguousarray(sample_sizes, dtype=np.uint32)
    out = np.zeros((ns.size, spec.n_targets + 4), dtype=np.float64)
    _check(lib().mcp_build_crit_table(low.byref(), ns.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint32(ns.size),
                                      out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


def t_ppf(p, df):
    return lib().mcp_t_ppf(p, df)


def norm_ppf(p):
    return lib().mcp_norm_ppf(p)


def chi2_ppf(p, k):
    return lib().mcp_chi2_ppf(p, k)


def f_ppf(p, dfn, dfd):
    return lib().mcp_f_ppf(p, dfn, dfd)


def generate_design(spec: S.SimulationSpec, max_n: int, n_sims: int, base_seed: int, sim_offset: int = 0):
    """`introspect::capture_sim0_data` (introspect.rs:104) for a replicate range.
    -> design (n_sims, max_n, p) f32, outcome (n_sims, max_n) f32."""
    low = spec.to_c()
    p = spec.n_predictors
    d = np.zeros((n_sims, p, max_n), dtype=np.float32)
    y = np.zeros((n_sims, max_n), dtype=np.float32)
    _check(lib().mcp_generate_design(low.byref(), C.c_uint32(max_n), C.c_uint32(n_sims), C.c_uint64(base_seed),
                                     C.c_uint64(sim_offset), d.ctypes.data_as(C.POINTER(C.c_float)),
                                     y.ctypes.data_as(C.POINTER(C.c_float))))
    return np.transpose(d, (0, 2, 1)), y


def fit_design(design: np.ndarray, outcome: np.ndarray):
    """`introspect::fit_provided_data` (introspect.rs:174), OLS arm: design (n_fits, n, p), outcome (n_fits, n)."""
    design = np.asarray(design, dtype=np.float32)
    if design.ndim == 2:
        design, outcome = design[None], np.asarray(outcome)[None]
    nf, n, p = design.shape
    x = np.ascontiguousarray(np.transpose(design, (0, 2, 1)))
    y = np.ascontiguousarray(outcome, dtype=np.float32)
    beta, se, tsq = (np.zeros((nf, p)) for _ in range(3))
    rss, sst = np.zeros(nf), np.zeros(nf)
    conv = np.zeros(nf, dtype=np.uint8)
    pd = C.POINTER(C.c_double)
    _check(lib().mcp_fit_design(x.ctypes.data_as(C.POINTER(C.c_float)), y.ctypes.data_as(C.POINTER(C.c_float)),
                                C.c_uint32(nf), C.c_uint32(n), C.c_uint32(p), beta.ctypes.data_as(pd),
                                se.ctypes.data_as(pd), tsq.ctypes.data_as(pd), rss.ctypes.data_as(pd),
                                sst.ctypes.data_as(pd), conv.ctypes.data_as(C.POINTER(C.c_uint8))))
    return dict(beta=beta, se=se, t_sq=tsq, rss=rss, sst=sst, converged=conv.astype(bool))


def icdf_probe(first_word: int, stride: int, n: int) -> np.ndarray:
    out = np.zeros(n, dtype=np.float32)
    _check(lib().mcp_icdf_probe(C.c_uint32(first_word), C.c_uint32(stride), C.c_uint32(n),
                                out.ctypes.data_as(C.POINTER(C.c_float))))
    return out
