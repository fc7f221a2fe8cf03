"""ctypes binding of the CPU ORACLE (`oracle/_build/libmcp_oracle.so`).

TEST INFRASTRUCTURE ONLY -- a plain-C restatement of the reference's simulate
loop (see the headers of oracle/*.c for the reference file:line each function
follows).  It is the checker for the HIP path; it is never the thing shipped
or measured, and nothing under `mcpower_amd/` imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from mcpower_amd import spec as S

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libmcp_oracle.so")
_lib = None


PORTABLE_CFLAGS = "-O2 -march=x86-64-v3 -ffp-contract=off"   # oracle/Makefile: the build that travels to the GPU box
NATIVE_CFLAGS = "-O3 -march=native -ffp-contract=off"          # the timed CPU baseline, built ON the timing host
_cflags = PORTABLE_CFLAGS


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds).  `make` is always consulted -- it rebuilds only when a source is newer
    than the library -- so an edited restatement can never be checked against a stale binary."""
    if force:
        subprocess.run(["make", "-C", _HERE, "clean"], capture_output=True)
    r = subprocess.run(["make", "-C", _HERE], capture_output=True, text=True)
    if r.returncode != 0:   # a failed rebuild must never leave a stale library in use
        raise RuntimeError("oracle build failed:\n" + r.stdout[-2000:] + r.stderr[-2000:])
    return _LIB_PATH


def use_native_build() -> str:
    """Rebuild the oracle `-O3 -march=native` on THIS host (the flags SURVEY 8(d) names for the timed CPU baseline) and
    bind it for the rest of the process.  Same IEEE arithmetic (no fast-math, no contraction): the tests' portable build
    and this one agree bit for bit; only `bench.py`'s cpu_baseline leg calls this.  Returns the flags in effect."""
    global _lib, _LIB_PATH, _cflags
    out = os.path.join(_HERE, "_build", "libmcp_oracle_native.so")
    r = subprocess.run(["make", "-C", _HERE, "OUT=_build/libmcp_oracle_native.so",
                        f"CFLAGS={NATIVE_CFLAGS} -fPIC -std=gnu11 -Wall -Wextra -pthread"], capture_output=True, text=True)
    if r.returncode == 0 and os.path.exists(out):
        _LIB_PATH, _lib, _cflags = out, None, NATIVE_CFLAGS
    return _cflags


def cflags() -> str:
    return _cflags


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        snap = os.environ.get("MCPO_LIB_PATH")   # worker processes of tools/oracle2_perdraw.py: a snapshot the parent built
        if snap:
            L = C.CDLL(snap)
        else:
            build()
            L = C.CDLL(_LIB_PATH)
        L.mcpo_splitmix64.restype = C.c_uint64
        L.mcpo_splitmix64.argtypes = [C.c_uint64]
        L.mcpo_pcg_mix64.restype = C.c_uint64
        L.mcpo_pcg_mix64.argtypes = [C.c_uint64, C.c_uint64]
        L.mcpo_u32_to_unit_f32.restype = C.c_float
        L.mcpo_u32_to_unit_f32.argtypes = [C.c_uint32]
        L.mcpo_norm_inv_cdf_f32.restype = C.c_float
        L.mcpo_norm_inv_cdf_f32.argtypes = [C.c_float]
        for name in ("mcpo_norm_ppf",):
            getattr(L, name).restype = C.c_double
            getattr(L, name).argtypes = [C.c_double]
        for name in ("mcpo_t_ppf", "mcpo_chi2_ppf"):
            getattr(L, name).restype = C.c_double
            getattr(L, name).argtypes = [C.c_double, C.c_double]
        L.mcpo_f_ppf.restype = C.c_double
        L.mcpo_f_ppf.argtypes = [C.c_double] * 3
        L.mcpo_q_tukey_ppf.restype = C.c_double
        L.mcpo_q_tukey_ppf.argtypes = [C.c_double] * 3
        L.mcpo_phi.restype = C.c_double
        L.mcpo_phi.argtypes = [C.c_double]
        L.mcpo_marginal_uniform.restype = C.c_double
        L.mcpo_marginal_uniform.argtypes = [C.c_double] * 3
        L.mcpo_last_error.restype = C.c_char_p
        L.mcpo_set_glmm_general_pirls_cap.argtypes = [C.c_int]
        L.mcpo_run_batch.restype = C.c_int
        L.mcpo_run_batch.argtypes = [C.POINTER(S.CSpec), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_uint64,
                                     C.POINTER(S.CRunOptions), C.c_uint32, C.POINTER(S.CBatchResult)]
        L.mcpo_free_result.argtypes = [C.POINTER(S.CBatchResult)]
        L.mcpo_generate_sim_data.restype = C.c_int
        L.mcpo_build_crit_table.restype = C.c_int
        L.mcpo_ols_fit.restype = C.c_int
        L.mcpo_glm_fit.restype = C.c_int
        L.mcpo_glm_fit_family.restype = C.c_int
        L.mcpo_lmm_fit.restype = C.c_int
        _lib = L
    return _lib


def _check(rc: int):
    if rc != 0:
        raise S.EngineError(rc, (lib().mcpo_last_error() or b"").decode() or f"oracle status {rc}")


def splitmix64(z: int) -> int:
    return lib().mcpo_splitmix64(C.c_uint64(z & (2**64 - 1)))


def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().mcpo_philox4x32_10(c, k, o)
    return [int(v) for v in o]


def u32_to_unit_f32_bits(w: int) -> int:
    return int(np.float32(lib().mcpo_u32_to_unit_f32(w)).view(np.uint32))


def norm_inv_cdf_f32(u: np.ndarray) -> np.ndarray:
    f = lib().mcpo_norm_inv_cdf_f32
    return np.array([f(float(x)) for x in np.asarray(u, dtype=np.float32)], dtype=np.float32)


def icdf_probe(first_word: int, stride: int, n: int) -> np.ndarray:
    out = np.zeros(n, dtype=np.float32)
    lib().mcpo_icdf_probe(C.c_uint32(first_word), C.c_uint32(stride), C.c_uint32(n),
                          out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def stream_f32(base_seed: int, sim_id: int, kind: str, n: int) -> np.ndarray:
    out = np.zeros(n, dtype=np.float32)
    k = {"normal": 0, "uniform": 1, "scenario_uniform": 2}[kind]
    lib().mcpo_stream_f32(C.c_uint64(base_seed), C.c_uint64(sim_id), C.c_int(k),
                          out.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint32(n))
    return out


def stream_categorical(base_seed: int, sim_id: int, probs, n: int) -> np.ndarray:
    p = np.ascontiguousarray(probs, dtype=np.float64)
    out = np.zeros(n, dtype=np.uint32)
    lib().mcpo_stream_categorical(C.c_uint64(base_seed), C.c_uint64(sim_id), p.ctypes.data_as(C.POINTER(C.c_double)),
                                  C.c_uint32(p.size), out.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint32(n))
    return out


def fill_normal_column(key, cls: int, col: int, n: int) -> np.ndarray:
    out = np.zeros(n, dtype=np.float32)
    k = (C.c_uint32 * 2)(*key)
    lib().mcpo_fill_normal_column(k, C.c_uint32(cls), C.c_uint32(col), out.ctypes.data_as(C.POINTER(C.c_float)),
                                  C.c_uint32(n))
    return out


def t_ppf(p, df):
    return lib().mcpo_t_ppf(p, df)


def norm_ppf(p):
    return lib().mcpo_norm_ppf(p)


def chi2_ppf(p, k):
    return lib().mcpo_chi2_ppf(p, k)


def f_ppf(p, dfn, dfd):
    return lib().mcpo_f_ppf(p, dfn, dfd)


def q_tukey_ppf(p, k, df):
    return lib().mcpo_q_tukey_ppf(p, k, df)


def phi(z):
    return lib().mcpo_phi(z)


def generate_sim_data(spec: S.SimulationSpec, base_seed: int, sim_id: int, max_n: int):
    """(design (max_n, p) f32 Fortran-ordered view, outcome (max_n,) f32, cluster_ids or None)."""
    low = spec.to_c()
    p = spec.n_predictors
    x = np.zeros((p, max_n), dtype=np.float32)  # column-major max_n x p
    y = np.zeros(max_n, dtype=np.float32)
    ids = np.zeros(max_n, dtype=np.uint32)
    rc = lib().mcpo_generate_sim_data(low.byref(), C.c_uint64(base_seed), C.c_uint64(sim_id), C.c_uint32(max_n),
                                      x.ctypes.data_as(C.POINTER(C.c_float)), y.ctypes.data_as(C.POINTER(C.c_float)),
                                      ids.ctypes.data_as(C.POINTER(C.c_uint32)))
    _check(rc)
    return x.T, y, (ids if spec.cluster is not None else None)


def generate_sim_data_ids(spec: S.SimulationSpec, base_seed: int, sim_id: int, max_n: int):
    """`generate_sim_data` plus the level ids of the extra groupings: (design, outcome, cluster_ids, [ids_g ...])."""
    low = spec.to_c()
    p = spec.n_predictors
    x = np.zeros((p, max_n), dtype=np.float32)
    y = np.zeros(max_n, dtype=np.float32)
    ids = np.zeros(max_n, dtype=np.uint32)
    ng = len(spec.cluster.extra_groupings) if spec.cluster is not None else 0
    ex = np.zeros((max(ng, 1), max_n), dtype=np.uint32)
    rc = lib().mcpo_generate_sim_data_ids(low.byref(), C.c_uint64(base_seed), C.c_uint64(sim_id), C.c_uint32(max_n),
                                          x.ctypes.data_as(C.POINTER(C.c_float)), y.ctypes.data_as(C.POINTER(C.c_float)),
                                          ids.ctypes.data_as(C.POINTER(C.c_uint32)),
                                          ex.ctypes.data_as(C.POINTER(C.c_uint32)))
    _check(rc)
    return x.T, y, (ids if spec.cluster is not None else None), [ex[g] for g in range(ng)]


def build_crit_table(spec: S.SimulationSpec, sample_sizes) -> np.ndarray:
    low = spec.to_c()
    ns = np.ascontiguousarray(sample_sizes, dtype=np.uint32)
    out = np.zeros((ns.size, spec.n_targets + 4), dtype=np.float64)
    rc = lib().mcpo_build_crit_table(low.byref(), ns.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint32(ns.size),
                                     out.ctypes.data_as(C.POINTER(C.c_double)))
    _check(rc)
    return out


def ols_fit(design: np.ndarray, outcome: np.ndarray):
    """OLS on provided bytes: design (n, p), outcome (n,). Returns dict(beta, se, t_sq, rss, sst, converged)."""
    n, p = design.shape
    x = np.ascontiguousarray(design.T, dtype=np.float32)
    y = np.ascontiguousarray(outcome, dtype=np.float32)
    beta, se, tsq = (np.zeros(p) for _ in range(3))
    rss, sst, conv = C.c_double(), C.c_double(), C.c_uint8()
    pd = C.POINTER(C.c_double)
    rc = lib().mcpo_ols_fit(x.ctypes.data_as(C.POINTER(C.c_float)), y.ctypes.data_as(C.POINTER(C.c_float)),
                            C.c_uint32(n), C.c_uint32(n), C.c_uint32(p), beta.ctypes.data_as(pd), se.ctypes.data_as(pd),
                            tsq.ctypes.data_as(pd), C.byref(rss), C.byref(sst), C.byref(conv))
    _check(rc)
    return dict(beta=beta, se=se, t_sq=tsq, rss=rss.value, sst=sst.value, converged=bool(conv.value))


GLM_FAMILIES = {"logit": 0, "probit": 1, "poisson": 2}


def glm_fit(design: np.ndarray, outcome: np.ndarray, beta_start=None, family: str = "logit"):
    """IRLS on provided bytes: design (n, p), outcome (n,); family logit / probit (y in {0,1}) or poisson (counts)."""
    n, p = design.shape
    x = np.ascontiguousarray(design.T, dtype=np.float32)
    y = np.ascontiguousarray(outcome, dtype=np.float32)
    beta, se, zsq = (np.zeros(p) for _ in range(3))
    dev, dev0, conv, nit = C.c_double(), C.c_double(), C.c_uint8(), C.c_uint32()
    pd = C.POINTER(C.c_double)
    b0 = None if beta_start is None else np.ascontiguousarray(beta_start, dtype=np.float64)
    rc = lib().mcpo_glm_fit_family(C.c_int(GLM_FAMILIES[family]), x.ctypes.data_as(C.POINTER(C.c_float)),
                                   y.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint32(n), C.c_uint32(n), C.c_uint32(p),
                                   None if b0 is None else b0.ctypes.data_as(pd), beta.ctypes.data_as(pd),
                                   se.ctypes.data_as(pd), zsq.ctypes.data_as(pd), C.byref(dev), C.byref(dev0),
                                   C.byref(conv), C.byref(nit))
    _check(rc)
    return dict(beta=beta, se=se, z_sq=zsq, deviance=dev.value, deviance_null=dev0.value,
                converged=bool(conv.value), n_iter=nit.value)


def truth_theta(spec: S.SimulationSpec, include_slopes: bool = True):
    """`truth_theta` (mixed_workspace.rs:112-198): per grouping the column-major vech of chol(diag(tau) R diag(tau))."""
    out = []
    groups = [(spec.cluster.tau_squared, spec.cluster.slopes)] + [
        (g["tau_squared"], g.get("slopes", []) or []) for g in spec.cluster.extra_groupings]
    for tau_sq, slopes in groups:
        if not include_slopes or not slopes:
            out.append(max(tau_sq, 0.0) ** 0.5)
            continue
        q = 1 + len(slopes)
        tau = np.array([max(tau_sq, 0.0) ** 0.5] + [max(s["variance"], 0.0) ** 0.5 for s in slopes])
        d = tau[:, None] * np.array(S.re_correlation_matrix(slopes)).reshape(q, q) * tau[None, :]
        l = np.zeros((q, q))
        for j in range(q):                      # chol_lower (data_gen.rs:55-75)
            diag = d[j, j] - (l[j, :j] ** 2).sum()
            l[j, j] = max(diag, 0.0) ** 0.5
            for i in range(j + 1, q):
                if l[j, j] > 0.0:
                    l[i, j] = (d[i, j] - (l[i, :j] * l[j, :j]).sum()) / l[j, j]
        out += [l[r, c] for c in range(q) for r in range(c, q)]
    return out


def lmm_general_fit(design: np.ndarray, outcome: np.ndarray, term_ids, term_q, slope_cols, theta_start=None, elim: int = 0):
    """General REML LMM on provided bytes.  term_ids: list of (n,) 0-based level ids, primary first;
    term_q: RE width per term (1 + slopes); slope_cols: per term the design columns of its slopes."""
    n, p = design.shape
    x = np.ascontiguousarray(design.T, dtype=np.float32)
    y = np.ascontiguousarray(outcome, dtype=np.float32)
    nt = len(term_ids)
    ids = np.ascontiguousarray(np.stack([np.asarray(v, dtype=np.uint32) for v in term_ids]))
    q = np.asarray(term_q, dtype=np.uint32)
    levels = np.asarray([int(v.max()) + 1 for v in ids], dtype=np.uint32)
    cols = np.zeros((nt, S.MAX_RE_SLOPES), dtype=np.uint32)
    for t, cs in enumerate(slope_cols):
        cols[t, :len(cs)] = cs
    n_theta = int(sum(int(v) * (int(v) + 1) // 2 for v in q))
    beta, se, zsq = (np.zeros(p) for _ in range(3))
    theta = np.zeros(max(n_theta, 1))
    vars_out = np.zeros((nt, 3, 3))
    s2 = C.c_double()
    conv, bh, ne, pinned = C.c_uint8(), C.c_uint8(), C.c_uint32(), C.c_uint64()
    pd, pu = C.POINTER(C.c_double), C.POINTER(C.c_uint32)
    t0 = None if theta_start is None else np.ascontiguousarray(theta_start, dtype=np.float64)
    rc = lib().mcpo_lmm_general_fit(x.ctypes.data_as(C.POINTER(C.c_float)), y.ctypes.data_as(C.POINTER(C.c_float)),
                                    C.c_uint32(n), C.c_uint32(n), C.c_uint32(p), C.c_uint32(nt), q.ctypes.data_as(pu),
                                    levels.ctypes.data_as(pu), ids.ctypes.data_as(pu), cols.ctypes.data_as(pu),
                                    C.c_uint32(elim), None if t0 is None else t0.ctypes.data_as(pd), beta.ctypes.data_as(pd),
                                    se.ctypes.data_as(pd), zsq.ctypes.data_as(pd), theta.ctypes.data_as(pd),
                                    C.byref(s2), vars_out.ctypes.data_as(pd), C.byref(conv), C.byref(bh),
                                    C.byref(pinned), C.byref(ne))
    _check(rc)
    return dict(beta=beta, se=se, z_sq=zsq, theta=theta[:n_theta], sigma_sq=s2.value, cov=vars_out,
                converged=bool(conv.value), boundary_hit=conv.value and bh.value, pinned=pinned.value, n_eval=ne.value)


def glmm_general_fit(design: np.ndarray, outcome: np.ndarray, term_ids, term_q, slope_cols, family: str = "logit",
                     theta_start=None, beta_start=None, wald_se: str = "rx", elim: int = 0):
    """GLMM (Laplace, "rx" SE) with any carried random-effect structure on provided bytes; arguments as lmm_general_fit.
    elim: the term that leads the index layout of the random effects (a cost choice, not a model choice)."""
    n, p = design.shape
    x = np.ascontiguousarray(design.T, dtype=np.float32)
    y = np.ascontiguousarray(outcome, dtype=np.float32)
    nt = len(term_ids)
    ids = np.ascontiguousarray(np.stack([np.asarray(v, dtype=np.uint32) for v in term_ids]))
    q = np.asarray(term_q, dtype=np.uint32)
    levels = np.asarray([int(v.max()) + 1 for v in ids], dtype=np.uint32)
    cols = np.zeros((nt, S.MAX_RE_SLOPES), dtype=np.uint32)
    for t, cs in enumerate(slope_cols):
        cols[t, :len(cs)] = cs
    n_theta = int(sum(int(v) * (int(v) + 1) // 2 for v in q))
    beta, se = np.zeros(p), np.zeros(p)
    theta, vars_out = np.zeros(max(n_theta, 1)), np.zeros((nt, 3, 3))
    conv, bh, ne, pinned = C.c_uint8(), C.c_uint8(), C.c_uint32(), C.c_uint64()
    pd, pu = C.POINTER(C.c_double), C.POINTER(C.c_uint32)
    t0 = None if theta_start is None else np.ascontiguousarray(theta_start, dtype=np.float64)
    b0 = None if beta_start is None else np.ascontiguousarray(beta_start, dtype=np.float64)
    rc = lib().mcpo_glmm_general_fit(C.c_int(GLM_FAMILIES[family]), x.ctypes.data_as(C.POINTER(C.c_float)),
                                     y.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint32(n), C.c_uint32(n), C.c_uint32(p),
                                     C.c_uint32(nt), q.ctypes.data_as(pu), levels.ctypes.data_as(pu), ids.ctypes.data_as(pu),
                                     cols.ctypes.data_as(pu), None if t0 is None else t0.ctypes.data_as(pd),
                                     None if b0 is None else b0.ctypes.data_as(pd), beta.ctypes.data_as(pd),
                                     se.ctypes.data_as(pd), theta.ctypes.data_as(pd), vars_out.ctypes.data_as(pd),
                                     C.byref(conv), C.byref(bh), C.byref(pinned), C.byref(ne),
                                     C.c_uint32((1 if wald_se == "hessian" else 0) | (int(elim) << 16)))
    _check(rc)
    return dict(beta=beta, se=se, theta=theta[:n_theta], cov=vars_out, converged=bool(conv.value), boundary_hit=bh.value,
                pinned=pinned.value, n_eval=ne.value)


def glmm_fit(design: np.ndarray, outcome: np.ndarray, cluster_ids: np.ndarray, family: str = "logit", theta_start: float = 0.0,
             beta_start=None, wald_se: str = "rx", nagq: int = 1):
    """Random-intercept GLMM (Laplace or nagq-point adaptive Gauss-Hermite; Wald SE mode "rx" or "hessian") on provided bytes."""
    n, p = design.shape
    x = np.ascontiguousarray(design.T, dtype=np.float32)
    y = np.ascontiguousarray(outcome, dtype=np.float32)
    ids = np.ascontiguousarray(cluster_ids, dtype=np.uint32)
    ncl = int(ids.max()) + 1
    beta, se, zsq = (np.zeros(p) for _ in range(3))
    th, t2 = C.c_double(), C.c_double()
    conv, bh, ne = C.c_uint8(), C.c_uint8(), C.c_uint32()
    pd = C.POINTER(C.c_double)
    b0 = None if beta_start is None else np.ascontiguousarray(beta_start, dtype=np.float64)
    rc = lib().mcpo_glmm_fit(C.c_int(GLM_FAMILIES[family]), x.ctypes.data_as(C.POINTER(C.c_float)),
                             y.ctypes.data_as(C.POINTER(C.c_float)), ids.ctypes.data_as(C.POINTER(C.c_uint32)),
                             C.c_uint32(n), C.c_uint32(n), C.c_uint32(p), C.c_uint32(ncl), C.c_double(theta_start),
                             None if b0 is None else b0.ctypes.data_as(pd), beta.ctypes.data_as(pd), se.ctypes.data_as(pd),
                             zsq.ctypes.data_as(pd), C.byref(th), C.byref(t2), C.byref(conv), C.byref(bh), C.byref(ne),
                             C.c_uint32((1 if wald_se == "hessian" else 0) | (int(nagq) << 8)))
    _check(rc)
    return dict(beta=beta, se=se, z_sq=zsq, theta=th.value, tau_sq=t2.value, converged=bool(conv.value),
                boundary_hit=bh.value, n_eval=ne.value)


def gh_rule(k: int):
    """k-point Gauss-Hermite rule for the standard normal weight: (nodes, weights), sum w = 1."""
    z, w = np.zeros(k), np.zeros(k)
    pd = C.POINTER(C.c_double)
    lib().mcpo_gh_rule(C.c_uint32(k), z.ctypes.data_as(pd), w.ctypes.data_as(pd))
    return z, w


def glmm_slopes_fit(design: np.ndarray, outcome: np.ndarray, cluster_ids: np.ndarray, slope_cols, family: str = "logit",
                    theta_start=None, beta_start=None, wald_se: str = "rx", nagq: int = 1):
    """GLMM with random slopes on the primary grouping (Laplace; Wald SE mode "rx" or "hessian") on provided bytes."""
    n, p = design.shape
    x = np.ascontiguousarray(design.T, dtype=np.float32)
    y = np.ascontiguousarray(outcome, dtype=np.float32)
    ids = np.ascontiguousarray(cluster_ids, dtype=np.uint32)
    ncl, q = int(ids.max()) + 1, 1 + len(slope_cols)
    cols = np.asarray(list(slope_cols) + [0, 0], dtype=np.uint32)
    beta, se = np.zeros(p), np.zeros(p)
    theta, vars_out = np.zeros(6), np.zeros((3, 3))
    conv, bh, ne, pinned = C.c_uint8(), C.c_uint8(), C.c_uint32(), C.c_uint64()
    pd = C.POINTER(C.c_double)
    t0 = None if theta_start is None else np.ascontiguousarray(theta_start, dtype=np.float64)
    b0 = None if beta_start is None else np.ascontiguousarray(beta_start, dtype=np.float64)
    rc = lib().mcpo_glmm_slopes_fit(C.c_int(GLM_FAMILIES[family]), x.ctypes.data_as(C.POINTER(C.c_float)),
                                    y.ctypes.data_as(C.POINTER(C.c_float)), ids.ctypes.data_as(C.POINTER(C.c_uint32)),
                                    C.c_uint32(n), C.c_uint32(n), C.c_uint32(p), C.c_uint32(ncl), C.c_uint32(q),
                                    cols.ctypes.data_as(C.POINTER(C.c_uint32)), None if t0 is None else t0.ctypes.data_as(pd),
                                    None if b0 is None else b0.ctypes.data_as(pd), beta.ctypes.data_as(pd),
                                    se.ctypes.data_as(pd), theta.ctypes.data_as(pd), vars_out.ctypes.data_as(pd),
                                    C.byref(conv), C.byref(bh), C.byref(pinned), C.byref(ne),
                                    C.c_uint32((1 if wald_se == "hessian" else 0) | (int(nagq) << 8)))
    _check(rc)
    return dict(beta=beta, se=se, theta=theta[:q * (q + 1) // 2], cov=vars_out[:q, :q], converged=bool(conv.value),
                boundary_hit=bh.value, pinned=pinned.value, n_eval=ne.value)


def lmm_fit(design: np.ndarray, outcome: np.ndarray, cluster_ids: np.ndarray, theta_start: float = 0.0):
    """Random-intercept REML on provided bytes: design (n, p), outcome (n,), 0-based cluster ids (n,)."""
    n, p = design.shape
    x = np.ascontiguousarray(design.T, dtype=np.float32)
    y = np.ascontiguousarray(outcome, dtype=np.float32)
    ids = np.ascontiguousarray(cluster_ids, dtype=np.uint32)
    ncl = int(ids.max()) + 1
    beta, se, zsq = (np.zeros(p) for _ in range(3))
    th, s2, t2 = C.c_double(), C.c_double(), C.c_double()
    conv, bh, ne = C.c_uint8(), C.c_uint8(), C.c_uint32()
    pd = C.POINTER(C.c_double)
    rc = lib().mcpo_lmm_fit(x.ctypes.data_as(C.POINTER(C.c_float)), y.ctypes.data_as(C.POINTER(C.c_float)),
                            ids.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint32(n), C.c_uint32(n), C.c_uint32(p),
                            C.c_uint32(ncl), C.c_double(theta_start), beta.ctypes.data_as(pd), se.ctypes.data_as(pd),
                            zsq.ctypes.data_as(pd), C.byref(th), C.byref(s2), C.byref(t2), C.byref(conv), C.byref(bh),
                            C.byref(ne))
    _check(rc)
    return dict(beta=beta, se=se, z_sq=zsq, theta=th.value, sigma_sq=s2.value, tau_sq=t2.value,
                converged=bool(conv.value), boundary_hit=int(bh.value), n_eval=ne.value)


def run_batch(spec: S.SimulationSpec, sample_sizes, n_sims: int, base_seed: int, *, sim_offset: int = 0,
              stat_capture: bool = False, n_threads: int = 1) -> S.BatchResult:
    """Oracle twin of `mcpower_amd.engine.run_batch` (reference batch.rs:130)."""
    low = spec.to_c()
    ns = np.ascontiguousarray(sample_sizes, dtype=np.uint32)
    opts = S.CRunOptions()
    opts.flags = S.WANT_PER_SIM | (S.WANT_STAT_CAPTURE if stat_capture else 0)
    opts.sim_offset = sim_offset
    res = S.CBatchResult()
    rc = lib().mcpo_run_batch(low.byref(), ns.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint32(ns.size),
                              C.c_uint32(n_sims), C.c_uint64(base_seed), C.byref(opts), C.c_uint32(n_threads),
                              C.byref(res))
    _check(rc)
    try:
        return S.result_from_c(res)
    finally:
        lib().mcpo_free_result(C.byref(res))


def set_glmm_general_pirls_cap(cap: int) -> None:
    """Test hook of fit_glmm_general.c: Fisher steps per evaluation (0 restores the default 60)."""
    lib().mcpo_set_glmm_general_pirls_cap(int(cap))
