"""Per-draw restatement of validation/oracle2_power.R:38-66 on the CPU oracle (test infrastructure, not product).

The reference's harness refits K datasets per cell and condition: draw k is `create_data()` under
`.debug_seed = base + k` (= replicate 0 of `base_seed = splitmix64(base + k)`, introspect.rs:104-155), fitted by
`load_data(d, wald_se = "rx" | "hessian")` (= fit_provided_data, truth start).  validation/data/oracle2_<cell>.rds
froze, over the kept draws k = 1 .. 50 000: sd(beta-hat) (`sd_true`), mean SE per flavour (`mean_se_rx`,
`mean_se_hess`) and the per-draw rejection rates (`power_rx`, `power_hessian`, `typeI_rx`, `typeI_hessian`;
null draws at base + 1 000 000 + k).  The data plane is bit-exact, so at K = 50 000 these are SAME-DATA comparisons:
sd(beta-hat) isolates the optimiser's endpoint, mean_se_rx the "rx" covariance, mean_se_hess the finite-difference
Hessian -- which the two find_power rates of tests/test_oracle2_power.py cannot separate.

  python tools/oracle2_perdraw.py --cells hiICC_int --k 50000 --procs 8 [--knob NAME=VALUE ...]
"""
import argparse
import json
import math
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

Z_CRIT = 1.959963984540054


def _setup(cell_name, beta):
    from mcpower_amd import _engine, contract as K
    from test_oracle2_power import ORACLE2, cell_contract
    c = ORACLE2["cells"][cell_name]
    contracts = K.decode_contracts(cell_contract(c["cell"], beta))
    spec = _engine.lower_contracts(contracts, 0)[0]["spec"]
    return c, spec


def _fit(oracle, spec, cell, x, y, ids, extra_ids, mode):
    cl = spec.cluster
    th = oracle.truth_theta(spec, True)
    b0 = np.asarray(spec.effect_sizes, dtype=np.float64)
    pcols = list(spec.cluster_slope_design_cols)
    if cl.extra_groupings:
        ecols = [list(c) for c in spec.extra_slope_cols]
        term_ids = [ids] + list(extra_ids)
        term_q = [1 + len(pcols)] + [1 + len(c) for c in ecols]
        return oracle.glmm_general_fit(x, y, term_ids, term_q, [pcols] + ecols, family="logit", theta_start=th, beta_start=b0,
                                       wald_se=mode)
    if pcols:
        return oracle.glmm_slopes_fit(x, y, ids, pcols, theta_start=th, beta_start=b0, wald_se=mode)
    return oracle.glmm_fit(x, y, ids, theta_start=float(th[0]), beta_start=b0, wald_se=mode)


def _work(args):
    cell_name, beta, base, k_from, k_to, knobs = args
    for kv in knobs:
        name, val = kv.split("=")
        os.environ["MCPO_KNOB_" + name.upper()] = val
    from oracle import oracle
    c, spec = _setup(cell_name, beta)
    n = c["cell"]["n"]
    out = []
    for k in range(k_from, k_to + 1):
        seed = oracle.splitmix64(base + k)
        x, y, ids, ex = oracle.generate_sim_data_ids(spec, seed, 0, n)
        fr = _fit(oracle, spec, c, x, y, ids, ex, "rx")
        fh = _fit(oracle, spec, c, x, y, ids, ex, "hessian")
        ok = fr["converged"] and fh["converged"]
        sr, sh, bh = fr["se"][1], fh["se"][1], fh["beta"][1]
        if not ok or not (np.isfinite(sr) and np.isfinite(sh) and sr > 0 and sh > 0 and np.isfinite(bh)):
            out.append((k, np.nan, np.nan, np.nan, np.nan, 0))
        else:
            out.append((k, bh, sr, sh, fr["beta"][1], fh.get("n_eval", 0)))
    return out


def run(cell_name, which, K, procs, knobs):
    from test_oracle2_power import ORACLE2
    c = ORACLE2["cells"][cell_name]
    beta = c["cell"]["beta_star"] if which == "alt" else 0.0
    base = c["cell"]["seed"] + (0 if which == "alt" else 1000000)
    chunk = max(1, K // (procs * 8))
    jobs = [(cell_name, beta, base, a, min(K, a + chunk - 1), knobs) for a in range(1, K + 1, chunk)]
    with mp.Pool(procs) as pool:
        parts = pool.map(_work, jobs)
    rows = np.array([r for p in parts for r in p], dtype=np.float64)
    keep = ~np.isnan(rows[:, 1])
    return rows, keep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", default="hiICC_int")
    ap.add_argument("--k", type=int, default=2000)
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--knob", action="append", default=[])
    ap.add_argument("--save", default="")
    a = ap.parse_args()
    from test_oracle2_power import ORACLE2
    from oracle import oracle
    import shutil, tempfile
    snap = os.path.join(tempfile.mkdtemp(prefix="mcpo_"), "libmcp_oracle.so")
    shutil.copy(oracle.build(), snap)       # workers bind this snapshot: the tree may be edited while a long run is going
    os.environ["MCPO_LIB_PATH"] = snap
    for name in a.cells.split(","):
        fz = ORACLE2["cells"][name]["frozen"]
        alt, ka = run(name, "alt", a.k, a.procs, a.knob)
        null, kn = run(name, "null", a.k, a.procs, a.knob)
        b, sr, sh = alt[ka, 1], alt[ka, 2], alt[ka, 3]
        nb, nsr, nsh = null[kn, 1], null[kn, 2], null[kn, 3]
        got = dict(sd_true=float(np.std(b, ddof=1)), mean_se_rx=float(sr.mean()), mean_se_hess=float(sh.mean()),
                   power_rx=float((np.abs(b / sr) > Z_CRIT).mean()), power_hessian=float((np.abs(b / sh) > Z_CRIT).mean()),
                   typeI_rx=float((np.abs(nb / nsr) > Z_CRIT).mean()), typeI_hessian=float((np.abs(nb / nsh) > Z_CRIT).mean()),
                   kept_alt=int(ka.sum()), kept_null=int(kn.sum()), mean_evals=float(alt[ka, 5].mean()))
        print(f"== {name}  K = {a.k}  knobs {a.knob}")
        for key in ("sd_true", "mean_se_rx", "mean_se_hess", "power_rx", "power_hessian", "typeI_rx", "typeI_hessian",
                    "kept_alt", "kept_null"):
            ref = fz[key]
            g = got[key]
            rel = (g / ref - 1.0) * 100 if ref else float("nan")
            print(f"  {key:14s} {g:12.6f}  frozen {ref:12.6f}  diff {g - ref:+.6f} ({rel:+.3f} %)")
        print(f"  mean evals (hessian fit) {got['mean_evals']:.1f}")
        if a.save:
            np.savez_compressed(f"{a.save}_{name}.npz", alt=alt, null=null)
        sys.stdout.flush()


if __name__ == "__main__":
    main()
