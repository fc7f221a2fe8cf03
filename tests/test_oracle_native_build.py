"""The `-O3 -march=native` oracle build that bench.py's cpu_baseline leg times must give the portable build's results.

Regression: gcc 11.4 vectorised the simplex shrink loop of oracle/fit_glmm_general.c with an aligned load at a 16-byte
aligned address, and the (1 + x1 | g) + (1 | item) bench shape died with SIGSEGV on its first shrink step -- only in the
native build, so no parity test saw it.  Each build runs in its own process (use_native_build rebinds the library)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
from oracle import oracle
from mcpower_amd import workloads as W
if sys.argv[1] == "native":
    assert "-march=native" in oracle.use_native_build()
out = {}
for fac, n in (("spec_glmm_crossed", 480), ("spec_glmm_slope", 600), ("spec_lmm_crossed", 480)):
    o = oracle.run_batch(getattr(W, fac)(), np.array([n], dtype=np.uint32), 6, W.BASE_SEED, n_threads=2)
    out[fac] = {k: np.asarray(v).ravel().tolist() for k, v in vars(o).items() if isinstance(v, np.ndarray) and v.dtype.kind in "iu"}
print("RESULT " + json.dumps(out))
"""


def _run(kind):
    r = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT}, kind], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, f"{kind} oracle build died (rc {r.returncode}): {r.stderr[-400:]}"
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_native_build_matches_portable_build_on_bench_shapes():
    portable = _run("portable")
    native = _run("native")
    assert all("uncorrected" in v and "converged" in v for v in portable.values()) and len(portable) == 3
    assert portable == native
