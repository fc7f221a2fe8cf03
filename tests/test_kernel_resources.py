"""The mixed-model kernels must not touch scratch memory.

`tools/kernel_resources.sh` reads the gfx950 code object embedded in each host object: per kernel the private segment
size (= scratch bytes per lane), VGPRs and LDS.  The call-structured optimiser drivers of round 2 left 536-916 B of
scratch per lane in every instantiation of glmm_kernel / glmm_slopes_kernel / lmm_general_kernel (33 GB of scratch writes
per 40 000-fit launch, profiles/r02_c_lmm_slope_pmc.csv); since round 3 the fit is one state machine around a single
inlined evaluation (mcp_fit_fsm.hpp) and the once-per-fit tail works in LDS (mcp_glmm_tail.hpp).  This test pins that:
a change that brings a spill or a call back fails here, on the CPU, before any GPU time is spent."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "mcpower_amd", "csrc", "_build")
# object -> (kernel name, minimum number of instantiations expected in it)
OBJECTS = {
    "glmm_kernels.o": ("glmm_kernel", 14),
    "glmm_slopes_kernels.o": ("glmm_slopes_kernel", 26),
    "glmm_general_kernels.o": ("glmm_general_kernel", 14),
    "lmm_general_kernels.o": ("lmm_general_kernel", 28),
    "lmm_general_scen_kernels.o": ("lmm_general_kernel", 28),
}


@pytest.mark.parametrize("obj", sorted(OBJECTS))
def test_mixed_model_kernels_use_no_scratch(obj, engine):
    path = os.path.join(BUILD, obj)
    if not os.path.exists(path):
        engine.build()
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "kernel_resources.sh"), path], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    name, at_least = OBJECTS[obj]
    rows = [ln for ln in r.stdout.splitlines() if re.search(r"\b%s<" % name, ln)]
    assert len(rows) >= at_least, (obj, len(rows))
    for ln in rows:
        m = re.search(r"scratch=(\d+) vgpr=(\d+)", ln)
        assert m, ln
        assert int(m.group(1)) == 0, f"scratch in use: {ln}"
        assert int(m.group(2)) <= 512, ln
