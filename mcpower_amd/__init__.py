"""mcpower_amd -- MI355X-native engine for MCPower's simulate loop.

Package contents (only what the hot path needs):
  csrc/      HIP kernels (gfx950) + the C-ABI host layer -> libmcpower_hip.so
  spec.py    host-side mirror of the reference's SimulationSpec / BatchResult
  engine.py  ctypes binding mirroring engine_core::{run_batch, run_batch_st, run_batch_st_capture}
  dist.py    one-process-per-GPU sharding + the single counter all-reduce (RCCL)
"""
from .spec import (BatchResult, ClusterSpec, CorrectionMethod, CritValues, Distribution, EngineError,  # noqa: F401
                   EstimatorSpec, LinkKind, OutcomeKind, PosthocSpec, ResidualDist, ScenarioPerturbations,
                   SimulationSpec, Status)
