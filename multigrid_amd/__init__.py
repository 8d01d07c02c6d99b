"""multigrid_amd -- MI355X (gfx950) 2D Poisson geometric-multigrid V-cycle.

Host-side mirror of the reference's solver interface (main.cpp) over the HIP
library libmgx.so; see include/mgx.h for the C ABI.  No CPU fallback exists:
without the built library or without a GPU, calls raise.
"""
import torch  # noqa: F401  (loads the HIP runtime the library binds to)

from . import _lib
from ._lib import COARSE_EXACT, COARSE_REFERENCE, MgxError, SCHEDULE_DEFAULT, SCHEDULE_FUSED, SCHEDULE_STEPWISE
from .solver import (FullMultiGrid, MGOutputs, MultiGrid, MultiPlan, Plan, calcNorm, coarse_exact, coarsen, colmajor, direct_solve, fused_down, fused_up,
                     gridtype, prolong_add, refine, relax, residual, residual_norm, solve_host, to_device, to_numpy)

__all__ = [
    "FullMultiGrid", "COARSE_EXACT", "COARSE_REFERENCE", "coarse_exact", "MgxError", "SCHEDULE_DEFAULT", "SCHEDULE_FUSED", "SCHEDULE_STEPWISE", "MGOutputs", "MultiGrid", "MultiPlan", "Plan",
    "calcNorm", "coarsen", "colmajor", "direct_solve", "fused_down", "fused_up", "gridtype", "prolong_add", "refine", "relax", "residual",
    "residual_norm", "solve_host", "to_device", "to_numpy",
]
