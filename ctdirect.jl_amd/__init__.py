"""MI355X-native collocation engine behind CTDirect.jl's NLP-callback boundary.

The compute path is libctdirect_hip.so (hand-written HIP for gfx950, C ABI in include/ctdirect_hip.h); this package
is the thin host-side mirror of the reference's DOCP interface plus the multi-GPU stitching helper.
"""
from . import _lib
from .docp import (DOCP, MultiDeviceDOCP, PATTERN_MODES, PROBLEMS, SCHEMES, CTDirectError, DOCP_Hessian_csr, DOCP_Hessian_pattern, DOCP_Jacobian_csr, DOCP_Jacobian_pattern, VALUE_ORDERS, constraints,
                   constraints_bounds, get_time_grid, gradient, initial_guess, jit_check, objective, ocp_source, pinned_empty,
                   register_ocp, unpack_solution, variables_bounds)

__all__ = ["DOCP", "MultiDeviceDOCP", "PROBLEMS", "SCHEMES", "PATTERN_MODES", "CTDirectError", "DOCP_Hessian_pattern", "DOCP_Jacobian_pattern", "DOCP_Jacobian_csr", "DOCP_Hessian_csr", "VALUE_ORDERS", "constraints",
           "constraints_bounds", "get_time_grid", "unpack_solution", "gradient", "initial_guess", "jit_check", "objective", "ocp_source", "register_ocp",
           "variables_bounds", "pinned_empty", "build"]


def build(jobs=8):
    """Compile the HIP extension in-tree (gfx950)."""
    return _lib.build(jobs)
