"""ctypes binding of libctdirect_hip.so -- the C ABI declared in include/ctdirect_hip.h.

The library is the product: if it is missing or fails to load, importing the engine fails loudly.  There is no
Python / NumPy / PyTorch compute fallback anywhere in this package.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CTD_LIB_PATH") or os.path.join(_HERE, "libctdirect_hip.so")      # (override: A/B runs of experiment builds)
CSRC = os.path.join(_HERE, "csrc")

# status codes (include/ctdirect_hip.h)
CTD_OK, CTD_EINVAL, CTD_EGRID, CTD_ESCHEME, CTD_EPATTERN, CTD_EPROBLEM, CTD_ENODEVICE, CTD_EHIP, CTD_ENOMEM, CTD_ERCCL = range(10)


class ctd_desc(C.Structure):
    _fields_ = [("problem", C.c_int32), ("scheme", C.c_int32), ("pattern_mode", C.c_int32), ("device", C.c_int32),
                ("grid_size", C.c_int64), ("time_grid", C.POINTER(C.c_double)), ("time_grid_len", C.c_int64),
                ("step_begin", C.c_int64), ("step_end", C.c_int64), ("stream", C.c_void_p),
                ("stream_mode", C.c_int32), ("control_steps", C.c_int32), ("value_order", C.c_int32), ("reserved0", C.c_int32)]


class ctd_init(C.Structure):
    _fields_ = [("use_problem_default", C.c_int32), ("state", C.POINTER(C.c_double)),
                ("control", C.POINTER(C.c_double)), ("variable", C.POINTER(C.c_double)),
                ("n_samples", C.c_int64), ("t_samples", C.POINTER(C.c_double)),
                ("state_samples", C.POINTER(C.c_double)), ("control_samples", C.POINTER(C.c_double))]


class ctd_ocp_def(C.Structure):
    _fields_ = [("name", C.c_char_p), ("n", C.c_int32), ("m", C.c_int32), ("nv", C.c_int32), ("npath", C.c_int32),
                ("nbc", C.c_int32), ("it0", C.c_int32), ("itf", C.c_int32), ("t0", C.c_double), ("tf", C.c_double),
                ("maximize", C.c_int32), ("reserved", C.c_int32), ("dynamics", C.POINTER(C.c_char_p)),
                ("lagrange", C.c_char_p), ("mayer", C.c_char_p), ("path", C.POINTER(C.c_char_p)),
                ("boundary", C.POINTER(C.c_char_p)), ("constants", C.c_char_p)] + \
               [(k, C.POINTER(C.c_double)) for k in ("state_lb", "state_ub", "control_lb", "control_ub", "variable_lb",
                                                     "variable_ub", "path_lb", "path_ub", "boundary_lb", "boundary_ub")]


# every symbol include/ctdirect_hip.h declares: name -> (restype, argtypes)
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)
_vp = C.c_void_p
SYMBOLS = {
    "ctd_create": (C.c_int32, [C.POINTER(ctd_desc), C.POINTER(_vp)]),
    "ctd_destroy": (C.c_int32, [_vp]),
    "ctd_last_error": (C.c_char_p, [_vp]),
    "ctd_strerror": (C.c_char_p, [C.c_int32]),
    "ctd_host_alloc": (C.c_int32, [C.POINTER(C.c_void_p), C.c_size_t]),
    "ctd_host_free": (C.c_int32, [C.c_void_p]),
    "ctd_sizes": (C.c_int32, [_vp, _ip, _ip, _ip, _ip]),
    "ctd_dims": (C.c_int32, [_vp, _ip]),
    "ctd_time_grid": (C.c_int32, [_vp, _dp, _dp]),
    "ctd_time_grid_at": (C.c_int32, [_vp, _dp, _dp]),
    "ctd_butcher": (C.c_int32, [_vp, _dp, _dp, _dp]),
    "ctd_bounds": (C.c_int32, [_vp, _dp, _dp, _dp, _dp]),
    "ctd_initial_guess": (C.c_int32, [_vp, _dp, C.POINTER(ctd_init)]),
    "ctd_jac_structure": (C.c_int32, [_vp, _ip, _ip]),
    "ctd_jac_csc": (C.c_int32, [_vp, _ip, _ip]),
    "ctd_jac_csr": (C.c_int32, [_vp, _ip, _ip]),
    "ctd_value_order": (C.c_int32, [_vp, C.POINTER(C.c_int32)]),
    "ctd_dropped_nonzeros": (C.c_int32, [_vp, _ip]),
    "ctd_obj": (C.c_int32, [_vp, _dp, _dp]),
    "ctd_grad": (C.c_int32, [_vp, _dp, _dp]),
    "ctd_grad_dev": (C.c_int32, [_vp, _vp, _vp]),
    "ctd_cons": (C.c_int32, [_vp, _dp, _dp]),
    "ctd_jac_coord": (C.c_int32, [_vp, _dp, _dp]),
    "ctd_cons_jac": (C.c_int32, [_vp, _dp, _dp, _dp]),
    "ctd_cons_jac_dev": (C.c_int32, [_vp, _vp, _vp, _vp]),
    "ctd_cons_jac_dev_async": (C.c_int32, [_vp, _vp, _vp, _vp]),
    "ctd_obj_dev": (C.c_int32, [_vp, _vp, _dp]),
    "ctd_obj_dev_async": (C.c_int32, [_vp, _vp, _vp]),
    "ctd_grad_dev_async": (C.c_int32, [_vp, _vp, _vp]),
    "ctd_grad_shard_dev_async": (C.c_int32, [_vp, _vp, _vp]),
    "ctd_sync": (C.c_int32, [_vp]),
    "ctd_set_stream": (C.c_int32, [_vp, _vp]),
    "ctd_shard_info": (C.c_int32, [_vp, _ip]),
    "ctd_time_cons_jac_dev": (C.c_int32, [_vp, _vp, _vp, _vp, C.c_int32, _dp]),
    "ctd_launch_info": (C.c_int32, [_vp, _ip]),
    "ctd_debug_stamps": (C.c_int32, [_vp, _vp, _vp, _vp, C.POINTER(C.c_uint64), C.c_int64]),
    "ctd_register_ocp": (C.c_int32, [C.POINTER(ctd_ocp_def), C.POINTER(C.c_int32)]),
    "ctd_ocp_source": (C.c_int32, [C.c_int32, C.c_char_p, C.c_int64]),
    "ctd_jit_check": (C.c_int32, [C.c_int32, C.c_int32]),
    "ctd_hess_structure": (C.c_int32, [_vp, _ip, _ip]),
    "ctd_hess_csc": (C.c_int32, [_vp, _ip, _ip]),
    "ctd_hess_csr": (C.c_int32, [_vp, _ip, _ip]),
    "ctd_hess_coord": (C.c_int32, [_vp, _dp, _dp, C.c_double, _dp]),
    "ctd_hess_coord_dev": (C.c_int32, [_vp, _vp, _vp, C.c_double, _vp]),
    "ctd_hess_coord_dev_async": (C.c_int32, [_vp, _vp, _vp, C.c_double, _vp]),
    "ctd_time_hess_dev": (C.c_int32, [_vp, _vp, _vp, C.c_double, _vp, C.c_int32, _dp]),
    "ctd_hess_launch_info": (C.c_int32, [_vp, _ip]),
    "ctd_hess_kernel_info": (C.c_int32, [_vp, _ip]),
    "ctd_hess_shard_info": (C.c_int32, [_vp, _ip]),
    "ctd_hess_debug_stamps": (C.c_int32, [_vp, _vp, _vp, C.c_double, _vp, C.POINTER(C.c_uint64), C.c_int64]),
    "ctd_eval_all_dev_async": (C.c_int32, [_vp, _vp, _vp, C.c_double, _vp, _vp, _vp, _vp, _vp]),
    # one transcription on several GPUs of one process
    "ctd_create_sharded": (C.c_int32, [C.POINTER(ctd_desc), C.POINTER(C.c_int32), C.c_int32, C.POINTER(_vp)]),
    "ctd_sharded_destroy": (C.c_int32, [_vp]),
    "ctd_sharded_last_error": (C.c_char_p, [_vp]),
    "ctd_sharded_handle": (C.c_int32, [_vp, C.c_int32, C.POINTER(_vp)]),
    "ctd_sharded_shard_info": (C.c_int32, [_vp, C.c_int32, _ip]),
    "ctd_cons_jac_sharded_dev_async": (C.c_int32, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.c_int32, C.c_int32]),
    "ctd_sharded_sync": (C.c_int32, [_vp]),
    "ctd_dev_alloc": (C.c_int32, [C.c_int32, C.c_size_t, C.POINTER(_vp)]),
    "ctd_dev_free": (C.c_int32, [C.c_int32, _vp]),
    "ctd_dev_copy": (C.c_int32, [C.c_int32, _vp, _vp, C.c_size_t, C.c_int32]),
    # sharded iterate read in place + buffers shared between the processes of a node
    "ctd_set_x_shards": (C.c_int32, [_vp, C.c_int32, _ip, C.POINTER(_vp), C.c_int32]),
    "ctd_ipc_export": (C.c_int32, [C.c_int32, _vp, _vp, _ip]),
    "ctd_ipc_open": (C.c_int32, [C.c_int32, _vp, C.POINTER(_vp)]),
    "ctd_ipc_close": (C.c_int32, [C.c_int32, _vp]),
    "ctd_ipc_probe": (C.c_int32, [C.c_int32, _vp, C.c_size_t]),
    "ctd_stitch_c": (C.c_int32, [_vp, _vp, C.c_int32, C.c_int32, _vp]),
    "ctd_shard_steps": (C.c_int32, [C.c_int64, C.c_int32, C.c_int32, _ip, _ip]),
}


def build(jobs=8):
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-C", CSRC, "-s", f"-j{jobs}"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension has not been built (run `python -c 'import "
                "__graft_entry__ as g; g.build()'` or `make -C ctdirect.jl_amd/csrc`). There is no fallback path.")
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 (same soname as /opt/rocm's).  If torch is
        # going to be used in this process it must be loaded first so that this library binds to the same runtime
        # (two runtimes in one process do not see each other's devices, streams or allocations).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            if os.environ.get("CTD_LIB_PATH") and not hasattr(L, name):
                continue                  # (A/B runs against an older experiment build: it may lack the newest entry points)
            f = getattr(L, name)          # AttributeError if the library does not export the symbol
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib
