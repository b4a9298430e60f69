"""Host-side mirror of the reference interface for the collocation hot path.

Names, argument meaning and error behaviour follow CTDirect.jl (the reference is Julia; Julia is not available in
this image, so the mirror above the C ABI is Python -- INTEGRATION.md holds the Julia `ccall` shim a maintainer
would add).  What is mirrored:

    CTDirect.DOCP(ocp, grid_size, control_steps, scheme, time_grid)          src/DOCP_data.jl:293-365
    docp.dims / docp.flags / docp.time / docp.bounds / dim_NLP_*             src/DOCP_data.jl:24-30,88-94,147-152,235-240
    CTDirect.__constraints!(c, xu, docp)                                     src/DOCP_functions.jl:80-115
    CTDirect.__objective(xu, docp)                                           src/DOCP_functions.jl:23-54
    CTDirect.__variables_bounds!(docp), __constraints_bounds!(docp)          src/DOCP_variables.jl:21-63, DOCP_functions.jl:163-191
    CTDirect.__initial_guess(docp, init)                                     src/DOCP_variables.jl:122-145
    CTDirect.DOCP_Jacobian_pattern(docp)                                     src/ode/{trapeze,midpoint,irk,irk_stagewise}.jl
    NLPModels: obj, cons!, jac_structure!, jac_coord!                        served by ADNLPModels in the reference
                                                                             (src/collocation.jl:137-149)

All arithmetic happens in the HIP library; this module only moves pointers.  NumPy arrays go through the
host-pointer entry points, torch CUDA(=HIP) tensors through the device-pointer ones.
"""
import ctypes as C
from types import SimpleNamespace

import numpy as np

from . import _lib

SCHEMES = {
    "trapeze": 0,
    "midpoint": 1,
    "gauss_legendre_1": 2,
    "gauss_legendre_2_constant_control": 3,
    "gauss_legendre_3_constant_control": 4,
    "gauss_legendre_2": 5,
    "gauss_legendre_3": 6,
    "euler": 7,
    "euler_implicit": 8,
}
# the reference's aliases (src/DOCP_data.jl:315-320)
SCHEME_ALIASES = {"euler_explicit": "euler", "euler_forward": "euler", "euler_backward": "euler_implicit"}
PROBLEMS = {
    "goddard": 0,
    "goddard_all": 1,
    "double_integrator_path": 2,
    "quadrotor": 3,
    "quadrotor12": 4,
    "stagewise_scalar": 5,
    "estimate_initial_condition": 6,
    "estimate_rotation_rate": 7,
    "least_squares_with_constraint": 8,
    "double_integrator_freet0tf": 9,
}
PATTERN_MODES = {"manual": 0, "reference_manual": 0, "structural": 1, "optimized": 2}
VALUE_ORDERS = {"csc": 0, "csr": 1}


class CTDirectError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"[ctd status {status}] {message}")
        self.status = status


def _raise(status, msg):
    # same exception kinds as the reference: ArgumentError for the grid (src/DOCP_data.jl:186-189) -> ValueError
    if status == _lib.CTD_EGRID:
        raise ValueError(msg)
    raise CTDirectError(status, msg)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _host_out(a, n, name):
    """A caller-supplied NumPy output of the host-pointer entry points: the library writes n doubles through the raw pointer,
    so anything but a C-contiguous float64 array of exactly n entries would corrupt host memory."""
    if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags["C_CONTIGUOUS"] and a.flags["WRITEABLE"] and a.size == n):
        raise ValueError(f"{name} must be a writeable C-contiguous float64 NumPy array of {n} entries")
    return a


def _is_tensor(x):
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda")


def register_ocp(name, *, dynamics, n=None, m=0, nv=0, lagrange=None, mayer=None, path=(), boundary=(), constants=None,
                 t0=0.0, tf=1.0, it0=-1, itf=-1, maximize=False, state_box=None, control_box=None, variable_box=None,
                 path_bounds=None, boundary_bounds=None):
    """Defines an OCP at run time from arithmetic expressions (the stand-in for the Julia closures of a CTModels.Model,
    include/ctdirect_hip.h `ctd_ocp_def`) and returns `name`, usable as the `ocp` argument of DOCP.

    dynamics / path: expressions in t, x1..xn, u1..um, v1..vnv; mayer / boundary: in x0_k, xf_k, vk; `constants` a dict of
    numbers (constants) and strings (aliases: named sub-expressions, e.g. the `aux = ...` lines of a CTParser @def block).
    Functions: exp log sin cos tan atan tanh sqrt abs asin acos sinh cosh floor, max(a, b), min(a, b); `^` takes a constant exponent.
    Boxes and bounds are (lb, ub) pairs of sequences (None = free boxes / equality-with-zero rows); it0 / itf are the
    0-based positions of a free initial / final time inside v.  The kernels are compiled with hiprtc when a DOCP is built."""
    L = _lib.lib()
    dynamics = list(dynamics)
    n = len(dynamics) if n is None else int(n)
    path, boundary = list(path), list(boundary)
    d = _lib.ctd_ocp_def()
    keep = []

    def strs(items):
        arr = (C.c_char_p * max(len(items), 1))(*[s.encode() for s in items])
        keep.append(arr)
        return C.cast(arr, C.POINTER(C.c_char_p))

    def dbl(seq, dim, fill):
        a = np.full(dim, fill, dtype=np.float64) if seq is None else np.ascontiguousarray(seq, dtype=np.float64)
        if a.size != dim:
            raise ValueError(f"bound array has {a.size} entries, expected {dim}")
        keep.append(a)
        return _dp(a) if dim else None

    d.name = name.encode()
    d.n, d.m, d.nv, d.npath, d.nbc = n, int(m), int(nv), len(path), len(boundary)
    d.it0, d.itf, d.t0, d.tf, d.maximize = int(it0), int(itf), float(t0), float(tf), int(bool(maximize))
    d.dynamics, d.path, d.boundary = strs(dynamics), strs(path), strs(boundary)
    d.lagrange = lagrange.encode() if lagrange else None
    d.mayer = mayer.encode() if mayer else None
    # numbers are constants; strings are aliases: named sub-expressions ("aux = 543 + 186*cos(x4) + ...") substituted where used
    d.constants = "; ".join(f"{k}={v if isinstance(v, str) else repr(float(v))}" for k, v in constants.items()).encode() if constants else None
    inf = float("inf")
    for key, box, dim in (("state", state_box, n), ("control", control_box, int(m)), ("variable", variable_box, int(nv))):
        setattr(d, key + "_lb", dbl(None if box is None else box[0], dim, -inf))
        setattr(d, key + "_ub", dbl(None if box is None else box[1], dim, inf))
    for key, bnd, dim in (("path", path_bounds, len(path)), ("boundary", boundary_bounds, len(boundary))):
        setattr(d, key + "_lb", dbl(None if bnd is None else bnd[0], dim, 0.0))
        setattr(d, key + "_ub", dbl(None if bnd is None else bnd[1], dim, 0.0))
    pid = C.c_int32()
    st = L.ctd_register_ocp(C.byref(d), C.byref(pid))
    if st != _lib.CTD_OK:
        _raise(st, L.ctd_last_error(None).decode())
    PROBLEMS[name] = pid.value
    return name


def ocp_source(name):
    """The functor text generated for a run-time OCP (diagnostics)."""
    L = _lib.lib()
    cap = 1 << 16
    for _ in range(2):
        buf = C.create_string_buffer(cap)
        st = L.ctd_ocp_source(PROBLEMS[name], buf, len(buf))
        if st == _lib.CTD_OK:
            return buf.value.decode()
        msg = L.ctd_last_error(None).decode()
        if "needs" not in msg:
            _raise(st, msg)
        cap = int(msg.split("needs")[1].split()[0])
    _raise(st, msg)


def jit_check(name, scheme):
    """Compile-only check (no GPU needed): the kernels of `scheme` build for gfx950 for a run-time OCP."""
    L = _lib.lib()
    st = L.ctd_jit_check(PROBLEMS[name], SCHEMES[scheme] if isinstance(scheme, str) else int(scheme))
    if st != _lib.CTD_OK:
        _raise(st, L.ctd_last_error(None).decode())


class DOCP:
    """Discretised OCP handle; mirrors `CTDirect.DOCP(ocp, grid_size, control_steps, scheme, time_grid)` (src/DOCP_data.jl:293).

    `ocp` is the name (or id) of a problem of the compiled registry (include/ctdirect_hip.h) -- the reference takes
    a CTModels.Model with Julia closures, which cannot cross the C ABI to the GPU.
    `device` is the HIP device ordinal; -1 builds a host-only handle (sizes, bounds, patterns, initial guess).
    `steps=(begin, end)` restricts the handle to a shard of the time grid (multi-GPU).
    `stream`: "torch" (default) launches on torch's current stream of `device` when torch sees that GPU, so the
    callbacks are ordered with the caller's tensor work; "own" gives the handle a private stream; an integer is a raw
    hipStream_t.
    `control_steps` > 1: the direct-shooting layout (src/direct_shooting.jl:55-71; :midpoint only): `control_steps` controls per
    time step, dynamics summed over the control sub-steps (midpoint.jl:137-155).
    `value_order`: "csc" (default: the reference's SparseArrays.sparse order, what jac_coord! fills) or "csr" (the same entries by
    rows: `jac_structure`, `jac_coord`, `cons_jac` follow it; `DOCP_Jacobian_csr` gives rowptr / colind).
    """

    def __init__(self, ocp, grid_size=250, scheme="midpoint", time_grid=None, *, pattern="manual", device=0,
                 steps=None, stream="torch", control_steps=1, value_order="csc"):
        L = _lib.lib()
        self.problem_name = ocp if isinstance(ocp, str) else {v: k for k, v in PROBLEMS.items()}.get(int(ocp), str(ocp))
        pid = PROBLEMS[ocp] if isinstance(ocp, str) else int(ocp)
        if isinstance(scheme, str):
            scheme = SCHEME_ALIASES.get(scheme, scheme)
            if scheme not in SCHEMES:
                _raise(_lib.CTD_ESCHEME, f"Unknown discretization method: {scheme}")    # src/DOCP_data.jl:342-349
            sid = SCHEMES[scheme]
        else:
            sid = int(scheme)
        self.scheme = scheme
        d = _lib.ctd_desc()
        d.problem, d.scheme = pid, sid
        d.pattern_mode = PATTERN_MODES[pattern] if isinstance(pattern, str) else int(pattern)
        d.device = int(device)
        self._tg = None
        if time_grid is not None:
            self._tg = np.ascontiguousarray(time_grid, dtype=np.float64)
            d.time_grid = _dp(self._tg)
            d.time_grid_len = len(self._tg)
            d.grid_size = len(self._tg) - 1
        else:
            d.time_grid = None
            d.time_grid_len = 0
            d.grid_size = int(grid_size)
        if steps is not None:
            d.step_begin, d.step_end = int(steps[0]), int(steps[1])
        d.stream, d.stream_mode = None, 0
        d.control_steps = int(control_steps)
        d.value_order = VALUE_ORDERS[value_order] if isinstance(value_order, str) else int(value_order)
        d.reserved0 = 0
        self.value_order = {0: "csc", 1: "csr"}.get(d.value_order, str(d.value_order))
        self.control_steps = max(1, int(control_steps))
        if int(device) >= 0 and stream != "own":
            if stream == "torch":
                try:
                    import torch
                    if torch.cuda.is_available():
                        d.stream = C.c_void_p(torch.cuda.current_stream(int(device)).cuda_stream)
                        d.stream_mode = 1
                except ImportError:
                    pass
            elif stream is not None:
                d.stream, d.stream_mode = C.c_void_p(int(stream)), 1
        h = C.c_void_p()
        st = L.ctd_create(C.byref(d), C.byref(h))
        if st != _lib.CTD_OK:
            _raise(st, L.ctd_last_error(None).decode())
        self._h = h
        self.device = int(device)
        self._load_static()
        if d.pattern_mode == 0 and self.dropped_nonzeros() > 0:
            # the reference's :manual pattern leaves structural nonzeros out here (trapeze.jl:203: dynamics rows x variables;
            # euler.jl:231: implicit Euler's path rows x previous control): values at the pattern's positions are exact, but
            # a solver fed this Jacobian misses those entries
            import warnings
            warnings.warn(f"pattern='manual' reproduces the reference's DOCP_Jacobian_pattern, which omits {self.dropped_nonzeros()} "
                          f"structural nonzeros for {self.problem_name} / {self.scheme}; pass pattern='structural' (or 'optimized') "
                          "to solve with a complete Jacobian", stacklevel=2)

    # ---- static data ------------------------------------------------------------------------------------
    def _load_static(self):
        L = _lib.lib()
        nvar, ncon, nnzj, nnzh = (C.c_int64() for _ in range(4))
        self._ck(L.ctd_sizes(self._h, C.byref(nvar), C.byref(ncon), C.byref(nnzj), C.byref(nnzh)))
        self.dim_NLP_variables, self.dim_NLP_constraints = nvar.value, ncon.value
        self.nnzj, self.nnzh = nnzj.value, nnzh.value
        o = np.zeros(16, dtype=np.int64)
        self._ck(L.ctd_dims(self._h, _ip(o)))
        self.dims = SimpleNamespace(NLP_x=int(o[0]), NLP_u=int(o[1]), NLP_v=int(o[2]), path_cons=int(o[3]),
                                    boundary_cons=int(o[4]))
        self.flags = SimpleNamespace(freet0=bool(o[11]), freetf=bool(o[12]), lagrange=bool(o[13]), mayer=bool(o[14]),
                                     max=bool(o[15]))
        stage = int(o[9])
        self.discretization = SimpleNamespace(_step_variables_block=int(o[6]), _state_stage_eqs_block=int(o[7]),
                                              _step_pathcons_block=int(o[8]), stage=stage, _final_control=bool(o[10]))
        steps = int(o[5])
        nrm = np.zeros(steps + 1)
        fx = np.zeros(steps + 1)
        self._ck(L.ctd_time_grid(self._h, _dp(nrm), _dp(fx)))
        self.time = SimpleNamespace(steps=steps, control_steps=getattr(self, "control_steps", 1), normalized_grid=nrm, fixed_grid=fx)
        if stage > 0:
            a = np.zeros(stage * stage); b = np.zeros(stage); c = np.zeros(stage)
            self._ck(L.ctd_butcher(self._h, _dp(a), _dp(b), _dp(c)))
            self.discretization.butcher_a = a.reshape(stage, stage)
            self.discretization.butcher_b, self.discretization.butcher_c = b, c
        sh = np.zeros(8, dtype=np.int64)
        self._ck(L.ctd_shard_info(self._h, _ip(sh)))
        self.shard = SimpleNamespace(step_begin=int(sh[0]), step_end=int(sh[1]), c_row_begin=int(sh[2]), c_row_end=int(sh[3]),
                                     vals_main_begin=int(sh[4]), vals_main_end=int(sh[5]), owns_first=bool(sh[6]),
                                     owns_last=bool(sh[7]))
        self._bounds = None

    def _ck(self, st):
        if st != _lib.CTD_OK:
            _raise(st, _lib.lib().ctd_last_error(self._h).decode() or _lib.lib().ctd_strerror(st).decode())

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().ctd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def bounds(self):
        """docp.bounds with var_l, var_u, con_l, con_u (filled by __variables_bounds! / __constraints_bounds!)."""
        if self._bounds is None:
            lv = np.zeros(self.dim_NLP_variables); uv = np.zeros(self.dim_NLP_variables)
            lc = np.zeros(self.dim_NLP_constraints); uc = np.zeros(self.dim_NLP_constraints)
            self._ck(_lib.lib().ctd_bounds(self._h, _dp(lv), _dp(uv), _dp(lc), _dp(uc)))
            self._bounds = SimpleNamespace(var_l=lv, var_u=uv, con_l=lc, con_u=uc)
        return self._bounds

    def launch_info(self):
        o = np.zeros(8, dtype=np.int64)
        self._ck(_lib.lib().ctd_launch_info(self._h, _ip(o)))
        return dict(grid=int(o[0]), block=int(o[1]), lds_bytes=int(o[2]), steps_per_tile=int(o[3]),
                    csc_period=int(o[4]), edge_entries=int(o[5]), direct_tiles=int(o[6]), workgroups_per_cu=int(o[7]))

    def dropped_nonzeros(self):
        n = C.c_int64()
        self._ck(_lib.lib().ctd_dropped_nonzeros(self._h, C.byref(n)))
        return n.value

    # ---- NLPModels-style callbacks --------------------------------------------------------------------------
    def _check_x(self, x):
        n = x.numel() if _is_tensor(x) else x.size
        if n != self.dim_NLP_variables:
            raise ValueError(f"x has {n} entries, expected dim_NLP_variables = {self.dim_NLP_variables}")

    def _dev_ptr(self, t, n, name):
        import torch
        if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.numel() == n):
            raise ValueError(f"{name} must be a contiguous float64 tensor of {n} entries on the handle's GPU")
        if t.device.index != self.device:
            raise ValueError(f"{name} lives on cuda:{t.device.index}, handle is bound to device {self.device}")
        return C.c_void_p(t.data_ptr())

    def cons_jac(self, x, c=None, vals=None, sync=True):
        """Fused cons!(nlp, x, c) + jac_coord!(nlp, x, vals): the benchmarked call.  Returns (c, vals)."""
        L = _lib.lib()
        self._check_x(x)
        if _is_tensor(x):
            import torch
            if c is None:
                c = torch.empty(self.dim_NLP_constraints, dtype=torch.float64, device=x.device)
            if vals is None:
                vals = torch.empty(self.nnzj, dtype=torch.float64, device=x.device)
            fn = L.ctd_cons_jac_dev if sync else L.ctd_cons_jac_dev_async
            self._ck(fn(self._h, self._dev_ptr(x, self.dim_NLP_variables, "x"),
                        self._dev_ptr(c, self.dim_NLP_constraints, "c"), self._dev_ptr(vals, self.nnzj, "vals")))
            return c, vals
        x = np.ascontiguousarray(x, dtype=np.float64)
        c = np.empty(self.dim_NLP_constraints) if c is None else _host_out(c, self.dim_NLP_constraints, "c")
        vals = np.empty(self.nnzj) if vals is None else _host_out(vals, self.nnzj, "vals")
        self._ck(L.ctd_cons_jac(self._h, _dp(x), _dp(c), _dp(vals)))
        return c, vals

    def bind_cons_jac(self, x, c, vals, sync=False):
        """Pre-binds the device pointers of (x, c, vals) and returns a zero-argument callable that enqueues one fused
        evaluation -- the per-iteration path of a GPU-resident solver loop, without the Python argument checks."""
        L = _lib.lib()
        fn = L.ctd_cons_jac_dev if sync else L.ctd_cons_jac_dev_async
        h = self._h
        px = self._dev_ptr(x, self.dim_NLP_variables, "x")
        pc = self._dev_ptr(c, self.dim_NLP_constraints, "c")
        pv = self._dev_ptr(vals, self.nnzj, "vals")
        ck = self._ck

        def call():
            st = fn(h, px, pc, pv)
            if st:
                ck(st)
        return call

    def set_x_shards(self, step_begins, x_ptrs, self_index):
        """Sharded iterate read in place (`ctd_set_x_shards`): from now on the constraint / Jacobian kernels of this shard
        handle load the entries other shards own (next shard's first node, previous shard's last block, X_1, X_{N+1}) straight
        from `x_ptrs[k]` -- raw device pointers of the shards' full-length buffers (peer- or IPC-mapped).  `step_begins` has
        one entry per shard plus N.  `step_begins=None` switches back."""
        L = _lib.lib()
        if step_begins is None:
            self._ck(L.ctd_set_x_shards(self._h, 0, None, None, 0))
            return
        G = len(x_ptrs)
        sb = np.ascontiguousarray(step_begins, dtype=np.int64)
        assert sb.size == G + 1
        arr = (C.c_void_p * G)(*[int(p) if p else None for p in x_ptrs])
        self._ck(L.ctd_set_x_shards(self._h, G, _ip(sb), arr, int(self_index)))

    def stitch_c(self, comm, n_ranks, rank, c):
        """`ctd_stitch_c`: all-gather of the row blocks of the device tensor c over the ncclComm_t `comm` (an integer / c_void_p:
        the host's RCCL communicator), inside the library -- the one-process-per-GPU route that needs no torch.distributed."""
        self._ck(_lib.lib().ctd_stitch_c(self._h, C.c_void_p(int(comm)), int(n_ranks), int(rank),
                                         self._dev_ptr(c, self.dim_NLP_constraints, "c")))
        return c

    def cons(self, x, c=None):
        """cons!(nlp, x, c) = __constraints!(c, x, docp); returns c (the reference's closure must return c too)."""
        L = _lib.lib()
        self._check_x(x)
        if _is_tensor(x):
            import torch
            if c is None:
                c = torch.empty(self.dim_NLP_constraints, dtype=torch.float64, device=x.device)
            self._ck(L.ctd_cons_jac_dev(self._h, self._dev_ptr(x, self.dim_NLP_variables, "x"),
                                        self._dev_ptr(c, self.dim_NLP_constraints, "c"), None))
            return c
        x = np.ascontiguousarray(x, dtype=np.float64)
        c = np.empty(self.dim_NLP_constraints) if c is None else _host_out(c, self.dim_NLP_constraints, "c")
        self._ck(L.ctd_cons(self._h, _dp(x), _dp(c)))
        return c

    def jac_coord(self, x, vals=None):
        """jac_coord!(nlp, x, vals): Jacobian values in the CSC order of DOCP_Jacobian_pattern."""
        L = _lib.lib()
        self._check_x(x)
        if _is_tensor(x):
            import torch
            if vals is None:
                vals = torch.empty(self.nnzj, dtype=torch.float64, device=x.device)
            self._ck(L.ctd_cons_jac_dev(self._h, self._dev_ptr(x, self.dim_NLP_variables, "x"), None,
                                        self._dev_ptr(vals, self.nnzj, "vals")))
            return vals
        x = np.ascontiguousarray(x, dtype=np.float64)
        vals = np.empty(self.nnzj) if vals is None else _host_out(vals, self.nnzj, "vals")
        self._ck(L.ctd_jac_coord(self._h, _dp(x), _dp(vals)))
        return vals

    def obj(self, x):
        """obj(nlp, x) = __objective(x, docp).  For a sharded handle: the shard's partial sum."""
        L = _lib.lib()
        self._check_x(x)
        f = C.c_double()
        if _is_tensor(x):
            self._ck(L.ctd_obj_dev(self._h, self._dev_ptr(x, self.dim_NLP_variables, "x"), C.byref(f)))
        else:
            x = np.ascontiguousarray(x, dtype=np.float64)
            self._ck(L.ctd_obj(self._h, _dp(x), C.byref(f)))
        return f.value

    def obj_async(self, x, f):
        """Enqueue obj(nlp, x) with the value written to the 1-element device tensor `f` (no host copy, no wait)."""
        self._check_x(x)
        self._ck(_lib.lib().ctd_obj_dev_async(self._h, self._dev_ptr(x, self.dim_NLP_variables, "x"), self._dev_ptr(f, 1, "f")))
        return f

    def grad(self, x, g=None, sync=True):
        """grad!(nlp, x, g): gradient of __objective (ReverseDiff over the closure in the reference, src/collocation.jl:127)."""
        L = _lib.lib()
        self._check_x(x)
        if _is_tensor(x):
            import torch
            if g is None:
                g = torch.empty(self.dim_NLP_variables, dtype=torch.float64, device=x.device)
            fn = L.ctd_grad_dev if sync else L.ctd_grad_dev_async
            self._ck(fn(self._h, self._dev_ptr(x, self.dim_NLP_variables, "x"), self._dev_ptr(g, self.dim_NLP_variables, "g")))
            return g
        x = np.ascontiguousarray(x, dtype=np.float64)
        g = np.empty(self.dim_NLP_variables) if g is None else _host_out(g, self.dim_NLP_variables, "g")
        self._ck(L.ctd_grad(self._h, _dp(x), _dp(g)))
        return g

    def grad_shard(self, x, g, sync=False):
        """`ctd_grad_shard_dev_async`: the gradient entries of THIS shard's own variables into the full-length device tensor g
        (+ the shard's partial sums of d/dv in the nv tail entries), from a sharded iterate read in place -- no all-gathered x."""
        self._check_x(x)
        self._ck(_lib.lib().ctd_grad_shard_dev_async(self._h, self._dev_ptr(x, self.dim_NLP_variables, "x"),
                                                     self._dev_ptr(g, self.dim_NLP_variables, "g")))
        if sync:
            self.sync()
        return g

    def eval_all(self, x, y=None, obj_weight=1.0, f=None, g=None, c=None, vals=None, hvals=None, sync=False):
        """One solver iteration in one call (`ctd_eval_all_dev_async`): objective -> f[0], gradient -> g, constraints -> c,
        Jacobian values -> vals, Hessian values of the Lagrangian -> hvals, for device tensors; outputs left None are skipped.
        The callbacks run side by side on the GPU."""
        self._check_x(x)
        P = lambda t, n, name: None if t is None else self._dev_ptr(t, n, name)      # noqa: E731
        self._ck(_lib.lib().ctd_eval_all_dev_async(self._h, self._dev_ptr(x, self.dim_NLP_variables, "x"),
                                                   P(y, self.dim_NLP_constraints, "y"), float(obj_weight), P(f, 1, "f"),
                                                   P(g, self.dim_NLP_variables, "g"), P(c, self.dim_NLP_constraints, "c"),
                                                   P(vals, self.nnzj, "vals"), P(hvals, self.nnzh, "hvals")))
        if sync:
            self.sync()

    def sync(self):
        self._ck(_lib.lib().ctd_sync(self._h))

    def set_stream(self, stream=None):
        """Launch on `stream` from now on (a torch.cuda.Stream, a raw hipStream_t or None = torch's current stream), e.g.
        inside `torch.cuda.graph(...)` to record a whole solver iteration into one HIP graph."""
        if stream is None:
            import torch
            stream = torch.cuda.current_stream(self.device)
        raw = getattr(stream, "cuda_stream", stream)
        self._ck(_lib.lib().ctd_set_stream(self._h, C.c_void_p(int(raw))))

    def time_cons_jac(self, x, c, vals, iters=20):
        """Mean duration (ms) of one fused-kernel launch, from HIP events recorded by each dispatch on the handle's stream."""
        ms = C.c_double()
        self._ck(_lib.lib().ctd_time_cons_jac_dev(self._h, self._dev_ptr(x, self.dim_NLP_variables, "x"),
                                                  self._dev_ptr(c, self.dim_NLP_constraints, "c"),
                                                  self._dev_ptr(vals, self.nnzj, "vals"), int(iters), C.byref(ms)))
        return ms.value

    def debug_stamps(self, x, c, vals, sub=False):
        """Diagnostics: per-workgroup phase stamps of one launch, array [grid, 6, 2] (realtime 100 MHz, shader cycles)."""
        grid = self.launch_info()["grid"]
        out = np.zeros(grid * (28 if sub else 12), dtype=np.uint64)
        self._ck(_lib.lib().ctd_debug_stamps(self._h, self._dev_ptr(x, self.dim_NLP_variables, "x"),
                                             self._dev_ptr(c, self.dim_NLP_constraints, "c"),
                                             self._dev_ptr(vals, self.nnzj, "vals"),
                                             out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size))
        if sub:      # experiment builds (-DCTD_SUBSTAMPS): [grid, wave 0/1, 8] cycle stamps inside the evaluation phase
            return out[:grid * 12].reshape(grid, 6, 2), out[grid * 12:].reshape(grid, 2, 8)
        return out.reshape(grid, 6, 2)

    # ---- Hessian of the Lagrangian ------------------------------------------------------------------------
    def hess_structure(self):
        """hess_structure!(nlp, rows, cols): 1-based COO (row >= col) of the lower triangle of DOCP_Hessian_pattern."""
        rows = np.zeros(self.nnzh, dtype=np.int64)
        cols = np.zeros(self.nnzh, dtype=np.int64)
        self._ck(_lib.lib().ctd_hess_structure(self._h, _ip(rows), _ip(cols)))
        return rows, cols

    def hess_coord(self, x, y, obj_weight=1.0, vals=None, sync=True):
        """hess_coord!(nlp, x, y, vals; obj_weight): obj_weight * d2 f + sum_i y_i d2 c_i on hess_structure()."""
        L = _lib.lib()
        self._check_x(x)
        if _is_tensor(x):
            import torch
            if vals is None:
                vals = torch.empty(self.nnzh, dtype=torch.float64, device=x.device)
            fn = L.ctd_hess_coord_dev if sync else L.ctd_hess_coord_dev_async
            self._ck(fn(self._h, self._dev_ptr(x, self.dim_NLP_variables, "x"), self._dev_ptr(y, self.dim_NLP_constraints, "y"),
                        float(obj_weight), self._dev_ptr(vals, self.nnzh, "vals")))
            return vals
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        if y.size != self.dim_NLP_constraints:
            raise ValueError(f"y has {y.size} entries, expected dim_NLP_constraints = {self.dim_NLP_constraints}")
        vals = np.empty(self.nnzh) if vals is None else _host_out(vals, self.nnzh, "vals")
        self._ck(L.ctd_hess_coord(self._h, _dp(x), _dp(y), float(obj_weight), _dp(vals)))
        return vals

    def time_hess(self, x, y, vals, obj_weight=1.0, iters=20):
        """Mean duration (ms) of one Hessian-kernel launch (per-dispatch HIP events on the handle's stream)."""
        ms = C.c_double()
        self._ck(_lib.lib().ctd_time_hess_dev(self._h, self._dev_ptr(x, self.dim_NLP_variables, "x"),
                                              self._dev_ptr(y, self.dim_NLP_constraints, "y"), float(obj_weight),
                                              self._dev_ptr(vals, self.nnzh, "vals"), int(iters), C.byref(ms)))
        return ms.value

    def hess_debug_stamps(self, x, y, vals, obj_weight=1.0):
        """Diagnostics: per-workgroup phase stamps of one Hessian launch, array [grid, 5, 2] (realtime 100 MHz, cycles)."""
        grid = self.hess_launch_info()["grid"]
        out = np.zeros(grid * 10, dtype=np.uint64)
        self._ck(_lib.lib().ctd_hess_debug_stamps(self._h, self._dev_ptr(x, self.dim_NLP_variables, "x"),
                                                  self._dev_ptr(y, self.dim_NLP_constraints, "y"), float(obj_weight),
                                                  self._dev_ptr(vals, self.nnzh, "vals"),
                                                  out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size))
        return out.reshape(grid, 5, 2)

    def hess_shard_info(self):
        """(vals_main_begin, vals_main_end, positions of the V x V entries): on a sharded handle hess_coord leaves partial
        sums in the V x V entries, to be added over the shards."""
        o = np.zeros(13, dtype=np.int64)
        self._ck(_lib.lib().ctd_hess_shard_info(self._h, _ip(o)))
        return int(o[0]), int(o[1]), o[3:3 + int(o[2])].copy()

    def hess_launch_info(self):
        o = np.zeros(10, dtype=np.int64)
        self._ck(_lib.lib().ctd_hess_launch_info(self._h, _ip(o)))
        return dict(grid=int(o[0]), block=int(o[1]), lds_bytes=int(o[2]), steps_per_tile=int(o[3]), csc_period=int(o[4]),
                    edge_entries=int(o[5]), stage_lanes=int(o[6]), path_lanes=int(o[7]), boundary_lanes=int(o[8]),
                    segment_terms=int(o[9]))

    def hess_kernel_info(self):
        """which Hessian kernel takes the regular steps: dict(kernel='tile' | 'step', grid=workgroups)"""
        o = np.zeros(2, dtype=np.int64)
        self._ck(_lib.lib().ctd_hess_kernel_info(self._h, _ip(o)))
        return dict(kernel="step" if o[0] == 1 else "tile", grid=int(o[1]))

    def jac_structure(self):
        """jac_structure!(nlp, rows, cols): 1-based COO in CSC order."""
        rows = np.zeros(self.nnzj, dtype=np.int64)
        cols = np.zeros(self.nnzj, dtype=np.int64)
        self._ck(_lib.lib().ctd_jac_structure(self._h, _ip(rows), _ip(cols)))
        return rows, cols


# ---- free functions named after the reference's --------------------------------------------------------------
def constraints(c, xu, docp):
    """`CTDirect.__constraints!(c, xu, docp)`; returns c."""
    return docp.cons(xu, c)


def objective(xu, docp):
    """`CTDirect.__objective(xu, docp)`."""
    return docp.obj(xu)


def gradient(xu, docp, g=None):
    """Gradient of `CTDirect.__objective(xu, docp)` (NLPModels `grad!`)."""
    return docp.grad(xu, g)


def variables_bounds(docp):
    """`CTDirect.__variables_bounds!(docp)` -> (var_l, var_u)."""
    return docp.bounds.var_l, docp.bounds.var_u


def constraints_bounds(docp):
    """`CTDirect.__constraints_bounds!(docp)` -> (lb, ub)."""
    return docp.bounds.con_l, docp.bounds.con_u


class _Pinned:
    """owner of one ctd_host_alloc block (freed when the last array viewing it goes away)"""

    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        st = _lib.lib().ctd_host_alloc(C.byref(self.ptr), nbytes)
        if st:
            _raise(st, _lib.lib().ctd_last_error(None).decode())

    def __del__(self):
        try:
            _lib.lib().ctd_host_free(self.ptr)
        except Exception:
            pass


def pinned_empty(n, dtype=np.float64):
    """NumPy array in page-locked host memory (`ctd_host_alloc`): passing such arrays to the host-pointer calls
    (`docp.cons_jac(x, c, vals)` with NumPy arguments, ...) lets the copies run as direct DMA at PCIe rate."""
    dtype = np.dtype(dtype)
    own = _Pinned(int(n) * dtype.itemsize)
    buf = (C.c_char * (int(n) * dtype.itemsize)).from_address(own.ptr.value)
    a = np.frombuffer(buf, dtype=dtype, count=int(n))
    _pinned_owners[a.__array_interface__["data"][0]] = own      # keeps the block alive as long as the module does ...
    import weakref
    weakref.finalize(a, _pinned_owners.pop, a.__array_interface__["data"][0], None)   # ... or until the array is collected
    return a


_pinned_owners = {}


def initial_guess(docp, init=None):
    """`CTDirect.__initial_guess(docp, init)`.  init: None (everything 0.1), "problem" (the problem file's init
    tuple) or a dict with optional constant `state`, `control`, `variable` entries; with a `time` entry (K increasing
    times) `state` [K, n] and `control` [K, m] are trajectories -- the interpolated / warm-start guesses of the reference
    (test/ci/test_initial_guess.jl), e.g. the T, X, U of `unpack_solution` -- interpolated linearly at the node (and stage)
    times."""
    x0 = np.zeros(docp.dim_NLP_variables)
    ci = _lib.ctd_init()
    keep = []
    if init == "problem":
        ci.use_problem_default = 1
    elif isinstance(init, dict) and init.get("time") is not None:
        t = np.ascontiguousarray(init["time"], dtype=np.float64)
        keep.append(t)
        ci.n_samples, ci.t_samples = len(t), _dp(t)
        for key, dim in (("state", docp.dims.NLP_x), ("control", docp.dims.NLP_u)):
            if init.get(key) is not None and dim > 0:
                a = np.ascontiguousarray(init[key], dtype=np.float64).reshape(len(t), dim)
                keep.append(a)
                setattr(ci, key + "_samples", _dp(a))
        if init.get("variable") is not None:
            a = np.ascontiguousarray(init["variable"], dtype=np.float64)
            keep.append(a)
            ci.variable = _dp(a)
    elif isinstance(init, dict):
        for key in ("state", "control", "variable"):
            if init.get(key) is not None:
                a = np.ascontiguousarray(init[key], dtype=np.float64)
                keep.append(a)
                setattr(ci, key, _dp(a))
    docp._ck(_lib.lib().ctd_initial_guess(docp._h, _dp(x0), C.byref(ci)))
    return x0


def DOCP_Jacobian_pattern(docp):
    """`CTDirect.DOCP_Jacobian_pattern(docp)` as 0-based CSC arrays (colptr, rowval) of the Bool sparse matrix."""
    colptr = np.zeros(docp.dim_NLP_variables + 1, dtype=np.int64)
    rowval = np.zeros(docp.nnzj, dtype=np.int64)
    docp._ck(_lib.lib().ctd_jac_csc(docp._h, _ip(colptr), _ip(rowval)))
    return colptr, rowval


def DOCP_Jacobian_csr(docp):
    """The same pattern as 0-based CSR arrays (rowptr, colind): the order of the values of a DOCP built with value_order="csr"."""
    rowptr = np.zeros(docp.dim_NLP_constraints + 1, dtype=np.int64)
    colind = np.zeros(docp.nnzj, dtype=np.int64)
    docp._ck(_lib.lib().ctd_jac_csr(docp._h, _ip(rowptr), _ip(colind)))
    return rowptr, colind


def DOCP_Hessian_csr(docp):
    """(rowptr, colind) of the UPPER triangle by rows -- the CSR reading of the value array hess_coord fills (the lower triangle by
    columns of a symmetric matrix is its upper triangle by rows)."""
    rowptr = np.zeros(docp.dim_NLP_variables + 1, dtype=np.int64)
    colind = np.zeros(docp.nnzh, dtype=np.int64)
    docp._ck(_lib.lib().ctd_hess_csr(docp._h, _ip(rowptr), _ip(colind)))
    return rowptr, colind


def DOCP_Hessian_pattern(docp):
    """Lower triangle of `CTDirect.DOCP_Hessian_pattern(docp)` as 0-based CSC arrays (colptr, rowval) -- the part of the
    symmetric Bool pattern ADNLPModels keeps for hess_structure!."""
    colptr = np.zeros(docp.dim_NLP_variables + 1, dtype=np.int64)
    rowval = np.zeros(docp.nnzh, dtype=np.int64)
    docp._ck(_lib.lib().ctd_hess_csc(docp._h, _ip(colptr), _ip(rowval)))
    return colptr, rowval


def get_time_grid(xu, docp):
    """`CTDirect.get_time_grid(xu, docp)` (src/DOCP_data.jl:437-458): t_i = t0 + tau_i (tf - t0), t0 / tf fixed or entries of v."""
    xu = np.ascontiguousarray(xu.detach().cpu().numpy() if _is_tensor(xu) else xu, dtype=np.float64)
    grid = np.zeros(docp.time.steps + 1)
    docp._ck(_lib.lib().ctd_time_grid_at(docp._h, _dp(xu), _dp(grid)))
    return grid


def _state_control_variable(docp, data):
    """X[N+1, n], U[N+1, m], v read from any vector with the NLP variable layout (the primal iterate or a vector of bound
    multipliers) with the reference's getters (`getter(...; val=:state | :control | :variable)`, src/ode/common.jl:50-84)"""
    n, m, nv = docp.dims.NLP_x, docp.dims.NLP_u, docp.dims.NLP_v
    N, blk = docp.time.steps, docp.discretization._step_variables_block
    X = np.stack([data[i * blk:i * blk + n] for i in range(N + 1)])
    stage = docp.discretization.stage
    stagewise = docp.scheme in ("gauss_legendre_2", "gauss_legendre_3")
    cs = getattr(docp.time, "control_steps", 1)
    if cs > 1:
        # getter(...; val = :control) with several controls per step (src/ode/common.jl:84-98): N control_steps + 1 rows, the
        # controls of every step in order, then get_OCP_control_at_time_step(N + 1) = the first control of the last step
        U = np.zeros((N * cs + 1, m))
        for i in range(N):
            for j in range(cs):
                U[i * cs + j] = data[i * blk + n + j * m:i * blk + n + (j + 1) * m]
        U[N * cs] = data[(N - 1) * blk + n:(N - 1) * blk + n + m]
        return X, U, data[len(data) - nv:].copy()
    U = np.zeros((N + 1, m))
    if m:
        b = docp.discretization.butcher_b if stagewise else None
        for i in range(N + 1):
            j = i if (i < N or docp.discretization._final_control) else N - 1     # u(t_f) = U_N convention
            if docp.scheme == "euler_implicit":
                j = max(i - 1, 0) if i > 0 else 0                                # u(t_i) = U_{i-1}, u(t_0) = U_0
            o = j * blk + n
            U[i] = sum(b[s] * data[o + s * m:o + (s + 1) * m] for s in range(stage)) if stagewise else data[o:o + m]
    return X, U, data[len(data) - nv:].copy()


def unpack_solution(docp, x, multipliers=None, multipliers_L=None, multipliers_U=None):
    """The arrays `CTDirect.build_OCP_solution` hands to CTModels (src/DOCP_data.jl:514-633) from an NLP solution, with the
    reference's getter conventions (src/ode/common.jl:7-104): X[N+1, n], U[N+1, m] (control of the scheme at every node,
    final control duplicated when the scheme has none), v, and from the constraint multipliers the costate P[N, n] (the
    multipliers of the state-equation rows), the path-constraint duals divided by the step length and the boundary duals;
    from the bound multipliers (`multipliers_L`, `multipliers_U`, nvar entries each) the state / control / variable box
    duals, read with the same getters as the primal arrays."""
    x = np.ascontiguousarray(x.detach().cpu().numpy() if _is_tensor(x) else x, dtype=np.float64)
    N = docp.time.steps
    T = get_time_grid(x, docp)
    X, U, v = _state_control_variable(docp, x)
    out = dict(T=T, X=X, U=U, v=v)
    cs = getattr(docp.time, "control_steps", 1)
    # time grid of the control rows (src/DOCP_data.jl:557-567): T_control[k] = T[i] + (j - 1) h_i / control_steps, last = T[end]
    out["T_control"] = np.concatenate([(T[:-1, None] + np.arange(cs)[None, :] * (np.diff(T)[:, None] / cs)).ravel(), T[-1:]]) if cs > 1 else T
    for z, tag in ((multipliers_L, "lb"), (multipliers_U, "ub")):
        zz = np.zeros_like(x) if z is None else np.ascontiguousarray(z, dtype=np.float64)
        ZX, ZU, Zv = _state_control_variable(docp, zz)
        out["state_constraints_%s_dual" % tag], out["control_constraints_%s_dual" % tag] = ZX, ZU
        out["variable_constraints_%s_dual" % tag] = Zv
    if multipliers is not None:
        y = np.ascontiguousarray(multipliers, dtype=np.float64)
        n = docp.dims.NLP_x
        eqs, p, bc = docp.discretization._state_stage_eqs_block, docp.dims.path_cons, docp.dims.boundary_cons
        cb = eqs + p
        out["P"] = np.stack([y[i * cb:i * cb + n] for i in range(N)])
        h = np.diff(T)
        raw = np.stack([y[i * cb + eqs:i * cb + eqs + p] for i in range(N)] + [y[N * cb:N * cb + p]]) if p else np.zeros((N + 1, 0))
        out["path_constraints_dual"] = raw / np.concatenate([h, h[-1:]])[:, None]
        out["boundary_constraints_dual"] = y[N * cb + p:N * cb + p + bc].copy()
    return out


class MultiDeviceDOCP:
    """One transcription sharded by time step over several GPUs of THIS process (`ctd_create_sharded`): the single-process
    counterpart of `dist.ShardedDOCP`.  Buffers are full-length on every device (global indexing); shard k writes its rows
    of c and its CSC ranges of the Jacobian values.  `devices` may repeat an ordinal (tests on a one-GPU box)."""

    # include/ctdirect_hip.h: X_SHARDED copies the halo entries (peer copies ordered by events, any topology); X_SHARDED_IN_PLACE lets
    # the kernels load them from the owner's buffer (needs peer access; falls back to the copies when a pair of devices has none)
    X_IN_PLACE, X_SHARDED, X_FROM_DEVICE0, X_SHARDED_COPY, X_SHARDED_IN_PLACE = 0, 1, 2, 3, 4

    def __init__(self, ocp, grid_size, scheme, devices, time_grid=None, pattern="manual", stream="torch", value_order="csc"):
        L = _lib.lib()
        pid = PROBLEMS[ocp] if isinstance(ocp, str) else int(ocp)
        scheme = SCHEME_ALIASES.get(scheme, scheme) if isinstance(scheme, str) else scheme
        d = _lib.ctd_desc()
        d.problem, d.scheme = pid, SCHEMES[scheme] if isinstance(scheme, str) else int(scheme)
        d.pattern_mode = PATTERN_MODES[pattern] if isinstance(pattern, str) else int(pattern)
        d.device = -1
        d.value_order = VALUE_ORDERS[value_order] if isinstance(value_order, str) else int(value_order)
        self._tg = None
        if time_grid is not None:
            self._tg = np.ascontiguousarray(time_grid, dtype=np.float64)
            d.time_grid, d.time_grid_len, d.grid_size = _dp(self._tg), len(self._tg), len(self._tg) - 1
        else:
            d.time_grid, d.time_grid_len, d.grid_size = None, 0, int(grid_size)
        self.devices = [int(k) for k in devices]
        arr = (C.c_int32 * len(self.devices))(*self.devices)
        h = C.c_void_p()
        st = L.ctd_create_sharded(C.byref(d), arr, len(self.devices), C.byref(h))
        if st != _lib.CTD_OK:
            _raise(st, L.ctd_sharded_last_error(None).decode())
        self._s = h
        self.shards = []
        for k in range(len(self.devices)):
            o = np.zeros(10, dtype=np.int64)
            self._ck(L.ctd_sharded_shard_info(self._s, k, _ip(o)))
            self.shards.append(SimpleNamespace(device=int(o[1]), step_begin=int(o[2]), step_end=int(o[3]), c_row_begin=int(o[4]),
                                               c_row_end=int(o[5]), vals_main_begin=int(o[6]), vals_main_end=int(o[7])))
        nvar, ncon, nnzj, nnzh = (C.c_int64() for _ in range(4))
        h0 = C.c_void_p()
        self._ck(L.ctd_sharded_handle(self._s, 0, C.byref(h0)))
        L.ctd_sizes(h0, C.byref(nvar), C.byref(ncon), C.byref(nnzj), C.byref(nnzh))
        self.dim_NLP_variables, self.dim_NLP_constraints, self.nnzj, self.nnzh = nvar.value, ncon.value, nnzj.value, nnzh.value
        # stream="torch" (default, like DOCP): every shard launches on torch's current stream of its device, so the engine's
        # copies and kernels are ordered with the caller's tensor work on x / c / vals (the private non-blocking streams of
        # stream="own" are not: the caller then synchronises itself)
        self._streams = None
        if stream == "torch":
            self.bind_torch_streams()

    def bind_torch_streams(self):
        import torch
        L = _lib.lib()
        cur = [torch.cuda.current_stream(dev).cuda_stream for dev in self.devices]
        if cur != self._streams:
            for k, st in enumerate(cur):
                hk = C.c_void_p()
                self._ck(L.ctd_sharded_handle(self._s, k, C.byref(hk)))
                self._ck(L.ctd_set_stream(hk, C.c_void_p(st)))
            self._streams = cur

    def _ck(self, st):
        if st != _lib.CTD_OK:
            _raise(st, _lib.lib().ctd_sharded_last_error(self._s).decode() or _lib.lib().ctd_strerror(st).decode())

    def _ptrs(self, tensors, n, name):
        import torch
        if tensors is None:
            return None
        assert len(tensors) == len(self.devices)
        for k, t in enumerate(tensors):
            if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.numel() == n and t.device.index == self.devices[k]):
                raise ValueError(f"{name}[{k}] must be a contiguous float64 tensor of {n} entries on cuda:{self.devices[k]}")
        return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])

    def cons_jac(self, x, c, vals, x_mode=0, stitch=False, sync=True):
        """x, c, vals: lists of one full-length tensor per shard (on that shard's device)."""
        if self._streams is not None:
            self.bind_torch_streams()          # (a cheap compare; rebinds when the caller switched torch streams)
        self._ck(_lib.lib().ctd_cons_jac_sharded_dev_async(self._s, self._ptrs(x, self.dim_NLP_variables, "x"),
                                                           self._ptrs(c, self.dim_NLP_constraints, "c"),
                                                           self._ptrs(vals, self.nnzj, "vals"), int(x_mode), int(bool(stitch))))
        if sync:
            self.sync()

    def sync(self):
        self._ck(_lib.lib().ctd_sharded_sync(self._s))

    def last_error(self):
        """`ctd_sharded_last_error`: the message of the last failed call -- or of the last fallback (X_SHARDED_IN_PLACE on devices
        without peer access takes the copying protocol and says so here)."""
        return _lib.lib().ctd_sharded_last_error(self._s).decode()

    def close(self):
        if getattr(self, "_s", None):
            _lib.lib().ctd_sharded_destroy(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
