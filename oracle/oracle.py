"""TEST INFRASTRUCTURE ONLY -- ctypes loader for the CPU oracle (oracle/ctd_oracle.cpp).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (ctdirect.jl_amd/) must never do so.  The oracle restates the reference's Julia
hot path (src/DOCP_functions.jl, src/ode/*.jl, src/DOCP_data.jl, src/DOCP_variables.jl) on the CPU;
see the header of ctd_oracle.cpp for the file:line map and the pinning status.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libctd_oracle.so")

# scheme ids, restated (reference scheme symbols: src/DOCP_data.jl:307-349)
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
    # oracle-only restatements (the engine takes these as run-time OCPs, tests/problem_folder_defs.py)
    "goddard_all_f0f1": 10,           # goddard_all with xdot = F0 + u F1 (test/problems/goddard.jl:7-15,44): the archived 28011 pattern
    "algal_bacterial": 11,            # test/problems/algal_bacterial.jl
}


def build(force=False):
    """Compile the oracle with g++ (a few seconds)."""
    src = [os.path.join(_HERE, f) for f in ("ctd_oracle.cpp", "dual.hpp", "dual2.hpp", "problems.hpp")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src if os.path.exists(s))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int64)
        L.orc_create.argtypes = [C.c_int, C.c_int, C.c_int64, dp, C.c_int64, C.POINTER(C.c_void_p)]
        L.orc_create.restype = C.c_int
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_last_error.restype = C.c_char_p
        L.orc_dims.argtypes = [C.c_void_p, ip]
        L.orc_flags.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.orc_butcher.argtypes = [C.c_void_p, dp, dp, dp]
        L.orc_grids.argtypes = [C.c_void_p, dp, dp]
        L.orc_bounds.argtypes = [C.c_void_p, dp, dp, dp, dp]
        L.orc_initial_guess.argtypes = [C.c_void_p, C.c_int, dp]
        L.orc_initial_guess_sampled.argtypes = [C.c_void_p, C.c_int64, dp, dp, dp, dp, dp]
        L.orc_constraints.argtypes = [C.c_void_p, dp, dp]
        L.orc_objective.argtypes = [C.c_void_p, dp]
        L.orc_objective.restype = C.c_double
        L.orc_gradient.argtypes = [C.c_void_p, dp, dp]
        L.orc_set_pattern_mode.argtypes = [C.c_void_p, C.c_int]
        L.orc_jac_nnz.argtypes = [C.c_void_p]
        L.orc_jac_nnz.restype = C.c_int64
        L.orc_jac_ncolors.argtypes = [C.c_void_p]
        L.orc_jac_ncolors.restype = C.c_int
        L.orc_jac_pattern.argtypes = [C.c_void_p, ip, ip]
        L.orc_jac_coord.argtypes = [C.c_void_p, dp, dp]
        L.orc_jac_coord_mt.argtypes = [C.c_void_p, dp, dp, C.c_int]
        L.orc_cons_jac_block.argtypes = [C.c_void_p, dp, dp, dp, C.c_int]
        L.orc_cons_jac_block.restype = C.c_int
        L.orc_hess_coord_block.argtypes = [C.c_void_p, dp, dp, C.c_double, dp, ip, C.c_int]
        L.orc_hess_coord_block.restype = C.c_int
        L.orc_jac_column.argtypes = [C.c_void_p, dp, C.c_int64, dp]
        L.orc_hess_nnz.argtypes = [C.c_void_p, ip, ip]
        L.orc_hess_lower_nnz.argtypes = [C.c_void_p]
        L.orc_hess_lower_nnz.restype = C.c_int64
        L.orc_hess_pattern.argtypes = [C.c_void_p, ip, ip]
        L.orc_hess_coord.argtypes = [C.c_void_p, dp, dp, C.c_double, dp, ip]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


class OracleDOCP:
    """CPU restatement of `CTDirect.DOCP(ocp, grid_size, 1, scheme, time_grid)` (src/DOCP_data.jl:293-365)."""

    def __init__(self, problem, scheme, grid_size=None, time_grid=None, control_steps=1):
        L = lib()
        pid = PROBLEMS[problem] if isinstance(problem, str) else int(problem)
        sid = SCHEMES[scheme] if isinstance(scheme, str) else int(scheme)
        h = C.c_void_p()
        self.control_steps = int(control_steps)          # DOCP(ocp, grid_size, control_steps, scheme, time_grid), src/DOCP_data.jl:293
        if time_grid is not None:
            tg = np.ascontiguousarray(time_grid, dtype=np.float64)
            st = L.orc_create_cs(pid, sid, len(tg) - 1, _dp(tg), len(tg), self.control_steps, C.byref(h))
        else:
            st = L.orc_create_cs(pid, sid, int(grid_size), None, 0, self.control_steps, C.byref(h))
        if st == 2:
            raise ValueError(L.orc_last_error().decode())       # Julia: ArgumentError (src/DOCP_data.jl:187)
        if st != 0:
            raise RuntimeError(L.orc_last_error().decode())     # Julia: error(...)   (src/DOCP_data.jl:343)
        self._h = h
        d = np.zeros(12, dtype=np.int64)
        L.orc_dims(h, _ip(d))
        (self.n, self.m, self.nv, self.path_cons, self.boundary_cons, self.steps, self.dim_NLP_variables,
         self.dim_NLP_constraints, self.step_variables_block, self.state_stage_eqs_block, self.stage,
         fc) = (int(v) for v in d)
        self.final_control = bool(fc)
        f = np.zeros(5, dtype=np.int32)
        L.orc_flags(h, f.ctypes.data_as(C.POINTER(C.c_int32)))
        self.freet0, self.freetf, self.lagrange, self.mayer, self.max = (bool(v) for v in f)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().orc_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- data
    def butcher(self):
        a = np.zeros(9); b = np.zeros(3); c = np.zeros(3)
        lib().orc_butcher(self._h, _dp(a), _dp(b), _dp(c))
        s = self.stage
        return a.reshape(3, 3)[:s, :s].copy(), b[:s].copy(), c[:s].copy()

    def grids(self):
        nrm = np.zeros(self.steps + 1); fx = np.zeros(self.steps + 1)
        lib().orc_grids(self._h, _dp(nrm), _dp(fx))
        return nrm, fx

    def bounds(self):
        lv = np.zeros(self.dim_NLP_variables); uv = np.zeros(self.dim_NLP_variables)
        lc = np.zeros(self.dim_NLP_constraints); uc = np.zeros(self.dim_NLP_constraints)
        lib().orc_bounds(self._h, _dp(lv), _dp(uv), _dp(lc), _dp(uc))
        return lv, uv, lc, uc

    def initial_guess_sampled(self, T, X=None, U=None, v=None):
        """__initial_guess with init.state / init.control given as trajectories sampled at the times T (linear
        interpolation, end values held): X [K, n], U [K, m], v [nv], each optional."""
        T = np.ascontiguousarray(T, dtype=np.float64)
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (X, U, v)]
        x0 = np.zeros(self.dim_NLP_variables)
        lib().orc_initial_guess_sampled(self._h, len(T), _dp(T), *[None if a is None else _dp(a) for a in arrs], _dp(x0))
        return x0

    def initial_guess(self, use_problem_init=True):
        x0 = np.zeros(self.dim_NLP_variables)
        lib().orc_initial_guess(self._h, int(use_problem_init), _dp(x0))
        return x0

    # -- callbacks
    def constraints(self, xu):
        xu = np.ascontiguousarray(xu, dtype=np.float64)
        assert xu.size == self.dim_NLP_variables
        c = np.full(self.dim_NLP_constraints, 666.666)   # sentinel idea from test/benchmark.jl:106,126
        lib().orc_constraints(self._h, _dp(xu), _dp(c))
        return c

    def objective(self, xu):
        xu = np.ascontiguousarray(xu, dtype=np.float64)
        assert xu.size == self.dim_NLP_variables
        return float(lib().orc_objective(self._h, _dp(xu)))

    def gradient(self, xu):
        xu = np.ascontiguousarray(xu, dtype=np.float64)
        g = np.zeros(self.dim_NLP_variables)
        lib().orc_gradient(self._h, _dp(xu), _dp(g))
        return g

    # -- Jacobian
    def set_pattern_mode(self, mode):
        """0 = REFERENCE_MANUAL (DOCP_Jacobian_pattern as written), 1 = STRUCTURAL (adds trapeze dyn x v, hazard H1)."""
        lib().orc_set_pattern_mode(self._h, int(mode))

    def jac_nnz(self):
        return int(lib().orc_jac_nnz(self._h))

    def jac_ncolors(self):
        return int(lib().orc_jac_ncolors(self._h))

    def jac_pattern(self):
        """0-based CSC (colptr[nvar+1], rowval[nnz]) of DOCP_Jacobian_pattern(docp)."""
        nnz = self.jac_nnz()
        colptr = np.zeros(self.dim_NLP_variables + 1, dtype=np.int64)
        rowval = np.zeros(nnz, dtype=np.int64)
        lib().orc_jac_pattern(self._h, _ip(colptr), _ip(rowval))
        return colptr, rowval

    def jac_coord(self, xu):
        """Values in the pattern's CSC order via coloured one-partial dual passes (ADNLPModels-style)."""
        xu = np.ascontiguousarray(xu, dtype=np.float64)
        vals = np.zeros(self.jac_nnz())
        lib().orc_jac_coord(self._h, _dp(xu), _dp(vals))
        return vals

    def jac_coord_mt(self, xu, nthreads):
        """jac_coord with the colours spread over `nthreads` host threads (same values; multi-core baseline)."""
        xu = np.ascontiguousarray(xu, dtype=np.float64)
        vals = np.zeros(self.jac_nnz())
        lib().orc_jac_coord_mt(self._h, _dp(xu), _dp(vals), int(nthreads))
        return vals

    def cons_jac_block(self, xu, nthreads=1):
        """Block mode (oracle/ctd_oracle.cpp cons_jac_block): fused (c, Jacobian values on the pattern), one time step at a
        time on dense local duals, OpenMP over the steps.  Not the reference's algorithm: the best-effort CPU figure of
        bench.py and a fast independent check of every Jacobian entry at full size.  Returns None for schemes without it."""
        xu = np.ascontiguousarray(xu, dtype=np.float64)
        if not hasattr(self, "_blk_c"):
            self._blk_c = np.zeros(self.dim_NLP_constraints)
            self._blk_v = np.zeros(self.jac_nnz())
        ok = lib().orc_cons_jac_block(self._h, _dp(xu), _dp(self._blk_c), _dp(self._blk_v), int(nthreads))
        return (self._blk_c, self._blk_v) if ok else None

    def jac_dense(self, xu):
        """Dense Jacobian, one dual pass per column (small problems only)."""
        xu = np.ascontiguousarray(xu, dtype=np.float64)
        J = np.zeros((self.dim_NLP_constraints, self.dim_NLP_variables))
        col = np.zeros(self.dim_NLP_constraints)
        for j in range(self.dim_NLP_variables):
            lib().orc_jac_column(self._h, _dp(xu), j, _dp(col))
            J[:, j] = col
        return J

    def hess_nnz(self):
        full = np.zeros(1, dtype=np.int64); lower = np.zeros(1, dtype=np.int64)
        lib().orc_hess_nnz(self._h, _ip(full), _ip(lower))
        return int(full[0]), int(lower[0])

    def hess_pattern(self):
        """0-based CSC (colptr, rowval) of the lower triangle of DOCP_Hessian_pattern(docp)."""
        nnz = int(lib().orc_hess_lower_nnz(self._h))
        colptr = np.zeros(self.dim_NLP_variables + 1, dtype=np.int64)
        rowval = np.zeros(nnz, dtype=np.int64)
        lib().orc_hess_pattern(self._h, _ip(colptr), _ip(rowval))
        return colptr, rowval

    def hess_coord_block(self, xu, y, obj_weight=1.0, nthreads=1, return_dropped=False):
        """Block mode of hess_coord (oracle/ctd_oracle.cpp hessian_block): the same second-order sweep one time step at a time,
        OpenMP over the steps -- affordable at the full BASELINE sizes.  None for schemes without it (implicit Euler)."""
        xu = np.ascontiguousarray(xu, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        assert xu.size == self.dim_NLP_variables and y.size == self.dim_NLP_constraints
        vals = np.zeros(int(lib().orc_hess_lower_nnz(self._h)))
        dropped = np.zeros(2, dtype=np.int64)
        if not lib().orc_hess_coord_block(self._h, _dp(xu), _dp(y), float(obj_weight), _dp(vals), _ip(dropped), int(nthreads)):
            return None
        return (vals, (int(dropped[0]), int(dropped[1]))) if return_dropped else vals

    def hess_coord(self, xu, y, obj_weight=1.0, return_dropped=False):
        """hess_coord!(nlp, x, y, vals; obj_weight): sparse second-order forward sweep over objective and constraints."""
        xu = np.ascontiguousarray(xu, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        assert xu.size == self.dim_NLP_variables and y.size == self.dim_NLP_constraints
        vals = np.zeros(int(lib().orc_hess_lower_nnz(self._h)))
        dropped = np.zeros(2, dtype=np.int64)
        lib().orc_hess_coord(self._h, _dp(xu), _dp(y), float(obj_weight), _dp(vals), _ip(dropped))
        return (vals, (int(dropped[0]), int(dropped[1]))) if return_dropped else vals
