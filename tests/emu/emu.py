"""TEST INFRASTRUCTURE ONLY -- ctypes loader of the serial kernel-logic emulator (tests/emu/ctd_emu.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libctd_emu.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        # (one builder at a time: pytest-xdist workers would otherwise run make -- and load a half-written library -- concurrently)
        import fcntl
        with open(os.path.join(_HERE, ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        # CTD_EMU_LIB: an instrumented build of the same source (e.g. -fsanitize=address,undefined; run pytest with the
        # sanitizer runtime preloaded)
        L = C.CDLL(os.environ.get("CTD_EMU_LIB") or _LIB)
        L.emu_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def _tg(time_grid):
    if time_grid is None:
        return None, 0
    a = np.ascontiguousarray(time_grid, dtype=np.float64)
    return a, len(a)


def sizes(problem, scheme, mode, N, time_grid=None):
    tg, n = _tg(time_grid)
    out = np.zeros(4, dtype=np.int64)
    st = lib().emu_sizes(problem, scheme, mode, C.c_int64(N or 0), tg.ctypes.data_as(C.c_void_p) if tg is not None else None,
                         C.c_int64(n), out.ctypes.data_as(C.c_void_p))
    if st:
        raise RuntimeError(f"emu status {st}: {lib().emu_last_error().decode()}")
    return [int(v) for v in out]


def csc(problem, scheme, mode, N, time_grid=None):
    nvar, ncon, nnz, _ = sizes(problem, scheme, mode, N, time_grid)
    tg, n = _tg(time_grid)
    colptr = np.zeros(nvar + 1, dtype=np.int64)
    rowval = np.zeros(nnz, dtype=np.int64)
    st = lib().emu_csc(problem, scheme, mode, C.c_int64(N or 0), tg.ctypes.data_as(C.c_void_p) if tg is not None else None,
                       C.c_int64(n), colptr.ctypes.data_as(C.c_void_p), rowval.ctypes.data_as(C.c_void_p))
    if st:
        raise RuntimeError(f"emu status {st}: {lib().emu_last_error().decode()}")
    return colptr, rowval


def csr(problem, scheme, mode, N, time_grid=None):
    """(rowptr, colind, info) of the pattern by rows; info = dict(reg_first, reg_last, Lseg, vr, HL, HH) of the model built with
    the current value order (`with emu.value_order(1)`)"""
    nvar, ncon, nnz, _ = sizes(problem, scheme, mode, N, time_grid)
    tg, n = _tg(time_grid)
    rowptr = np.zeros(ncon + 1, dtype=np.int64)
    colind = np.zeros(nnz, dtype=np.int64)
    info = np.zeros(6, dtype=np.int64)
    st = lib().emu_csr(problem, scheme, mode, C.c_int64(N or 0), tg.ctypes.data_as(C.c_void_p) if tg is not None else None,
                       C.c_int64(n), rowptr.ctypes.data_as(C.c_void_p), colind.ctypes.data_as(C.c_void_p), info.ctypes.data_as(C.c_void_p))
    if st:
        raise RuntimeError(f"emu status {st}: {lib().emu_last_error().decode()}")
    return rowptr, colind, dict(zip(("reg_first", "reg_last", "Lseg", "vr", "HL", "HH"), (int(v) for v in info)))


def shard_range(problem, scheme, mode, N, step_begin, step_end, time_grid=None):
    tg, n = _tg(time_grid)
    out = np.zeros(2, dtype=np.int64)
    st = lib().emu_shard_range(problem, scheme, mode, C.c_int64(N or 0), tg.ctypes.data_as(C.c_void_p) if tg is not None else None,
                               C.c_int64(n), C.c_int64(step_begin), C.c_int64(step_end), out.ctypes.data_as(C.c_void_p))
    if st:
        raise RuntimeError(f"emu status {st}: {lib().emu_last_error().decode()}")
    return int(out[0]), int(out[1])


def cons_jac(problem, scheme, mode, N, x, time_grid=None, tile=0, nthr=64, step_begin=0, step_end=0, c=None, vals=None):
    nvar, ncon, nnz, _ = sizes(problem, scheme, mode, N, time_grid)
    tg, n = _tg(time_grid)
    x = np.ascontiguousarray(x, dtype=np.float64)
    assert x.size == nvar
    if c is None:
        c = np.full(ncon, 666.666)
    if vals is None:
        vals = np.full(nnz, 666.666)
    st = lib().emu_cons_jac(problem, scheme, mode, C.c_int64(N or 0), tg.ctypes.data_as(C.c_void_p) if tg is not None else None,
                            C.c_int64(n), tile, nthr, C.c_int64(step_begin), C.c_int64(step_end),
                            x.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p))
    if st:
        raise RuntimeError(f"emu status {st}: {lib().emu_last_error().decode()}")
    return c, vals


def cons_jac_sharded(problem, scheme, mode, N, x, G, time_grid=None, tile=0, nthr=64):
    """All G shards of the engine's balanced split, each evaluating from a buffer that holds only its own variables (NaN
    elsewhere) with the other shards' entries read in place through the XHalo table (ctd_set_x_shards)."""
    nvar, ncon, nnz, _ = sizes(problem, scheme, mode, N, time_grid)
    tg, n = _tg(time_grid)
    x = np.ascontiguousarray(x, dtype=np.float64)
    assert x.size == nvar
    c = np.full(ncon, 666.666)
    vals = np.full(nnz, 666.666)
    st = lib().emu_cons_jac_sharded(problem, scheme, mode, C.c_int64(N or 0), tg.ctypes.data_as(C.c_void_p) if tg is not None else None,
                                    C.c_int64(n), tile, nthr, G, x.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p),
                                    vals.ctypes.data_as(C.c_void_p))
    if st:
        raise RuntimeError(f"emu status {st}: {lib().emu_last_error().decode()}")
    return c, vals


def hess_csc(problem, scheme, mode, N, time_grid=None):
    nvar = sizes(problem, scheme, mode, N, time_grid)[0]
    tg, n = _tg(time_grid)
    tgp = tg.ctypes.data_as(C.c_void_p) if tg is not None else None
    L = lib()
    L.emu_hess_nnz.restype = C.c_int64
    nnz = int(L.emu_hess_nnz(problem, scheme, mode, C.c_int64(N or 0), tgp, C.c_int64(n)))
    if nnz < 0:
        raise RuntimeError(f"emu: {L.emu_last_error().decode()}")
    colptr = np.zeros(nvar + 1, dtype=np.int64)
    rowval = np.zeros(nnz, dtype=np.int64)
    st = L.emu_hess_csc(problem, scheme, mode, C.c_int64(N or 0), tgp, C.c_int64(n), colptr.ctypes.data_as(C.c_void_p),
                        rowval.ctypes.data_as(C.c_void_p))
    if st:
        raise RuntimeError(f"emu status {st}: {L.emu_last_error().decode()}")
    return colptr, rowval


def hess(problem, scheme, mode, N, x, y, obj_weight=1.0, time_grid=None, tile=0, nthr=64, step_begin=0, step_end=0, vals=None):
    nvar, ncon, _, _ = sizes(problem, scheme, mode, N, time_grid)
    tg, n = _tg(time_grid)
    tgp = tg.ctypes.data_as(C.c_void_p) if tg is not None else None
    L = lib()
    L.emu_hess_nnz.restype = C.c_int64
    nnz = int(L.emu_hess_nnz(problem, scheme, mode, C.c_int64(N or 0), tgp, C.c_int64(n)))
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    assert x.size == nvar and y.size == ncon
    if vals is None:
        vals = np.full(nnz, 666.666)
    st = L.emu_hess(problem, scheme, mode, C.c_int64(N or 0), tgp, C.c_int64(n), tile, nthr, x.ctypes.data_as(C.c_void_p),
                    y.ctypes.data_as(C.c_void_p), C.c_double(obj_weight), vals.ctypes.data_as(C.c_void_p),
                    C.c_int64(step_begin), C.c_int64(step_end))
    if st:
        raise RuntimeError(f"emu status {st}: {L.emu_last_error().decode()}")
    return vals


def sym_check(expr, n, m, nv, point):
    """(largest difference between the symbolic second derivatives of `expr` and central differences of its symbolic first
    derivatives at `point` = (t, x, u, v), number of DAG nodes): ctd_sym.hpp"""
    point = np.ascontiguousarray(point, dtype=np.float64)
    assert point.size == 1 + n + m + nv
    err, nn = C.c_double(), C.c_int64()
    st = lib().emu_sym_check(expr.encode(), n, m, nv, point.ctypes.data_as(C.c_void_p), C.byref(err), C.byref(nn))
    if st:
        raise RuntimeError(lib().emu_last_error().decode())
    return err.value, nn.value


class control_steps:
    """`with emu.control_steps(cs): ...` -- the models the emulator builds inside use DOCP(..., control_steps = cs, ...)"""

    def __init__(self, cs):
        self.cs = int(cs)

    def __enter__(self):
        lib().emu_set_control_steps(self.cs)

    def __exit__(self, *a):
        lib().emu_set_control_steps(1)


class value_order:
    """`with emu.value_order(1): ...` -- the models the emulator builds inside use ctd_desc.value_order = CTD_ORDER_CSR"""

    def __init__(self, order):
        self.order = int(order)

    def __enter__(self):
        lib().emu_set_value_order(self.order)

    def __exit__(self, *a):
        lib().emu_set_value_order(0)
