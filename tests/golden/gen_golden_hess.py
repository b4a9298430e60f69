#!/usr/bin/env python3
"""Generate the golden HESSIAN fixtures tests/golden/hess_*.json (run once, output committed).

Same idea as gen_golden.py (which this script imports for the problem files, the discretisations and the input
vectors): the reference holds no test that inspects a Hessian value -- it only pins nnzh (6519) --
so the values are pinned by an independent statement: the Python/mpmath restatement of __objective and __constraints!
pushed through a sparse SECOND-order forward number at 50 digits.  What is differentiated is exactly what ADNLPModels
differentiates for hess_coord!(nlp, x, y, vals; obj_weight): the Lagrangian  obj_weight * f(x) + sum_i y_i c_i(x)
of the closures handed over at src/collocation.jl:137-149.

Each fixture: problem/scheme/grid, xu (the input of the fixture of the same tag), the multipliers y, obj_weight, and
every nonzero entry (row >= col, 0-based) of the exact Hessian, correctly rounded to double.

Usage:  python tests/golden/gen_golden_hess.py      (about a minute)
"""
import json
import math
import os
import sys

from mpmath import mp, mpf

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402

mp.dps = 50


class Du2:
    """value + sparse gradient {i: d/dx_i} + sparse lower-triangular Hessian {(i, j), i >= j}"""
    __slots__ = ("v", "d", "h")

    def __init__(self, v, d=None, h=None):
        self.v = mpf(v)
        self.d = d if d is not None else {}
        self.h = h if h is not None else {}

    @staticmethod
    def lift(x):
        return x if isinstance(x, Du2) else Du2(x)

    @staticmethod
    def _lin(ca, a, cb, b):
        r = {k: ca * x for k, x in a.items()}
        for k, x in b.items():
            r[k] = r.get(k, mpf(0)) + cb * x
        return r

    def __add__(self, o):
        o = Du2.lift(o)
        return Du2(self.v + o.v, Du2._lin(1, self.d, 1, o.d), Du2._lin(1, self.h, 1, o.h))

    __radd__ = __add__

    def __sub__(self, o):
        o = Du2.lift(o)
        return Du2(self.v - o.v, Du2._lin(1, self.d, -1, o.d), Du2._lin(1, self.h, -1, o.h))

    def __rsub__(self, o):
        return Du2.lift(o) - self

    def __neg__(self):
        return Du2(-self.v, {k: -x for k, x in self.d.items()}, {k: -x for k, x in self.h.items()})

    def __mul__(self, o):
        o = Du2.lift(o)
        h = Du2._lin(o.v, self.h, self.v, o.h)
        for i, a in self.d.items():
            for j, b in o.d.items():
                k = (i, j) if i >= j else (j, i)
                t = a * b
                h[k] = h.get(k, mpf(0)) + (2 * t if i == j else t)
        return Du2(self.v * o.v, Du2._lin(o.v, self.d, self.v, o.d), h)

    __rmul__ = __mul__

    def chain(self, f0, f1, f2):
        h = {k: f1 * x for k, x in self.h.items()}
        items = list(self.d.items())
        for i, a in items:
            for j, b in items:
                if i >= j:
                    h[(i, j)] = h.get((i, j), mpf(0)) + f2 * a * b
        return Du2(f0, {k: f1 * x for k, x in self.d.items()}, h)

    def __truediv__(self, o):
        o = Du2.lift(o)
        q = 1 / o.v
        return self * o.chain(q, -q * q, 2 * q * q * q)

    def __rtruediv__(self, o):
        return Du2.lift(o) / self

    def __pow__(self, k):
        assert k == 2
        return self * self


def dexp2(x):
    e = mp.exp(x.v)
    return x.chain(e, e, e)


def dsin2(x):
    s, c = mp.sin(x.v), mp.cos(x.v)
    return x.chain(s, c, -s)


def dcos2(x):
    s, c = mp.sin(x.v), mp.cos(x.v)
    return x.chain(c, -s, -c)


def multipliers(ncon):
    """deterministic multipliers of mixed sign and magnitude (closed form, plain floats)"""
    return [float(math.sin(0.7 * r + 0.3) * (1 + 0.5 * math.cos(0.13 * r))) for r in range(ncon)]


OBJ_WEIGHT = 0.75


def run_case(tag, prob, scheme, N=None, time_grid=None, control_steps=1):
    # inputs: identical to the first-order fixture of the same tag (gen_golden.fill_inputs is deterministic)
    gg.Du, gg.dexp, gg.dsin, gg.dcos = gg_first_order
    d = gg.Docp(prob, scheme, N=N, time_grid=time_grid, control_steps=control_steps)
    xu = gg.fill_inputs(d)
    gg.Du, gg.dexp, gg.dsin, gg.dcos = Du2, dexp2, dsin2, dcos2
    z = [Du2(mpf(x), {j: mpf(1)}) for j, x in enumerate(xu)]
    y = multipliers(d.ncon)
    lag = OBJ_WEIGHT * Du2.lift(d.objective(z))
    for r, cr in enumerate(d.constraints(z)):
        lag = lag + mpf(y[r]) * Du2.lift(cr)
    entries = [[i, j, gg.hexf(v)] for (i, j), v in sorted(lag.h.items()) if v != 0]
    out = {
        "tag": tag, "problem": prob.name, "scheme": scheme, "grid_size": d.N,
        "time_grid": [gg.hexf(t) for t in time_grid] if time_grid is not None else None,
        "xu": [gg.hexf(x) for x in xu], "y": [gg.hexf(v) for v in y], "obj_weight": OBJ_WEIGHT,
        **({"control_steps": control_steps} if control_steps != 1 else {}),
        "hess_nonzeros": entries,
        "note": "lower triangle (row >= col, 0-based) of obj_weight*d2f + sum y_i d2c_i; correctly rounded doubles "
                "(C99 hex) of a 50-digit mpmath evaluation",
    }
    path = os.path.join(HERE, "hess_" + tag + ".json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(f"hess_{tag}: nvar={d.nvar} ncon={d.ncon} nnz(true)={len(entries)} -> {os.path.relpath(path)}")


gg_first_order = (gg.Du, gg.dexp, gg.dsin, gg.dcos)


def main_control_steps():
    """control_steps > 1 (direct shooting layout, midpoint scheme; same tags and inputs as gen_golden.main_control_steps):
    `python gen_golden_hess.py cs` writes only these"""
    try:
        run_case("cs2_goddard_midpoint_N4", gg.Goddard(), "midpoint", N=4, control_steps=2)
        run_case("cs3_goddard_midpoint_nonuniform", gg.Goddard(), "midpoint", time_grid=[0.0, 0.1, 0.45, 1.0], control_steps=3)
        run_case("cs2_dip_midpoint_N4", gg.DoubleIntegratorPath(), "midpoint", N=4, control_steps=2)
        run_case("cs3_dip_midpoint_nonuniform", gg.DoubleIntegratorPath(), "midpoint", time_grid=[0.0, 0.3, 0.5, 0.6, 1.0], control_steps=3)
        run_case("cs2_goddard_all_midpoint_N3", gg.GoddardAll(), "midpoint", N=3, control_steps=2)
        run_case("cs3_freet0tf_midpoint_N3", gg.DoubleIntegratorFreeT0Tf(), "midpoint", N=3, control_steps=3)
        run_case("cs2_quadrotor_midpoint_N2", gg.Quadrotor8(), "midpoint", N=2, control_steps=2)
        run_case("cs3_quadrotor_midpoint_N2", gg.Quadrotor8(), "midpoint", N=2, control_steps=3)
        # more than 3 controls per step: the kernels sum the points of a step before the emission (hess_sums_stages)
        run_case("cs5_dip_midpoint_nonuniform", gg.DoubleIntegratorPath(), "midpoint", time_grid=[0.0, 0.3, 0.5, 0.6, 1.0], control_steps=5)
        run_case("cs5_goddard_all_midpoint_N3", gg.GoddardAll(), "midpoint", N=3, control_steps=5)
    finally:
        gg.Du, gg.dexp, gg.dsin, gg.dcos = gg_first_order


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "cs":
        return main_control_steps()
    try:
        run_case("goddard_midpoint_N4", gg.Goddard(), "midpoint", N=4)
        run_case("goddard_trapeze_N4", gg.Goddard(), "trapeze", N=4)
        run_case("goddard_gl2_N4", gg.Goddard(), "gauss_legendre_2", N=4)
        run_case("goddard_gl3_N3", gg.Goddard(), "gauss_legendre_3", N=3)
        run_case("goddard_gl2cc_N3", gg.Goddard(), "gauss_legendre_2_constant_control", N=3)
        run_case("goddard_gl3cc_nonuniform", gg.Goddard(), "gauss_legendre_3_constant_control", time_grid=[0.0, 0.1, 0.45, 1.0])
        run_case("goddard_gl1_N3", gg.Goddard(), "gauss_legendre_1", N=3)
        run_case("goddard_all_trapeze_N4", gg.GoddardAll(), "trapeze", N=4)
        run_case("goddard_all_gl2_N3", gg.GoddardAll(), "gauss_legendre_2", N=3)
        run_case("dip_midpoint_N4", gg.DoubleIntegratorPath(), "midpoint", N=4)
        run_case("dip_trapeze_N3", gg.DoubleIntegratorPath(), "trapeze", N=3)
        run_case("dip_gl3_N2", gg.DoubleIntegratorPath(), "gauss_legendre_3", N=2)
        run_case("quadrotor_gl3_N2", gg.Quadrotor8(), "gauss_legendre_3", N=2)
        run_case("quadrotor_midpoint_N3", gg.Quadrotor8(), "midpoint", N=3)
        run_case("quadrotor12_trapeze_N2", gg.Quadrotor12(), "trapeze", N=2)
        run_case("rotrate_gl3cc_N2", gg.EstimateRotationRate(), "gauss_legendre_3_constant_control", N=2)
        run_case("lsq_trapeze_N3", gg.LeastSquaresConstraint(), "trapeze", N=3)
        run_case("lsq_gl2_N3", gg.LeastSquaresConstraint(), "gauss_legendre_2", N=3)
        run_case("freet0tf_midpoint_N3", gg.DoubleIntegratorFreeT0Tf(), "midpoint", N=3)
        run_case("freet0tf_gl2_N2", gg.DoubleIntegratorFreeT0Tf(), "gauss_legendre_2", N=2)
        run_case("freet0tf_trapeze_N3", gg.DoubleIntegratorFreeT0Tf(), "trapeze", N=3)
        run_case("scalar_gauss_legendre_2_perturbed", gg.StagewiseScalar(), "gauss_legendre_2", time_grid=[0.0, 0.2, 0.6, 1.0])
        run_case("goddard_euler_N4", gg.Goddard(), "euler", N=4)
        run_case("goddard_all_euler_implicit_N4", gg.GoddardAll(), "euler_implicit", N=4)
        run_case("dip_euler_implicit_N3", gg.DoubleIntegratorPath(), "euler_implicit", N=3)
        run_case("quadrotor_euler_N3", gg.Quadrotor8(), "euler", N=3)
        run_case("quadrotor_euler_implicit_N2", gg.Quadrotor8(), "euler_implicit", N=2)
        run_case("lsq_euler_implicit_N3", gg.LeastSquaresConstraint(), "euler_implicit", N=3)
        run_case("freet0tf_euler_N3", gg.DoubleIntegratorFreeT0Tf(), "euler", N=3)
        run_case("freet0tf_euler_implicit_N3", gg.DoubleIntegratorFreeT0Tf(), "euler_implicit", N=3)
    finally:
        gg.Du, gg.dexp, gg.dsin, gg.dcos = gg_first_order
    main_control_steps()


if __name__ == "__main__":
    main()
