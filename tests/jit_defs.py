"""Run-time OCP definitions used by the JIT tests: the registry problems restated as expressions (to compare the
hiprtc-compiled kernels with the built-in ones) and one problem that exists nowhere else (checked against an on-the-fly
50-digit mpmath evaluation through tests/golden/gen_golden.py)."""
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import gen_golden as gg  # noqa: E402
from mpmath import mpf  # noqa: E402

import ctdirect_jl_amd as ct  # noqa: E402

INF = float("inf")

# name of the registry twin -> keyword arguments of ct.register_ocp (expressions follow the problem files cited in
# ctdirect.jl_amd/csrc/ctd_problems.hpp)
TWINS = {
    "goddard": dict(
        dynamics=["x2", "-Cd*x2^2*exp(-beta*(x1-1))/x3 - 1/x1^2 + u1*(Tmax/x3)", "u1*(-b*Tmax)"], m=1, nv=1, mayer="xf_1",
        boundary=["x0_1", "x0_2", "x0_3", "xf_3"], constants=dict(Cd=310, beta=500, b=2, Tmax=3.5), itf=0, maximize=True,
        state_box=([1, 0, 0.6], [1.1, 0.1, 1]), control_box=([0], [1]), variable_box=([0.01], [INF]),
        boundary_bounds=([1, 0, 1, 0.6], [1, 0, 1, 0.6])),
    "goddard_all": dict(
        dynamics=["x2", "-Cd*x2^2*exp(-beta*(x1-1))/x3 - 1/x1^2 + u1*Tmax/x3", "-b*Tmax*u1"], m=1, nv=1, mayer="xf_1",
        path=["x2", "u1", "x1 + x2 + x3 + u1 + v1"], boundary=["x0_1", "x0_2", "x0_3", "xf_3"],
        constants=dict(Cd=310, beta=500, b=2, Tmax=3.5), itf=0, maximize=True,
        state_box=([1, 0, 0], [INF, INF, 1]), control_box=([0], [INF]), variable_box=([0.01], [INF]),
        path_bounds=([-INF, -INF, 0], [0.1, 1, INF]), boundary_bounds=([1, 0, 1, 0.6], [1, 0, 1, 0.6])),
    "double_integrator_path": dict(
        dynamics=["x2", "u1"], m=1, lagrange="u1^2", path=["x1 + 0.1*x2^2"], boundary=["x0_1", "x0_2", "xf_1", "xf_2"],
        t0=0.0, tf=2.0, control_box=([-5], [5]), path_bounds=([-INF], [1.05]), boundary_bounds=([0, 0, 1, 0], [0, 0, 1, 0])),
    "quadrotor": dict(
        dynamics=["x4", "x5", "x6", "(cos(u4)*sin(x8)*cos(x7) + sin(u4)*sin(x7))*u1",
                  "(sin(u4)*sin(x8)*cos(x7) - cos(u4)*sin(x7))*u1", "(cos(x8)*cos(x7))*u1 - g", "u2", "u3"],
        m=4, nv=1, lagrange="1e-8*(x7^2 + x8^2 + u4^2 + u1^2) + 1e2*u4^2", mayer="v1", path=["cos(x8)*cos(x7)"],
        boundary=[f"x0_{i}" for i in range(1, 9)] + [f"xf_{i}" for i in range(1, 7)], constants=dict(g=9.81), itf=0,
        state_box=([-INF] * 6 + [-math.pi / 2] * 2, [INF] * 6 + [math.pi / 2] * 2),
        control_box=([0, -3, -3, -INF], [9.18 * 5, 3, 3, INF]), variable_box=([0.1], [INF]),
        path_bounds=([math.cos(1.1 / 2)], [INF]),
        boundary_bounds=([0, 0, 2.5, 0, 0, 0, 0, 0, 0.01, 5, 2.5, 0, 0, 0], [0, 0, 2.5, 0, 0, 0, 0, 0, 0.01, 5, 2.5, 0, 0, 0])),
    "double_integrator_freet0tf": dict(
        dynamics=["x2", "u1"], m=1, nv=2, mayer="v1", boundary=["x0_1", "x0_2", "xf_1", "xf_2", "v2 - v1"], it0=0, itf=1,
        maximize=True, control_box=([-1], [1]), variable_box=([0.05, 0.05], [10, 10]),
        boundary_bounds=([0, 0, 1, 0, 0.01], [0, 0, 1, 0, INF])),
    "least_squares_with_constraint": dict(
        dynamics=["-x2", "x1"], nv=2, lagrange="(t - 0.5)^2*((x1 - 0.7)^2 + (x2 - 0.7)^2)", mayer="0.01*(v1^2 + v2^2)",
        path=["x1^2 + x2^2"], boundary=["x0_1 - v1", "x0_2 - v2"], t0=0.0, tf=1.0, path_bounds=([-INF], [2.0])),
}

_registered = {}


def twin(name):
    """registers the expression twin of a registry problem once; returns its run-time name"""
    if name not in _registered:
        _registered[name] = ct.register_ocp(name + "_rt", **TWINS[name])
    return _registered[name]


# ---- a problem that only exists as expressions: forced Van der Pol with a parameter, time-dependent dynamics and cost,
# a state-control path constraint that depends on the parameter, a nonlinear boundary constraint and a Bolza cost
VDP = dict(
    dynamics=["x2", "v1*(1 - x1^2)*x2 - x1 + u1 + 0.1*sin(3*t)"], m=1, nv=2,
    lagrange="x1^2 + x2^2 + (1 + 0.5*t)*u1^2 + 0.01*v1*u1", mayer="v1^2 + xf_1*xf_2 + 0.3*v2",
    path=["x2 + 0.2*x1^2 + 0.05*v1*u1", "sqrt(1 + x1^2) - t*u1"], boundary=["x0_1", "x0_2", "xf_1^2 + xf_2^2", "v2 - 0.1*x0_1"],
    itf=1, t0=0.25, variable_box=([0.1, 0.5], [3.0, 6.0]), control_box=([-2], [2]),
    path_bounds=([-INF, 0.0], [1.5, INF]), boundary_bounds=([1, 0, 0, 1.0], [1, 0, 0.5, INF]))


class VdpMp(gg.Problem):
    """the same problem for the mpmath restatement (tests/golden/gen_golden.py): Du / dsin are looked up at call time so
    that the second-order number of gen_golden_hess.py can be swapped in"""
    name = "vdp_rt"
    n, m, nv, p, bc = 2, 1, 2, 2, 4
    freetf, lagrange, mayer = True, True, True

    def t0(self, v): return gg.Du(mpf("0.25"))
    def tf(self, v): return v[1]

    def dynamics(self, t, x, u, v):
        return [x[1] + 0, v[0] * (1 - x[0] ** 2) * x[1] - x[0] + u[0] + mpf("0.1") * gg.dsin(3 * gg.Du.lift(t))]

    def lagr(self, t, x, u, v):
        return x[0] ** 2 + x[1] ** 2 + (1 + mpf("0.5") * t) * u[0] ** 2 + mpf("0.01") * v[0] * u[0]

    def may(self, x0, xf, v): return v[0] ** 2 + xf[0] * xf[1] + mpf("0.3") * v[1]

    def path(self, t, x, u, v):
        return [x[1] + mpf("0.2") * x[0] ** 2 + mpf("0.05") * v[0] * u[0], dsqrt(1 + x[0] ** 2) - t * u[0]]

    def boundary(self, x0, xf, v): return [x0[0] + 0, x0[1] + 0, xf[0] ** 2 + xf[1] ** 2, v[1] - mpf("0.1") * x0[0]]


def dsqrt(x):
    """sqrt for whichever dual type gen_golden currently uses (first order Du or the second-order Du2)"""
    from mpmath import mp
    s = mp.sqrt(x.v)
    if hasattr(x, "chain"):
        return x.chain(s, 1 / (2 * s), -1 / (4 * s * x.v))
    return gg.Du(s, [a / (2 * s) for a in x.d])


# ---- a Lagrange problem with NO boundary and NO path rows: the case in which the leftover block of the stagewise / Euler
# Jacobian patterns (irk_stagewise.jl:550-552, euler.jl:257-259: last row x column n, hazard H2 of SURVEY.md section 8) is
# a real extra entry instead of being merged into the boundary rows
PENDULUM = dict(dynamics=["x2", "-sin(x1) + u1 - 0.05*x2"], m=1, lagrange="x1^2 + 0.2*x2^2 + 0.1*u1^2", t0=0.0, tf=1.5,
                control_box=([-3], [3]))


class PendulumMp(gg.Problem):
    name = "pendulum_rt"
    n, m, nv, p, bc = 2, 1, 0, 0, 0
    lagrange = True

    def tf(self, v): return gg.Du(mpf("1.5"))

    def dynamics(self, t, x, u, v): return [x[1] + 0, -gg.dsin(x[0]) + u[0] - mpf("0.05") * x[1]]

    def lagr(self, t, x, u, v): return x[0] ** 2 + mpf("0.2") * x[1] ** 2 + mpf("0.1") * u[0] ** 2


# ---- problems of the reference's solve catalogue (test/ci/test_all_ocp.jl, test/problems/*.jl) restated as expressions, with the
# catalogued objective of each problem file: solved end to end through the callbacks in tests/test_gpu_solve_catalogue.py
CATALOGUE = {
    # test/problems/beam.jl:4-18
    "beam": (dict(dynamics=["x2", "u1"], m=1, lagrange="u1^2", boundary=["x0_1", "x0_2", "xf_1", "xf_2"], t0=0.0, tf=1.0,
                  state_box=([0, -INF], [0.1, INF]), control_box=([-10], [10]), boundary_bounds=([0, 1, 0, -1], [0, 1, 0, -1])),
             8.898598),
    # test/problems/fuller.jl:4-15
    "fuller": (dict(dynamics=["x2", "u1"], m=1, lagrange="x1^2", boundary=["x0_1", "x0_2", "xf_1", "xf_2"], t0=0.0, tf=3.5,
                    control_box=([-1], [1]), boundary_bounds=([0, 1, 0, 0], [0, 1, 0, 0])), 2.683944e-1),
    # test/problems/jackson.jl:4-26
    "jackson": (dict(dynamics=["-u1*(k1*x1 - k2*x2)", "u1*(k1*x1 - k2*x2) - (1 - u1)*k3*x2", "(1 - u1)*k3*x2"], m=1, mayer="xf_3",
                     maximize=True, boundary=["x0_1", "x0_2", "x0_3"], constants=dict(k1=1, k2=10, k3=1), t0=0.0, tf=4.0,
                     state_box=([0, 0, 0], [1.1, 1.1, 1.1]), control_box=([0], [1]), boundary_bounds=([1, 0, 0], [1, 0, 0])),
                0.192011),
    # test/problems/vanderpol.jl:4-18
    "vanderpol": (dict(dynamics=["x2", "epsilon*omega*(1 - x1^2)*x2 - omega^2*x1 + u1"], m=1,
                       lagrange="0.5*(x1^2 + x2^2 + u1^2)", boundary=["x0_1", "x0_2"], constants=dict(omega=1, epsilon=1),
                       t0=0.0, tf=2.0, boundary_bounds=([1, 0], [1, 0])), 1.047921),
    # test/problems/simple_integrator.jl:5-16
    "simple_integrator": (dict(dynamics=["-x1 - u1 + u2"], m=2, lagrange="(u1 + u2)^2", boundary=["x0_1", "xf_1"], t0=0.0, tf=1.0,
                               control_box=([0, 0], [INF, INF]), boundary_bounds=([-1, 0], [-1, 0])), 3.13e-1),
    # test/problems/bolza.jl:4-17
    "bolza_freetf": (dict(dynamics=["v1*u1"], m=1, nv=1, lagrange="0.5*u1^2", mayer="v1", boundary=["x0_1", "xf_1"], itf=0,
                          state_box=([0], [INF]), variable_box=([0.1], [INF]), boundary_bounds=([0, 1], [0, 1])), 1.476),
    # test/problems/robbins.jl:4-20
    "robbins": (dict(dynamics=["x2", "x3", "u1"], m=1, lagrange="alpha*x1 + beta*x1^2 + gamma*u1^2",
                     boundary=["x0_1", "x0_2", "x0_3", "xf_1", "xf_2", "xf_3"], constants=dict(alpha=3, beta=0, gamma=0.5),
                     t0=0.0, tf=10.0, state_box=([0, -INF, -INF], [INF, INF, INF]),
                     boundary_bounds=([1, -2, 0, 0, 0, 0], [1, -2, 0, 0, 0, 0])), 19.4),
    # test/problems/double_integrator.jl:4-18
    "double_integrator_tf": (dict(dynamics=["x2", "u1"], m=1, nv=1, mayer="v1", boundary=["x0_1", "x0_2", "xf_1", "xf_2"], itf=0,
                                  control_box=([-1], [1]), variable_box=([0.05], [INF]),
                                  boundary_bounds=([0, 0, 1, 0], [0, 0, 1, 0])), 2.0),
    # test/problems/moonlander.jl:7-82 (the problem the reference's "quadrotor" testset actually solves, test_all_ocp.jl:90-92);
    # F_tot = R(theta) [0, F1 + F2]: ddp1 = -sin(theta) (F1 + F2) / m, ddp2 = cos(theta) (F1 + F2) / m - g
    "moonlander": (dict(dynamics=["x3", "x4", "-sin(x5)*(u1 + u2)/mass", "cos(x5)*(u1 + u2)/mass - g", "x6", "(1/I)*(D/2)*(u2 - u1)"],
                        m=2, nv=1, mayer="v1", itf=0, constants=dict(mass=1.0, g=9.81, I=0.1, D=1.0),
                        boundary=[f"x0_{i}" for i in range(1, 7)] + ["xf_1", "xf_2", "xf_3", "xf_4"],
                        control_box=([0, 0], [2 * 9.81, 2 * 9.81]), variable_box=([0.1], [INF]),
                        boundary_bounds=([0] * 6 + [5, 5, 0, 0], [0] * 6 + [5, 5, 0, 0])), 9.62e-1, dict(control=[5.0, 5.0])),
}


def catalogue(name):
    """(run-time problem name, catalogued objective, init of the problem file or None)"""
    key = name + "_cat"
    entry = CATALOGUE[name]
    if key not in _registered:
        _registered[key] = ct.register_ocp(key, **entry[0])
    return _registered[key], entry[1], (entry[2] if len(entry) > 2 else None)


# ---- every function of the expression grammar in one problem (log, tan, atan, tanh, abs beside exp / sin / cos / sqrt)
FUNCS = dict(
    dynamics=["x2 + log(1 + x1^2) - 0.3*tan(0.5*u1)", "atan(x1*x2) - tanh(v1*x2) + u1 + 0.1*abs(x1 - 0.2)"], m=1, nv=1,
    lagrange="log(2 + u1^2) + tanh(x1)^2 + 0.05*abs(x2)", mayer="atan(xf_1) + v1*tan(0.3*xf_2)",
    path=["tan(0.4*x1) + abs(u1 + 0.7)"], boundary=["x0_1", "log(1 + x0_2^2)", "tanh(xf_1) + xf_2"], t0=0.0, tf=1.2,
    path_bounds=([-INF], [3.0]), boundary_bounds=([0.5, 0.0, 0.0], [0.5, 0.5, 1.0]))


def _mp_unary(x, f0, f1, f2):
    """unary function for whichever dual type gen_golden currently uses (first order Du or the second-order Du2)"""
    if hasattr(x, "chain"):
        return x.chain(f0, f1, f2)
    return gg.Du(f0, [a * f1 for a in x.d])


def dlog(x):
    from mpmath import mp
    x = gg.Du.lift(x)
    return _mp_unary(x, mp.log(x.v), 1 / x.v, -1 / x.v ** 2)


def dtan(x):
    from mpmath import mp
    x = gg.Du.lift(x)
    t = mp.tan(x.v)
    return _mp_unary(x, t, 1 + t * t, 2 * t * (1 + t * t))


def datan(x):
    from mpmath import mp
    x = gg.Du.lift(x)
    q = 1 / (1 + x.v ** 2)
    return _mp_unary(x, mp.atan(x.v), q, -2 * x.v * q * q)


def dtanh(x):
    from mpmath import mp
    x = gg.Du.lift(x)
    t = mp.tanh(x.v)
    return _mp_unary(x, t, 1 - t * t, -2 * t * (1 - t * t))


def dabs(x):
    x = gg.Du.lift(x)
    return _mp_unary(x, abs(x.v), mpf(1) if x.v > 0 else (mpf(-1) if x.v < 0 else mpf(0)), mpf(0))


class FuncsMp(gg.Problem):
    name = "funcs_rt"
    n, m, nv, p, bc = 2, 1, 1, 1, 3
    lagrange, mayer = True, True

    def tf(self, v): return gg.Du(mpf("1.2"))

    def dynamics(self, t, x, u, v):
        return [x[1] + dlog(1 + x[0] ** 2) - mpf("0.3") * dtan(mpf("0.5") * u[0]),
                datan(x[0] * x[1]) - dtanh(v[0] * x[1]) + u[0] + mpf("0.1") * dabs(x[0] - mpf("0.2"))]

    def lagr(self, t, x, u, v): return dlog(2 + u[0] ** 2) + dtanh(x[0]) ** 2 + mpf("0.05") * dabs(x[1])

    def may(self, x0, xf, v): return datan(xf[0]) + v[0] * dtan(mpf("0.3") * xf[1])

    def path(self, t, x, u, v): return [dtan(mpf("0.4") * x[0]) + dabs(u[0] + mpf("0.7"))]

    def boundary(self, x0, xf, v): return [x0[0] + 0, dlog(1 + x0[1] ** 2), dtanh(xf[0]) + xf[1]]


# ---- four optimisation variables, both times free and NOT at the ends of v (it0 = 2, itf = 1): every V x V entry, the
# composite time directions and the K x V terms of the Hessian at once
FOURV = dict(
    dynamics=["x2*v1 + 0.2*sin(t)", "-x1*(1 + v4^2) + u1*v1 - 0.1*x2^3"], m=1, nv=4, it0=2, itf=1,
    lagrange="0.5*u1^2 + v4*x1^2 + 0.1*t*x2", mayer="xf_1^2 + v1*v4 + 0.5*(v2 - v3)^2",
    path=["x1*u1 + v4*t"], boundary=["x0_1", "x0_2 - v1", "xf_1*xf_2", "v2 - v3"],
    variable_box=([0.1, 0.5, -1.0, -2.0], [3.0, 6.0, 0.4, 2.0]), control_box=([-2], [2]),
    path_bounds=([-INF], [2.0]), boundary_bounds=([1, 0, 0, 0.2], [1, 0, 0.5, INF]))


class FourVMp(gg.Problem):
    name = "fourv_rt"
    n, m, nv, p, bc = 2, 1, 4, 1, 4
    freet0, freetf, lagrange, mayer = True, True, True, True

    def t0(self, v): return v[2]
    def tf(self, v): return v[1]

    def dynamics(self, t, x, u, v):
        return [x[1] * v[0] + mpf("0.2") * gg.dsin(gg.Du.lift(t)), -x[0] * (1 + v[3] ** 2) + u[0] * v[0] - mpf("0.1") * x[1] ** 2 * x[1]]

    def lagr(self, t, x, u, v): return mpf("0.5") * u[0] ** 2 + v[3] * x[0] ** 2 + mpf("0.1") * t * x[1]

    def may(self, x0, xf, v): return xf[0] ** 2 + v[0] * v[3] + mpf("0.5") * (v[1] - v[2]) ** 2

    def path(self, t, x, u, v): return [x[0] * u[0] + v[3] * t]

    def boundary(self, x0, xf, v): return [x0[0] + 0, x0[1] - v[0], xf[0] * xf[1], v[1] - v[2]]
