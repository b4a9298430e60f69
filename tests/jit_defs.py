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
    # the 12-state quadrotor (ctd_problems.hpp Quadrotor12OCP): 16 dynamics directions = four chunks -- the generated functor carries
    # its dynamics split by rows into four parts (DYN_PARTS), one wave each, like the built-in problem
    "quadrotor12": dict(
        dynamics=["x4", "x5", "x6", "(cos(x9)*sin(x8)*cos(x7) + sin(x9)*sin(x7))*u1", "(sin(x9)*sin(x8)*cos(x7) - cos(x9)*sin(x7))*u1",
                  "(cos(x8)*cos(x7))*u1 - g", "x10 + sin(x7)*(sin(x8)/cos(x8))*x11 + cos(x7)*(sin(x8)/cos(x8))*x12",
                  "cos(x7)*x11 - sin(x7)*x12", "(sin(x7)*x11 + cos(x7)*x12)/cos(x8)", "((Jy - Jz)*x11*x12 + u2)/Jx",
                  "((Jz - Jx)*x12*x10 + u3)/Jy", "((Jx - Jy)*x10*x11 + u4)/Jz"],
        m=4, nv=1, lagrange="1e-8*(x7^2 + x8^2 + u1^2) + 1e-2*(u2^2 + u3^2 + u4^2) + 1e2*x9^2", mayer="v1", path=["cos(x8)*cos(x7)"],
        boundary=[f"x0_{i}" for i in range(1, 13)] + [f"xf_{i}" for i in range(1, 9)] + [f"xf_{i}" for i in range(10, 13)],
        constants=dict(g=9.81, Jx=0.03, Jy=0.03, Jz=0.06), itf=0,
        state_box=([-INF] * 6 + [-math.pi / 2] * 2 + [-INF] * 4, [INF] * 6 + [math.pi / 2] * 2 + [INF] * 4),
        control_box=([0, -1, -1, -1], [45.9, 1, 1, 1]), variable_box=([0.1], [INF]),
        path_bounds=([math.cos(0.55)], [INF]),
        boundary_bounds=([0, 0, 2.5] + [0] * 9 + [0.01, 5, 2.5] + [0] * 8, [0, 0, 2.5] + [0] * 9 + [0.01, 5, 2.5] + [0] * 8)),
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


def _glider():
    """test/problems/glider.jl:8-100 (hang glider in a thermal updraft, COPS): the helper values of its `dynamics` inlined"""
    r = "((x1/r_0 - 2.5)^2)"
    w = f"(x4 - u_c*(1 - {r})*exp(-{r}))"
    v = f"sqrt(x3^2 + {w}^2)"
    D = f"(0.5*(c0 + c1*u1^2)*rho*S*{v}^2)"
    L = f"(0.5*u1*rho*S*{v}^2)"
    tf0 = (900.0 - 1000.0) / -1.288
    return (dict(dynamics=["x3", "x4", f"(-{L}*({w}/{v}) - {D}*(x3/{v}))/mass", f"(({L}*(x3/{v}) - {D}*({w}/{v}))/mass) - g"],
                 m=1, nv=1, itf=0, t0=0.0, mayer="xf_1", maximize=True,
                 constants=dict(u_c=2.5, r_0=100.0, mass=100.0, g=9.81, c0=0.034, c1=0.069662, S=14.0, rho=1.13),
                 boundary=["x0_1", "x0_2", "x0_3", "x0_4", "xf_2", "xf_3", "xf_4"],
                 boundary_bounds=([0.0, 1000.0, 13.23, -1.288, 900.0, 13.23, -1.288],) * 2,
                 state_box=([0, -INF, 0, -INF], [INF] * 4), control_box=([0.0], [1.4]), variable_box=([10.0], [INF])),
            1.25e3,
            dict(time=[0.0, tf0], state=[[0.0, 1000.0, 13.23, -1.288], [13.23 * tf0, 900.0, 13.23, -1.288]],
                 control=[[0.7], [0.7]], variable=[tf0]))


def _space_shuttle():
    """test/problems/space_shuttle.jl:8-150 (re-entry, maximal cross range; no heating constraint)"""
    import math
    h, v, adeg = "(x1*1e5)", "(x4*1e4)", "(u1*180/pi)"
    cD, cL = f"(b0 + b1*{adeg} + b2*{adeg}^2)", f"(a0 + a1*{adeg})"
    rho = f"(rho0*exp(-{h}/hr))"
    D, L = f"(0.5*{cD}*S*{rho}*{v}^2)", f"(0.5*{cL}*S*{rho}*{v}^2)"
    r = f"(Re + {h})"
    g = f"(mu/{r}^2)"
    dyn = [f"{v}*sin(x5)/1e5", f"({v}/{r})*cos(x5)*sin(x6)/cos(x3)", f"({v}/{r})*cos(x5)*cos(x6)", f"(-({D}/mass) - {g}*sin(x5))/1e4",
           f"({L}/(mass*{v}))*cos(u2) + cos(x5)*(({v}/{r}) - ({g}/{v}))",
           f"(1/(mass*{v}*cos(x5)))*{L}*sin(u2) + ({v}/({r}*cos(x3)))*cos(x5)*sin(x6)*sin(x3)"]
    d2r = math.radians
    x0 = [2.6, 0.0, 0.0, 2.56, d2r(-1), d2r(90)]
    xT = [0.8, 0.0, 0.0, 0.25, d2r(-5), d2r(90)]
    return (dict(dynamics=dyn, m=2, nv=1, itf=0, t0=0.0, mayer="xf_3", maximize=True,
                 constants=dict(pi=math.pi, mass=203000.0 / 32.174, rho0=0.002378, hr=23800.0, Re=20902900.0, mu=0.14076539e17, S=2690.0,
                                a0=-0.20704, a1=0.029244, b0=0.07854, b1=-0.61592e-2, b2=0.621408e-3),
                 boundary=[f"x0_{i}" for i in range(1, 7)] + ["xf_1", "xf_4", "xf_5"],
                 boundary_bounds=(x0 + [0.8, 0.25, d2r(-5)],) * 2,
                 state_box=([0, -INF, d2r(-89), 0, d2r(-89), -INF], [INF, INF, d2r(89), INF, d2r(89), INF]),
                 control_box=([d2r(-90), d2r(-89)], [d2r(90), d2r(1)]), variable_box=([1750.0], [2250.0])),
            d2r(34.18),
            dict(time=[0.0, 500.0], state=[x0, xT], control=[[0.0, 0.0], [0.0, 0.0]], variable=[500.0]))


def _truck_trailer():
    """test/problems/truck_trailer.jl:6-130 (truck with two trailers, minimum time + alignment)"""
    import math
    b01, b12 = "(x3 - x4)", "(x4 - x5)"
    dth0 = "(x6/L0*tan(x7))"
    dth1 = f"(x6/L1*sin({b01}) - M0/L1*cos({b01})*{dth0})"
    v1 = f"(x6*cos({b01}) + M0*sin({b01})*{dth0})"
    dth2 = f"({v1}/L2*sin({b12}) - M1/L2*cos({b12})*{dth1})"
    v2 = f"({v1}*cos({b12}) + M1*sin({b12})*{dth1})"
    hp = math.pi / 2
    return (dict(dynamics=[f"{v2}*cos(x5)", f"{v2}*sin(x5)", dth0, dth1, dth2, "u1", "u2"], m=2, nv=1, itf=0, t0=0.0,
                 lagrange=f"{b01}^2 + {b12}^2", mayer="v1", constants=dict(L0=0.4, M0=0.1, L1=1.1, M1=0.2, L2=0.8),
                 path=[b01, b12], path_bounds=([-hp, -hp], [hp, hp]),
                 boundary=["x0_1", "x0_2", "x0_3", "x0_4", "x0_5", "xf_1", "xf_2", "xf_5", "xf_3 - xf_4", "xf_4 - xf_5"],
                 boundary_bounds=([0, 0, 0, 0, 0, 0.0, -2.0, hp, 0.0, 0.0],) * 2,
                 state_box=([-INF, -INF, -hp, -hp, -INF, -0.2, -math.pi / 6], [INF, INF, hp, hp, INF, 0.2, math.pi / 6]),
                 control_box=([-1.0, -math.pi / 10], [1.0, math.pi / 10]), variable_box=([1.0], [1000.0])),
            59.28, dict(variable=[10.0]))


CATALOGUE.update({
    # test/problems/double_integrator.jl:100-113
    "double_integrator_nobounds": (dict(dynamics=["x2", "u1"], m=1, lagrange="0.5*u1^2", boundary=["x0_1", "x0_2", "xf_1", "xf_2"],
                                        t0=0.0, tf=1.0, boundary_bounds=([1, -2, 0, 0], [1, -2, 0, 0])), 2.0),
    # test/problems/electric_vehicle.jl:8-63
    "electric_vehicle": (dict(dynamics=["x2", "h1*u1 - h2*x2^2 - h0 - (p0 + p1*x1 + p2*x1^2 + p3*x1^3)"], m=1,
                              lagrange="b1*u1*x2 + b2*u1^2", t0=0.0, tf=1.0,
                              constants=dict(b1=1e3, b2=1e3, h0=0.1, h1=1.0, h2=1e-3, p0=3.0, p1=0.4, p2=-1.0, p3=0.1),
                              boundary=["x0_1", "x0_2", "xf_1", "xf_2"], boundary_bounds=([0, 0, 10.0, 0], [0, 0, 10.0, 0]),
                              state_box=([0, 0], [INF, INF])), 1.23e6,
                         dict(time=[0.0, 1.0], state=[[0.0, 1.0], [10.0, 1.0]], control=[[0.5], [0.5]])),
    "glider": _glider(),
    # test/problems/insurance.jl:7-62 (Bocop's non-audit insurance example; the reference solves it with :trapeze only).
    # alpha = 4: m^(alpha/2) = m^2, m^(alpha/2 - 1) = m;  k = 0: epsilon = 0;  sigma = 0
    "insurance": (dict(dynamics=["(1 - gamma*t*(2*x2/(1 + x2^2)^2)/u5)*u1", "u1", "x1*(lambda*exp(-lambda*t) + exp(-lambda*10)/10)"],
                       m=5, nv=1, t0=0.0, tf=10.0, lagrange="u4*(lambda*exp(-lambda*t) + exp(-lambda*10)/10)", maximize=True,
                       constants={"gamma": 0.2, "lambda": 0.25, "h0": 1.5, "w": 1.0, "s": 10.0},
                       path=["u2 - (w - v1 + x1 - x2)", "u3 - (h0 - gamma*t*(1 - x2^2/(1 + x2^2)))", "u4 - (1 - exp(-s*u2) + u3)",
                             "u5 - s*exp(-s*u2)"], path_bounds=([0] * 4, [0] * 4),
                       boundary=["x0_1", "x0_2", "x0_3", "v1 - xf_3"], boundary_bounds=([0, 0.001, 0, 0], [0, 0.001, 0, 0]),
                       state_box=([0, 0, -INF], [1.1, 1.1, INF]), control_box=([0, 0, 0, 0, 1e-8], [25, INF, INF, INF, INF]),
                       variable_box=([0], [INF])), 2.059511),
    "space_shuttle": _space_shuttle(),
    "truck_trailer": _truck_trailer(),
    # problems of the catalogue that are in the compiled registry (solved through the built-in kernels)
    "goddard_all": ("registry:goddard_all", 1.01257, "problem"),                         # test/problems/goddard.jl:87-158
    "double_integrator_freet0tf": ("registry:double_integrator_freet0tf", 8.0, None),     # test/problems/double_integrator.jl:79-99
})


def catalogue(name):
    """(run-time problem name, catalogued objective, init of the problem file or None)"""
    key = name + "_cat"
    entry = CATALOGUE[name]
    if isinstance(entry[0], str):                     # a registry problem
        return entry[0].split(":", 1)[1], entry[1], entry[2]
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
