#!/usr/bin/env python3
"""Generate the golden value fixtures under tests/golden/ (run once, output committed).

The reference (CTDirect.jl, Julia) cannot be executed in the build container (no Julia), and it has no test
that inspects a Jacobian VALUE.  This script is therefore a third, independent statement
of the hot path -- pure Python over mpmath at 50 significant digits with a dense forward-mode dual number --
written from the Julia sources:

    __constraints!         src/DOCP_functions.jl:80-115         stepPathConstraints!  :122-140
    __objective            src/DOCP_functions.jl:23-54
    trapeze                src/ode/trapeze.jl:50-71 (work), :78-110 (integral), :118-142 (step)
    midpoint               src/ode/midpoint.jl:47-72, :79-97, :124-140
    GL constant control    src/ode/irk.jl:179-228, :236-308
    GL stagewise           src/ode/irk_stagewise.jl:173-224 (getters), :344-384, :394-460
    time grid              src/DOCP_data.jl:176-214, :437-458
    problems               test/problems/goddard.jl, quadrotor.jl, double_integrator.jl, autonomous_system.jl,
                           test/ci/test_discretization_stagewise.jl:1-14  (+ the two build-defined problems)

Each fixture holds: the problem/scheme/grid, the input vector xu (exact doubles), and c(xu), objective(xu),
grad objective, and every structurally nonzero entry (row, col, value) of the dense Jacobian dc/dxu, all
correctly rounded to double from the 50-digit result.  Both the C++ oracle (oracle/) and the HIP engine are
tested against these files.

Usage:  python tests/golden/gen_golden.py        (rewrites tests/golden/*.json; takes about a minute)
"""
import json
import math
import os
import sys

from mpmath import mp, mpf

mp.dps = 50
HERE = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------------------------
# dense forward-mode dual over mpf
# --------------------------------------------------------------------------------------------
class Du:
    __slots__ = ("v", "d")
    NV = 0

    def __init__(self, v, d=None):
        self.v = mpf(v)
        self.d = d if d is not None else [mpf(0)] * Du.NV

    @staticmethod
    def lift(x):
        return x if isinstance(x, Du) else Du(x)

    def __add__(self, o):
        o = Du.lift(o)
        return Du(self.v + o.v, [a + b for a, b in zip(self.d, o.d)])

    __radd__ = __add__

    def __sub__(self, o):
        o = Du.lift(o)
        return Du(self.v - o.v, [a - b for a, b in zip(self.d, o.d)])

    def __rsub__(self, o):
        return Du.lift(o) - self

    def __neg__(self):
        return Du(-self.v, [-a for a in self.d])

    def __mul__(self, o):
        o = Du.lift(o)
        return Du(self.v * o.v, [a * o.v + self.v * b for a, b in zip(self.d, o.d)])

    __rmul__ = __mul__

    def __truediv__(self, o):
        o = Du.lift(o)
        q = self.v / o.v
        return Du(q, [(a - q * b) / o.v for a, b in zip(self.d, o.d)])

    def __rtruediv__(self, o):
        return Du.lift(o) / self

    def __pow__(self, k):
        assert k == 2
        return self * self


def dexp(x):
    e = mp.exp(x.v)
    return Du(e, [e * a for a in x.d])


def dsin(x):
    s, c = mp.sin(x.v), mp.cos(x.v)
    return Du(s, [c * a for a in x.d])


def dcos(x):
    s, c = mp.sin(x.v), mp.cos(x.v)
    return Du(c, [-s * a for a in x.d])


# --------------------------------------------------------------------------------------------
# problems (Python restatement of the Julia problem files)
# --------------------------------------------------------------------------------------------
class Problem:
    n = m = nv = p = bc = 0
    freet0 = freetf = lagrange = mayer = False

    def t0(self, v): return Du(0)
    def tf(self, v): raise NotImplementedError
    def dynamics(self, t, x, u, v): raise NotImplementedError
    def lagr(self, t, x, u, v): return Du(0)
    def may(self, x0, xf, v): return Du(0)
    def path(self, t, x, u, v): return []
    def boundary(self, x0, xf, v): return []


class Goddard(Problem):                      # test/problems/goddard.jl:7-49
    name = "goddard"
    n, m, nv, p, bc = 3, 1, 1, 0, 4
    freetf, mayer = True, True
    Cd, beta, b, Tmax = 310, 500, 2, mpf("3.5")

    def tf(self, v): return v[0]

    def dynamics(self, t, x, u, v):
        r, vv, mm = x
        D = self.Cd * vv ** 2 * dexp(-self.beta * (r - 1))
        F0 = [vv, -D / mm - 1 / r ** 2, Du(0)]
        F1 = [Du(0), self.Tmax / mm, Du(-self.b * self.Tmax)]
        return [F0[i] + u[0] * F1[i] for i in range(3)]

    def may(self, x0, xf, v): return xf[0]

    def boundary(self, x0, xf, v): return [x0[0], x0[1], x0[2], xf[2]]


class GoddardAll(Goddard):                   # test/problems/goddard.jl:87-158
    name = "goddard_all"
    p = 3

    def dynamics(self, t, x, u, v):
        D = self.Cd * x[1] ** 2 * dexp(-self.beta * (x[0] - 1))
        return [x[1] + 0, -D / x[2] - 1 / x[0] ** 2 + u[0] * self.Tmax / x[2], -self.b * self.Tmax * u[0]]

    def path(self, t, x, u, v):
        return [x[1] + 0, u[0] + 0, x[0] + x[1] + x[2] + u[0] + v[0]]


class DoubleIntegratorPath(Problem):         # double_integrator.jl:42-58 + build-defined path q + 0.1 w^2
    name = "double_integrator_path"
    n, m, nv, p, bc = 2, 1, 0, 1, 4
    lagrange = True

    def tf(self, v): return Du(2)
    def dynamics(self, t, x, u, v): return [x[1] + 0, u[0] + 0]
    def lagr(self, t, x, u, v): return u[0] ** 2
    def path(self, t, x, u, v): return [x[0] + mpf("0.1") * x[1] ** 2]
    def boundary(self, x0, xf, v): return [x0[0], x0[1], xf[0], xf[1]]


class Quadrotor8(Problem):                   # test/problems/quadrotor.jl:7-105
    name = "quadrotor"
    n, m, nv, p, bc = 8, 4, 1, 1, 14
    freetf, lagrange, mayer = True, True, True
    g = mpf("9.81")

    def tf(self, v): return v[0]

    def dynamics(self, t, x, u, v):
        p1, p2, p3, v1, v2, v3, phi, th = x
        at, phid, thd, psi = u
        cr, sr, cp, sp, cy, sy = dcos(phi), dsin(phi), dcos(th), dsin(th), dcos(psi), dsin(psi)
        R = [[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
             [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
             [-sp, cp * sr, cp * cr]]
        vec = [Du(0), Du(0), at]
        at_ = [R[i][0] * vec[0] + R[i][1] * vec[1] + R[i][2] * vec[2] for i in range(3)]
        g_ = [0, 0, -self.g]
        a = [g_[i] + at_[i] for i in range(3)]
        return [v1 + 0, v2 + 0, v3 + 0, a[0], a[1], a[2], phid + 0, thd + 0]

    def lagr(self, t, x, u, v):
        return mpf("1e-8") * (x[6] ** 2 + x[7] ** 2 + u[3] ** 2 + u[0] ** 2) + (mpf("1e2") * (u[3] - 0) ** 2)

    def may(self, x0, xf, v): return v[0]
    def path(self, t, x, u, v): return [dcos(x[7]) * dcos(x[6])]
    def boundary(self, x0, xf, v): return [x0[i] for i in range(8)] + [xf[i] for i in range(6)]


class Quadrotor12(Problem):                  # build-defined (DESIGN.md "quadrotor-12")
    name = "quadrotor12"
    n, m, nv, p, bc = 12, 4, 1, 1, 23
    freetf, lagrange, mayer = True, True, True
    g, Jx, Jy, Jz = mpf("9.81"), mpf("0.03"), mpf("0.03"), mpf("0.06")

    def tf(self, v): return v[0]

    def dynamics(self, t, x, u, v):
        phi, th, psi, w1, w2, w3 = x[6], x[7], x[8], x[9], x[10], x[11]
        at = u[0]
        cr, sr, cp, sp, cy, sy = dcos(phi), dsin(phi), dcos(th), dsin(th), dcos(psi), dsin(psi)
        R13 = cy * sp * cr + sy * sr
        R23 = sy * sp * cr - cy * sr
        R33 = cp * cr
        tp = sp / cp
        return [x[3] + 0, x[4] + 0, x[5] + 0,
                R13 * at, R23 * at, -self.g + R33 * at,
                w1 + sr * tp * w2 + cr * tp * w3,
                cr * w2 - sr * w3,
                (sr * w2 + cr * w3) / cp,
                ((self.Jy - self.Jz) * w2 * w3 + u[1]) / self.Jx,
                ((self.Jz - self.Jx) * w3 * w1 + u[2]) / self.Jy,
                ((self.Jx - self.Jy) * w1 * w2 + u[3]) / self.Jz]

    def lagr(self, t, x, u, v):
        return (mpf("1e-8") * (x[6] ** 2 + x[7] ** 2 + u[0] ** 2)
                + mpf("1e-2") * (u[1] ** 2 + u[2] ** 2 + u[3] ** 2) + mpf("1e2") * x[8] ** 2)

    def may(self, x0, xf, v): return v[0]
    def path(self, t, x, u, v): return [dcos(x[7]) * dcos(x[6])]

    def boundary(self, x0, xf, v):
        return [x0[i] for i in range(12)] + [xf[i] for i in range(8)] + [xf[9 + i] for i in range(3)]


class StagewiseScalar(Problem):              # test/ci/test_discretization_stagewise.jl:1-14
    name = "stagewise_scalar"
    n, m, nv, p, bc = 1, 1, 0, 0, 2
    lagrange = True

    def tf(self, v): return Du(1)
    def dynamics(self, t, x, u, v): return [u[0] + 0]
    def lagr(self, t, x, u, v): return u[0] ** 2
    def boundary(self, x0, xf, v): return [x0[0], xf[0]]


class EstimateRotationRate(Problem):         # autonomous_system.jl:46-87
    name = "estimate_rotation_rate"
    n, m, nv, p, bc = 2, 0, 1, 0, 2
    mayer = True

    def tf(self, v): return Du(1)
    def dynamics(self, t, x, u, v): return [v[0] * (-x[1]), v[0] * x[0]]
    def may(self, x0, xf, v): return (xf[0] - 0) ** 2 + (xf[1] - 1) ** 2 + mpf("0.01") * v[0] ** 2
    def boundary(self, x0, xf, v): return [x0[0] - 1, x0[1] - 0]


class EstimateInitialCondition(Problem):     # autonomous_system.jl:6-43
    name = "estimate_initial_condition"
    n, m, nv, p, bc = 2, 0, 2, 0, 2
    mayer = True

    def tf(self, v): return Du(mpf(math.pi / 2))      # Float64 pi/2, as the Julia literal
    def dynamics(self, t, x, u, v): return [-x[1], x[0] + 0]
    def may(self, x0, xf, v): return (xf[0] - 0) ** 2 + (xf[1] - 1) ** 2
    def boundary(self, x0, xf, v): return [x0[0] - v[0], x0[1] - v[1]]


class LeastSquaresConstraint(Problem):       # autonomous_system.jl:90-138
    name = "least_squares_with_constraint"
    n, m, nv, p, bc = 2, 0, 2, 1, 2
    lagrange, mayer = True, True

    def tf(self, v): return Du(1)
    def dynamics(self, t, x, u, v): return [-x[1], x[0] + 0]
    def lagr(self, t, x, u, v): return (t - mpf("0.5")) ** 2 * ((x[0] - mpf("0.7")) ** 2 + (x[1] - mpf("0.7")) ** 2)
    def may(self, x0, xf, v): return mpf("0.01") * (v[0] ** 2 + v[1] ** 2)
    def path(self, t, x, u, v): return [x[0] ** 2 + x[1] ** 2]
    def boundary(self, x0, xf, v): return [x0[0] - v[0], x0[1] - v[1]]


class DoubleIntegratorFreeT0Tf(Problem):     # double_integrator.jl:79-99 (+ tf - t0 as 5th boundary row)
    name = "double_integrator_freet0tf"
    n, m, nv, p, bc = 2, 1, 2, 0, 5
    freet0, freetf, mayer = True, True, True

    def t0(self, v): return v[0]
    def tf(self, v): return v[1]
    def dynamics(self, t, x, u, v): return [x[1] + 0, u[0] + 0]
    def may(self, x0, xf, v): return v[0]
    def boundary(self, x0, xf, v): return [x0[0], x0[1], xf[0], xf[1], v[1] - v[0]]


# --------------------------------------------------------------------------------------------
# schemes
# --------------------------------------------------------------------------------------------
def butcher(s):
    """Float64 tables exactly as the Julia constructors compute them (irk.jl:41-43,77-79,111-119)."""
    q3, q15 = math.sqrt(3), math.sqrt(15)
    if s == 1:
        return [[0.5]], [1.0], [0.5]
    if s == 2:
        return ([[0.25, (0.25 - q3 / 6)], [(0.25 + q3 / 6), 0.25]], [0.5, 0.5], [0.5 - q3 / 6, 0.5 + q3 / 6])
    return ([[(5.0 / 36.0), (2 / 9 - q15 / 15), (5 / 36 - q15 / 30)],
             [(5.0 / 36.0 + q15 / 24), (2.0 / 9.0), (5.0 / 36.0 - q15 / 24)],
             [(5 / 36 + q15 / 30), (2 / 9 + q15 / 15), (5.0 / 36.0)]],
            [5.0 / 18.0, 4.0 / 9.0, 5.0 / 18.0],
            [0.5 - 0.1 * q15, 0.5, 0.5 + 0.1 * q15])


SCHEMES = {
    "trapeze": ("trapeze", 0, False),
    "midpoint": ("midpoint", 0, False),
    "gauss_legendre_1": ("irk", 1, False),
    "gauss_legendre_2_constant_control": ("irk", 2, False),
    "gauss_legendre_3_constant_control": ("irk", 3, False),
    "gauss_legendre_2": ("irk", 2, True),
    "gauss_legendre_3": ("irk", 3, True),
    "euler": ("euler_explicit", 0, False),           # src/ode/euler.jl (explicit=true)
    "euler_implicit": ("euler_implicit", 0, False),  # src/ode/euler.jl (explicit=false)
}


class Docp:
    def __init__(self, prob, scheme, N=None, time_grid=None, control_steps=1):
        self.P = prob
        self.scheme = scheme
        self.cs = control_steps            # DOCPtime.control_steps (direct shooting): midpoint.jl:47-72, :99-116, :137-155
        assert control_steps == 1 or scheme == "midpoint"
        self.kind, self.s, self.stagewise = SCHEMES[scheme]
        if time_grid is None:
            self.N = N
            self.tau = [(i / N) for i in range(N + 1)]          # LinRange(0,1,N+1)
        else:
            self.N = len(time_grid) - 1
            t0, tf = time_grid[0], time_grid[-1]
            if t0 != 0 or tf != 1:
                self.tau = [(t - t0) / (tf - t0) for t in time_grid]
            else:
                self.tau = list(time_grid)
        P = prob
        n, m, nv = P.n, P.m, P.nv
        s = self.s
        if self.kind == "trapeze":
            self.blk, self.eqs, self.final_control = n + m, n, True
            self.nvar = self.N * self.blk + n + nv + m
        elif self.kind in ("midpoint", "euler_explicit", "euler_implicit"):
            self.blk, self.eqs, self.final_control = n + m * self.cs, n, False       # midpoint.jl:20
            self.nvar = self.N * self.blk + n + nv
        else:
            self.a, self.b, self.c = butcher(s)
            cu = m * s if self.stagewise else m
            self.cu = cu
            self.blk, self.eqs, self.final_control = n + cu + s * n, n * (1 + s), False
            self.nvar = self.N * self.blk + n + nv
        self.ncon = self.N * (self.eqs + P.p) + P.p + P.bc

    # 1-based getters, as in the reference
    def V(self, xu): return xu[self.nvar - self.P.nv:]
    def X(self, xu, i): return xu[(i - 1) * self.blk:(i - 1) * self.blk + self.P.n]

    def Ugen(self, xu, i, j=1):
        if not self.final_control and i == self.N + 1:
            i = self.N
        o = (i - 1) * self.blk + self.P.n + (j - 1) * self.P.m        # common.jl:140-155
        return xu[o:o + self.P.m]

    def Ustage(self, xu, i, j):
        o = (i - 1) * self.blk + self.P.n + (j - 1) * self.P.m
        return xu[o:o + self.P.m]

    def K(self, xu, i, j):
        o = (i - 1) * self.blk + self.P.n + self.cu + (j - 1) * self.P.n
        return xu[o:o + self.P.n]

    def Uctl(self, xu, i):
        """control seen by path constraints (stagewise: b-weighted average, irk_stagewise.jl:197-205)"""
        if self.stagewise:
            if i == self.N + 1:
                i = self.N
            ui = [mpf(self.b[0]) * a for a in self.Ustage(xu, i, 1)]
            for j in range(2, self.s + 1):
                ui = [a + mpf(self.b[j - 1]) * b for a, b in zip(ui, self.Ustage(xu, i, j))]
            return ui
        if self.kind == "euler_explicit":                       # euler.jl:59-72
            return self.Ugen(xu, i)
        if self.kind == "euler_implicit":                       # u(t_1) = U_1, u(t_i) = U_{i-1}
            j = (2 if i == 1 else i) - 1
            o = (j - 1) * self.blk + self.P.n
            return xu[o:o + self.P.m]
        return self.Ugen(xu, i)

    def grid(self, xu):
        v = self.V(xu)
        t0, tf = self.P.t0(v), self.P.tf(v)
        return [t0 + mpf(tau) * (tf - t0) for tau in self.tau]

    def constraints(self, xu):
        P, N, n = self.P, self.N, self.P.n
        T = self.grid(xu)
        v = self.V(xu)
        c = []
        if self.kind == "trapeze":
            work = [P.dynamics(T[i - 1], self.X(xu, i), self.Ugen(xu, i), v) for i in range(1, N + 2)]
        elif self.kind == "midpoint":
            work = []
            for i in range(1, N + 1):
                ts = mpf("0.5") * (T[i - 1] + T[i])
                xs = [mpf("0.5") * (a + b) for a, b in zip(self.X(xu, i), self.X(xu, i + 1))]
                work.append([P.dynamics(ts, xs, self.Ugen(xu, i, j), v) for j in range(1, self.cs + 1)])
        elif self.kind in ("euler_explicit", "euler_implicit"):       # euler.jl:79-105
            work = []
            for i in range(1, N + 1):
                idx = i if self.kind == "euler_explicit" else i + 1
                work.append(P.dynamics(T[idx - 1], self.X(xu, idx), self.Uctl(xu, idx), v))
        for i in range(1, N + 1):
            ti, tip1 = T[i - 1], T[i]
            xi, xip1 = self.X(xu, i), self.X(xu, i + 1)
            if self.kind == "trapeze":
                hh = mpf("0.5") * (tip1 - ti)
                c += [xip1[k] - (xi[k] + hh * (work[i - 1][k] + work[i][k])) for k in range(n)]
            elif self.kind == "midpoint":                                 # midpoint.jl:124-156
                hi = (tip1 - ti) / self.cs
                x_next = list(xi)
                for j in range(self.cs):
                    x_next = [a + hi * f for a, f in zip(x_next, work[i - 1][j])]
                c += [xip1[k] - x_next[k] for k in range(n)]
            elif self.kind in ("euler_explicit", "euler_implicit"):
                hi = tip1 - ti
                c += [xip1[k] - (xi[k] + hi * work[i - 1][k]) for k in range(n)]
            else:
                hi = tip1 - ti
                sumbk = None
                stage_rows = []
                for j in range(1, self.s + 1):
                    tij = ti + mpf(self.c[j - 1]) * hi
                    kij = self.K(xu, i, j)
                    uij = self.Ustage(xu, i, j) if self.stagewise else self.Ugen(xu, i)
                    if j == 1:
                        sumbk = [mpf(self.b[0]) * a for a in kij]
                    else:
                        sumbk = [a + mpf(self.b[j - 1]) * b for a, b in zip(sumbk, kij)]
                    xij = list(xi)
                    for l in range(1, self.s + 1):
                        kil = self.K(xu, i, l)
                        xij = [a + hi * mpf(self.a[j - 1][l - 1]) * b for a, b in zip(xij, kil)]
                    f = P.dynamics(tij, xij, uij, v)
                    stage_rows += [kij[k] - f[k] for k in range(n)]
                c += [xip1[k] - (xi[k] + hi * sumbk[k]) for k in range(n)]
                c += stage_rows
            if P.p > 0:
                c += P.path(ti, xi, self.Uctl(xu, i), v)
        if P.p > 0:
            c += P.path(T[N], self.X(xu, N + 1), self.Uctl(xu, N + 1), v)
        if P.bc > 0:
            c += P.boundary(self.X(xu, 1), self.X(xu, N + 1), v)
        assert len(c) == self.ncon
        return c

    def objective(self, xu):
        P, N = self.P, self.N
        T = self.grid(xu)
        v = self.V(xu)
        obj = Du(0)
        if P.mayer:
            obj = obj + P.may(self.X(xu, 1), self.X(xu, N + 1), v)
        if P.lagrange:
            val = Du(0)
            if self.kind == "trapeze":
                val = val + (T[1] - T[0]) / 2 * P.lagr(T[0], self.X(xu, 1), self.Ugen(xu, 1), v)
                for i in range(2, N + 1):
                    val = val + (T[i] - T[i - 2]) / 2 * P.lagr(T[i - 1], self.X(xu, i), self.Ugen(xu, i), v)
                val = val + (T[N] - T[N - 1]) / 2 * P.lagr(T[N], self.X(xu, N + 1), self.Ugen(xu, N + 1), v)
            elif self.kind == "midpoint" and self.cs > 1:                 # midpoint.jl:99-116
                for i in range(1, N + 1):
                    hi = (T[i] - T[i - 1]) / self.cs
                    xs = [mpf("0.5") * (a + b) for a, b in zip(self.X(xu, i), self.X(xu, i + 1))]
                    for j in range(1, self.cs + 1):
                        tij = T[i - 1] + (j - mpf("0.5")) * hi
                        val = val + hi * P.lagr(tij, xs, self.Ugen(xu, i, j), v)
            elif self.kind == "midpoint":
                for i in range(1, N + 1):
                    hi = T[i] - T[i - 1]
                    ts = mpf("0.5") * (T[i - 1] + T[i])
                    xs = [mpf("0.5") * (a + b) for a, b in zip(self.X(xu, i), self.X(xu, i + 1))]
                    val = val + hi * P.lagr(ts, xs, self.Ugen(xu, i), v)
            elif self.kind in ("euler_explicit", "euler_implicit"):      # euler.jl:112-134
                for i in range(1, N + 1):
                    idx = i if self.kind == "euler_explicit" else i + 1
                    hi = T[i] - T[i - 1]
                    val = val + hi * P.lagr(T[idx - 1], self.X(xu, idx), self.Uctl(xu, idx), v)
            else:
                for i in range(1, N + 1):
                    ti = T[i - 1]
                    hi = T[i] - ti
                    xi = self.X(xu, i)
                    loc = Du(0)
                    for j in range(1, self.s + 1):
                        tij = ti + mpf(self.c[j - 1]) * hi
                        uij = self.Ustage(xu, i, j) if self.stagewise else self.Ugen(xu, i)
                        xij = list(xi)
                        for l in range(1, self.s + 1):
                            xij = [a + hi * mpf(self.a[j - 1][l - 1]) * b for a, b in zip(xij, self.K(xu, i, l))]
                        loc = loc + mpf(self.b[j - 1]) * P.lagr(tij, xij, uij, v)
                    val = val + hi * loc
            obj = obj + val
        return obj


# --------------------------------------------------------------------------------------------
# deterministic inputs (closed form) -- plain Python floats
# --------------------------------------------------------------------------------------------
def fill_inputs(d):
    """Returns xu as a list of Python floats.  State/control/variable by closed form in tau, stage variables
    K_i^j = f(t_ij, x_i, u_ij) * (1 + small) so residuals are small but nonzero."""
    P, N = d.P, d.N
    xu = [0.0] * d.nvar
    name = P.name

    def state(tau):
        if name.startswith("goddard"):
            return [1 + 0.01 * tau, 0.1 * math.sin(math.pi * tau), 1 - 0.4 * tau]
        if name in ("double_integrator_path", "double_integrator_freet0tf"):
            return [tau * tau, 2 * tau]
        if name == "quadrotor":
            return [0.01 * tau + 0.01 * math.sin(3 * tau), 5 * tau, 2.5 + 0.01 * math.sin(5 * tau),
                    0.1 * math.sin(2 * tau), 5 + 0.1 * math.cos(tau), 0.02 * math.sin(4 * tau),
                    0.1 * math.sin(6 * tau), 0.15 * math.cos(5 * tau)]
        if name == "quadrotor12":
            return [0.01 * tau + 0.01 * math.sin(3 * tau), 5 * tau, 2.5 + 0.01 * math.sin(5 * tau),
                    0.1 * math.sin(2 * tau), 5 + 0.1 * math.cos(tau), 0.02 * math.sin(4 * tau),
                    0.1 * math.sin(6 * tau), 0.15 * math.cos(5 * tau), 0.2 * math.sin(2 * tau + 0.3),
                    0.3 * math.sin(7 * tau), 0.25 * math.cos(3 * tau), 0.1 * math.sin(9 * tau + 1)]
        if name == "stagewise_scalar":
            return [tau * tau + 0.01 * math.sin(5 * tau)]
        return [math.cos(1.3 * tau) + 0.1, math.sin(1.3 * tau) - 0.05]      # oscillators

    def control(tau, j):
        if name.startswith("goddard"):
            return [0.5 + 0.5 * math.cos(7 * tau + j)]
        if name in ("double_integrator_path", "double_integrator_freet0tf", "stagewise_scalar"):
            return [2 * math.cos(3 * tau + 0.1 * j)]
        if name == "quadrotor":
            return [10 + math.sin(4 * tau + j), 0.3 * math.cos(3 * tau + j), 0.2 * math.sin(5 * tau + j), 0.05 * math.cos(2 * tau + j)]
        if name == "quadrotor12":
            return [10 + math.sin(4 * tau + j), 0.03 * math.cos(3 * tau + j), 0.02 * math.sin(5 * tau + j), 0.01 * math.cos(2 * tau + j)]
        return []

    var = {"goddard": [0.2], "goddard_all": [0.2], "quadrotor": [1.0], "quadrotor12": [1.0],
           "estimate_rotation_rate": [1.4], "estimate_initial_condition": [0.9, 0.1],
           "least_squares_with_constraint": [0.8, 0.2], "double_integrator_freet0tf": [0.3, 2.1]}.get(name, [])
    for k in range(P.nv):
        xu[d.nvar - P.nv + k] = var[k]
    for i in range(1, N + 2):
        tau = d.tau[i - 1]
        o = (i - 1) * d.blk
        xu[o:o + P.n] = state(tau)
        if P.m > 0 and d.kind != "irk" and (i <= N or d.final_control):
            xu[o + P.n:o + P.n + P.m] = control(tau, 0)
            for j in range(2, getattr(d, "cs", 1) + 1):           # further controls of the step (control_steps > 1)
                tj = tau + (j - 1) / d.cs * (d.tau[i] - tau)
                xu[o + P.n + (j - 1) * P.m:o + P.n + j * P.m] = control(tj, j - 1)
    if d.kind == "irk":
        # controls, then stage variables from a float evaluation of the dynamics at the stage points
        Du.NV = 0
        vv = [Du(x) for x in xu[d.nvar - P.nv:]]
        t0, tf = float(P.t0(vv).v), float(P.tf(vv).v)
        for i in range(1, N + 1):
            o = (i - 1) * d.blk
            ti = t0 + d.tau[i - 1] * (tf - t0)
            hi = (t0 + d.tau[i] * (tf - t0)) - ti
            for j in range(1, d.s + 1):
                tau_ij = d.tau[i - 1] + d.c[j - 1] * (d.tau[i] - d.tau[i - 1])
                if P.m > 0:
                    if d.stagewise:
                        xu[o + P.n + (j - 1) * P.m:o + P.n + j * P.m] = control(tau_ij, j)
                    elif j == 1:
                        xu[o + P.n:o + P.n + P.m] = control(d.tau[i - 1], 0)
            for j in range(1, d.s + 1):
                uo = o + P.n + ((j - 1) * P.m if d.stagewise else 0)
                f = P.dynamics(Du(ti + d.c[j - 1] * hi), [Du(x) for x in xu[o:o + P.n]], [Du(x) for x in xu[uo:uo + P.m]], vv)
                ko = o + P.n + d.cu + (j - 1) * P.n
                xu[ko:ko + P.n] = [float(fk.v) * (1 + 0.01 * math.sin(i + j + k)) + 0.001 * math.cos(i * j + k)
                                   for k, fk in enumerate(f)]
    return [float(x) for x in xu]


def hexf(x):
    return float(x).hex()


def run_case(tag, prob, scheme, N=None, time_grid=None, xu=None, control_steps=1):
    d = Docp(prob, scheme, N=N, time_grid=time_grid, control_steps=control_steps)
    if xu is None:
        xu = fill_inputs(d)
    assert len(xu) == d.nvar
    Du.NV = d.nvar
    z = []
    for j, x in enumerate(xu):
        der = [mpf(0)] * d.nvar
        der[j] = mpf(1)
        z.append(Du(mpf(x), der))
    c = d.constraints(z)
    obj = d.objective(z)
    jac = []
    for r, cr in enumerate(c):
        cr = Du.lift(cr)
        for j in range(d.nvar):
            if cr.d[j] != 0:
                jac.append([r, j, hexf(cr.d[j])])
    out = {
        "tag": tag, "problem": prob.name, "scheme": scheme,
        "grid_size": d.N, "time_grid": [hexf(t) for t in time_grid] if time_grid is not None else None,
        "dims": {"n": prob.n, "m": prob.m, "nv": prob.nv, "path": prob.p, "boundary": prob.bc,
                 "nvar": d.nvar, "ncon": d.ncon, "step_variables_block": d.blk, "state_stage_eqs_block": d.eqs},
        "xu": [hexf(x) for x in xu],
        **({"control_steps": control_steps} if control_steps != 1 else {}),
        "c": [hexf(Du.lift(x).v) for x in c],
        "objective": hexf(obj.v),
        "gradient": [[j, hexf(g)] for j, g in enumerate(obj.d) if g != 0],
        "jac_nonzeros": jac,
        "note": "values are correctly rounded doubles (C99 hex) of a 50-digit mpmath evaluation; rows/cols 0-based",
    }
    path = os.path.join(HERE, tag + ".json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(f"{tag}: nvar={d.nvar} ncon={d.ncon} nnz(true)={len(jac)} -> {os.path.relpath(path)}")


def exact_stagewise_xu(d):
    """x = t^2, u = 2t, K = 2 t_ij on the given grid (test/ci/test_discretization_stagewise.jl:20-42)."""
    xu = [0.0] * d.nvar
    T = d.tau                      # grid [0,.2,.6,1] is already normalized and t in [0,1]
    for i in range(1, d.N + 2):
        xu[(i - 1) * d.blk] = T[i - 1] ** 2
    for i in range(1, d.N + 1):
        ti = T[i - 1]
        hi = T[i] - ti
        for j in range(1, d.s + 1):
            tij = ti + d.c[j - 1] * hi
            xu[(i - 1) * d.blk + 1 + (j - 1)] = 2 * tij
            xu[(i - 1) * d.blk + 1 + d.s + (j - 1)] = 2 * tij
    return xu


def main_control_steps():
    """control_steps > 1 (direct shooting layout, midpoint scheme): `python gen_golden.py cs` writes only these"""
    run_case("cs2_goddard_midpoint_N4", Goddard(), "midpoint", N=4, control_steps=2)
    run_case("cs3_goddard_midpoint_nonuniform", Goddard(), "midpoint", time_grid=[0.0, 0.1, 0.45, 1.0], control_steps=3)
    run_case("cs2_dip_midpoint_N4", DoubleIntegratorPath(), "midpoint", N=4, control_steps=2)
    run_case("cs3_dip_midpoint_nonuniform", DoubleIntegratorPath(), "midpoint", time_grid=[0.0, 0.3, 0.5, 0.6, 1.0], control_steps=3)
    run_case("cs2_goddard_all_midpoint_N3", GoddardAll(), "midpoint", N=3, control_steps=2)
    run_case("cs3_freet0tf_midpoint_N3", DoubleIntegratorFreeT0Tf(), "midpoint", N=3, control_steps=3)
    run_case("cs2_quadrotor_midpoint_N2", Quadrotor8(), "midpoint", N=2, control_steps=2)


def main():
    if sys.argv[1:] == ["cs"]:
        return main_control_steps()
    g = [0.0, 0.2, 0.6, 1.0]
    # G1: the reference's exact-feasible trajectory (c == 0, objective == 4/3)
    for sch in ("gauss_legendre_2", "gauss_legendre_3"):
        d = Docp(StagewiseScalar(), sch, time_grid=g)
        run_case(f"scalar_{sch}_exact", StagewiseScalar(), sch, time_grid=g, xu=exact_stagewise_xu(d))
        run_case(f"scalar_{sch}_perturbed", StagewiseScalar(), sch, time_grid=g)
    run_case("goddard_midpoint_N4", Goddard(), "midpoint", N=4)
    run_case("goddard_trapeze_N4", Goddard(), "trapeze", N=4)
    run_case("goddard_gl2_N4", Goddard(), "gauss_legendre_2", N=4)
    run_case("goddard_gl3_N3", Goddard(), "gauss_legendre_3", N=3)
    run_case("goddard_gl2cc_N3", Goddard(), "gauss_legendre_2_constant_control", N=3)
    run_case("goddard_gl3cc_nonuniform", Goddard(), "gauss_legendre_3_constant_control", time_grid=[0.0, 0.1, 0.45, 1.0])
    run_case("goddard_gl1_N3", Goddard(), "gauss_legendre_1", N=3)
    run_case("goddard_all_trapeze_N4", GoddardAll(), "trapeze", N=4)
    run_case("goddard_all_gl2_N3", GoddardAll(), "gauss_legendre_2", N=3)
    run_case("goddard_all_midpoint_nonuniform", GoddardAll(), "midpoint", time_grid=[0.0, 0.3, 0.5, 0.6, 1.0])
    run_case("dip_midpoint_N4", DoubleIntegratorPath(), "midpoint", N=4)
    run_case("dip_trapeze_N3", DoubleIntegratorPath(), "trapeze", N=3)
    run_case("dip_gl3_N2", DoubleIntegratorPath(), "gauss_legendre_3", N=2)
    run_case("quadrotor_gl3_N2", Quadrotor8(), "gauss_legendre_3", N=2)
    run_case("quadrotor_midpoint_N3", Quadrotor8(), "midpoint", N=3)
    run_case("quadrotor12_gl3_N2", Quadrotor12(), "gauss_legendre_3", N=2)
    run_case("quadrotor12_trapeze_N2", Quadrotor12(), "trapeze", N=2)
    run_case("rotrate_gl3cc_N2", EstimateRotationRate(), "gauss_legendre_3_constant_control", N=2)
    run_case("rotrate_midpoint_N3", EstimateRotationRate(), "midpoint", N=3)
    run_case("initcond_trapeze_N3", EstimateInitialCondition(), "trapeze", N=3)
    run_case("lsq_trapeze_N3", LeastSquaresConstraint(), "trapeze", N=3)
    run_case("lsq_gl2_N3", LeastSquaresConstraint(), "gauss_legendre_2", N=3)
    run_case("freet0tf_midpoint_N3", DoubleIntegratorFreeT0Tf(), "midpoint", N=3)
    run_case("freet0tf_gl2_N2", DoubleIntegratorFreeT0Tf(), "gauss_legendre_2", N=2)
    run_case("freet0tf_trapeze_N3", DoubleIntegratorFreeT0Tf(), "trapeze", N=3)
    run_case("goddard_euler_N4", Goddard(), "euler", N=4)
    run_case("goddard_all_euler_implicit_N4", GoddardAll(), "euler_implicit", N=4)
    run_case("goddard_all_euler_nonuniform", GoddardAll(), "euler", time_grid=[0.0, 0.3, 0.5, 0.6, 1.0])
    run_case("dip_euler_implicit_N3", DoubleIntegratorPath(), "euler_implicit", N=3)
    run_case("quadrotor_euler_N3", Quadrotor8(), "euler", N=3)
    run_case("quadrotor_euler_implicit_N2", Quadrotor8(), "euler_implicit", N=2)
    run_case("lsq_euler_implicit_N3", LeastSquaresConstraint(), "euler_implicit", N=3)
    run_case("freet0tf_euler_N3", DoubleIntegratorFreeT0Tf(), "euler", N=3)
    run_case("freet0tf_euler_implicit_N3", DoubleIntegratorFreeT0Tf(), "euler_implicit", N=3)
    main_control_steps()


if __name__ == "__main__":
    main()
