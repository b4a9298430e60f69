"""The rest of the reference's problem folder (/root/reference/test/problems/*.jl) restated as run-time OCPs -- the problems round 3
left out because the expression grammar lacked what they use (VERDICT r03 "missing" 3): action, algal_bacterial, bioreactor (1 day,
N days), parametric, schlogl, swimmer / swimmer2 -- plus goddard_all with its dynamics written F0 + u F1, the form whose traced
Jacobian pattern has the archived 28011 / 280011 entries (test/archives/AD_backend.md:63).

Expressions follow the Julia code OPERATOR BY OPERATOR (an `:optimized` pattern is the operator-level dependence of the code as
written); `constants` holds numbers and ALIASES (named sub-expressions: the `aux = ...` lines of the @def blocks and the helper
functions phi / rho / mu, light, growth, p_relu of the problem files).  Each entry: (register_ocp kwargs, catalogued objective or
None, init or None).  The same kwargs feed tests/expr_mp.py::ExprMp, the independent mpmath restatement."""
import ctdirect_jl_amd as ct

INF = float("inf")

# ---- test/problems/algal_bacterial.jl:3-52 --------------------------------------------------------------------------
_ALGAL = dict(
    dynamics=["u2*(s_in - x1) - phi*x2/gamma",                     # s   :27
              "((1 - u1)*phi - u2)*x2",                            # e   :28
              "u1*beta*phi*x2 - rho*x5 - u2*x3",                   # v   :29
              "rho - mu*x4",                                       # q   :30
              "(mu - u2)*x5",                                      # c   :31
              "u2*x5"],                                            # obj = d*c :32
    m=2, mayer="xf_6", maximize=True, t0=0.0, tf=20.0,
    constants=dict(s_in=0.5, beta=23e-3, gamma=0.44, dmax=1.5, phimax=6.48, ks=0.09, rhomax=27.3e-3, kv=0.57e-3, mumax=1.0211,
                   qmin=2.7628e-3,
                   phi="phimax*x1/(ks + x1)", rho="rhomax*x3/(kv + x3)", mu="mumax*(1 - qmin/x4)"),      # :17-19
    boundary=[f"x0_{k}" for k in range(1, 7)],
    boundary_bounds=([0.1629, 0.0487, 0.0003, 0.0177, 0.035, 0.0],) * 2,                                   # x(t0) == x0 :22,40
    state_box=([0, 0, 0, 2.7628e-3, 0, 0], [INF] * 6),                                                     # :41
    control_box=([0, 0], [1, 1.5]))                                                                        # :42

# ---- test/problems/action.jl:3-41 -----------------------------------------------------------------------------------
_ACTION = dict(
    dynamics=["u1", "u2"], m=2, t0=0.0, tf=50.0,
    lagrange="sqrt(sqrt((unorm2*fnorm2)^2 + eps^2)) - dotuf",                                               # asqrt(unorm2*fnorm2) - dotuf :12,19
    constants=dict(eps=1e-1, f1="x1 - x1^3 - 10*x1*x2^2", f2="-(1 - x1^2)*x2",                             # :7
                   unorm2="u1^2 + u2^2", fnorm2="f1^2 + f2^2", dotuf="u1*f1 + u2*f2"),                      # :16-18
    boundary=["x0_1", "x0_2", "xf_1", "xf_2"], boundary_bounds=([-1, 0, 1, 0],) * 2)                        # :26-27

# ---- test/problems/bioreactor.jl:9-20 (growth, light), :23-61 (1 day), :64-107 (N days) -------------------------------
_BIO_CONST = dict(beta=1, c=2, gamma=1, Ks=0.05, mu2m=0.1, mubar=1, r=0.005, halfperiod=5, pi=3.141592653589793,
                  days="t/(halfperiod*2)", tau="(days - floor(days))*2*pi", light="max(0, sin(tau))^2",     # :15-20
                  mu="light*mubar", mu2="mu2m*x2/(x2 + Ks)")                                                # :10-12, :45-46
_BIO_DYN = ["mu*x1/(1 + x1) - (r + u1)*x1", "-mu2*x3 + u1*beta*(gamma*x1 - x2)", "(mu2 - u1*beta)*x3"]      # :52-56
_BIO1 = dict(dynamics=_BIO_DYN, m=1, t0=0.0, tf=10.0, lagrange="mu2*x3/(beta + c)", maximize=True, constants=_BIO_CONST,
             boundary=["x0_1", "x0_3", "x0_1 - xf_1", "x0_2 - xf_2", "x0_3 - xf_3"],                         # :49-51
             boundary_bounds=([1, 1, 0, 0, 0], [INF, INF, 0, 0, 0]),
             state_box=([0, 0, 0.001], [INF] * 3), control_box=([0], [1]))
_BION = dict(dynamics=_BIO_DYN, m=1, t0=0.0, tf=300.0, lagrange="mu2*x3/(beta + c)", maximize=True, constants=_BIO_CONST,
             boundary=["x0_1", "x0_2", "x0_3"], boundary_bounds=([0.05, 0.5, 0.5], [0.25, 5, 3]),           # :88-90
             state_box=([0, 0, 0.001], [INF] * 3), control_box=([0], [1]))

# ---- test/problems/parametric.jl:3-33 (rho = 1) ----------------------------------------------------------------------
_PARAM = dict(
    dynamics=["v1*(u1 + 2)", "(T - v1)*u2"], m=2, nv=1, t0=0.0, tf=1.0,
    lagrange="-rho*(v1*m1^2 + (T - v1)*m2^2)", mayer="-(xf_2 - 2)^3",                                        # :23 (the integral enters with a minus)
    constants=dict(rho=1, mu=10, T=2, m1="log(abs(1 + exp(mu*(1 - x1))))/mu", m2="log(abs(1 + exp(mu*(1 - x2))))/mu"),   # :6-8
    boundary=["x0_1", "x0_2", "xf_1"], boundary_bounds=([0, 1, 1],) * 2,
    control_box=([-1, -1], [1, 1]), variable_box=([0], [2]))

# ---- test/problems/schlogl.jl:3-39 (the third bracket reads u0 where u2 is meant: restated as written) ---------------
_SCHLOGL = dict(
    dynamics=["u1 - u2 + u3 - u4"], m=4, nv=1, t0=0.0, itf=0,
    lagrange="u1*log(abs(u1/k0)) - (u1 - k0) + u2*log(abs(u2/(k1*x1))) - (u2 - k1*x1) + u3*log(abs(u3/(k2*x1^2))) - (u1 - k2*x1^2)"
             " + u4*log(abs(u4/(k3*x1^3))) - (u4 - k3*x1^3)",
    constants=dict(k0=6, k1=11, k2=6, k3=1), boundary=["x0_1", "xf_1"], boundary_bounds=([1, 2],) * 2,
    state_box=([0.5], [INF]), control_box=([0.1] * 4, [INF] * 4), variable_box=([0.02], [1.0]))

# ---- test/problems/swimmer.jl:5-153 (swimmer2.jl is the same problem with named state components) ---------------------
_SW = dict(th="x3", b1="x4", b3="x5", a1="u1", a2="u2")
_SW["aux"] = ("543 + 186*cos(b1) + 37*cos(2*b1) + 12*cos(b1 - 2*b3) + 30*cos(b1 - b3) + 2*cos(2*(b1 - b3)) + 12*cos(2*b1 - b3) + 186*cos(b3)"
              " + 37*cos(2*b3) - 6*cos(b1 + b3) - 3*cos(2*(b1 + b3)) - 6*cos(2*b1 + b3) - 6*cos(b1 + 2*b3)")
_SW["g11"] = ("(-42*sin(b1 - th) - 2*sin(2*b1 - th) - 24*sin(th) - 300*sin(b1 + th) - 12*sin(2*b1 + th) - 6*sin(b1 - th - 2*b3)"
              " - sin(2*b1 - th - 2*b3) + 4*sin(th - 2*b3) - 12*sin(b1 + th - 2*b3) - sin(2*b1 + th - 2*b3) + 18*sin(b1 - th - b3)"
              " + 8*sin(th - b3) - 54*sin(b1 + th - b3) - 2*sin(2*b1 + th - b3) - 18*sin(b1 - th + b3) - 38*sin(th + b3) - 90*sin(b1 + th + b3)"
              " - 6*sin(b1 - th + 2*b3) - 18*sin(th + 2*b3) - 30*sin(b1 + th + 2*b3)) / (4*aux)")
_SW["g12"] = ("(-42*cos(b1 - th) - 2*cos(2*b1 - th) + 24*cos(th) + 300*cos(b1 + th) + 12*cos(2*b1 + th) - 6*cos(b1 - th - 2*b3)"
              " - cos(2*b1 - th - 2*b3) - 4*cos(th - 2*b3) + 12*cos(b1 + th - 2*b3) + cos(2*b1 + th - 2*b3) + 18*cos(b1 - th - b3)"
              " - 8*cos(th - b3) + 54*cos(b1 + th - b3) + 2*cos(2*b1 + th - b3) - 18*cos(b1 - th + b3) + 38*cos(th + b3) + 90*cos(b1 + th + b3)"
              " - 6*cos(b1 - th + 2*b3) + 18*cos(th + 2*b3) + 30*cos(b1 + th + 2*b3)) / (4*aux)")
_SW["g13"] = ("-(105 + 186*cos(b1) + 2*cos(2*b1) + 12*cos(b1 - 2*b3) + 30*cos(b1 - b3) + cos(2*(b1 - b3)) - 4*cos(2*b3) - 6*cos(b1 + b3)"
              " - 6*cos(b1 + 2*b3)) / (2*aux)")
_SW["g21"] = ("(8*sin(b1 - th) + 4*sin(2*b1 - th) + 24*sin(th) + 38*sin(b1 + th) + 18*sin(2*b1 + th) - 2*sin(b1 - th - 2*b3)"
              " - sin(2*b1 - th - 2*b3) - 2*sin(th - 2*b3) - sin(2*b1 + th - 2*b3) - 54*sin(b1 - th - b3) - 12*sin(2*b1 - th - b3)"
              " - 42*sin(th - b3) + 18*sin(b1 + th - b3) - 6*sin(2*b1 + th - b3) + 18*sin(b1 - th + b3) + 6*sin(2*b1 - th + b3)"
              " + 300*sin(th + b3) + 90*sin(b1 + th + b3) + 30*sin(2*b1 + th + b3) + 12*sin(th + 2*b3)) / (4*aux)")
_SW["g22"] = ("(8*cos(b1 - th) + 4*cos(2*b1 - th) - 24*cos(th) - 38*cos(b1 + th) - 18*cos(2*b1 + th) - 2*cos(b1 - th - 2*b3)"
              " - cos(2*b1 - th - 2*b3) + 2*cos(th - 2*b3) + cos(2*b1 + th - 2*b3) - 54*cos(b1 - th - b3) - 12*cos(2*b1 - th - b3)"
              " + 42*cos(th - b3) - 18*cos(b1 + th - b3) + 6*cos(2*b1 + th - b3) + 18*cos(b1 - th + b3) + 6*cos(2*b1 - th + b3)"
              " - 300*cos(th + b3) - 90*cos(b1 + th + b3) - 30*cos(2*b1 + th + b3) - 12*cos(th + 2*b3)) / (4*aux)")
_SW["g23"] = ("-(105 - 4*cos(2*b1) + 30*cos(b1 - b3) + cos(2*(b1 - b3)) + 12*cos(2*b1 - b3) + 186*cos(b3) + 2*cos(2*b3) - 6*cos(b1 + b3)"
              " - 6*cos(2*b1 + b3)) / (2*aux)")
_SWIMMER = dict(
    dynamics=["g11*a1 + g21*a2", "g12*a1 + g22*a2", "g13*a1 + g23*a2", "a1", "a2"], m=2, t0=0.0, tf=25.0,   # :144
    mayer="xf_1", maximize=True, constants=_SW,
    boundary=["x0_1", "x0_2", "x0_3", "x0_4", "xf_2"],                                                       # :17-23
    boundary_bounds=([0, 0, -3.15, 0, 0], [0, 0, 0, INF, 0]),
    state_box=([-INF, -INF, -3.15, -1.5, -1.5], [INF, INF, 3.15, 1.5, 1.5]), control_box=([-1, -1], [1, 1]))

# ---- goddard_all (test/problems/goddard.jl:87-158) with its dynamics in the form of goddard.jl:7-15,44: F0(x) + u F1(x).  The
# traced pattern of THIS form has the (r-row, U_i) and (r-row, U_{i+1}) entries of `u * 0`: 28 Jacobian entries per trapeze step,
# the count the archive publishes (test/archives/AD_backend.md:63: 28011 / 280011); the f! of the current file gives 26.
_GODDARD_ALL_F0F1 = dict(
    dynamics=["x2 + u1*0", "(-Cd*x2^2*exp(-beta*(x1 - 1))/x3 - 1/x1^2) + u1*(Tmax/x3)", "0 + u1*(-b*Tmax)"], m=1, nv=1, itf=0,
    mayer="xf_1", maximize=True, constants=dict(Cd=310, beta=500, b=2, Tmax=3.5),
    path=["x2", "u1", "x1 + x2 + x3 + u1 + v1"], path_bounds=([-INF, -INF, 0], [0.1, 1, INF]),
    boundary=["x0_1", "x0_2", "x0_3", "xf_3"], boundary_bounds=([1, 0, 1, 0.6],) * 2,
    state_box=([1, 0, 0], [INF, INF, 1]), control_box=([0], [INF]), variable_box=([0.01], [INF]))

FOLDER = {
    "algal_bacterial": (_ALGAL, 5.45, None),                                  # obj :52 (5.4522 in test/archives/jump_ctdirect.md:52)
    "action": (_ACTION, None, "path"),
    "bioreactor_1day": (_BIO1, 0.614134, None),                               # :60
    "bioreactor_Ndays": (_BION, 19.0745, dict(state=[50, 50, 50])),           # :99-105
    "parametric": (_PARAM, -3.36e-1, None),                                   # :28
    "schlogl": (_SCHLOGL, None, None),
    "swimmer": (_SWIMMER, 0.984273, None),                                    # :147
    "swimmer2": (_SWIMMER, 0.984273, None),
    "goddard_all_f0f1": (_GODDARD_ALL_F0F1, 1.01257, dict(state=[1.01, 0.05, 0.8])),
}

_registered = {}


def folder(name):
    """(run-time problem name, catalogued objective, init) -- registers the problem once"""
    if name not in _registered:
        _registered[name] = ct.register_ocp(name + "_pf", **FOLDER[name][0])
    return _registered[name], FOLDER[name][1], FOLDER[name][2]


def mp_problem(name):
    """the independent 50-digit restatement of the same text (tests/expr_mp.py)"""
    from expr_mp import ExprMp
    return ExprMp(name + "_pf", **FOLDER[name][0])
