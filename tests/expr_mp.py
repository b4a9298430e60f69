"""TEST INFRASTRUCTURE ONLY -- an independent evaluator of the engine's OCP expression grammar (include/ctdirect_hip.h, ctd_ocp_def)
on the 50-digit dual numbers of tests/golden/gen_golden.py (first order `Du`) and gen_golden_hess.py (second order `Du2`).

`ExprMp(**register_ocp kwargs)` turns the keyword arguments a test hands to `ct.register_ocp` into a `gg.Problem`, so any run-time
OCP gets an on-the-fly mpmath restatement of c(x), the dense Jacobian, the objective, its gradient and the Hessian of the
Lagrangian (tests/test_gpu_jit.py::_mp_reference) WITHOUT going through the engine's parser: the text is tokenised and parsed
here, in Python, by a separate recursive descent.  Derivative conventions restated from the reference's AD stack: max / min take
the derivative of the selected operand, at a tie max follows its second argument and min its first (DiffRules / ForwardDiff);
floor has derivative zero; abs' = sign."""
import re

import gen_golden as gg
from mpmath import mp, mpf

_TOKEN = re.compile(r"\s*(?:(\d+\.?\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?)|([A-Za-z_][A-Za-z_0-9]*)|(.))")


def _unary(x, f0, f1, f2):
    """f(x) for whichever dual type gen_golden currently uses (tests swap the second-order number in)"""
    x = gg.Du.lift(x)
    if hasattr(x, "chain"):
        return x.chain(f0, f1, f2)
    return gg.Du(f0, [a * f1 for a in x.d])


def _sgn(v):
    return mpf(1) if v > 0 else (mpf(-1) if v < 0 else mpf(0))


def _fn(name, x):
    x = gg.Du.lift(x)
    v = x.v
    if name == "exp":
        e = mp.exp(v)
        return _unary(x, e, e, e)
    if name == "log":
        return _unary(x, mp.log(v), 1 / v, -1 / v ** 2)
    if name == "sin":
        return _unary(x, mp.sin(v), mp.cos(v), -mp.sin(v))
    if name == "cos":
        return _unary(x, mp.cos(v), -mp.sin(v), -mp.cos(v))
    if name == "tan":
        t = mp.tan(v)
        return _unary(x, t, 1 + t * t, 2 * t * (1 + t * t))
    if name == "atan":
        q = 1 / (1 + v * v)
        return _unary(x, mp.atan(v), q, -2 * v * q * q)
    if name == "tanh":
        t = mp.tanh(v)
        return _unary(x, t, 1 - t * t, -2 * t * (1 - t * t))
    if name == "sqrt":
        s = mp.sqrt(v)
        return _unary(x, s, 1 / (2 * s), -1 / (4 * s * v))
    if name == "abs":
        return _unary(x, abs(v), _sgn(v), mpf(0))
    if name == "asin":
        q = 1 / (1 - v * v)
        return _unary(x, mp.asin(v), mp.sqrt(q), v * q * mp.sqrt(q))
    if name == "acos":
        q = 1 / (1 - v * v)
        return _unary(x, mp.acos(v), -mp.sqrt(q), -v * q * mp.sqrt(q))
    if name == "sinh":
        return _unary(x, mp.sinh(v), mp.cosh(v), mp.sinh(v))
    if name == "cosh":
        return _unary(x, mp.cosh(v), mp.sinh(v), mp.cosh(v))
    if name == "floor":
        return _unary(x, mp.floor(v), mpf(0), mpf(0))
    raise ValueError(f"unknown function {name}")


def _pow(x, p):
    x = gg.Du.lift(x)
    if p == int(p) and 0 <= p <= 64:
        k = int(p)
        if k == 0:
            return gg.Du(1)
        r = x
        for _ in range(k - 1):
            r = r * x
        return r
    p = mpf(p)
    return _unary(x, x.v ** p, p * x.v ** (p - 1), p * (p - 1) * x.v ** (p - 2))


class _Parser:
    def __init__(self, text, names, aliases, depth=0):
        self.toks = [(m.group(1), m.group(2), m.group(3)) for m in _TOKEN.finditer(text) if m.group(0).strip()]
        self.i, self.names, self.aliases, self.depth = 0, names, aliases, depth

    def peek(self):
        return self.toks[self.i] if self.i < len(self.toks) else (None, None, None)

    def take(self, ch=None):
        t = self.peek()
        if ch is not None and t[2] != ch:
            raise ValueError(f"'{ch}' expected, got {t}")
        self.i += 1
        return t

    def expr(self):
        r = self.term()
        while self.peek()[2] in ("+", "-"):
            op = self.take()[2]
            b = self.term()
            r = r + b if op == "+" else r - b
        return r

    def term(self):
        r = self.unary()
        while self.peek()[2] in ("*", "/"):
            op = self.take()[2]
            b = self.unary()
            r = r * b if op == "*" else r / b
        return r

    def unary(self):
        if self.peek()[2] == "-":
            self.take()
            return -gg.Du.lift(self.unary())
        if self.peek()[2] == "+":
            self.take()
            return self.unary()
        return self.power()

    def const_atom(self):
        """a constant for the exponent of ^"""
        sign = 1
        while self.peek()[2] in ("+", "-"):
            if self.take()[2] == "-":
                sign = -sign
        v = gg.Du.lift(self.atom())
        assert not any(d != 0 for d in (v.d.values() if isinstance(v.d, dict) else v.d)), "exponent must be constant"
        return sign * v.v

    def power(self):
        r = self.atom()
        if self.peek()[2] == "^":
            self.take()
            r = _pow(r, self.const_atom())
        return r

    def atom(self):
        num, name, ch = self.peek()
        if num is not None:
            self.take()
            return gg.Du(mpf(num))
        if ch == "(":
            self.take()
            r = self.expr()
            self.take(")")
            return r
        if name is not None:
            self.take()
            if self.peek()[2] == "(":
                self.take()
                a = self.expr()
                if name in ("max", "min"):
                    self.take(",")
                    b = self.expr()
                    self.take(")")
                    a, b = gg.Du.lift(a), gg.Du.lift(b)
                    if name == "max":
                        return a if a.v > b.v else b
                    return b if a.v > b.v else a
                self.take(")")
                return _fn(name, a)
            if name in self.names:
                return self.names[name]
            if name in self.aliases:
                assert self.depth < 24
                p = _Parser(self.aliases[name], self.names, self.aliases, self.depth + 1)
                r = p.expr()
                assert p.i == len(p.toks), "trailing input in alias " + name
                return r
            raise ValueError(f"unknown name {name}")
        raise ValueError(f"unexpected token {self.peek()}")


def evaluate(text, names, aliases):
    p = _Parser(text, names, aliases)
    r = p.expr()
    assert p.i == len(p.toks), f"trailing input in {text!r}"
    return gg.Du.lift(r)


class ExprMp(gg.Problem):
    """gg.Problem from the keyword arguments of ct.register_ocp"""

    def __init__(self, name, *, dynamics, n=None, m=0, nv=0, lagrange=None, mayer=None, path=(), boundary=(), constants=None,
                 t0=0.0, tf=1.0, it0=-1, itf=-1, maximize=False, **_bounds):
        self.name = name
        self.dyn, self.lag, self.mayer_e, self.path_e, self.bnd = list(dynamics), lagrange, mayer, list(path), list(boundary)
        self.n = len(self.dyn) if n is None else n
        self.m, self.nv, self.p, self.bc = m, nv, len(self.path_e), len(self.bnd)
        self.freet0, self.freetf = it0 >= 0, itf >= 0
        self.lagrange, self.mayer = bool(lagrange), bool(mayer)
        self._t0, self._tf, self.it0, self.itf = t0, tf, it0, itf
        constants = constants or {}
        self.consts = {k: mpf(repr(float(v))) for k, v in constants.items() if not isinstance(v, str)}
        self.aliases = {k: v for k, v in constants.items() if isinstance(v, str)}

    def t0(self, v): return v[self.it0] if self.it0 >= 0 else gg.Du(mpf(repr(float(self._t0))))
    def tf(self, v): return v[self.itf] if self.itf >= 0 else gg.Du(mpf(repr(float(self._tf))))

    def _names(self, t=None, x=None, u=None, v=None, x0=None, xf=None):
        nm = {k: gg.Du(c) for k, c in self.consts.items()}
        if t is not None:
            nm["t"] = gg.Du.lift(t)
        for pre, vec in (("x", x), ("u", u), ("v", v)):
            if vec is not None:
                for k, e in enumerate(vec):
                    nm[f"{pre}{k + 1}"] = e
        for pre, vec in (("x0_", x0), ("xf_", xf)):
            if vec is not None:
                for k, e in enumerate(vec):
                    nm[f"{pre}{k + 1}"] = e
        return nm

    def dynamics(self, t, x, u, v):
        nm = self._names(t, x, u, v)
        return [evaluate(e, nm, self.aliases) for e in self.dyn]

    def lagr(self, t, x, u, v): return evaluate(self.lag, self._names(t, x, u, v), self.aliases) if self.lag else gg.Du(0)
    def may(self, x0, xf, v): return evaluate(self.mayer_e, self._names(v=v, x0=x0, xf=xf), self.aliases) if self.mayer_e else gg.Du(0)

    def path(self, t, x, u, v):
        nm = self._names(t, x, u, v)
        return [evaluate(e, nm, self.aliases) for e in self.path_e]

    def boundary(self, x0, xf, v):
        nm = self._names(v=v, x0=x0, xf=xf)
        return [evaluate(e, nm, self.aliases) for e in self.bnd]
