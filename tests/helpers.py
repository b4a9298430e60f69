"""Shared helpers for the test-suite: golden-fixture loading and deterministic input fills."""
import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# parity tolerance (BASELINE.json north_star / SURVEY.md section 8d): max |a - b| / max(1, |b|) <= 1e-10
TOL = 1e-10


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def hess_err(o, x, y, obj_weight, got, ref=None, idx=None):
    """Error of Hessian-of-the-Lagrangian values against the oracle on the backward-error scale: entries are sums of terms
    y_r d2c_r that cancel when the multipliers change sign from row to row (Goddard's drag: per-stage terms ~1e5 cancel to
    ~20), so two double-precision evaluations cannot agree to 1e-10 of the RESULT there.  Each entry is measured against
    max(1, |ref|, the same sum with |y| and |obj_weight|) -- the magnitude of what is summed (tests/test_gpu_hessian.py) --
    which keeps the 1e-10 bar.  `idx`: compare only these positions."""
    ref = o.hess_coord(x, y, obj_weight) if ref is None else ref
    mag = o.hess_coord(x, np.abs(y), abs(obj_weight))
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    if idx is not None:
        got, ref, mag = got[idx], ref[idx], mag[idx]
    if got.size == 0:
        return 0.0
    return float(np.max(np.abs(got - ref) / np.maximum(1.0, np.maximum(np.abs(ref), np.abs(mag)))))


def fx(h):
    return float.fromhex(h)


def golden_files():
    return sorted(f for f in glob.glob(os.path.join(GOLDEN_DIR, "*.json")) if not os.path.basename(f).startswith("hess_"))


def hess_golden_files():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "hess_*.json")))


def load_hess_golden(path):
    """Fixture of tests/golden/gen_golden_hess.py: xu, multipliers y, obj_weight and the exact lower-triangular Hessian
    of the Lagrangian as a dict {(row, col): value} (0-based, row >= col)."""
    with open(path) as f:
        g = json.load(f)
    out = dict(g)
    out["time_grid"] = [fx(t) for t in g["time_grid"]] if g["time_grid"] else None
    out["xu"] = np.array([fx(x) for x in g["xu"]])
    out["y"] = np.array([fx(x) for x in g["y"]])
    out["H"] = {(r, c): fx(v) for r, c, v in g["hess_nonzeros"]}
    return out


def hess_on_pattern(H, colptr, rowval):
    """values of the fixture's Hessian dict on a lower-triangular CSC pattern + the entries the pattern leaves out"""
    vals = np.zeros(len(rowval))
    seen = set()
    for j in range(len(colptr) - 1):
        for k in range(colptr[j], colptr[j + 1]):
            key = (int(rowval[k]), j)
            vals[k] = H.get(key, 0.0)
            seen.add(key)
    return vals, [k for k in H if k not in seen]


def load_golden(path):
    with open(path) as f:
        g = json.load(f)
    out = dict(g)
    out["time_grid"] = [fx(t) for t in g["time_grid"]] if g["time_grid"] else None
    out["xu"] = np.array([fx(x) for x in g["xu"]])
    out["c"] = np.array([fx(x) for x in g["c"]])
    out["objective"] = fx(g["objective"])
    nvar, ncon = g["dims"]["nvar"], g["dims"]["ncon"]
    J = np.zeros((ncon, nvar))
    for r, c, v in g["jac_nonzeros"]:
        J[r, c] = fx(v)
    out["J"] = J
    grad = np.zeros(nvar)
    for j, v in g["gradient"]:
        grad[j] = fx(v)
    out["gradient"] = grad
    return out


def csc_to_set(colptr, rowval):
    s = set()
    for j in range(len(colptr) - 1):
        for k in range(colptr[j], colptr[j + 1]):
            s.add((int(rowval[k]), j))
    return s


def dense_on_pattern(J, colptr, rowval):
    vals = np.zeros(len(rowval))
    for j in range(len(colptr) - 1):
        for k in range(colptr[j], colptr[j + 1]):
            vals[k] = J[rowval[k], j]
    return vals


# ---- xorshift64* (SURVEY.md section 8d seeded variant); bit-identical integer generator
def xorshift64star(seed, count):
    mask = (1 << 64) - 1
    x = seed & mask
    out = np.empty(count, dtype=np.float64)
    for i in range(count):
        x ^= x >> 12
        x ^= (x << 25) & mask
        x ^= x >> 27
        r = (x * 0x2545F4914F6CDD1D) & mask
        out[i] = ((r >> 11) / float(1 << 53)) * 2.0 - 1.0
    return out


def bench_inputs(d, perturb=0.0, seed=0x9E3779B97F4A7C15):
    """Closed-form feasible-ish inputs for a DOCP-like object `d` (needs: problem name via d.problem_name,
    n, m, nv, steps, step_variables_block, stage, stagewise, scheme_kind, dim_NLP_variables, butcher c).
    Vectorised numpy version of the fills in SURVEY.md section 8(d); used for mid/large sizes where the
    oracle (not mpmath) is the checker, so the exact fill formula only has to be the same on both sides."""
    N, blk, n, m, nv = d.steps, d.step_variables_block, d.n, d.m, d.nv
    nvar = d.dim_NLP_variables
    xu = np.full(nvar, 0.1)
    name = d.problem_name
    tau = np.arange(N + 1) / N

    def state(t):
        if name.startswith("goddard"):
            return [1 + 0.01 * t, 0.1 * np.sin(np.pi * t), 1 - 0.4 * t]
        if name.startswith("double_integrator"):
            return [t * t, 2 * t]
        if name == "quadrotor":
            return [0.01 * t, 5 * t, 2.5 + 0.01 * np.sin(5 * t), 0.1 * np.sin(2 * t), 5 + 0.1 * np.cos(t),
                    0.02 * np.sin(4 * t), 0.1 * np.sin(6 * t), 0.15 * np.cos(5 * t)]
        if name == "quadrotor12":
            return [0.01 * t, 5 * t, 2.5 + 0.01 * np.sin(5 * t), 0.1 * np.sin(2 * t), 5 + 0.1 * np.cos(t),
                    0.02 * np.sin(4 * t), 0.1 * np.sin(6 * t), 0.15 * np.cos(5 * t), 0.2 * np.sin(2 * t + 0.3),
                    0.3 * np.sin(7 * t), 0.25 * np.cos(3 * t), 0.1 * np.sin(9 * t + 1)]
        if name == "stagewise_scalar":
            return [t * t]
        return [np.cos(1.3 * t) + 0.1, np.sin(1.3 * t) - 0.05]

    def control(t, j):
        if name.startswith("goddard"):
            return [0.5 + 0.5 * np.cos(7 * t + j)]
        if name.startswith("double_integrator") or name == "stagewise_scalar":
            return [2 * np.cos(3 * t + 0.1 * j)]
        if name == "quadrotor":
            return [10 + np.sin(4 * t + j), 0.3 * np.cos(3 * t + j), 0.2 * np.sin(5 * t + j), 0.05 * np.cos(2 * t + j)]
        if name == "quadrotor12":
            return [10 + np.sin(4 * t + j), 0.03 * np.cos(3 * t + j), 0.02 * np.sin(5 * t + j), 0.01 * np.cos(2 * t + j)]
        return []

    var = {"goddard": [0.2], "goddard_all": [0.2], "quadrotor": [1.0], "quadrotor12": [1.0],
           "estimate_rotation_rate": [1.4], "estimate_initial_condition": [0.9, 0.1],
           "least_squares_with_constraint": [0.8, 0.2], "double_integrator_freet0tf": [0.3, 2.1]}.get(name, [])
    for k in range(nv):
        xu[nvar - nv + k] = var[k]
    st = state(tau)
    for k in range(n):
        xu[k:(N + 1) * blk:blk][:N + 1] = st[k] if np.ndim(st[k]) else np.full(N + 1, st[k])
    kind = d.scheme_kind
    if m > 0:
        if kind == "irk" and d.stagewise:
            s = d.stage
            cj = d.butcher_c
            for j in range(s):
                tj = tau[:-1] + cj[j] * (tau[1:] - tau[:-1])
                uj = control(tj, j + 1)
                for k in range(m):
                    xu[n + j * m + k:N * blk:blk] = uj[k]
        else:
            nodes = N + 1 if kind == "trapeze" else N
            u0 = control(tau[:nodes], 0)
            for k in range(m):
                xu[n + k:nodes * blk:blk][:nodes] = u0[k]
    if kind == "irk":
        # stage variables: smooth, O(1) values near the local slope of the state fill (not exact dynamics:
        # residuals are then O(1e-1), well conditioned, and identical on both sides by construction)
        s = d.stage
        cu = m * s if d.stagewise else m
        for j in range(s):
            for k in range(n):
                xu[n + cu + j * n + k:N * blk:blk] = 0.3 * np.sin(3 * tau[:-1] + k + 0.5 * j) + 0.05 * (k + 1)
    if perturb:
        xu = xu + perturb * xorshift64star(seed, nvar)
    return xu


def describe(d, problem, scheme):
    """Adapter giving bench_inputs() the few attributes it needs, from an OracleDOCP or a ctdirect DOCP."""
    from types import SimpleNamespace
    if hasattr(d, "dims"):     # product DOCP
        disc = d.discretization
        return SimpleNamespace(problem_name=problem, n=d.dims.NLP_x, m=d.dims.NLP_u, nv=d.dims.NLP_v, steps=d.time.steps,
                               step_variables_block=disc._step_variables_block, stage=disc.stage,
                               stagewise=scheme in ("gauss_legendre_2", "gauss_legendre_3"),
                               scheme_kind="trapeze" if scheme == "trapeze" else ("midpoint" if scheme == "midpoint" else "irk"),
                               dim_NLP_variables=d.dim_NLP_variables,
                               butcher_c=getattr(disc, "butcher_c", None))
    return SimpleNamespace(problem_name=problem, n=d.n, m=d.m, nv=d.nv, steps=d.steps,
                           step_variables_block=d.step_variables_block, stage=d.stage,
                           stagewise=scheme in ("gauss_legendre_2", "gauss_legendre_3"),
                           scheme_kind="trapeze" if scheme == "trapeze" else ("midpoint" if scheme == "midpoint" else "irk"),
                           dim_NLP_variables=d.dim_NLP_variables,
                           butcher_c=d.butcher()[2] if d.stage else None)
