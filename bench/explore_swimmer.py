#!/usr/bin/env python3
"""swimmer (test/problems/swimmer.jl, catalogued 0.984273 on the reference's default 250-step grid) through the GPU callbacks with
scipy's trust-constr: grid sizes / iteration caps / warm starts.  python bench/explore_swimmer.py"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from scipy.optimize import Bounds, NonlinearConstraint, minimize
from scipy.sparse import csc_matrix, coo_matrix
import ctdirect_jl_amd as ct
import problem_folder_defs as pf


def solve_tc(prob, scheme, N, init, maxiter, x0=None):
    d = ct.DOCP(prob, N, scheme, pattern="structural", device=0)
    nvar, ncon = d.dim_NLP_variables, d.dim_NLP_constraints
    lc, uc = ct.constraints_bounds(d); lv, uv = ct.variables_bounds(d)
    if x0 is None:
        x0 = np.clip(ct.initial_guess(d, init), lv, uv)
    colptr, rowval = ct.DOCP_Jacobian_pattern(d)
    hr, hc = d.hess_structure()
    sign = -1.0 if d.flags.max else 1.0
    jac = lambda x: csc_matrix((d.jac_coord(x), rowval, colptr), shape=(ncon, nvar))
    def sym(vals):
        lower = coo_matrix((vals, (hr - 1, hc - 1)), shape=(nvar, nvar)).tocsc()
        diag = coo_matrix((vals[hr == hc], (hr[hr == hc] - 1, hc[hr == hc] - 1)), shape=(nvar, nvar)).tocsc()
        return lower + lower.T - diag
    con = NonlinearConstraint(lambda x: d.cons(x), lc, uc, jac=jac, hess=lambda x, v: sym(d.hess_coord(x, v, 0.0)))
    res = minimize(lambda x: sign * d.obj(x), x0, jac=lambda x: sign * d.grad(x), hess=lambda x: sym(d.hess_coord(x, np.zeros(ncon), sign)),
                   constraints=[con], bounds=Bounds(lv, uv), method="trust-constr", options={"maxiter": maxiter, "gtol": 1e-8, "xtol": 1e-10})
    c = d.cons(res.x)
    viol = max(float(np.max(np.maximum(lc - c, 0.0))), float(np.max(np.maximum(c - uc, 0.0))))
    return sign * res.fun, viol, res, d


rt, want, init = pf.folder("swimmer")
prev = init
for N, it in ((100, 400), (150, 1500), (250, 2500)):
    t0 = time.time()
    obj, viol, res, d = solve_tc(rt, "midpoint", N, prev, it)
    print(f"swimmer/midpoint N={N} (warm start from the coarser grid): {obj:.6f} (cat {want}) viol {viol:.1e} status {res.status} nit {res.nit} {time.time()-t0:.0f}s", flush=True)
    sol = ct.unpack_solution(d, res.x)
    prev = dict(time=sol["T"], state=sol["X"], control=sol["U"], variable=sol["v"])
    d.close()
for N, it in ((150, 1500), (200, 2500)):
    t0 = time.time()
    obj, viol, res, d = solve_tc(rt, "midpoint", N, init, it)
    print(f"swimmer/midpoint N={N} (problem file's guess): {obj:.6f} (cat {want}) viol {viol:.1e} status {res.status} nit {res.nit} {time.time()-t0:.0f}s", flush=True)
    d.close()
