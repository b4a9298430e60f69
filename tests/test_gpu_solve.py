"""End-to-end check: an off-the-shelf NLP solver (scipy SLSQP) driven ONLY by the engine's
callbacks -- obj, grad!, cons!, jac_structure!, jac_coord!, bounds, initial guess, all through the C ABI on the GPU --
reproduces the objectives the reference catalogues for its test problems, with the reference's own acceptance rule
`objective(sol) ~ prob.obj rtol = 1e-2` (test/runtests.jl:5-11).

    goddard            obj 1.01257   test/problems/goddard.jl:48   (default Collocation scheme: midpoint)
    stagewise_scalar   obj 1.0       test/ci/test_discretization_stagewise.jl:14,108-113
    double integrator, min energy on [0, 2]: analytic u(t) = 1.5 - 1.5 t, cost 1.5 (the build's path constraint
    q + 0.1 w^2 <= 1.05 stays inactive); same kind of check as test/ci/test_modeler_solver.jl:49-65
"""
import numpy as np
import pytest
import scipy.sparse as sp
from scipy.optimize import minimize

import ctdirect_jl_amd as ct

pytestmark = pytest.mark.gpu


def solve(prob, scheme, N, init=None, maxiter=400, ftol=1e-8, restarts=1):
    d = ct.DOCP(prob, N, scheme, pattern="structural", device=0)
    rows, cols = d.jac_structure()
    shape = (d.dim_NLP_constraints, d.dim_NLP_variables)
    lc, uc = ct.constraints_bounds(d)
    lv, uv = ct.variables_bounds(d)
    x0 = ct.initial_guess(d, init)
    sign = -1.0 if d.flags.max else 1.0          # minimize = !docp.flags.max   (src/collocation.jl:145)

    def jac(x):
        return sp.csc_matrix((d.jac_coord(x), (rows - 1, cols - 1)), shape=shape).toarray()

    eq = np.where(lc == uc)[0]
    lo = np.where((lc != uc) & np.isfinite(lc))[0]
    up = np.where((lc != uc) & np.isfinite(uc))[0]
    cons = [{"type": "eq", "fun": lambda x: (d.cons(x) - lc)[eq], "jac": lambda x: jac(x)[eq]}]
    if len(lo):
        cons.append({"type": "ineq", "fun": lambda x: (d.cons(x) - lc)[lo], "jac": lambda x: jac(x)[lo]})
    if len(up):
        cons.append({"type": "ineq", "fun": lambda x: (uc - d.cons(x))[up], "jac": lambda x: -jac(x)[up]})
    bounds = [(None if not np.isfinite(a) else a, None if not np.isfinite(b) else b) for a, b in zip(lv, uv)]
    x, nit = x0, 0
    for _ in range(restarts):        # SLSQP's LSQ sub-problem sometimes gives up early; restarting from its iterate recovers
        res = minimize(lambda x: sign * d.obj(x), x, jac=lambda x: sign * d.grad(x), bounds=bounds, constraints=cons,
                       method="SLSQP", options={"maxiter": maxiter, "ftol": ftol})
        x = np.clip(res.x, lv, uv)
        nit += res.nit
        if res.success:
            break
    res.nit = nit
    c = d.cons(res.x)
    viol = max(float(np.max(np.maximum(lc - c, 0.0))), float(np.max(np.maximum(c - uc, 0.0))))
    return d, res, sign * res.fun, viol


def test_goddard_catalogued_objective():
    # Goddard (bang - singular - bang) is hard for a generic SQP code (the reference uses Ipopt) and SLSQP's trajectory is
    # sensitive to last-bit differences, so a few grid sizes are tried; the first converged solve must be feasible and
    # meet the reference's acceptance rule objective ~ 1.01257, rtol = 1e-2 (test/runtests.jl:5-11).  On MI355X the first
    # candidate converges: N = 100, 131 iterations, objective 1.012521, violation 4.9e-10.
    for N in (100, 80, 120, 60):
        d, res, obj, viol = solve("goddard", "midpoint", N, init="problem", restarts=2)
        print(f"goddard N={N}: success={res.success} objective={obj:.6f} iterations={res.nit} violation={viol:.2e}")
        if res.success:
            break
    assert res.success and viol <= 1e-6
    assert abs(obj - 1.01257) <= 1e-2 * 1.01257                       # test_problem: rtol = 1e-2
    assert abs(obj - 1.01257) <= 2e-4                                  # in fact much closer
    blk = d.discretization._step_variables_block
    assert np.allclose(res.x[0:3], [1.0, 0.0, 1.0], atol=1e-7)        # x(0) == x0
    assert abs(res.x[N * blk + 2] - 0.6) <= 1e-7                      # m(tf) == mf
    assert 0.01 <= res.x[-1]                                          # tf >= 0.01


@pytest.mark.parametrize("scheme", ["gauss_legendre_2", "gauss_legendre_3"])
def test_stagewise_scalar_solve(scheme):
    # reference: test/ci/test_discretization_stagewise.jl:103-116 (grid_size 20, objective 1.0, x(0) = 0, x(1) = 1)
    d, res, obj, viol = solve("stagewise_scalar", scheme, 20)
    assert res.success and viol <= 1e-8
    assert abs(obj - 1.0) <= 1e-2
    blk = d.discretization._step_variables_block
    assert abs(res.x[0]) <= 1e-8 and abs(res.x[20 * blk] - 1.0) <= 1e-4


def test_double_integrator_min_energy_analytic():
    N = 50
    d, res, obj, viol = solve("double_integrator_path", "midpoint", N)
    assert res.success and viol <= 1e-8
    assert abs(obj - 1.5) <= 1e-2 * 1.5
    blk = d.discretization._step_variables_block
    t_mid = (np.arange(N) + 0.5) * (2.0 / N)
    u = res.x[2:N * blk:blk]
    assert np.max(np.abs(u - (1.5 - 1.5 * t_mid))) <= 1e-2            # analytic control, atol 1e-2 as in the reference
    t = np.arange(N + 1) * (2.0 / N)
    q = res.x[0:(N + 1) * blk:blk]
    assert np.max(np.abs(q - (0.75 * t ** 2 - 0.25 * t ** 3))) <= 1e-2
