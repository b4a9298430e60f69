"""End-to-end second-order checks (SURVEY.md section 8 f1 + f3): solvers that consume the exact Hessian of the Lagrangian
served by ctd_hess_structure / ctd_hess_coord, next to the first-order callbacks, all through the C ABI on the GPU.

  * a Newton-KKT step on the min-energy double integrator (quadratic cost, linear dynamics): with the exact Hessian ONE
    step from the default initial guess lands on the solution -- checks values, symmetry handling and the sign convention
    L = obj_weight f + y'c  (NLPModels hess_coord!) in one go;
  * scipy's trust-constr interior-point method with exact Hessians on the catalogued problems, accepted with the
    reference's rule objective ~ prob.obj, rtol = 1e-2 (test/runtests.jl:5-11)."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from scipy.optimize import Bounds, NonlinearConstraint, minimize

import ctdirect_jl_amd as ct

pytestmark = pytest.mark.gpu


def _sym(d, vals):
    hr, hc = d.hess_structure()
    L = sp.coo_matrix((vals, (hr - 1, hc - 1)), shape=(d.dim_NLP_variables,) * 2).tocsr()
    return L + L.T - sp.diags(L.diagonal())


def _jac(d, x):
    jr, jc = d.jac_structure()
    return sp.csr_matrix((d.jac_coord(x), (jr - 1, jc - 1)), shape=(d.dim_NLP_constraints, d.dim_NLP_variables))


def test_newton_kkt_step_solves_the_quadratic_program():
    N = 40
    d = ct.DOCP("double_integrator_path", N, "gauss_legendre_2", device=0)
    nvar = d.dim_NLP_variables
    lc, uc = ct.constraints_bounds(d)
    eq = np.where(lc == uc)[0]                         # dynamics + boundary rows; the path rows stay inactive
    x = ct.initial_guess(d)
    y = np.zeros(d.dim_NLP_constraints)
    # KKT system of  min f(x)  s.t.  c_eq(x) = lc_eq  at (x, y):  [H J'; J 0] [dx; y+] = -[grad f; c - lc]
    H = _sym(d, d.hess_coord(x, y, 1.0))
    J = _jac(d, x)[eq]
    K = sp.bmat([[H, J.T], [J, None]], format="csc")
    rhs = -np.concatenate([d.grad(x), (d.cons(x) - lc)[eq]])
    sol = spla.spsolve(K, rhs)
    x1 = x + sol[:nvar]
    y1 = np.zeros_like(y)
    y1[eq] = sol[nvar:]
    # one step is exact for a QP: feasible, stationary, and it is the analytic solution (cost 1.5, u = 1.5 - 1.5 t)
    assert np.max(np.abs((d.cons(x1) - lc)[eq])) <= 1e-10
    assert np.max(np.abs(d.grad(x1) + _jac(d, x1).T @ y1)) <= 1e-9
    assert abs(d.obj(x1) - 1.5) <= 1e-6
    c1 = d.cons(x1)
    assert np.all(c1 <= uc + 1e-9)                     # q + 0.1 w^2 <= 1.05 indeed inactive
    # solution rebuild with the reference's conventions (src/ode/common.jl:7-104, src/DOCP_data.jl:514-633) against the analytic
    # extremal of min int u^2 on [0, 2]: u = 1.5 - 1.5 t, q = 0.75 t^2 - 0.25 t^3, w = 1.5 t - 0.75 t^2, costate (3, 3 - 3 t);
    # same check and tolerance as test/ci/test_modeler_solver.jl:49-65
    sol = ct.unpack_solution(d, x1, y1)
    T = sol["T"]
    assert np.allclose(T, np.linspace(0.0, 2.0, N + 1))
    assert np.allclose(sol["X"], np.stack([0.75 * T ** 2 - 0.25 * T ** 3, 1.5 * T - 0.75 * T ** 2], axis=1), atol=1e-2)
    assert np.allclose(sol["U"][:-1, 0], 1.5 - 1.5 * (T[:-1] + T[1:]) / 2, atol=2e-2)         # b-weighted stage average ~ midpoint value
    # the multipliers of the state-equation rows of step i are the discrete costate at the END of the step (exactly, for this
    # QP); the reference compares them with p(t_i) in norm at rtol 1e-2 on its 250-step grid, where the O(h) shift is below that
    assert np.allclose(sol["P"], np.stack([np.full(N, 3.0), 3.0 - 3.0 * T[1:]], axis=1), atol=1e-7)
    assert sol["path_constraints_dual"].shape == (N + 1, 1) and np.allclose(sol["path_constraints_dual"], 0.0)
    assert sol["boundary_constraints_dual"].shape == (4,)
    # the Hessian of this problem does not depend on x (checked at the solution, with the new multipliers)
    assert np.max(np.abs(d.hess_coord(x1, y1, 1.0) - d.hess_coord(x, y1, 1.0))) <= 1e-12


def test_newton_kkt_step_with_direct_shooting_layout():
    """The same quadratic program on the direct-shooting layout (`control_steps = 3` controls per step, midpoint scheme,
    src/direct_shooting.jl:55-71): ONE Newton-KKT step with the exact Hessian of that layout (one second-order point per
    control of the step) is feasible, stationary and reproduces the analytic extremal to the scheme's order."""
    N, cs = 40, 3
    d = ct.DOCP("double_integrator_path", N, "midpoint", device=0, control_steps=cs)
    nvar = d.dim_NLP_variables
    lc, uc = ct.constraints_bounds(d)
    eq = np.where(lc == uc)[0]
    x = ct.initial_guess(d)
    y = np.zeros(d.dim_NLP_constraints)
    H = _sym(d, d.hess_coord(x, y, 1.0))
    J = _jac(d, x)[eq]
    K = sp.bmat([[H, J.T], [J, None]], format="csc")
    sol = spla.spsolve(K, -np.concatenate([d.grad(x), (d.cons(x) - lc)[eq]]))
    x1 = x + sol[:nvar]
    y1 = np.zeros_like(y)
    y1[eq] = sol[nvar:]
    assert np.max(np.abs((d.cons(x1) - lc)[eq])) <= 1e-10
    assert np.max(np.abs(d.grad(x1) + _jac(d, x1).T @ y1)) <= 1e-9
    assert abs(d.obj(x1) - 1.5) <= 1e-3
    assert np.all(d.cons(x1) <= uc + 1e-9)
    out = ct.unpack_solution(d, x1, y1)
    T, Tc = out["T"], out["T_control"]
    assert Tc.shape == (N * cs + 1,) and np.allclose(Tc[:-1:cs], T[:-1])
    assert np.allclose(out["X"], np.stack([0.75 * T ** 2 - 0.25 * T ** 3, 1.5 * T - 0.75 * T ** 2], axis=1), atol=1e-2)
    # the controls of a step all see the step's midpoint state (midpoint.jl:57-69) and this cost is convex in them: they come out
    # equal, at the extremal u = 1.5 - 1.5 t of the step's midpoint
    U = x1[:N * (2 + cs)].reshape(N, 2 + cs)[:, 2:]
    assert np.max(np.abs(U - U[:, :1])) <= 1e-9
    assert np.allclose(U[:, 0], 1.5 - 1.5 * (T[:-1] + T[1:]) / 2, atol=2e-2)
    d.close()


def _trust_constr(prob, scheme, N, init=None, maxiter=500):
    d = ct.DOCP(prob, N, scheme, pattern="structural", device=0)
    ncon = d.dim_NLP_constraints
    lc, uc = ct.constraints_bounds(d)
    lv, uv = ct.variables_bounds(d)
    x0 = np.clip(ct.initial_guess(d, init), lv, uv)
    sign = -1.0 if d.flags.max else 1.0                # minimize = !docp.flags.max (src/collocation.jl:145)
    con = NonlinearConstraint(lambda x: d.cons(x), lc, uc, jac=lambda x: _jac(d, x),
                              hess=lambda x, v: _sym(d, d.hess_coord(x, v, 0.0)))
    res = minimize(lambda x: sign * d.obj(x), x0, jac=lambda x: sign * d.grad(x),
                   hess=lambda x: _sym(d, d.hess_coord(x, np.zeros(ncon), sign)), constraints=[con], bounds=Bounds(lv, uv),
                   method="trust-constr", options={"maxiter": maxiter, "gtol": 1e-8, "xtol": 1e-10})
    c = d.cons(res.x)
    viol = max(float(np.max(np.maximum(lc - c, 0.0))), float(np.max(np.maximum(c - uc, 0.0))))
    return d, res, sign * res.fun, viol


def test_second_order_solves_reach_the_catalogued_objectives():
    d, res, obj, viol = _trust_constr("double_integrator_path", "midpoint", 50)
    assert res.status in (1, 2) and viol <= 1e-8 and abs(obj - 1.5) <= 1e-2 * 1.5
    d, res, obj, viol = _trust_constr("stagewise_scalar", "gauss_legendre_2", 20)     # test_discretization_stagewise.jl:103-116
    assert res.status in (1, 2) and viol <= 1e-8 and abs(obj - 1.0) <= 1e-2
    # Goddard (test/problems/goddard.jl:48, obj 1.01257): the interior-point iteration creeps along the singular arc, so
    # it is stopped by the iteration cap; the iterate is feasible and the objective meets the reference's rtol = 1e-2
    d, res, obj, viol = _trust_constr("goddard", "midpoint", 60, init="problem", maxiter=1000)
    print(f"goddard trust-constr: status={res.status} nit={res.nit} objective={obj:.6f} violation={viol:.2e}")
    assert viol <= 1e-5 and abs(obj - 1.01257) <= 1e-2 * 1.01257
