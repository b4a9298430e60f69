"""SURVEY.md section 8 f3, literally: "a small in-repo interior-point loop through the ABI".  tests/ipm.py (a primal-dual interior-point
method of Ipopt's class: log barrier, slacks, one sparse KKT solve per iteration, inertia-free regularisation, filter line search -- or an
l1 merit function -- with second-order correction) sees the NLP only through the engine's callbacks -- obj, grad, cons, jac_structure /
jac_coord, hess_structure / hess_coord (all on the GPU through the C ABI), bounds, initial guess -- exactly what the reference hands to
Ipopt (src/solve.jl), and solves the reference's catalogued problems ON THE REFERENCE'S DEFAULT GRID (250 steps) to its own tolerance:
objective within rtol = 1e-2 of the value catalogued in the problem file (test/runtests.jl:5-11), KKT error <= 1e-8.

Eighteen problems of test/ci/test_all_ocp.jl and of test/problems/ converge this way in 0.1 - 1.8 s each (1 - 181 iterations); among
them four that scipy's solvers did not settle on this grid or at all (double_integrator_freet0tf, insurance, bioreactor_Ndays,
algal_bacterial on the archive's figure: 5.4530 against 5.4522 of test/archives/jump_ctdirect.md); plus the registry problems on the
stagewise Gauss-Legendre grids, and BASELINE.json's own bench workload (Goddard, Gauss-Legendre 2, 10 000 steps) end to end
(bench/solve_10k.py: 25 iterations, 8 s).  Four more -- moonlander, bioreactor_1day, space_shuttle and the swimmer (0.992069 against the
catalogued 0.984273: the problem scipy's solvers left 1 - 8 % off) -- converge in the loop's ELASTIC mode (test_interior_point_loop_in_elastic_mode).
That leaves ONE problem of the reference's catalogue and problem folder unsolved on the default grid: truck_trailer (the elastic form ends in
a local minimum of the infeasibility, t_f at its lower bound; scipy on 50 steps: within 8 %, tests/test_gpu_solve_catalogue.py;
profiles/r04_experiments.md section 7)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here)
sys.path.insert(0, os.path.dirname(here))

N_REF = 250          # the reference's default grid_size (src/collocation.jl:16-48)

CASES = [("cat", "beam", "midpoint"), ("cat", "fuller", "midpoint"), ("cat", "jackson", "midpoint"), ("cat", "vanderpol", "trapeze"),
         ("cat", "simple_integrator", "midpoint"), ("cat", "bolza_freetf", "midpoint"), ("cat", "robbins", "midpoint"),
         ("cat", "double_integrator_tf", "trapeze"), ("cat", "double_integrator_nobounds", "midpoint"), ("cat", "double_integrator_freet0tf", "midpoint"),
         ("cat", "electric_vehicle", "midpoint"), ("cat", "goddard_all", "midpoint"), ("cat", "glider", "midpoint"), ("cat", "insurance", "trapeze"),
         ("pf", "algal_bacterial", "midpoint"), ("pf", "bioreactor_Ndays", "midpoint"), ("pf", "parametric", "midpoint"), ("pf", "goddard_all_f0f1", "midpoint"),
         ("reg", "goddard", "trapeze"), ("reg", "goddard", "midpoint"), ("reg", "double_integrator_path", "gauss_legendre_2"),
         # the stagewise Gauss-Legendre transcriptions (stage variables K; Goddard's singular arc has negative curvature there: filter line search)
         ("reg", "goddard", "gauss_legendre_2"), ("reg", "goddard", "gauss_legendre_3"), ("reg", "goddard_all", "gauss_legendre_2"),
         ("cat", "beam", "gauss_legendre_2"), ("cat", "vanderpol", "gauss_legendre_3"), ("reg", "quadrotor", "midpoint")]


@pytest.mark.parametrize("kind,name,scheme", CASES, ids=[f"{n}-{s}" for _, n, s in CASES])
def test_interior_point_loop_on_the_default_grid(kind, name, scheme):
    import ctdirect_jl_amd as ct
    import ipm
    import jit_defs
    import problem_folder_defs as pf
    if kind == "cat":
        prob, want, init = jit_defs.catalogue(name)
    elif kind == "pf":
        prob, want, init = pf.folder(name)
    else:
        prob, want, init = name, {"goddard": 1.01257, "goddard_all": 1.01257, "double_integrator_path": 1.5, "quadrotor": None}[name], "problem"
    d = ct.DOCP(prob, N_REF, scheme, pattern="structural", device=0)
    lv, uv = ct.variables_bounds(d)
    x0 = np.clip(ct.initial_guess(d, init), lv, uv)
    with np.errstate(all="ignore"):
        r = ipm.solve_auto(ipm.NLP.from_docp(d, x0, ct), max_iter=600, time_limit=120)
    print(f"{name}/{scheme} N={N_REF}: objective {r.obj:.6f} (catalogue {want}), violation {r.violation:.1e}, KKT error {r.kkt:.1e}, {r.iters} iterations")
    assert r.status == 0 and r.kkt <= 1e-8
    assert r.violation <= 1e-6
    assert want is None or abs(r.obj - want) <= 1e-2 * abs(want)
    # the multipliers it ends with are those of the NLP as the callbacks define it: grad f + J'y - z_L + z_U = 0 (z: bound multipliers)
    import scipy.sparse as sp
    jr, jc = d.jac_structure()
    J = sp.csr_matrix((d.jac_coord(r.x), (jr - 1, jc - 1)), shape=(d.dim_NLP_constraints, d.dim_NLP_variables))
    sign = -1.0 if d.flags.max else 1.0
    res = sign * d.grad(r.x) + J.T @ r.y - r.zl + r.zu
    scale = max(1.0, np.max(np.abs(r.y), initial=0.0), np.max(r.zl, initial=0.0), np.max(r.zu, initial=0.0))
    assert np.max(np.abs(res), initial=0.0) <= 1e-6 * scale
    d.close()


def test_north_star_transcription_solved_end_to_end():
    """The 10 000-step Goddard transcription BASELINE.json's north_star names (trapeze: 40 005 variables, 30 004 constraints, 270 028
    Jacobian and 300 024 Hessian entries) solved by the in-repo interior-point loop through the GPU callbacks: 21 iterations, a few seconds,
    of which the callbacks -- ~165 calls through the host-pointer entry points, PCIe both ways -- are a few tenths of a second; the rest is
    the host's sparse LU (a bordered factorisation: the free final time makes one dense row / column).  The CPU oracle needs about two
    minutes PER ITERATION for the same callbacks (profiles/r04_experiments.md section 7c)."""
    import time
    import ctdirect_jl_amd as ct
    import ipm
    d = ct.DOCP("goddard", 10000, "trapeze", pattern="structural", device=0)
    lv, uv = ct.variables_bounds(d)
    nlp = ipm.NLP.from_docp(d, np.clip(ct.initial_guess(d, "problem"), lv, uv), ct)
    t0 = time.time()
    with np.errstate(all="ignore"):
        r = ipm.solve_auto(nlp, max_iter=200, time_limit=300)
    el, cb = time.time() - t0, sum(nlp.seconds.values())
    print(f"goddard/trapeze N=10000: objective {r.obj:.7f}, {r.iters} iterations, {el:.1f} s total, {cb:.2f} s inside the callbacks {nlp.calls}")
    assert r.status == 0 and r.violation <= 1e-6
    assert abs(r.obj - 1.01257) <= 1e-3 * 1.01257          # (barrier shift mu x number of active bounds: 3e-5 at this size)
    assert cb <= 0.5 * el
    d.close()


@pytest.mark.parametrize("name,scheme,N", [("goddard", "gauss_legendre_2", 250), ("goddard_all", "midpoint", 250), ("goddard", "trapeze", 10000)])
def test_interior_point_loop_on_the_fused_iteration_call(name, scheme, N):
    """the same loop with ONE ctd_eval_all_dev_async per iteration (objective, gradient, constraints, Jacobian and Hessian values at the
    iterate and its multipliers in two launches, device buffers) instead of five host-pointer callbacks: same iterates, same solution"""
    import ctdirect_jl_amd as ct
    import ipm
    d = ct.DOCP(name, N, scheme, pattern="structural", device=0)
    lv, uv = ct.variables_bounds(d)
    x0 = np.clip(ct.initial_guess(d, "problem"), lv, uv)
    with np.errstate(all="ignore"):
        a = ipm.solve(ipm.NLP.from_docp(d, x0, ct), max_iter=300, linesearch="filter")
        nb = ipm.NLP.from_docp(d, x0, ct, fused=True)
        b = ipm.solve(nb, max_iter=300, linesearch="filter")
    print(f"{name}/{scheme} N={N}: {a.iters} / {b.iters} iterations, objective {a.obj:.9f} / {b.obj:.9f}; fused: {nb.calls}, {sum(nb.seconds.values()):.3f} s in callbacks")
    assert a.status == 0 and b.status == 0
    assert abs(a.obj - b.obj) <= 1e-7 * abs(a.obj) and abs(a.obj - 1.01257) <= 1e-3
    # (one gradient and one Jacobian at the initial point, for the scaling; afterwards only the fused call and the line search's c / f)
    assert nb.calls["eval_all"] == b.iters + 1 and "hess" not in nb.calls and nb.calls.get("jac", 0) <= 1 and nb.calls.get("grad", 0) <= 1
    d.close()


@pytest.mark.parametrize("kind,name,scheme,rho,iters", [("cat", "moonlander", "midpoint", 1e2, 600), ("pf", "bioreactor_1day", "midpoint", 1e4, 600),
                                                        ("cat", "space_shuttle", "trapeze", 1e6, 1200), ("pf", "swimmer", "midpoint", 1e4, 2000)])
def test_interior_point_loop_in_elastic_mode(kind, name, scheme, rho, iters):
    """The four catalogued problems whose constraints the plain loop cannot satisfy from the problem file's initial guess (it has no
    feasibility-restoration phase) in the l1-ELASTIC form  min f + rho sum(p + n),  cl <= c(x) - p + n <= cu,  p, n >= 0  (ipm.elastic):
    every point has a feasible completion, and at the solution the elastic variables vanish -- on the reference's default 250-step grid:
    moonlander 0.96182 (catalogued 0.962), bioreactor_1day 0.614111 (0.614134), space_shuttle 0.595692 (0.596554), and the swimmer
    0.992069 (0.984273: 0.8 %, the one scipy's solvers left 1 - 8 % off).  rho: the penalty (above the constraint multipliers)."""
    import ctdirect_jl_amd as ct
    import ipm
    import jit_defs
    import problem_folder_defs as pf
    prob, want, init = jit_defs.catalogue(name) if kind == "cat" else pf.folder(name)
    d = ct.DOCP(prob, N_REF, scheme, pattern="structural", device=0)
    lv, uv = ct.variables_bounds(d)
    x0 = np.clip(ct.initial_guess(d, init), lv, uv)
    with np.errstate(all="ignore"):
        r = ipm.solve_elastic(ipm.NLP.from_docp(d, x0, ct), rhos=(rho,), max_iter=iters, time_limit=150, linesearch="filter")
    print(f"{name}/{scheme} N={N_REF} elastic rho={rho:g}: objective {r.obj:.6f} (catalogue {want}), violation {r.violation:.1e}, KKT error {r.kkt:.1e}, "
          f"{r.iters} iterations, sum of the elastic variables {r.elastic_sum:.1e}")
    assert r.status == 0 and r.kkt <= 1e-8 and r.violation <= 1e-6
    assert r.elastic_sum <= 1e-3
    assert abs(r.obj - want) <= 1e-2 * abs(want)
    d.close()


def test_solution_rebuild_from_the_interior_point_solution():
    """test/ci/test_modeler_solver.jl:49-65 on its own terms: the min-energy double integrator on the reference's 250-step grid, solved by
    the in-repo loop through the GPU callbacks, rebuilt with the reference's getter conventions (`unpack_solution`): state, control and
    COSTATE (the multipliers of the state-equation rows, src/DOCP_data.jl:514-633) against the analytic extremal x = (0.75 t^2 - 0.25 t^3,
    1.5 t - 0.75 t^2), u = 1.5 - 1.5 t, p = (3, 3 - 3 t), in norm at rtol 1e-2 as the reference does."""
    import ctdirect_jl_amd as ct
    import ipm
    d = ct.DOCP("double_integrator_path", N_REF, "midpoint", pattern="structural", device=0)
    lv, uv = ct.variables_bounds(d)
    with np.errstate(all="ignore"):
        r = ipm.solve_auto(ipm.NLP.from_docp(d, np.clip(ct.initial_guess(d), lv, uv), ct), max_iter=200)
    assert r.status == 0 and abs(r.obj - 1.5) <= 1e-3
    sol = ct.unpack_solution(d, r.x, r.y)
    T = sol["T"]
    X = np.stack([0.75 * T ** 2 - 0.25 * T ** 3, 1.5 * T - 0.75 * T ** 2], axis=1)
    U = 1.5 - 1.5 * (T[:-1] + T[1:]) / 2
    P = np.stack([np.full(N_REF, 3.0), 3.0 - 3.0 * T[1:]], axis=1)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)      # noqa: E731
    assert rel(sol["X"], X) <= 1e-2 and rel(sol["U"][:-1, 0], U) <= 1e-2 and rel(sol["P"], P) <= 1e-2
    d.close()
