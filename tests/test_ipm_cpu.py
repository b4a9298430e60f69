"""CPU check of the test infrastructure itself: the in-repo interior-point loop (tests/ipm.py) on the CPU oracle's
callbacks -- the same loop tests/test_gpu_solve_ipm.py drives through the engine's GPU callbacks.  Reference values: the objectives
catalogued in the problem files (test/problems/goddard.jl:48: 1.01257; the min-energy double integrator: 1.5 analytically)."""
import numpy as np
import pytest

import ipm
from oracle.oracle import OracleDOCP


@pytest.mark.parametrize("prob,sch,N,maximize,want", [("goddard", "trapeze", 100, True, 1.01257), ("goddard_all", "midpoint", 60, True, 1.01257),
                                                      ("double_integrator_path", "gauss_legendre_2", 40, False, 1.5),
                                                      ("quadrotor", "midpoint", 250, False, None), ("goddard", "euler_implicit", 100, True, 1.01257),
                                                      ("goddard", "gauss_legendre_2", 50, True, 1.01257), ("goddard_all", "gauss_legendre_2", 50, True, 1.01257)])
def test_interior_point_loop_on_the_oracle(prob, sch, N, maximize, want):
    o = OracleDOCP(prob, sch, N)
    o.set_pattern_mode(1)
    with np.errstate(all="ignore"):
        r = ipm.solve_auto(ipm.NLP.from_oracle(o, o.initial_guess(True), maximize=maximize), max_iter=400)
    assert r.status == 0 and r.kkt <= 1e-8 and r.violation <= 1e-6
    if want is not None:
        assert abs(r.obj - want) <= 1e-2 * abs(want)
    # first-order optimality in the oracle's own terms
    cp, rv = o.jac_pattern()
    import scipy.sparse as sp
    J = sp.csc_matrix((o.jac_coord(r.x), rv, cp), shape=(o.dim_NLP_constraints, o.dim_NLP_variables))
    g = (-1.0 if maximize else 1.0) * o.gradient(r.x)
    res = g + J.T @ r.y - r.zl + r.zu
    assert np.max(np.abs(res)) <= 1e-6 * max(1.0, np.max(np.abs(r.y)), np.max(r.zl), np.max(r.zu))


def test_elastic_mode_on_the_oracle():
    """ipm.elastic / solve_elastic: the l1-elastic form reaches the same solution with vanishing elastic variables"""
    o = OracleDOCP("goddard", "trapeze", 100)
    o.set_pattern_mode(1)
    with np.errstate(all="ignore"):
        r = ipm.solve_elastic(ipm.NLP.from_oracle(o, o.initial_guess(True), maximize=True), rhos=(1e4,), max_iter=300, linesearch="filter")
    assert r.status == 0 and r.violation <= 1e-6 and r.elastic_sum <= 1e-3
    assert abs(r.obj - 1.01257) <= 1e-3
