"""GPU parity of `control_steps > 1` -- DOCP(ocp, grid_size, control_steps, :midpoint, time_grid), the direct-shooting layout
of the reference (src/direct_shooting.jl:55-71): `control_steps` controls per time step, dynamics summed over the control
sub-steps (src/ode/midpoint.jl:47-72,137-155), cost midpoint.jl:99-116.  HIP engine through the C ABI against the oracle
(and, in test_gpu_parity.py::test_fixture_parity_host_pointers, against the 50-digit mpmath fixtures tests/golden/cs*.json)."""
import numpy as np
import pytest

import ctdirect_jl_amd as ct
from helpers import TOL, bench_inputs, describe, hess_err, relerr

pytestmark = pytest.mark.gpu
SENT = 777.125


def _inputs(o, prob, cs, seed):
    x = bench_inputs(describe(o, prob, "midpoint"), perturb=1e-2)
    rng = np.random.default_rng(seed)
    n, m, blk = o.n, o.m, o.step_variables_block
    for j in range(1, cs):                      # the further controls of every step: near the first one, all different
        for k in range(m):
            x[n + j * m + k:o.steps * blk:blk] = x[n + k:o.steps * blk:blk] * (1 + 0.05 * j) + 0.01 * rng.standard_normal(o.steps)
    return x


CASES = [("goddard", 2), ("goddard", 3), ("double_integrator_path", 2), ("double_integrator_path", 3), ("goddard_all", 2),
         ("goddard_all", 3), ("double_integrator_freet0tf", 3), ("quadrotor", 2), ("quadrotor12", 2), ("quadrotor12", 3),
         ("estimate_rotation_rate", 2)]


@pytest.mark.parametrize("prob,cs", CASES, ids=[f"{p}-cs{c}" for p, c in CASES])
def test_control_steps_parity(oracle_lib, prob, cs):
    import torch
    rng = np.random.default_rng(5)
    for N, tg in ((1, None), (5, None), (257, None), (64, np.cumsum(rng.uniform(0.5, 1.5, 65)))):
        o = oracle_lib.OracleDOCP(prob, "midpoint", N, time_grid=tg, control_steps=cs)
        x = _inputs(o, prob, cs, N)
        xd = torch.from_numpy(x).cuda()
        for mode, name in ((0, "manual"), (1, "structural"), (2, "optimized")):
            o.set_pattern_mode(mode)
            d = ct.DOCP(prob, N, "midpoint", time_grid=tg, pattern=name, device=0, control_steps=cs)
            assert d.nnzj == o.jac_nnz()
            c = torch.full((d.dim_NLP_constraints,), SENT, dtype=torch.float64, device="cuda")
            v = torch.full((d.nnzj,), SENT, dtype=torch.float64, device="cuda")
            d.cons_jac(xd, c, v)
            assert not bool((c == SENT).any()) and not bool((v == SENT).any())           # every output written
            assert relerr(c.cpu().numpy(), o.constraints(x)) <= TOL, (N, name)
            assert relerr(v.cpu().numpy(), o.jac_coord(x)) <= TOL, (N, name)
            if mode == 1:
                f = o.objective(x)
                assert abs(d.obj(xd) - f) <= TOL * max(1.0, abs(f))
                assert relerr(d.grad(xd).cpu().numpy(), o.gradient(x)) <= TOL
                # host-pointer entry points and the one-call iteration (first-order callbacks) take the same kernels
                c2, v2 = d.cons_jac(x)
                assert np.array_equal(c2, c.cpu().numpy()) and np.array_equal(v2, v.cpu().numpy())
                g3 = torch.full((d.dim_NLP_variables,), SENT, dtype=torch.float64, device="cuda")
                c3, v3, f3 = torch.full_like(c, SENT), torch.full_like(v, SENT), torch.full((1,), SENT, dtype=torch.float64, device="cuda")
                d.eval_all(xd, None, 1.0, f3, g3, c3, v3, None, sync=True)
                assert torch.equal(c3, c) and torch.equal(v3, v) and abs(float(f3[0]) - f) <= TOL * max(1.0, abs(f))
            # hess_structure / hess_coord: one stage-type point per control of the step (all three patterns; the optimized one is
            # the oracle's traced pattern), every entry written, 1e-10 on the backward-error scale of helpers.hess_err
            hp, hr = o.hess_pattern()
            hp2, hr2 = ct.DOCP_Hessian_pattern(d)
            assert d.nnzh == len(hr) and np.array_equal(hp, hp2) and np.array_equal(hr, hr2), (N, name)
            y = rng.standard_normal(d.dim_NLP_constraints)
            hv = torch.full((d.nnzh,), SENT, dtype=torch.float64, device="cuda")
            d.hess_coord(xd, torch.from_numpy(y).cuda(), 0.7, hv)
            assert not bool((hv == SENT).any())
            want, dropped = o.hess_coord(x, y, 0.7, return_dropped=True)
            assert dropped == (0, 0)
            assert hess_err(o, x, y, 0.7, hv.cpu().numpy(), ref=want) <= TOL, (N, name)
            if mode == 1:       # the one-call iteration with the Hessian: same values
                h3 = torch.full_like(hv, SENT)
                d.eval_all(xd, torch.from_numpy(y).cuda(), 0.7, None, None, None, None, h3, sync=True)
                assert torch.equal(h3, hv)
            d.close()


def test_control_steps_shards_compose(oracle_lib):
    """a control_steps handle restricted to a shard of the grid writes exactly its rows / CSC ranges"""
    import torch
    prob, cs, N = "goddard_all", 3, 301
    o = oracle_lib.OracleDOCP(prob, "midpoint", N, control_steps=cs)
    o.set_pattern_mode(1)
    x = _inputs(o, prob, cs, 1)
    xd = torch.from_numpy(x).cuda()
    c = torch.full((o.dim_NLP_constraints,), SENT, dtype=torch.float64, device="cuda")
    v = torch.full((o.jac_nnz(),), SENT, dtype=torch.float64, device="cuda")
    for a, b in ((0, 100), (100, 101), (101, 301)):
        d = ct.DOCP(prob, N, "midpoint", pattern="structural", device=0, control_steps=cs, steps=(a, b))
        d.cons_jac(xd, c, v)
        d.close()
    assert relerr(c.cpu().numpy(), o.constraints(x)) <= TOL and relerr(v.cpu().numpy(), o.jac_coord(x)) <= TOL


def test_control_steps_runtime_ocp_any_count(oracle_lib):
    """OCPs registered at run time compile the midpoint kernel for ANY number of controls per step (hiprtc); the compiled
    registry holds 1, 2 and 3.  manual_test.jl of the reference runs direct shooting with control_steps = 10."""
    import torch
    import jit_defs
    rt = jit_defs.twin("goddard")
    for cs in (4, 10):
        N = 33
        o = oracle_lib.OracleDOCP("goddard", "midpoint", N, control_steps=cs)
        o.set_pattern_mode(1)
        x = _inputs(o, "goddard", cs, cs)
        xd = torch.from_numpy(x).cuda()
        d = ct.DOCP(rt, N, "midpoint", pattern="structural", device=0, control_steps=cs)
        c, v = d.cons_jac(xd)
        assert relerr(c.cpu().numpy(), o.constraints(x)) <= TOL and relerr(v.cpu().numpy(), o.jac_coord(x)) <= TOL
        f = o.objective(x)
        assert abs(d.obj(xd) - f) <= TOL * max(1.0, abs(f)) and relerr(d.grad(xd).cpu().numpy(), o.gradient(x)) <= TOL
        d.close()
    # hess_coord: the Hessian kernel hiprtc builds for the run-time OCP (symbolic stage functions; the twin with a Lagrange cost
    # takes the second-order numbers, its quadrature points have times of their own); more than 3 controls per step: the
    # points of a step are summed before the emission (hess_sums_stages)
    for name, cs in (("goddard", 2), ("goddard", 3), ("double_integrator_path", 3), ("goddard", 4), ("goddard", 10),
                     ("double_integrator_path", 7), ("quadrotor", 4)):
        rt = jit_defs.twin(name)
        N = 40
        o = oracle_lib.OracleDOCP(name, "midpoint", N, control_steps=cs)
        x = _inputs(o, name, cs, cs)
        y = np.random.default_rng(cs).standard_normal(o.dim_NLP_constraints)
        for mode, pname in ((0, "manual"), (2, "optimized")):
            o.set_pattern_mode(mode)
            d = ct.DOCP(rt, N, "midpoint", pattern=pname, device=0, control_steps=cs)
            hp, hr = o.hess_pattern()
            hp2, hr2 = ct.DOCP_Hessian_pattern(d)
            assert np.array_equal(hp, hp2) and np.array_equal(hr, hr2)
            hv = d.hess_coord(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), 0.7)
            assert hess_err(o, x, y, 0.7, hv.cpu().numpy()) <= TOL, (name, cs, pname)
            d.close()
