"""GPU tests of the run-time OCP path (run with -m gpu on an MI355X): OCPs handed over as expressions (ctd_register_ocp),
kernels compiled with hiprtc at ctd_create and launched through the same C-ABI entry points.

  * registry twins: a problem of the compiled registry restated as expressions gives the same constraints, Jacobian,
    objective, gradient and Hessian as the built-in handle (<= 1e-10; operation order inside an expression may differ
    from the hand-written functor by a rounding) and as the oracle;
  * a problem that exists only as expressions (forced Van der Pol, free final time, time- and parameter-dependent
    dynamics / cost / path constraints, nonlinear boundary constraint, sqrt) against an on-the-fly 50-digit mpmath evaluation
    of the reference's formulas (tests/golden/gen_golden.py machinery) on every scheme."""
import numpy as np
import pytest
from mpmath import mpf

import ctdirect_jl_amd as ct
import jit_defs
from jit_defs import gg
from helpers import TOL, bench_inputs, describe, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


PAIRS = [(p, s) for p in sorted(jit_defs.TWINS) for s in ct.SCHEMES]


@pytest.mark.parametrize("prob,sch", PAIRS, ids=[f"{p}-{s}" for p, s in PAIRS])
def test_runtime_twin_matches_registry_and_oracle(oracle_lib, torch_cuda, prob, sch):
    torch = torch_cuda
    if prob == "quadrotor" and sch not in ("midpoint", "gauss_legendre_2", "gauss_legendre_3", "trapeze"):
        pytest.skip("quadrotor twin: four schemes are enough (compile time)")
    if prob == "quadrotor12" and sch not in ("midpoint", "gauss_legendre_3", "trapeze"):
        pytest.skip("quadrotor12 twin: three schemes are enough (compile time)")
    rt = jit_defs.twin(prob)
    rng = np.random.default_rng(21)
    for N in (4, 41):
        # structural pattern: on the reference's manual trapeze pattern a coloured Jacobian is wrong (hazard H1)
        a, b = ct.DOCP(prob, N, sch, pattern="structural", device=0), ct.DOCP(rt, N, sch, pattern="structural", device=0)
        o = oracle_lib.OracleDOCP(prob, sch, N)
        o.set_pattern_mode(1)
        x = bench_inputs(describe(o, prob, sch), perturb=1e-2)
        y = rng.standard_normal(o.dim_NLP_constraints)
        xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
        ca, va = a.cons_jac(xd)
        cb, vb = b.cons_jac(xd)
        assert relerr(cb.cpu().numpy(), ca.cpu().numpy()) <= TOL and relerr(vb.cpu().numpy(), va.cpu().numpy()) <= TOL
        assert relerr(cb.cpu().numpy(), o.constraints(x)) <= TOL and relerr(vb.cpu().numpy(), o.jac_coord(x)) <= TOL
        assert abs(b.obj(xd) - o.objective(x)) <= TOL * max(1.0, abs(o.objective(x)))
        assert relerr(b.grad(xd).cpu().numpy(), o.gradient(x)) <= TOL
        hb = b.hess_coord(xd, yd, 0.7).cpu().numpy()
        # the structural-sparsity probe walks the expressions of a run-time OCP and the functor of a registry entry: same
        # nonzero structure, hence the same number of terms and of second-order eval lanes -- except at stage points, where
        # a run-time OCP runs its symbolically differentiated stage function on ONE lane
        ia, ib = a.hess_launch_info(), b.hess_launch_info()
        assert all(ia[k] == ib[k] for k in ("path_lanes", "boundary_lanes", "segment_terms", "edge_entries")), (ia, ib)
        assert ib["stage_lanes"] == 1, (ia, ib)
        assert relerr(hb, a.hess_coord(xd, yd, 0.7).cpu().numpy()) <= TOL
        assert relerr(hb, o.hess_coord(x, y, 0.7)) <= TOL
        # host-pointer entry points take the same path
        c2, v2 = b.cons_jac(x)
        assert np.array_equal(c2, cb.cpu().numpy()) and np.array_equal(v2, vb.cpu().numpy())
        a.close(); b.close()


def _mp_reference(P, scheme, N, xu, y, sigma, control_steps=1):
    """c, dense Jacobian, objective, gradient (first-order dense dual) and the Hessian of the Lagrangian (sparse
    second-order number) of the mpmath restatement"""
    import gen_golden_hess as gh
    d = gg.Docp(P, scheme, N=N, control_steps=control_steps)
    nvar = d.nvar
    gg.Du.NV = nvar
    z = []
    for j, v in enumerate(xu):
        der = [mpf(0)] * nvar
        der[j] = mpf(1)
        z.append(gg.Du(mpf(float(v)), der))
    c = [gg.Du.lift(e) for e in d.constraints(z)]
    obj = gg.Du.lift(d.objective(z))
    cval = np.array([float(e.v) for e in c])
    J = np.array([[float(g) for g in e.d] for e in c])
    grad = np.array([float(g) for g in obj.d])
    first = (gg.Du, gg.dexp, gg.dsin, gg.dcos)
    try:
        gg.Du, gg.dexp, gg.dsin, gg.dcos = gh.Du2, gh.dexp2, gh.dsin2, gh.dcos2
        z2 = [gh.Du2(mpf(float(v)), {j: mpf(1)}) for j, v in enumerate(xu)]
        lag = mpf(sigma) * gh.Du2.lift(d.objective(z2))
        for r, e in enumerate(d.constraints(z2)):
            lag = lag + mpf(float(y[r])) * gh.Du2.lift(e)
        H = {k: float(v) for k, v in lag.h.items()}
    finally:
        gg.Du, gg.dexp, gg.dsin, gg.dcos = first
    return d, cval, J, float(obj.v), grad, H


@pytest.mark.parametrize("sch", list(ct.SCHEMES))
def test_expression_only_problem_against_mpmath(torch_cuda, sch):
    torch = torch_cuda
    name = "vdp_rt" if "vdp_rt" in ct.PROBLEMS else ct.register_ocp("vdp_rt", **jit_defs.VDP)
    N = 5
    d = ct.DOCP(name, N, sch, pattern="structural", device=0)
    rng = np.random.default_rng(33)
    x = 0.4 + 0.3 * rng.standard_normal(d.dim_NLP_variables)
    x[-2:] = [1.3, 2.1]                                            # v = (mu, tf)
    y = rng.standard_normal(d.dim_NLP_constraints)
    md, cref, Jref, fref, gref, Href = _mp_reference(jit_defs.VdpMp(), sch, N, x, y, 0.6)
    assert (md.nvar, md.ncon) == (d.dim_NLP_variables, d.dim_NLP_constraints)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    c, vals = d.cons_jac(xd)
    assert relerr(c.cpu().numpy(), cref) <= TOL
    rows, cols = d.jac_structure()
    assert relerr(vals.cpu().numpy(), Jref[rows - 1, cols - 1]) <= TOL
    pat = set(zip(rows - 1, cols - 1))
    euler = sch.startswith("euler")      # the reference's Euler patterns leave true nonzeros out (tests/test_oracle_goldens.py)
    assert euler or all((r, cc) in pat for r, cc in zip(*np.nonzero(Jref)))  # structural pattern holds every true nonzero
    assert abs(d.obj(xd) - fref) <= TOL * max(1.0, abs(fref))
    assert relerr(d.grad(xd).cpu().numpy(), gref) <= TOL
    hr, hc = d.hess_structure()
    hv = d.hess_coord(xd, yd, 0.6).cpu().numpy()
    want = np.array([Href.get((int(r) - 1, int(cc) - 1), 0.0) for r, cc in zip(hr, hc)])
    assert relerr(hv, want) <= TOL
    hpat = set(zip(hr - 1, hc - 1))
    dropped = [k for k, v in Href.items() if v != 0.0 and k not in hpat]
    assert euler or not dropped, dropped
    d.close()


@pytest.mark.parametrize("cs", [2, 3, 5])
def test_expression_only_problem_with_several_controls_per_step(torch_cuda, cs):
    """The direct-shooting layout of a problem whose dynamics AND Lagrange cost read the time, with a free final time and a
    parameter: the dynamics of every control's point at the step's midpoint, its quadrature point at t_i + (j - 1/2) h / cs
    (midpoint.jl:57,110) -- all callbacks against the 50-digit restatement (cs = 5: the points of a step are summed before
    the Hessian's emission)."""
    torch = torch_cuda
    name = "vdp_rt" if "vdp_rt" in ct.PROBLEMS else ct.register_ocp("vdp_rt", **jit_defs.VDP)
    N = 6
    d = ct.DOCP(name, N, "midpoint", pattern="structural", device=0, control_steps=cs)
    rng = np.random.default_rng(40 + cs)
    x = 0.4 + 0.3 * rng.standard_normal(d.dim_NLP_variables)
    x[-2:] = [1.3, 2.1]                                            # v = (mu, tf)
    y = rng.standard_normal(d.dim_NLP_constraints)
    md, cref, Jref, fref, gref, Href = _mp_reference(jit_defs.VdpMp(), "midpoint", N, x, y, 0.6, control_steps=cs)
    assert (md.nvar, md.ncon) == (d.dim_NLP_variables, d.dim_NLP_constraints)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    c, vals = d.cons_jac(xd)
    rows, cols = d.jac_structure()
    assert relerr(c.cpu().numpy(), cref) <= TOL and relerr(vals.cpu().numpy(), Jref[rows - 1, cols - 1]) <= TOL
    assert abs(d.obj(xd) - fref) <= TOL * max(1.0, abs(fref)) and relerr(d.grad(xd).cpu().numpy(), gref) <= TOL
    hr, hc = d.hess_structure()
    hv = d.hess_coord(xd, yd, 0.6).cpu().numpy()
    want = np.array([Href.get((int(r) - 1, int(cc) - 1), 0.0) for r, cc in zip(hr, hc)])
    assert relerr(hv, want) <= TOL
    hpat = set(zip(hr - 1, hc - 1))
    assert not [k for k, v in Href.items() if v != 0.0 and k not in hpat]
    d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sch", ["gauss_legendre_2", "gauss_legendre_3", "euler", "euler_implicit", "midpoint", "trapeze",
                                 "gauss_legendre_2_constant_control"])
def test_lagrange_problem_without_boundary_rows(torch_cuda, sch):
    """bc = 0, p = 0, Lagrange cost: the leftover block of the stagewise / Euler patterns (hazard H2) is then a REAL extra
    entry (last row, column n) of the reference pattern; the engine lists it (bug-compatible pattern) and writes the exact
    partial there -- 0 unless the last row really depends on X_1 -- beside the true values everywhere else"""
    torch = torch_cuda
    name = "pendulum_rt" if "pendulum_rt" in ct.PROBLEMS else ct.register_ocp("pendulum_rt", **jit_defs.PENDULUM)
    for N in (1, 4):
        d = ct.DOCP(name, N, sch, device=0)                            # reference (manual) pattern
        rng = np.random.default_rng(5)
        x = 0.3 + 0.4 * rng.standard_normal(d.dim_NLP_variables)
        y = rng.standard_normal(d.dim_NLP_constraints)
        md, cref, Jref, fref, gref, Href = _mp_reference(jit_defs.PendulumMp(), sch, N, x, y, 0.9)
        assert (md.nvar, md.ncon) == (d.dim_NLP_variables, d.dim_NLP_constraints)
        rows, cols = d.jac_structure()
        pat = set(zip(rows - 1, cols - 1))
        leftover = (d.dim_NLP_constraints - 1, 1)                      # 0-based (ncon, n)
        has = sch in ("gauss_legendre_2", "gauss_legendre_3", "euler", "euler_implicit")
        assert (leftover in pat) == (has or Jref[leftover] != 0.0), (sch, N)
        xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
        c, vals = d.cons_jac(xd)
        assert relerr(c.cpu().numpy(), cref) <= TOL
        assert relerr(vals.cpu().numpy(), Jref[rows - 1, cols - 1]) <= TOL
        assert all((r, cc) in pat for r, cc in zip(*np.nonzero(Jref))) or sch == "euler_implicit"
        assert abs(d.obj(xd) - fref) <= TOL * max(1.0, abs(fref))
        assert relerr(d.grad(xd).cpu().numpy(), gref) <= TOL
        hr, hc = d.hess_structure()
        hv = d.hess_coord(xd, yd, 0.9).cpu().numpy()
        want = np.array([Href.get((int(r) - 1, int(cc) - 1), 0.0) for r, cc in zip(hr, hc)])
        assert relerr(hv, want) <= TOL
        d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sch", ["gauss_legendre_2", "midpoint", "trapeze", "euler_implicit"])
def test_every_function_of_the_grammar_against_mpmath(torch_cuda, sch):
    """log, tan, atan, tanh and abs (value, first and second derivative) through the generated functor, in dynamics, both
    costs, a path row and boundary rows, against the 50-digit evaluation"""
    torch = torch_cuda
    name = "funcs_rt" if "funcs_rt" in ct.PROBLEMS else ct.register_ocp("funcs_rt", **jit_defs.FUNCS)
    N = 4
    d = ct.DOCP(name, N, sch, pattern="structural", device=0)
    rng = np.random.default_rng(12)
    x = 0.3 * rng.standard_normal(d.dim_NLP_variables)
    x[-1] = 0.8
    y = rng.standard_normal(d.dim_NLP_constraints)
    md, cref, Jref, fref, gref, Href = _mp_reference(jit_defs.FuncsMp(), sch, N, x, y, 0.7)
    assert (md.nvar, md.ncon) == (d.dim_NLP_variables, d.dim_NLP_constraints)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    c, vals = d.cons_jac(xd)
    rows, cols = d.jac_structure()
    assert relerr(c.cpu().numpy(), cref) <= TOL and relerr(vals.cpu().numpy(), Jref[rows - 1, cols - 1]) <= TOL
    assert abs(d.obj(xd) - fref) <= TOL * max(1.0, abs(fref)) and relerr(d.grad(xd).cpu().numpy(), gref) <= TOL
    hr, hc = d.hess_structure()
    want = np.array([Href.get((int(r) - 1, int(cc) - 1), 0.0) for r, cc in zip(hr, hc)])
    assert relerr(d.hess_coord(xd, yd, 0.7).cpu().numpy(), want) <= TOL
    d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sch", ["gauss_legendre_3", "gauss_legendre_2_constant_control", "midpoint", "trapeze", "euler"])
def test_four_optimisation_variables_with_both_times_free(torch_cuda, sch):
    """nv = 4 (the ABI's maximum), t0 = v3 and tf = v2: all ten V x V Hessian entries, the composite time directions and the
    K x V terms, against the 50-digit evaluation"""
    torch = torch_cuda
    name = "fourv_rt" if "fourv_rt" in ct.PROBLEMS else ct.register_ocp("fourv_rt", **jit_defs.FOURV)
    for N in (3, 26):
        d = ct.DOCP(name, N, sch, pattern="structural", device=0)
        rng = np.random.default_rng(8)
        x = 0.4 + 0.3 * rng.standard_normal(d.dim_NLP_variables)
        x[-4:] = [1.2, 1.9, 0.15, 0.7]                              # v = (a, tf, t0, b)
        y = rng.standard_normal(d.dim_NLP_constraints)
        if N == 3:
            md, cref, Jref, fref, gref, Href = _mp_reference(jit_defs.FourVMp(), sch, N, x, y, 0.8)
            assert (md.nvar, md.ncon) == (d.dim_NLP_variables, d.dim_NLP_constraints)
        xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
        c, vals = d.cons_jac(xd)
        rows, cols = d.jac_structure()
        hr, hc = d.hess_structure()
        hv = d.hess_coord(xd, yd, 0.8).cpu().numpy()
        if N == 3:
            assert relerr(c.cpu().numpy(), cref) <= TOL and relerr(vals.cpu().numpy(), Jref[rows - 1, cols - 1]) <= TOL
            assert abs(d.obj(xd) - fref) <= TOL * max(1.0, abs(fref)) and relerr(d.grad(xd).cpu().numpy(), gref) <= TOL
            want = np.array([Href.get((int(r) - 1, int(cc) - 1), 0.0) for r, cc in zip(hr, hc)])
            assert relerr(hv, want) <= TOL
            assert sum(1 for r, cc in zip(hr, hc) if r > d.dim_NLP_variables - 4 and cc > d.dim_NLP_variables - 4) == 10
        else:
            # larger grid: the symbolic stage functions against the second-order forward numbers of the generic path
            # (and the symbolic first derivatives of the constraint / Jacobian kernel against the forward duals)
            import os
            os.environ["CTD_HESS_SYM"] = os.environ["CTD_DYN_SYM"] = "0"
            try:
                name0 = "fourv_rt0" if "fourv_rt0" in ct.PROBLEMS else ct.register_ocp("fourv_rt0", **jit_defs.FOURV)
            finally:
                os.environ.pop("CTD_HESS_SYM", None)
                os.environ.pop("CTD_DYN_SYM", None)
            assert "HAS_SYM_DYN = false" in ct.ocp_source(name0) and "HAS_SYM_DYN = true" in ct.ocp_source(name)
            d0 = ct.DOCP(name0, N, sch, pattern="structural", device=0)
            assert d0.hess_launch_info()["stage_lanes"] > 1 and d.hess_launch_info()["stage_lanes"] == 1
            assert relerr(hv, d0.hess_coord(xd, yd, 0.8).cpu().numpy()) <= TOL
            c0, v0 = d0.cons_jac(xd)
            assert relerr(c.cpu().numpy(), c0.cpu().numpy()) <= TOL and relerr(vals.cpu().numpy(), v0.cpu().numpy()) <= TOL
            d0.close()
        d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("prob,sch", [("quadrotor", "gauss_legendre_2"), ("goddard_all", "trapeze"), ("quadrotor", "midpoint")])
def test_sparse_and_dense_eval_blocks_agree(torch_cuda, monkeypatch, prob, sch):
    """Run-time OCPs store only the structurally nonzero partials of their dynamics (DynNZ specialisation in the generated functor);
    CTD_DENSE_EVAL=1 at registration keeps the dense blocks.  Same values, entry for entry, in all three patterns -- and the sparse
    variant really is sparse (smaller tiles' LDS, a DynNZ specialisation in its source)."""
    torch = torch_cuda
    a = jit_defs.twin(prob)
    monkeypatch.setenv("CTD_DENSE_EVAL", "1")
    name = prob + "_dense_rt"
    b = name if name in ct.PROBLEMS else ct.register_ocp(name, **jit_defs.TWINS[prob])
    monkeypatch.delenv("CTD_DENSE_EVAL")
    assert "struct DynNZ<UserOCP>" in ct.ocp_source(a) and "struct DynNZ<UserOCP>" not in ct.ocp_source(b)
    for pattern in ("manual", "structural", "optimized"):
        da, db = ct.DOCP(a, 57, sch, pattern=pattern, device=0), ct.DOCP(b, 57, sch, pattern=pattern, device=0)
        x = torch.from_numpy(bench_inputs(describe(da, prob, sch), perturb=1e-2)).cuda()
        ca, va = da.cons_jac(x)
        cb, vb = db.cons_jac(x)
        assert da.nnzj == db.nnzj and torch.equal(ca, cb)
        assert relerr(va.cpu().numpy(), vb.cpu().numpy()) <= 1e-15            # (a structural zero is 0.0 either way, possibly -0.0)
        ia, ib = da.launch_info(), db.launch_info()
        assert ia["lds_bytes"] / max(1, ia["steps_per_tile"]) < ib["lds_bytes"] / max(1, ib["steps_per_tile"])
        da.close(); db.close()
