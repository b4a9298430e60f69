"""Maximum sizes (round 4): a transcription whose Jacobian AND Hessian hold more than 2^31 values each -- Goddard, Gauss-Legendre 3,
N = 2^24 = 16 777 216 time steps: nvar 251 658 244, ncon 201 326 596, nnzj 3 221 225 500 (25.8 GB of values), nnzh 2 264 924 185
(18.1 GB) -- evaluated on one MI355X in both value orders.  No oracle finishes at this size, so the check is size-independent
and EXACT: with a power-of-two number of steps, the same step block in every step and an autonomous OCP every regular step
computes the same numbers (t_f = 1/4: the node times t_i = tau_i t_f and their differences h are exact), and a 16-step
transcription with t_f scaled by 2^-20 has the same step length h bit for bit, so

  * every value of the big Jacobian equals, bit for bit, the value of the 16-step Jacobian at the corresponding position (first
    step, a regular step, last step, tail rows; the d/dt_f entries after an exact scaling by 2^-20), which in turn is checked
    against the oracle;
  * no output position is left unwritten (NaN sentinel), in 64-bit index territory included (positions beyond 2^31 and 2^32);
  * c(x): the same, row for row;  Hessian values: periodic over the regular steps, every position written.

Creating the handle is O(1) in N on the host apart from the time table (bounds are written on demand, the V columns of the CSC
tables are generated piecewise): ~3 s at this size."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here)
sys.path.insert(0, os.path.dirname(here))

LOG2N = int(os.environ.get("CTD_MAX_LOG2N", "24"))
PROB, SCH = "goddard", "gauss_legendre_3"
NM = 16                                     # the small twin


def _x(blockvals, N, final, tf):
    return np.concatenate([np.tile(blockvals, N), final, [tf]])


def _chunks_equal_first(v, base, seg, count, first, chunk=1 << 18):
    """v[base + k seg : base + (k + 1) seg] == first for k in [0, count), bit for bit (NaN never equals)"""
    k = 0
    while k < count:
        c = min(chunk, count - k)
        blk = v[base + k * seg: base + (k + c) * seg].view(c, seg)
        if not bool((blk == first).all()):
            bad = int((blk != first).any(dim=1).nonzero()[0])
            return k + bad
        k += c
    return -1


@pytest.mark.parametrize("order", ["csr", "csc"])
def test_more_than_2_to_31_jacobian_values(order):
    import ctdirect_jl_amd as ct
    from helpers import TOL, bench_inputs, describe, relerr
    from oracle.oracle import OracleDOCP
    N = 1 << LOG2N
    scale = float(NM) / float(N)                       # exact power of two
    mid = ct.DOCP(PROB, NM, SCH, device=0, value_order=order)
    L = mid.discretization
    blk, cb, n = L._step_variables_block, L._state_stage_eqs_block + L._step_pathcons_block, mid.dims.NLP_x
    xb = bench_inputs(describe(mid, PROB, SCH), perturb=1e-3)
    B, F, tf = xb[3 * blk:4 * blk].copy(), xb[3 * blk:3 * blk + n].copy(), 0.25           # t_f a power of two: every t_i = tau_i t_f, hence every h, is exact
    x_mid, x_big = _x(B, NM, F, tf * scale), _x(B, N, F, tf)
    # --- the 16-step twin: engine against the oracle (CSC order on the oracle's side)
    c_mid, v_mid = mid.cons_jac(x_mid)
    o = OracleDOCP(PROB, SCH, NM)
    vref = o.jac_coord(x_mid)
    colptr, rowval = ct.DOCP_Jacobian_pattern(mid)
    cols = np.repeat(np.arange(len(colptr) - 1), np.diff(colptr))
    if order == "csr":
        perm = np.lexsort((cols, rowval))
        vref, ent_rows, ent_cols = vref[perm], rowval[perm], cols[perm]
    else:
        ent_rows, ent_cols = rowval, cols
    assert relerr(c_mid, o.constraints(x_mid)) <= TOL and relerr(v_mid, vref) <= TOL
    nnz_m, ncon_m, nvar_m = mid.nnzj, mid.dim_NLP_constraints, mid.dim_NLP_variables
    v_off_m = nvar_m - 1
    # position map small -> big: an entry belongs to a step s (by its row in CSR order, by its column in CSC order; V-column entries by
    # their row) or to the tail; steps 0, 1 map to themselves, steps >= NM - 2 and the tail to the END, regular steps to step 1's image
    big = ct.DOCP(PROB, N, SCH, device=0, value_order=order)
    nnz_b, ncon_b, nvar_b = big.nnzj, big.dim_NLP_constraints, big.dim_NLP_variables
    assert nnz_b > 2 ** 31 or LOG2N < 24
    per_step = (nnz_b - nnz_m) // (N - NM)
    assert nnz_m + per_step * (N - NM) == nnz_b
    xd = torch.from_numpy(x_big).cuda()
    c = torch.full((ncon_b,), float("nan"), dtype=torch.float64, device="cuda")
    v = torch.full((nnz_b,), float("nan"), dtype=torch.float64, device="cuda")
    big.cons_jac(xd, c, v)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        big.cons_jac(xd, c, v, sync=False)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"\n{PROB}/{SCH} N=2^{LOG2N} ({order}): nnzj {nnz_b}, {8e-9 * (nvar_b + ncon_b + nnz_b):.2f} GB per evaluation in {ms:.3f} ms = "
          f"{8e-9 * (nvar_b + ncon_b + nnz_b) / ms:.2f} TB/s")
    # c: rows of steps periodic, equal to the twin's; tail rows equal
    c_m = torch.from_numpy(np.asarray(c_mid)).cuda()
    assert bool((c[:cb] == c_m[:cb]).all()) and _chunks_equal_first(c, cb, cb, N - 2, c_m[cb:2 * cb]) == -1
    assert bool((c[(N - 1) * cb:] == c_m[(NM - 1) * cb:]).all())
    # Jacobian values
    vm = torch.from_numpy(np.asarray(v_mid)).cuda()
    isv = torch.from_numpy((ent_cols == v_off_m) & (ent_rows < NM * cb)).cuda()          # d/dt_f of a step's rows: proportional to 1 / N
    vm_scaled = torch.where(isv, vm * scale, vm)
    if order == "csr":
        rowptr, _ = ct.DOCP_Jacobian_csr(mid)
        s1, s2, sl = int(rowptr[cb]), int(rowptr[2 * cb]), int(rowptr[(NM - 1) * cb])
        seg = s2 - s1
        assert seg == per_step and sl == s1 + (NM - 2) * seg
        assert bool((v[:s1] == vm_scaled[:s1]).all()), "first step"
        bad = _chunks_equal_first(v, s1, seg, N - 2, vm_scaled[s1:s2])
        assert bad == -1, f"regular step {bad + 1} differs from step 1"
        assert bool((v[nnz_b - (nnz_m - sl):] == vm_scaled[sl:]).all()), "last step + tail rows"
    else:
        nv_col = int(colptr[v_off_m])                                  # first value of the V column
        s1, s2, sl = int(colptr[blk]), int(colptr[2 * blk]), int(colptr[(NM - 1) * blk])
        seg = s2 - s1
        assert sl == s1 + (NM - 2) * seg
        vr = (int(colptr[v_off_m + 1]) - nv_col - int(((ent_cols == v_off_m) & (ent_rows >= NM * cb)).sum())) // NM
        assert seg + vr == per_step
        assert bool((v[:s1] == vm_scaled[:s1]).all()), "first step's columns"
        bad = _chunks_equal_first(v, s1, seg, N - 2, vm_scaled[s1:s2])
        assert bad == -1, f"regular step {bad + 1} differs from step 1"
        nvb = nnz_b - (nnz_m - nv_col) - (N - NM) * vr                 # first value of the big V column
        assert bool((v[s1 + (N - 2) * seg:nvb] == vm_scaled[sl:nv_col]).all()), "last step's columns + final state"
        bad = _chunks_equal_first(v, nvb, vr, N, vm_scaled[nv_col + vr:nv_col + 2 * vr]) if vr else -1
        assert bad == -1, f"V column: step {bad}"
        assert bool((v[nvb + N * vr:] == vm_scaled[nv_col + NM * vr:]).all()), "V column tail rows"
    del v, c
    big.close()
    mid.close()
    torch.cuda.empty_cache()


def test_more_than_2_to_31_hessian_values():
    import ctdirect_jl_amd as ct
    from helpers import bench_inputs, describe
    N = 1 << LOG2N
    mid = ct.DOCP(PROB, NM, SCH, device=0)
    L = mid.discretization
    blk, cb, n = L._step_variables_block, L._state_stage_eqs_block + L._step_pathcons_block, mid.dims.NLP_x
    xb = bench_inputs(describe(mid, PROB, SCH), perturb=1e-3)
    B, F = xb[3 * blk:4 * blk].copy(), xb[3 * blk:3 * blk + n].copy()
    hcolptr, _ = ct.DOCP_Hessian_pattern(mid)
    s1, s2 = int(hcolptr[blk]), int(hcolptr[2 * blk])
    seg = s2 - s1
    assert int(hcolptr[(NM - 1) * blk]) == s1 + (NM - 2) * seg
    nnzh_m = mid.nnzh
    mid.close()
    big = ct.DOCP(PROB, N, SCH, device=0)
    nnzh = big.nnzh
    assert nnzh > 2 ** 31 or LOG2N < 24
    assert nnzh_m + seg * (N - NM) == nnzh
    xd = torch.from_numpy(_x(B, N, F, 0.25)).cuda()
    ystep = 0.5 + 0.3 * np.cos(np.arange(cb))
    tail = big.dim_NLP_constraints - N * cb
    yd = torch.cat([torch.from_numpy(ystep).cuda().repeat(N), torch.full((tail,), 0.7, dtype=torch.float64, device="cuda")])
    h = torch.full((nnzh,), float("nan"), dtype=torch.float64, device="cuda")
    big.hess_coord(xd, yd, 0.9, h)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        big.hess_coord(xd, yd, 0.9, h, sync=False)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    gb = 8e-9 * (big.dim_NLP_variables + big.dim_NLP_constraints + nnzh)
    print(f"\n{PROB}/{SCH} N=2^{LOG2N} Hessian: nnzh {nnzh}, {gb:.2f} GB per evaluation in {ms:.3f} ms = {gb / ms:.2f} TB/s")
    assert not bool(torch.isnan(h[:s1]).any()) and not bool(torch.isnan(h[s1 + (N - 2) * seg:]).any())
    bad = _chunks_equal_first(h, s1, seg, N - 2, h[s1:s2].clone())
    assert bad == -1, f"regular step {bad + 1} differs from step 1"
    assert not bool(torch.isnan(h[s1:s2]).any())
    big.close()


def _long_grid_case(prob, sch, N, orders):
    import ctdirect_jl_amd as ct
    from helpers import TOL, bench_inputs, describe, relerr
    from oracle.oracle import OracleDOCP
    o = OracleDOCP(prob, sch, N)
    x = bench_inputs(describe(o, prob, sch), perturb=1e-3)
    ref = o.cons_jac_block(x, min(16, os.cpu_count() or 1))
    if ref is None:                        # (implicit Euler has no block mode: the coloured passes)
        ref = (o.constraints(x), o.jac_coord(x))
    for order in orders:
        d = ct.DOCP(prob, N, sch, device=0, value_order=order)
        li = d.launch_info()
        assert li["block"] == 512 and li["steps_per_tile"] >= 64 and li["lds_bytes"] <= (74 if sch in ("midpoint", "euler", "euler_implicit") else 64) * 1024, li
        c = torch.full((d.dim_NLP_constraints,), float("nan"), dtype=torch.float64, device="cuda")
        v = torch.full((d.nnzj,), float("nan"), dtype=torch.float64, device="cuda")
        d.cons_jac(torch.from_numpy(x).cuda(), c, v)
        vref = ref[1]
        if order == "csr":                 # the same entries by rows
            rows, cols = d.jac_structure()
            cp, rv = o.jac_pattern()
            oc = np.repeat(np.arange(len(cp) - 1), np.diff(cp))
            perm = np.lexsort((oc, rv))
            assert np.array_equal(rows - 1, rv[perm]) and np.array_equal(cols - 1, oc[perm])
            vref = vref[perm]
        assert relerr(c.cpu().numpy(), ref[0]) <= TOL, (prob, sch, order)
        assert relerr(v.cpu().numpy(), vref) <= TOL, (prob, sch, order)
        d.close()


def test_long_grid_geometry_against_the_oracle():
    """Grids of 8 rounds of resident workgroups and more take another launch geometry (eight waves per workgroup, the largest tile
    whose records fit 64 KiB: `ctd_create`, CTD_LONG_GRID).  2^20 steps of the bench OCP: EVERY value of c and of the Jacobian, both
    value orders, against the oracle's block mode (the reference's `constraints` template differentiated one step at a time on dense
    local duals, OpenMP over the steps)."""
    _long_grid_case("goddard", "gauss_legendre_2", 1 << 20, ("csc", "csr"))


@pytest.mark.parametrize("prob,sch,log2n", [("goddard", "gauss_legendre_3", 17), ("double_integrator_freet0tf", "gauss_legendre_3", 17),
                                            ("goddard_all", "gauss_legendre_2", 17), ("double_integrator_path", "gauss_legendre_2", 17),
                                            ("goddard_all", "gauss_legendre_1", 17), ("double_integrator_path", "gauss_legendre_3", 17),
                                            ("double_integrator_path", "midpoint", 19), ("goddard", "midpoint", 19), ("goddard_all", "midpoint", 19),
                                            ("goddard", "euler", 19), ("double_integrator_path", "euler_implicit", 18)])
def test_long_grid_geometry_other_ocps(prob, sch, log2n, monkeypatch):
    """the same geometry forced on 2^17 / 2^19 steps (CTD_LONG_GRID_ROUNDS = 1: every grid of more than one round) for the other narrow
    OCPs, both drivers (direct: Goddard, double integrators; staged: goddard_all) and the midpoint scheme"""
    monkeypatch.setenv("CTD_LONG_GRID_ROUNDS", "1")
    _long_grid_case(prob, sch, 1 << log2n, ("csc", "csr"))
