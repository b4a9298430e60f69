"""GPU parity tests (run with -m gpu on an MI355X).  Every call goes through the C ABI of libctdirect_hip.so; the
checker is the CPU oracle (oracle/) and the 50-digit mpmath fixtures (tests/golden/).

Tolerance (BASELINE.json north_star):  max |gpu - ref| / max(1, |ref|) <= 1e-10 for constraint
values, Jacobian values and the objective; sparsity patterns bit-exact (checked on CPU in test_abi_cpu.py, re-checked
here for the handles used).
"""
import os

import numpy as np
import pytest

import ctdirect_jl_amd as ct
from helpers import TOL, bench_inputs, dense_on_pattern, describe, golden_files, hess_err, load_golden, relerr

pytestmark = pytest.mark.gpu

SENT = 666.666   # unwritten-output sentinel, idea from the reference's test/benchmark.jl:106,126


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _rel(a, b):
    return abs(a - b) / max(1.0, abs(b))


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: p.split("/")[-1][:-5])
def test_fixture_parity_host_pointers(path):
    """Every golden fixture through the host-pointer entry points (ctd_cons, ctd_jac_coord, ctd_cons_jac, ctd_obj)."""
    g = load_golden(path)
    for pattern in ("manual", "structural"):
        d = ct.DOCP(g["problem"], g["grid_size"], g["scheme"], time_grid=g["time_grid"], pattern=pattern, device=0,
                    control_steps=g.get("control_steps", 1))
        cp, rv = ct.DOCP_Jacobian_pattern(d)
        ref = dense_on_pattern(g["J"], cp, rv)
        c = np.full(d.dim_NLP_constraints, SENT)
        vals = np.full(d.nnzj, SENT)
        d.cons_jac(g["xu"], c, vals)
        assert not np.any(c == SENT) and not np.any(vals == SENT)
        assert relerr(c, g["c"]) <= TOL
        assert relerr(vals, ref) <= TOL
        assert relerr(d.cons(g["xu"]), g["c"]) <= TOL
        assert relerr(d.jac_coord(g["xu"]), ref) <= TOL
        assert _rel(d.obj(g["xu"]), g["objective"]) <= TOL
        assert relerr(d.grad(g["xu"]), g["gradient"]) <= TOL          # grad! against the mpmath gradient
        d.close()


def test_reference_known_answer_on_gpu():
    """The reference's own known-answer test (test/ci/test_discretization_stagewise.jl:79-100): on the exact feasible
    trajectory x = t^2, u = 2t, K = 2 t_ij the constraints vanish (atol 1e-12) and the objective is 4/3 (atol 1e-12)."""
    grid = [0.0, 0.2, 0.6, 1.0]
    for sch in ("gauss_legendre_2", "gauss_legendre_3"):
        d = ct.DOCP("stagewise_scalar", scheme=sch, time_grid=grid, device=0)
        blk, s = d.discretization._step_variables_block, d.discretization.stage
        T, cc = d.time.fixed_grid, d.discretization.butcher_c
        xu = np.zeros(d.dim_NLP_variables)
        for i in range(4):
            xu[i * blk] = T[i] ** 2
        for i in range(3):
            for j in range(s):
                tij = T[i] + cc[j] * (T[i + 1] - T[i])
                xu[i * blk + 1 + j] = 2 * tij
                xu[i * blk + 1 + s + j] = 2 * tij
        c = d.cons(xu)
        assert np.max(np.abs(c - d.bounds.con_l)) <= 1e-12 and np.max(np.abs(c - d.bounds.con_u)) <= 1e-12
        assert abs(d.obj(xu) - 4.0 / 3.0) <= 1e-12


PAIRS = [(p, s) for p in ct.PROBLEMS for s in ct.SCHEMES]


@pytest.mark.parametrize("prob,sch", PAIRS, ids=[f"{p}-{s}" for p, s in PAIRS])
def test_oracle_parity_midsize(oracle_lib, torch_cuda, prob, sch):
    """All registry problems x all schemes at sizes that exercise edge-only (N < 5), single-tile and multi-tile
    launches, uniform and ragged (non-uniform) grids, device-pointer entry points."""
    torch = torch_cuda
    rng = np.random.default_rng(11)
    for N, tg in ((1, None), (3, None), (5, None), (64, None), (257, None), (1000, None),
                  (101, np.cumsum(rng.uniform(0.2, 1.8, 102)))):
        o = oracle_lib.OracleDOCP(prob, sch, N, time_grid=tg)
        o.set_pattern_mode(1)
        d = ct.DOCP(prob, N, sch, time_grid=tg, pattern="structural", device=0)
        x = bench_inputs(describe(o, prob, sch), perturb=1e-3)
        xd = torch.from_numpy(x).cuda()
        c = torch.full((d.dim_NLP_constraints,), SENT, dtype=torch.float64, device="cuda")
        v = torch.full((d.nnzj,), SENT, dtype=torch.float64, device="cuda")
        d.cons_jac(xd, c, v)
        c, v = c.cpu().numpy(), v.cpu().numpy()
        assert not np.any(c == SENT) and not np.any(v == SENT)
        assert relerr(c, o.constraints(x)) <= TOL
        assert relerr(v, o.jac_coord(x)) <= TOL
        assert _rel(d.obj(xd), o.objective(x)) <= TOL
        if N <= 257:          # the oracle's gradient is one dual pass per variable: keep it to the smaller grids
            assert relerr(d.grad(xd).cpu().numpy(), o.gradient(x)) <= TOL
        # REFERENCE_MANUAL pattern: same values on the (possibly smaller) pattern
        dm = ct.DOCP(prob, N, sch, time_grid=tg, pattern="manual", device=0)
        if dm.nnzj == d.nnzj:
            assert np.array_equal(dm.jac_coord(xd).cpu().numpy(), v)
        else:
            import scipy.sparse as sp
            cps, rvs = ct.DOCP_Jacobian_pattern(d)
            cpm, rvm = ct.DOCP_Jacobian_pattern(dm)
            Js = sp.csc_matrix((v, rvs, cps), shape=(d.dim_NLP_constraints, d.dim_NLP_variables))
            Jm = sp.csc_matrix((dm.jac_coord(xd).cpu().numpy(), rvm, cpm), shape=Js.shape)
            mask = sp.csc_matrix((np.ones(len(rvm)), rvm, cpm), shape=Js.shape)
            assert abs(Js.multiply(mask) - Jm).max() == 0.0
            # entries only the STRUCTURAL pattern holds (trapeze dynamics rows x V, hazard H1): either they are true
            # nonzeros the manual pattern drops (free times / v-dependent dynamics) and the handle reports them, or
            # the problem does not depend on v there and their values are exactly zero
            extra = Js - Js.multiply(mask)
            # (block count for the trapeze hazard; for implicit Euler only the (path row, control) pairs the path functions
            # really couple are counted: at most the block)
            assert 0 <= dm.dropped_nonzeros() <= d.nnzj - dm.nnzj
            assert dm.dropped_nonzeros() in (0, d.nnzj - dm.nnzj) or sch == "euler_implicit"
            if dm.dropped_nonzeros() == 0:
                assert abs(extra).max() == 0.0
            else:
                assert abs(extra).max() > 0.0
        d.close()
        dm.close()


@pytest.mark.parametrize("tile,block", [(1, 64), (3, 128), (16, 256), (64, 256)])
def test_tile_and_block_shapes(oracle_lib, torch_cuda, monkeypatch, tile, block):
    """Results do not depend on the launch geometry (steps per workgroup, workgroup size)."""
    torch = torch_cuda
    monkeypatch.setenv("CTD_TILE", str(tile))
    monkeypatch.setenv("CTD_BLOCK", str(block))
    for prob, sch, N in (("goddard", "gauss_legendre_2", 333), ("goddard_all", "trapeze", 200),
                         ("quadrotor", "midpoint", 150), ("quadrotor12", "gauss_legendre_3", 40)):
        o = oracle_lib.OracleDOCP(prob, sch, N)
        o.set_pattern_mode(1)
        d = ct.DOCP(prob, N, sch, pattern="structural", device=0)
        assert d.launch_info()["steps_per_tile"] <= tile and d.launch_info()["block"] == block
        assert d.launch_info()["lds_bytes"] <= 80 * 1024          # a tile that does not fit is shrunk
        x = bench_inputs(describe(o, prob, sch), perturb=1e-3)
        c, v = d.cons_jac(torch.from_numpy(x).cuda())
        assert relerr(c.cpu().numpy(), o.constraints(x)) <= TOL
        assert relerr(v.cpu().numpy(), o.jac_coord(x)) <= TOL
        d.close()


def test_shards_compose_on_gpu(torch_cuda):
    """Time-step shards (multi-GPU partition) write disjoint rows / CSC ranges and together
    reproduce the unsharded evaluation bit for bit."""
    torch = torch_cuda
    for prob, sch, N in (("goddard", "gauss_legendre_3", 1000), ("goddard_all", "trapeze", 777),
                         ("double_integrator_path", "midpoint", 1203)):
        full = ct.DOCP(prob, N, sch, device=0)
        x = torch.from_numpy(bench_inputs(describe(full, prob, sch), perturb=1e-3)).cuda()
        cf, vf = full.cons_jac(x)
        c = torch.full_like(cf, SENT)
        v = torch.full_like(vf, SENT)
        cuts = [0, N // 3, N // 2 + 1, N]
        for a, b in zip(cuts[:-1], cuts[1:]):
            sh = ct.DOCP(prob, N, sch, steps=(a, b), device=0)
            c2 = torch.full_like(cf, SENT)
            v2 = torch.full_like(vf, SENT)
            sh.cons_jac(x, c2, v2)
            both = (c != SENT) & (c2 != SENT)
            cbn = N * (full.discretization._state_stage_eqs_block + full.discretization._step_pathcons_block)
            assert not bool(both[:cbn].any()) and not bool(((v != SENT) & (v2 != SENT)).any())
            assert torch.equal(c[both], c2[both])          # tail rows: written by every shard, identical
            rows = torch.nonzero(c2[:cbn] != SENT).flatten()
            assert int(rows.min()) == sh.shard.c_row_begin and int(rows.max()) == sh.shard.c_row_end - 1
            assert not bool((c2[cbn:] == SENT).any())
            c = torch.where(c2 != SENT, c2, c)
            v = torch.where(v2 != SENT, v2, v)
            sh.close()
        assert torch.equal(c, cf) and torch.equal(v, vf)
        full.close()


def test_gradient_full_size_directional(oracle_lib, torch_cuda):
    """grad! at the bench size: g . d against central differences of the GPU objective and, cheaply, against the oracle's
    directional derivative (one dual pass is O(N), unlike its full gradient)."""
    torch = torch_cuda
    for prob, sch, N in (("quadrotor", "gauss_legendre_3", 20000), ("double_integrator_path", "midpoint", 100000),
                         ("least_squares_with_constraint", "trapeze", 50000)):
        d = ct.DOCP(prob, N, sch, device=0)
        x = bench_inputs(describe(d, prob, sch), perturb=1e-3)
        g = d.grad(torch.from_numpy(x).cuda()).cpu().numpy()
        assert np.all(np.isfinite(g))
        rng = np.random.default_rng(3)
        dirv = rng.uniform(-0.5, 0.5, x.size)
        eps = 1e-6
        fd = (d.obj(x + eps * dirv) - d.obj(x - eps * dirv)) / (2 * eps)
        assert abs(g @ dirv - fd) / max(1.0, abs(fd)) <= 1e-6
        d.close()


def test_idempotent_and_async(torch_cuda):
    """Repeated and asynchronous launches give bit-identical outputs (no atomics, one writer per output)."""
    torch = torch_cuda
    d = ct.DOCP("goddard", 5000, "gauss_legendre_2", device=0)
    x = torch.from_numpy(bench_inputs(describe(d, "goddard", "gauss_legendre_2"), perturb=1e-3)).cuda()
    c0, v0 = d.cons_jac(x)
    c1 = torch.empty_like(c0)
    v1 = torch.empty_like(v0)
    for _ in range(5):
        d.cons_jac(x, c1, v1, sync=False)
    d.sync()
    assert torch.equal(c0, c1) and torch.equal(v0, v1)
    assert d.time_cons_jac(x, c1, v1, iters=3) > 0.0


# ---- BASELINE.json configurations at full size ------------------------------------------------------------------------
FULL = [("goddard", "trapeze", 100), ("goddard_all", "trapeze", 100),                 # config 1 / 1'
        ("goddard", "gauss_legendre_2", 10000),                                       # config 2 (bench workload)
        ("double_integrator_path", "midpoint", 100000)]                               # config 3


@pytest.mark.parametrize("prob,sch,N", FULL, ids=[f"{p}-{s}-{n}" for p, s, n in FULL])
def test_baseline_configs_full_parity(oracle_lib, torch_cuda, prob, sch, N):
    torch = torch_cuda
    o = oracle_lib.OracleDOCP(prob, sch, N)
    o.set_pattern_mode(1)
    d = ct.DOCP(prob, N, sch, pattern="structural", device=0)
    x = bench_inputs(describe(o, prob, sch), perturb=1e-3)
    xd = torch.from_numpy(x).cuda()
    c, v = d.cons_jac(xd)
    assert relerr(c.cpu().numpy(), o.constraints(x)) <= TOL
    assert relerr(v.cpu().numpy(), o.jac_coord(x)) <= TOL
    assert _rel(d.obj(xd), o.objective(x)) <= TOL


BIG = [("goddard", "gauss_legendre_3", 80000),      # config 4 (15.36 M Jacobian entries)
       ("quadrotor", "gauss_legendre_3", 20000),    # config 5' (reference's quadrotor, 28.6 M entries)
       ("quadrotor12", "gauss_legendre_3", 20000)]  # config 5  (59.06 M entries)


@pytest.mark.parametrize("prob,sch,N", BIG, ids=[f"{p}-{s}-{n}" for p, s, n in BIG])
def test_baseline_configs_full_size_properties(oracle_lib, torch_cuda, prob, sch, N):
    """At the largest sizes the oracle's coloured Jacobian is too slow / large for a unit test, so:
       (1) c(x) against the oracle at full size (cheap: O(N));
       (2) every output written (sentinel), outputs finite;
       (3) the Jacobian is the derivative of the constraints: J d == (c(x + e d) - c(x - e d)) / 2e along a random
           direction d, with J applied from (rows, cols, vals) -- a size-independent property;
       (4) windows of steps against the oracle run on the same window (non-uniform sub-grid, tf scaled by the window
           width): c rows and all non-V Jacobian entries of those steps."""
    torch = torch_cuda
    d = ct.DOCP(prob, N, sch, device=0)
    o = oracle_lib.OracleDOCP(prob, sch, N)
    x = bench_inputs(describe(o, prob, sch), perturb=1e-3)
    xd = torch.from_numpy(x).cuda()
    c = torch.full((d.dim_NLP_constraints,), SENT, dtype=torch.float64, device="cuda")
    v = torch.full((d.nnzj,), SENT, dtype=torch.float64, device="cuda")
    d.cons_jac(xd, c, v)
    assert not bool((c == SENT).any()) and not bool((v == SENT).any())
    assert bool(torch.isfinite(c).all()) and bool(torch.isfinite(v).all())
    assert relerr(c.cpu().numpy(), o.constraints(x)) <= TOL                              # (1)
    assert _rel(d.obj(xd), o.objective(x)) <= TOL
    if d.nnzj <= 16_000_000:          # config 4: the oracle's coloured Jacobian is still affordable -> direct parity
        assert relerr(v.cpu().numpy(), o.jac_coord(x)) <= TOL
    # EVERY Jacobian entry at full size (configs 5' and 5 included) against the oracle's block mode: the same `constraints`
    # template differentiated one time step at a time on dense local duals (bit-identical to its coloured passes on the
    # sizes where both run, tests/test_oracle_goldens.py), OpenMP over the steps
    blk_ref = o.cons_jac_block(x, min(16, os.cpu_count() or 1))
    assert blk_ref is not None
    assert relerr(c.cpu().numpy(), blk_ref[0]) <= TOL and relerr(v.cpu().numpy(), blk_ref[1]) <= TOL
    # (3) directional derivative: J d (scipy, from the pattern + GPU values) vs central differences of the GPU's c(x)
    import scipy.sparse as sp
    rows, cols = d.jac_structure()
    cp = np.zeros(d.dim_NLP_variables + 1, dtype=np.int64)
    np.cumsum(np.bincount(cols - 1, minlength=d.dim_NLP_variables), out=cp[1:])
    vh = v.cpu().numpy()
    J = sp.csc_matrix((vh, rows - 1, cp), shape=(d.dim_NLP_constraints, d.dim_NLP_variables))
    rng = np.random.default_rng(5)
    dirv = rng.uniform(-0.5, 0.5, d.dim_NLP_variables)
    Jd = J @ dirv
    eps = 1e-6
    xp = torch.from_numpy(x + eps * dirv).cuda()
    xm = torch.from_numpy(x - eps * dirv).cuda()
    fd = ((d.cons(xp) - d.cons(xm)) / (2 * eps)).cpu().numpy()
    err = float(np.max(np.abs(Jd - fd) / np.maximum(1.0, np.abs(fd))))
    assert err <= 2e-5, err          # truncation error of the difference quotient (third derivatives ~ 500^3 for Goddard)
    del J
    # (4) windows against the oracle
    blk = d.discretization._step_variables_block
    cb = d.discretization._state_stage_eqs_block + d.discretization._step_pathcons_block
    nv = d.dims.NLP_v
    W = 6
    ch = c.cpu().numpy()
    tau = d.time.normalized_grid
    for s0 in (1, N // 2, N - W - 1):
        sub = tau[s0:s0 + W + 1]
        ow = oracle_lib.OracleDOCP(prob, sch, time_grid=sub)
        xw = np.concatenate([x[s0 * blk:(s0 + W) * blk + d.dims.NLP_x], x[len(x) - nv:]])
        assert xw.size == ow.dim_NLP_variables
        width = sub[-1] - sub[0]
        xw[-1] = x[-1] * width if nv else 0.0          # tf' = tf * window width keeps every h_i (autonomous dynamics)
        cw = ow.constraints(xw)
        assert relerr(ch[s0 * cb:(s0 + W) * cb], cw[:W * cb]) <= 1e-9
        cpw, rvw = ow.jac_pattern()
        vw = ow.jac_coord(xw)
        for jl in range(blk, (W - 1) * blk):           # interior columns of the window (skip its first/last step)
            jg = s0 * blk + jl
            seg_g = vh[cp[jg]:cp[jg + 1]]
            seg_w = vw[cpw[jl]:cpw[jl + 1]]
            assert np.array_equal(rows[cp[jg]:cp[jg + 1]] - 1 - s0 * cb, rvw[cpw[jl]:cpw[jl + 1]])
            assert relerr(seg_g, seg_w) <= 1e-9
    d.close()


def test_device_resident_iteration_without_host_syncs(oracle_lib, torch_cuda):
    """all five callbacks of one solver iteration enqueued on the stream with no host synchronisation in between
    (ctd_*_dev_async), results read back once at the end -- the call pattern of a device-resident solver loop"""
    torch = torch_cuda
    prob, sch, N = "goddard_all", "gauss_legendre_2", 500
    o = oracle_lib.OracleDOCP(prob, sch, N)
    o.set_pattern_mode(1)
    d = ct.DOCP(prob, N, sch, pattern="structural", device=0)
    x = bench_inputs(describe(o, prob, sch), perturb=1e-2)
    y = np.cos(0.41 * np.arange(o.dim_NLP_constraints))
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    f = torch.full((1,), SENT, dtype=torch.float64, device="cuda")
    g = torch.full((d.dim_NLP_variables,), SENT, dtype=torch.float64, device="cuda")
    c = torch.full((d.dim_NLP_constraints,), SENT, dtype=torch.float64, device="cuda")
    v = torch.full((d.nnzj,), SENT, dtype=torch.float64, device="cuda")
    hv = torch.full((d.nnzh,), SENT, dtype=torch.float64, device="cuda")
    for _ in range(3):
        d.obj_async(xd, f)
        d.grad(xd, g, sync=False)
        d.cons_jac(xd, c, v, sync=False)
        d.hess_coord(xd, yd, 1.0, hv, sync=False)
    d.sync()
    assert _rel(float(f[0]), o.objective(x)) <= TOL
    assert relerr(g.cpu().numpy(), o.gradient(x)) <= TOL
    assert relerr(c.cpu().numpy(), o.constraints(x)) <= TOL and relerr(v.cpu().numpy(), o.jac_coord(x)) <= TOL
    assert relerr(hv.cpu().numpy(), o.hess_coord(x, y, 1.0)) <= TOL


def test_whole_iteration_as_one_hip_graph(oracle_lib, torch_cuda):
    """the five callbacks of an iteration recorded once into a HIP graph (torch.cuda.CUDAGraph on ROCm) and replayed on new
    iterates: one graph launch instead of eight kernel launches"""
    torch = torch_cuda
    prob, sch, N = "goddard", "gauss_legendre_2", 2000
    o = oracle_lib.OracleDOCP(prob, sch, N)
    d = ct.DOCP(prob, N, sch, device=0)
    x0 = bench_inputs(describe(o, prob, sch), perturb=1e-2)
    y = np.cos(0.41 * np.arange(o.dim_NLP_constraints))
    xd, yd = torch.from_numpy(x0).cuda(), torch.from_numpy(y).cuda()
    f = torch.zeros(1, dtype=torch.float64, device="cuda")
    g = torch.zeros(d.dim_NLP_variables, dtype=torch.float64, device="cuda")
    c = torch.zeros(d.dim_NLP_constraints, dtype=torch.float64, device="cuda")
    v = torch.zeros(d.nnzj, dtype=torch.float64, device="cuda")
    hv = torch.zeros(d.nnzh, dtype=torch.float64, device="cuda")

    def iteration():
        d.obj_async(xd, f)
        d.grad(xd, g, sync=False)
        d.cons_jac(xd, c, v, sync=False)
        d.hess_coord(xd, yd, 1.0, hv, sync=False)

    iteration()                       # first use: uploads the Hessian tables (not capturable)
    d.sync()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(graph, stream=side):
        d.set_stream(side)
        iteration()
    d.set_stream(None)
    rng = np.random.default_rng(3)
    for _ in range(3):                # new iterate in the same buffers, one graph launch
        x = x0 + 1e-3 * rng.standard_normal(x0.size)
        xd.copy_(torch.from_numpy(x))
        graph.replay()
        torch.cuda.synchronize()
        assert _rel(float(f[0]), o.objective(x)) <= TOL and relerr(g.cpu().numpy(), o.gradient(x)) <= TOL
        assert relerr(c.cpu().numpy(), o.constraints(x)) <= TOL and relerr(v.cpu().numpy(), o.jac_coord(x)) <= TOL
        assert relerr(hv.cpu().numpy(), o.hess_coord(x, y, 1.0)) <= TOL


@pytest.mark.gpu
def test_page_locked_host_arrays_take_the_same_path(torch_cuda):
    """ctd_host_alloc memory (ct.pinned_empty) through the host-pointer entry points: same values as pageable arrays"""
    d = ct.DOCP("goddard", 300, "gauss_legendre_2", device=0)
    x = bench_inputs(describe(d, "goddard", "gauss_legendre_2"), perturb=1e-3)
    c, v = d.cons_jac(x)
    xp, cp, vp = ct.pinned_empty(x.size), ct.pinned_empty(c.size), ct.pinned_empty(v.size)
    xp[:] = x
    cp[:] = vp[:] = 777.0
    d.cons_jac(xp, cp, vp)
    assert np.array_equal(cp, c) and np.array_equal(vp, v)
    g = ct.pinned_empty(x.size)
    assert np.array_equal(d.grad(xp, g), d.grad(x))
    del xp, cp, vp, g
    d.close()


@pytest.mark.gpu
def test_c_program_evaluates_on_the_gpu(torch_cuda, tmp_path):
    """examples/cabi_demo.c (plain C against the ABI) runs one fused evaluation on device 0; same numbers as the Python mirror"""
    import subprocess
    from test_abi_cpu import _build_c_demo
    out = subprocess.run([_build_c_demo(tmp_path), "gpu"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    d = ct.DOCP("goddard", 100, "gauss_legendre_2", device=0)
    x0 = ct.initial_guess(d)
    c, v = d.cons_jac(x0)
    assert f"objective {d.obj(x0):.6f}  c[0] {c[0]:.6e}  vals[0] {v[0]:.6e}" in out.stdout, out.stdout
    # the multi-device entry points from plain C: two shards, iterate sent from shard 0, residual stitched on every shard
    assert "differs from the single-device c in 0 of 904 rows" in out.stdout, out.stdout
    # the Jacobian in CSR order from plain C (ctd_jac_csr + a handle with value_order = CTD_ORDER_CSR): every entry found in its row, same value
    assert "CSR order: rowptr[ncon] 11128, values differ from the CSC values in 0 of 11128 entries" in out.stdout, out.stdout
    d.close()


@pytest.mark.gpu
def test_distinct_handles_are_usable_from_concurrent_threads(torch_cuda):
    """"handle is not thread-safe, distinct handles are" (include/ctdirect_hip.h): four host threads, each with its own handle
    (own stream) -- two of them created concurrently for a run-time OCP, i.e. through the hiprtc cache -- evaluate at the
    same time through the host-pointer entry points and get the single-threaded results"""
    import threading
    import jit_defs
    jobs = [("goddard", "gauss_legendre_2", 400), ("quadrotor", "midpoint", 150), (jit_defs.twin("goddard"), "trapeze", 300),
            (jit_defs.twin("goddard"), "gauss_legendre_3", 200)]
    want, got, errs = {}, {}, []
    for k, (prob, sch, N) in enumerate(jobs):
        d = ct.DOCP(prob, N, sch, device=0, stream="own")
        x = 0.4 + 0.1 * np.sin(np.arange(d.dim_NLP_variables))
        y = np.cos(0.3 * np.arange(d.dim_NLP_constraints))
        want[k] = (x, y, d.cons_jac(x), d.grad(x), d.hess_coord(x, y, 0.5))
        d.close()

    def work(k):
        try:
            prob, sch, N = jobs[k]
            d = ct.DOCP(prob, N, sch, device=0, stream="own")
            x, y = want[k][0], want[k][1]
            for _ in range(40):
                out = (d.cons_jac(x), d.grad(x), d.hess_coord(x, y, 0.5))
            got[k] = out
            d.close()
        except Exception as e:      # noqa: BLE001
            errs.append((k, repr(e)))

    th = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for k in range(len(jobs)):
        (c0, v0), g0, h0 = want[k][2], want[k][3], want[k][4]
        (c1, v1), g1, h1 = got[k]
        assert np.array_equal(c0, c1) and np.array_equal(v0, v1) and np.array_equal(g0, g1) and np.array_equal(h0, h1), k


@pytest.mark.gpu
@pytest.mark.parametrize("prob", ["goddard", "goddard_all", "double_integrator_path", "quadrotor", "quadrotor12",
                                  "double_integrator_freet0tf", "least_squares_with_constraint"])
def test_optimized_pattern_values_on_gpu(oracle_lib, torch_cuda, prob):
    """CTD_PATTERN_OPTIMIZED (the sparsity the reference's default backend detects, src/collocation.jl:131-134): constraints,
    Jacobian values and Hessian values on the smaller patterns equal the oracle's on its traced patterns; every entry written."""
    torch = torch_cuda
    rng = np.random.default_rng(11)
    for sch, N in (("trapeze", 257), ("midpoint", 1000), ("euler", 64), ("gauss_legendre_2", 257), ("gauss_legendre_3", 64),
                   ("gauss_legendre_2_constant_control", 5), ("gauss_legendre_3", 3)):
        o = oracle_lib.OracleDOCP(prob, sch, N)
        o.set_pattern_mode(2)
        d = ct.DOCP(prob, N, sch, pattern="optimized", device=0)
        assert d.nnzj == o.jac_nnz() and d.nnzh == len(o.hess_pattern()[1])
        x = bench_inputs(describe(o, prob, sch), perturb=1e-2)
        y = rng.standard_normal(o.dim_NLP_constraints)
        xd = torch.from_numpy(x).cuda()
        c = torch.full((d.dim_NLP_constraints,), SENT, dtype=torch.float64, device="cuda")
        v = torch.full((d.nnzj,), SENT, dtype=torch.float64, device="cuda")
        h = torch.full((d.nnzh,), SENT, dtype=torch.float64, device="cuda")
        d.cons_jac(xd, c, v)
        d.hess_coord(xd, torch.from_numpy(y).cuda(), 0.7, h)
        assert not bool((c == SENT).any()) and not bool((v == SENT).any()) and not bool((h == SENT).any())
        assert relerr(c.cpu().numpy(), o.constraints(x)) <= TOL and relerr(v.cpu().numpy(), o.jac_coord(x)) <= TOL
        want, dropped = o.hess_coord(x, y, 0.7, return_dropped=True)
        assert dropped == (0, 0) and hess_err(o, x, y, 0.7, h.cpu().numpy(), ref=want) <= TOL      # (sign-changing y: backward-error scale)
        d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("prob", ["goddard_all", "double_integrator_path", "quadrotor"])
def test_implicit_euler_structural_pattern_is_complete(oracle_lib, torch_cuda, prob):
    """Implicit Euler evaluates the path constraints of node i with U_{i-1} (euler.jl:59-72) while the reference's pattern lists
    U_i (euler.jl:231).  The manual pattern is bug-compatible (and says how many true nonzeros it drops); the structural and
    optimized patterns hold the (path_i, U_{i-1}) entries: J d equals central differences of c along random directions, and
    the values equal the oracle's."""
    torch = torch_cuda
    import scipy.sparse as sp
    N = 300
    man = ct.DOCP(prob, N, "euler_implicit", device=-1)
    # (path row, control) pairs the path functions really couple (goddard_all: rows u and x1+x2+x3+u+v; the double integrator's
    # path constraint q + 0.1 w^2 has none; the quadrotor's cos(theta) cos(phi) has none)
    assert man.dropped_nonzeros() == (N - 1) * {"goddard_all": 2, "double_integrator_path": 0, "quadrotor": 0}[prob]
    for pattern, mode in (("structural", 1), ("optimized", 2)):
        d = ct.DOCP(prob, N, "euler_implicit", pattern=pattern, device=0)
        assert d.dropped_nonzeros() == 0
        o = oracle_lib.OracleDOCP(prob, "euler_implicit", N)
        o.set_pattern_mode(mode)
        x = bench_inputs(describe(o, prob, "euler_implicit"), perturb=1e-2)
        xd = torch.from_numpy(x).cuda()
        c, v = d.cons_jac(xd)
        assert relerr(c.cpu().numpy(), o.constraints(x)) <= TOL and relerr(v.cpu().numpy(), o.jac_coord(x)) <= TOL
        rows, cols = d.jac_structure()
        J = sp.coo_matrix((v.cpu().numpy(), (rows - 1, cols - 1)), shape=(d.dim_NLP_constraints, d.dim_NLP_variables)).tocsr()
        rng = np.random.default_rng(3)
        for _ in range(3):
            dirv = rng.uniform(-0.5, 0.5, d.dim_NLP_variables)
            eps = 1e-6
            fd = ((d.cons(torch.from_numpy(x + eps * dirv).cuda()) - d.cons(torch.from_numpy(x - eps * dirv).cuda())) / (2 * eps)).cpu().numpy()
            err = float(np.max(np.abs(J @ dirv - fd) / np.maximum(1.0, np.abs(fd))))
            assert err <= 2e-5, (pattern, err)
        d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("prob,sch,N", [("goddard", "gauss_legendre_2", 10000), ("goddard_all", "trapeze", 1001),
                                        ("double_integrator_path", "midpoint", 3000), ("quadrotor", "gauss_legendre_3", 300),
                                        ("least_squares_with_constraint", "gauss_legendre_2_constant_control", 64),
                                        ("double_integrator_freet0tf", "euler_implicit", 5), ("goddard", "midpoint", 3)])
def test_whole_iteration_in_two_launches(oracle_lib, torch_cuda, prob, sch, N):
    """ctd_eval_all_dev_async: objective, gradient, constraints + Jacobian values and Hessian values of one iterate from the
    horizontally fused kernel (+ finish kernel).  f, g, c and the Jacobian values are bit-identical to the single-purpose
    kernels (same bodies, same rounding mode), the Hessian values agree to rounding (its single-purpose build contracts
    multiply-adds); everything matches the oracle; subsets of the outputs work too."""
    torch = torch_cuda
    o = oracle_lib.OracleDOCP(prob, sch, N)
    o.set_pattern_mode(1)
    d = ct.DOCP(prob, N, sch, pattern="structural", device=0)
    x = bench_inputs(describe(o, prob, sch), perturb=1e-2)
    y = np.cos(0.41 * np.arange(o.dim_NLP_constraints))
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    new = lambda n: torch.full((n,), SENT, dtype=torch.float64, device="cuda")      # noqa: E731
    f, g, c, v, h = new(1), new(d.dim_NLP_variables), new(d.dim_NLP_constraints), new(d.nnzj), new(d.nnzh)
    for _ in range(2):
        d.eval_all(xd, yd, 0.9, f, g, c, v, h)
    d.sync()
    f2, g2, h2 = new(1), new(d.dim_NLP_variables), new(d.nnzh)
    d.obj_async(xd, f2)
    d.grad(xd, g2, sync=False)
    c2, v2 = d.cons_jac(xd)
    d.hess_coord(xd, yd, 0.9, h2)
    assert torch.equal(f, f2) and torch.equal(g, g2) and torch.equal(c, c2) and torch.equal(v, v2)
    assert relerr(h.cpu().numpy(), h2.cpu().numpy()) <= 1e-11
    assert _rel(float(f[0]), o.objective(x)) <= TOL and relerr(g.cpu().numpy(), o.gradient(x)) <= TOL
    assert relerr(c.cpu().numpy(), o.constraints(x)) <= TOL and relerr(v.cpu().numpy(), o.jac_coord(x)) <= TOL
    assert hess_err(o, x, y, 0.9, h.cpu().numpy()) <= TOL
    # subsets: only first-order callbacks; only the Hessian
    g3, c3, v3 = new(d.dim_NLP_variables), new(d.dim_NLP_constraints), new(d.nnzj)
    d.eval_all(xd, None, 1.0, None, g3, c3, v3, None, sync=True)
    assert torch.equal(g3, g2) and torch.equal(c3, c2) and torch.equal(v3, v2)
    h3 = new(d.nnzh)
    d.eval_all(xd, yd, 0.9, hvals=h3, sync=True)
    assert torch.equal(h3, h)
    d.close()
