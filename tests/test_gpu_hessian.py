"""GPU parity tests of the Hessian-of-the-Lagrangian row (run with -m gpu on an MI355X).  Every call goes through the C
ABI (ctd_hess_structure / ctd_hess_coord / ctd_hess_coord_dev); the checker is the CPU oracle's sparse second-order sweep
(oracle/dual2.hpp) and the 50-digit mpmath fixtures tests/golden/hess_*.json.

Tolerance: max |gpu - ref| / max(1, |ref|) <= 1e-10 on the values; the pattern (lower triangle of DOCP_Hessian_pattern)
is bit-exact (CPU: tests/test_hessian_cpu.py; re-checked here for the handles used)."""
import os

import numpy as np
import pytest

import ctdirect_jl_amd as ct
from helpers import TOL, bench_inputs, describe, hess_golden_files, hess_on_pattern, load_hess_golden, relerr

pytestmark = pytest.mark.gpu
SENT = 666.666


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


@pytest.mark.parametrize("path", hess_golden_files(), ids=lambda p: p.split("/")[-1][:-5])
def test_hessian_fixture_parity(torch_cuda, path):
    """Every Hessian fixture through the host-pointer and the device-pointer entry points."""
    torch = torch_cuda
    g = load_hess_golden(path)
    cs = g.get("control_steps", 1)
    prob = g["problem"]
    if cs > 3:          # (the compiled registry holds 1 - 3 controls per step; more: the problem's run-time twin, built by hiprtc)
        import jit_defs
        prob = jit_defs.twin(prob)
    d = ct.DOCP(prob, g["grid_size"], g["scheme"], time_grid=g["time_grid"], device=0, control_steps=cs)
    cp, rv = ct.DOCP_Hessian_pattern(d)
    want, outside = hess_on_pattern(g["H"], cp, rv)
    # (the Euler patterns of the reference, euler.jl:270-355, leave some true nonzeros out: test_oracle_goldens.py)
    assert not outside or g["scheme"].startswith("euler")
    vals = np.full(d.nnzh, SENT)
    d.hess_coord(g["xu"], g["y"], g["obj_weight"], vals)
    assert not np.any(vals == SENT)
    assert relerr(vals, want) <= TOL
    xd, yd = torch.from_numpy(g["xu"]).cuda(), torch.from_numpy(g["y"]).cuda()
    vd = torch.full((d.nnzh,), SENT, dtype=torch.float64, device="cuda")
    d.hess_coord(xd, yd, g["obj_weight"], vd)
    assert np.array_equal(vd.cpu().numpy(), vals)        # same kernel, same bits
    d.close()


PAIRS = [(p, s) for p in ct.PROBLEMS for s in ct.SCHEMES]


@pytest.mark.parametrize("prob,sch", PAIRS, ids=[f"{p}-{s}" for p, s in PAIRS])
def test_hessian_oracle_parity_midsize(oracle_lib, torch_cuda, prob, sch):
    """All registry problems x all schemes: edge-only sizes (N < 5), single-tile and multi-tile launches, ragged grid."""
    torch = torch_cuda
    rng = np.random.default_rng(13)
    for N, tg in ((1, None), (3, None), (5, None), (64, None), (257, None), (1000, None),
                  (29, np.cumsum(rng.uniform(0.5, 1.5, 30)))):
        o = oracle_lib.OracleDOCP(prob, sch, N, time_grid=tg)
        d = ct.DOCP(prob, N, sch, time_grid=tg, device=0)
        cp, rv = o.hess_pattern()
        cp2, rv2 = ct.DOCP_Hessian_pattern(d)
        assert np.array_equal(cp, cp2) and np.array_equal(rv, rv2)
        x = bench_inputs(describe(o, prob, sch), perturb=1e-2)
        y = rng.standard_normal(o.dim_NLP_constraints) * rng.choice([1e-2, 1.0, 10.0], o.dim_NLP_constraints)
        xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
        for sigma in (1.0, -0.3):
            vd = torch.full((d.nnzh,), SENT, dtype=torch.float64, device="cuda")
            d.hess_coord(xd, yd, sigma, vd)
            v = vd.cpu().numpy()
            assert not np.any(v == SENT)
            assert relerr(v, o.hess_coord(x, y, sigma)) <= TOL
        d.close()


def test_hessian_tile_shapes_and_idempotence(oracle_lib, torch_cuda, monkeypatch):
    """tile sizes forced through CTD_HESS_TILE give the same values; repeated / asynchronous launches are idempotent"""
    torch = torch_cuda
    prob, sch, N = "goddard_all", "midpoint", 203
    o = oracle_lib.OracleDOCP(prob, sch, N)
    x = bench_inputs(describe(o, prob, sch), perturb=1e-2)
    y = np.cos(np.arange(o.dim_NLP_constraints) * 0.37)
    want = o.hess_coord(x, y, 0.9)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    ref = None
    for tile in ("1", "3", "7", "32"):
        monkeypatch.setenv("CTD_HESS_TILE", tile)
        d = ct.DOCP(prob, N, sch, device=0)
        assert d.hess_launch_info()["steps_per_tile"] == int(tile)
        v1 = d.hess_coord(xd, yd, 0.9)
        v2 = torch.full_like(v1, SENT)
        d.hess_coord(xd, yd, 0.9, v2, sync=False)
        d.hess_coord(xd, yd, 0.9, v2, sync=False)
        d.sync()
        assert torch.equal(v1, v2)
        assert relerr(v1.cpu().numpy(), want) <= TOL
        d.close()


@pytest.mark.parametrize("prob,sch,N", [("goddard", "gauss_legendre_3", 1000), ("quadrotor", "gauss_legendre_3", 700),
                                        ("quadrotor12", "gauss_legendre_2", 300)])
def test_hessian_walk_modes_on_gpu(oracle_lib, torch_cuda, monkeypatch, prob, sch, N):
    """the walk over the step-periodic segment (all entries / entries with terms after a zero fill / entries with terms +
    explicit zero stores) and the number of edge workgroups do not change a bit of the result; every entry is written"""
    torch = torch_cuda
    o = oracle_lib.OracleDOCP(prob, sch, N)
    x = bench_inputs(describe(o, prob, sch), perturb=1e-2)
    y = np.cos(np.arange(o.dim_NLP_constraints) * 0.37)
    want = o.hess_coord_block(x, y, 0.9) if hasattr(o, "hess_coord_block") and N > 500 else o.hess_coord(x, y, 0.9)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    ref = None
    for compact, eb in (("0", "1"), ("1", "1"), ("2", "1"), ("", ""), ("1", "5")):
        if compact:
            monkeypatch.setenv("CTD_HESS_COMPACT", compact)
            monkeypatch.setenv("CTD_HESS_EDGE_BLOCKS", eb)
        else:
            monkeypatch.delenv("CTD_HESS_COMPACT", raising=False)
            monkeypatch.delenv("CTD_HESS_EDGE_BLOCKS", raising=False)
        d = ct.DOCP(prob, N, sch, device=0)
        v = torch.full((d.nnzh,), SENT, dtype=torch.float64, device="cuda")
        d.hess_coord(xd, yd, 0.9, v)
        assert not bool((v == SENT).any())
        if ref is None:
            ref = v.clone()
            assert relerr(v.cpu().numpy(), want) <= TOL
        assert torch.equal(v, ref)
        d.close()


@pytest.mark.parametrize("prob,sch", [("goddard_all", "midpoint"), ("goddard", "gauss_legendre_2"), ("quadrotor", "trapeze"),
                                      ("double_integrator_freet0tf", "euler_implicit"), ("goddard_all", "gauss_legendre_3_constant_control")])
def test_hessian_shards_compose_on_gpu(oracle_lib, torch_cuda, prob, sch):
    """time-step shards (what each rank of a multi-GPU run evaluates): entries outside the V x V block are written by exactly
    one shard, bit-identical to the full evaluation; the shards' V x V partials add up to the full entries"""
    torch = torch_cuda
    N = 301
    full = ct.DOCP(prob, N, sch, device=0)
    o = oracle_lib.OracleDOCP(prob, sch, N)
    x = bench_inputs(describe(o, prob, sch), perturb=1e-2)
    y = np.cos(0.37 * np.arange(o.dim_NLP_constraints))
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    ref = full.hess_coord(xd, yd, 0.9)
    vv = torch.as_tensor(full.hess_shard_info()[2], dtype=torch.long, device="cuda")
    notvv = torch.ones(full.nnzh, dtype=torch.bool, device="cuda")
    notvv[vv] = False
    acc = torch.full((full.nnzh,), SENT, dtype=torch.float64, device="cuda")
    vvsum = torch.zeros(len(vv), dtype=torch.float64, device="cuda")
    for a, b in ((0, 100), (100, 101), (101, 301)):
        sh = ct.DOCP(prob, N, sch, device=0, steps=(a, b))
        part = torch.full((full.nnzh,), SENT, dtype=torch.float64, device="cuda")
        sh.hess_coord(xd, yd, 0.9, part)
        wrote = (part != SENT) & notvv
        assert not bool((wrote & (acc != SENT)).any())
        acc[wrote] = part[wrote]
        vvsum += part[vv]
        sh.close()
    assert torch.equal(acc[notvv], ref[notvv])
    assert relerr(vvsum.cpu().numpy(), ref[vv].cpu().numpy()) <= 1e-12
    assert relerr(ref.cpu().numpy(), o.hess_coord(x, y, 0.9)) <= TOL


FULL = [("goddard", "trapeze", 100), ("goddard_all", "trapeze", 100), ("goddard", "gauss_legendre_2", 10000),
        ("double_integrator_path", "midpoint", 20000), ("goddard", "gauss_legendre_3", 80000)]


@pytest.mark.parametrize("prob,sch,N", FULL, ids=[f"{p}-{s}-{n}" for p, s, n in FULL])
def test_hessian_baseline_configs_direct_parity(oracle_lib, torch_cuda, prob, sch, N):
    """BASELINE.json configurations at full size against the oracle's sweep.

    (a) multipliers of one sign: plain 1e-10 criterion.
    (b) multipliers that change sign from row to row make single entries sums of cancelling terms (Goddard's drag gives
        d2f/dr2 ~ beta^2 D/m ~ 1e5 per stage, the stages cancel to ~20): there two double-precision evaluations cannot
        agree to 1e-10 of the RESULT -- at N = 10000 the 50-digit value of the worst entry is -19.918244250016, the oracle
        gives ...56929 (3.5e-10 off) and the engine ...44982 (2.5e-10 off).  The criterion is therefore taken relative to the
        magnitude of what is summed, |H|(|y|, |obj_weight|), the usual backward-error scale."""
    torch = torch_cuda
    o = oracle_lib.OracleDOCP(prob, sch, N)
    d = ct.DOCP(prob, N, sch, device=0)
    x = bench_inputs(describe(o, prob, sch), perturb=1e-3)
    xd = torch.from_numpy(x).cuda()
    wave = np.sin(0.7 * np.arange(o.dim_NLP_constraints) + 0.3)
    y = 0.6 + 0.4 * wave
    v = d.hess_coord(xd, torch.from_numpy(y).cuda(), 0.75).cpu().numpy()
    assert relerr(v, o.hess_coord(x, y, 0.75)) <= TOL                                                   # (a)
    v = d.hess_coord(xd, torch.from_numpy(wave).cuda(), -0.75).cpu().numpy()
    ref = o.hess_coord(x, wave, -0.75)
    scale = np.maximum(1.0, np.maximum(np.abs(ref), np.abs(o.hess_coord(x, np.abs(wave), 0.75))))
    assert float(np.max(np.abs(v - ref) / scale)) <= TOL                                                # (b)
    d.close()


STEP_CASES = [(p, s) for p in ("goddard", "goddard_all", "quadrotor", "quadrotor12", "double_integrator_path", "least_squares_with_constraint",
                                "double_integrator_freet0tf", "estimate_rotation_rate", "estimate_initial_condition", "stagewise_scalar")
              for s in ("gauss_legendre_2", "gauss_legendre_3", "gauss_legendre_2_constant_control", "gauss_legendre_3_constant_control")]


@pytest.mark.parametrize("prob,sch", STEP_CASES, ids=[f"{p}-{s}" for p, s in STEP_CASES])
def test_hessian_step_kernel(oracle_lib, torch_cuda, monkeypatch, prob, sch):
    """The lane-per-step kernel (Gauss-Legendre schemes with 2 / 3 stages, ctd_hess_step.hpp), forced at small
    sizes (CTD_HESS_STEP=2; by default it takes over from 9 000 / 28 000 steps): every entry written, values against the
    oracle and against the tile kernel, full grids and shards (partial waves, irregular first / last steps), ragged time grid."""
    torch = torch_cuda
    rng = np.random.default_rng(31)
    for N, tg in ((6, None), (64, None), (200, None), (333, np.cumsum(rng.uniform(0.5, 1.5, 334)))):
        o = oracle_lib.OracleDOCP(prob, sch, N, time_grid=tg)
        x = bench_inputs(describe(o, prob, sch), perturb=1e-2)
        y = rng.standard_normal(o.dim_NLP_constraints)
        want = o.hess_coord(x, y, 0.7)
        # random multipliers of both signs make single entries sums of cancelling terms: criterion relative to the magnitude of
        # what is summed (as in test_hessian_full_size_properties)
        scale = np.maximum(1.0, np.maximum(np.abs(want), np.abs(o.hess_coord(x, np.abs(y), 0.7))))
        xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
        res = {}
        for mode in ("0", "2"):
            monkeypatch.setenv("CTD_HESS_STEP", mode)
            d = ct.DOCP(prob, N, sch, time_grid=tg, device=0)
            v = torch.full((d.nnzh,), SENT, dtype=torch.float64, device="cuda")
            d.hess_coord(xd, yd, 0.7, v)
            assert not bool((v == SENT).any())
            res[mode] = v.cpu().numpy()
            assert float(np.max(np.abs(res[mode] - want) / scale)) <= TOL
            d.close()
        assert float(np.max(np.abs(res["2"] - res["0"]) / scale)) <= 1e-12
        # the other sparsity patterns: the same assembly function, positions mapped by (row, column)
        if N == 200:
            for pat in ("structural", "optimized"):
                vals = {}
                for mode in ("0", "2"):
                    monkeypatch.setenv("CTD_HESS_STEP", mode)
                    d = ct.DOCP(prob, N, sch, time_grid=tg, device=0, pattern=pat)
                    light = prob not in ("quadrotor", "quadrotor12")         # (the quadrotors keep the tile kernel: registers)
                    if prob != "estimate_initial_condition":                 # (linear dynamics: no regular segment in the optimized pattern)
                        assert d.hess_kernel_info()["kernel"] == ("step" if mode == "2" and light else "tile")
                    v = torch.full((d.nnzh,), SENT, dtype=torch.float64, device="cuda")
                    d.hess_coord(xd, yd, 0.7, v)
                    assert not bool((v == SENT).any())
                    vals[mode] = v.cpu().numpy()
                    d.close()
                sc = np.maximum(1.0, np.abs(vals["0"]))
                assert float(np.max(np.abs(vals["2"] - vals["0"]) / sc)) <= 1e-10
        # shards: disjoint outside the V x V block, V x V partials add up
        if N >= 64:
            monkeypatch.setenv("CTD_HESS_STEP", "2")
            full = ct.DOCP(prob, N, sch, time_grid=tg, device=0)
            _, _, vv = full.hess_shard_info()
            notvv = np.ones(full.nnzh, dtype=bool)
            notvv[vv] = False
            acc = np.full(full.nnzh, SENT)
            vvsum = np.zeros(len(vv))
            for a, b in ((0, 70), (70, 71), (71, N)) if N > 71 else ((0, 31), (31, N)):
                sh = ct.DOCP(prob, N, sch, time_grid=tg, device=0, steps=(a, b))
                part = torch.full((full.nnzh,), SENT, dtype=torch.float64, device="cuda")
                sh.hess_coord(xd, yd, 0.7, part)
                pn = part.cpu().numpy()
                wrote = (pn != SENT) & notvv
                assert not np.any(wrote & (acc != SENT))
                acc[wrote] = pn[wrote]
                vvsum += pn[vv]
                sh.close()
            assert not np.any(acc[notvv] == SENT)
            assert relerr(acc[notvv], res["2"][notvv]) <= 1e-14
            assert relerr(vvsum, res["2"][vv]) <= 1e-12
            full.close()


BIG = [("double_integrator_path", "midpoint", 100000), ("quadrotor", "gauss_legendre_3", 20000),
       ("quadrotor12", "gauss_legendre_3", 20000)]


@pytest.mark.parametrize("prob,sch,N", BIG, ids=[f"{p}-{s}-{n}" for p, s, n in BIG])
def test_hessian_full_size_properties(torch_cuda, prob, sch, N):
    """Where the oracle's sweep takes too long for a unit test: every entry written and finite, and the Hessian is the
    derivative of the Lagrangian's gradient -- H d == (g(x + e d) - g(x - e d)) / 2e with g = obj_weight grad f + J'y
    from the first-order callbacks of the same handle (a size-independent property)."""
    import scipy.sparse as sp
    torch = torch_cuda
    d = ct.DOCP(prob, N, sch, device=0)
    nvar, ncon = d.dim_NLP_variables, d.dim_NLP_constraints
    x = bench_inputs(describe(d, prob, sch), perturb=1e-3)
    rng = np.random.default_rng(17)
    y = rng.uniform(-1.0, 1.0, ncon)
    sigma = 0.8
    vd = torch.full((d.nnzh,), SENT, dtype=torch.float64, device="cuda")
    d.hess_coord(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), sigma, vd)
    assert not bool((vd == SENT).any()) and bool(torch.isfinite(vd).all())
    v = vd.cpu().numpy()
    # EVERY entry at full size against the oracle's block mode (the second-order sweep of `lagrangian_hessian` one time step
    # at a time, OpenMP over the steps; equal to the full sweep to 3e-15 where both run, tests/test_oracle_goldens.py).  Random
    # multipliers of both signs make single entries sums of cancelling terms: criterion relative to the magnitude of what is
    # summed, as in test_hessian_baseline_configs_direct_parity (b)
    from oracle.oracle import OracleDOCP
    o = OracleDOCP(prob, sch, N)
    ncpu = min(16, os.cpu_count() or 1)
    ref, dropped = o.hess_coord_block(x, y, sigma, ncpu, return_dropped=True)
    assert dropped == (0, 0)
    scale = np.maximum(1.0, np.maximum(np.abs(ref), np.abs(o.hess_coord_block(x, np.abs(y), sigma, ncpu))))
    assert float(np.max(np.abs(v - ref) / scale)) <= TOL
    rows, cols = d.hess_structure()
    Hl = sp.coo_matrix((v, (rows - 1, cols - 1)), shape=(nvar, nvar)).tocsr()
    diag = sp.diags(Hl.diagonal())
    jr, jc = d.jac_structure()

    def glag(xx):
        xd = torch.from_numpy(xx).cuda()
        J = sp.coo_matrix((d.jac_coord(xd).cpu().numpy(), (jr - 1, jc - 1)), shape=(ncon, nvar)).tocsr()
        return sigma * d.grad(xd).cpu().numpy() + J.T @ y

    dirv = rng.uniform(-0.5, 0.5, nvar)
    Hd = Hl @ dirv + Hl.T @ dirv - diag @ dirv
    eps = 1e-6
    fd = (glag(x + eps * dirv) - glag(x - eps * dirv)) / (2 * eps)
    err = float(np.max(np.abs(Hd - fd) / np.maximum(1.0, np.abs(fd))))
    assert err <= 2e-5, err
    d.close()
