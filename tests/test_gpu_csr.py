"""CSR value order on the MI355X (ctd_desc.value_order = CTD_ORDER_CSR; BASELINE north_star: "assembled into the block-banded sparse
KKT Jacobian ... in CSR on device").  The kernels are the ones the CSC order runs -- the host hands them a row-order emit template --
so every check is BIT-EXACT: the CSR values are the CSC values under the host permutation CSC -> CSR, for every registry problem x
every scheme x all three patterns at the sizes of test_oracle_parity_midsize, for BASELINE configs 2 - 5 at full size on both
patterns, for shards (ONE value range per shard, the ranges partition the array), for run-time OCPs (hiprtc-compiled kernels) and
through the multi-device C ABI.  The CSC values themselves are pinned against the oracle in tests/test_gpu_parity.py.  Reference
order being replaced: SparseArrays.sparse(Is, Js, ...), src/ode/irk_stagewise.jl:555-558, midpoint.jl:229-232, trapeze.jl:229-232."""
import numpy as np
import pytest

import ctdirect_jl_amd as ct
from helpers import bench_inputs, describe

pytestmark = pytest.mark.gpu
SENT = 777.25


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU; there is no CPU fallback"
    return torch


def perm_csc_to_csr(torch, d0):
    """perm[k_csr] = k_csc from the CSC handle's jac_structure (sorted by (row, col) on the GPU: the entries are unique)"""
    r0, c0 = d0.jac_structure()
    key = torch.from_numpy((r0 - 1) * d0.dim_NLP_variables + (c0 - 1)).cuda()
    perm = torch.argsort(key)
    return perm, r0, c0


def check_pair(torch, prob, N, sch, pattern, tg=None, structure=True, **kw):
    d0 = ct.DOCP(prob, N, sch, time_grid=tg, pattern=pattern, device=0, **kw)
    d1 = ct.DOCP(prob, N, sch, time_grid=tg, pattern=pattern, device=0, value_order="csr", **kw)
    assert d1.nnzj == d0.nnzj
    x = torch.from_numpy(bench_inputs(describe(d0, d0.problem_name.replace("_rt", ""), sch), perturb=1e-3)).cuda()
    c0 = torch.full((d0.dim_NLP_constraints,), SENT, dtype=torch.float64, device="cuda")
    v0 = torch.full((d0.nnzj,), SENT, dtype=torch.float64, device="cuda")
    c1, v1 = torch.full_like(c0, SENT), torch.full_like(v0, SENT)
    d0.cons_jac(x, c0, v0)
    d1.cons_jac(x, c1, v1)
    perm, r0, c0s = perm_csc_to_csr(torch, d0)
    assert not bool((v1 == SENT).any()) and not bool((c1 == SENT).any())          # every entry written
    assert torch.equal(c1, c0)
    assert torch.equal(v1, v0[perm]), (prob, N, sch, pattern)
    if structure:
        r1, c1s = d1.jac_structure()
        p = perm.cpu().numpy()
        assert np.array_equal(r1, r0[p]) and np.array_equal(c1s, c0s[p])
        rp, ci = ct.DOCP_Jacobian_csr(d1)
        assert np.array_equal(ci + 1, c1s) and np.array_equal(np.repeat(np.arange(len(rp) - 1), np.diff(rp)) + 1, r1)
    # the host-pointer entry points and jac_coord alone follow the same order
    if d0.nnzj < 200000:
        xh = x.cpu().numpy()
        assert np.array_equal(d1.jac_coord(xh), v1.cpu().numpy())
        assert torch.equal(d1.jac_coord(x), v1)
    d0.close(); d1.close()


PAIRS = [(p, s) for p in ct.PROBLEMS for s in ct.SCHEMES]


@pytest.mark.parametrize("prob,sch", PAIRS, ids=[f"{p}-{s}" for p, s in PAIRS])
def test_csr_equals_csc_permuted_midsize(torch_cuda, prob, sch):
    rng = np.random.default_rng(11)
    for N, tg in ((1, None), (3, None), (5, None), (64, None), (257, None), (1000, None), (101, np.cumsum(rng.uniform(0.2, 1.8, 102)))):
        for pattern in ("manual", "structural", "optimized"):
            try:
                check_pair(torch_cuda, prob, N, sch, pattern, tg=tg)
            except ct.CTDirectError:
                assert pattern == "optimized" and sch == "euler_implicit"      # (refused, never silently wrong: see test_gpu_parity)


FULL = [("goddard", "gauss_legendre_2", 10000), ("double_integrator_path", "midpoint", 100000), ("goddard", "gauss_legendre_3", 80000),
        ("quadrotor", "gauss_legendre_3", 20000), ("quadrotor12", "gauss_legendre_3", 20000), ("quadrotor12", "midpoint", 20000),
        ("goddard_all", "trapeze", 10000)]


@pytest.mark.parametrize("prob,sch,N", FULL, ids=[f"{p}-{s}-{n}" for p, s, n in FULL])
def test_csr_equals_csc_permuted_full_size(torch_cuda, prob, sch, N):
    for pattern in ("manual", "optimized"):
        check_pair(torch_cuda, prob, N, sch, pattern, structure=(N * 10 <= 1000000 and prob != "quadrotor12"))


@pytest.mark.parametrize("tile,block", [(1, 64), (3, 128), (16, 256), (64, 256)])
def test_csr_launch_geometry_independence(torch_cuda, monkeypatch, tile, block):
    monkeypatch.setenv("CTD_TILE", str(tile))
    monkeypatch.setenv("CTD_BLOCK", str(block))
    for prob, sch, N in (("goddard", "gauss_legendre_2", 333), ("goddard_all", "trapeze", 200), ("quadrotor", "midpoint", 150),
                         ("quadrotor12", "gauss_legendre_3", 40), ("double_integrator_path", "euler_implicit", 77)):
        check_pair(torch_cuda, prob, N, sch, "structural")


def test_csr_shards_one_range_each(torch_cuda):
    """SURVEY 8e "CSC caveat" gone: in CSR order a rank's rows are ONE contiguous range of the value array (V entries inline)."""
    torch = torch_cuda
    for prob, sch, N in (("goddard", "gauss_legendre_3", 1000), ("goddard_all", "trapeze", 777), ("double_integrator_path", "midpoint", 1203),
                         ("quadrotor", "gauss_legendre_2", 301), ("goddard_all", "euler_implicit", 90)):
        full = ct.DOCP(prob, N, sch, device=0, value_order="csr", pattern="structural")
        x = torch.from_numpy(bench_inputs(describe(full, prob, sch), perturb=1e-3)).cuda()
        cf, vf = full.cons_jac(x)
        v = torch.full_like(vf, SENT)
        cuts = [0, N // 3, N // 2 + 1, N]
        covered = 0
        for a, b in zip(cuts[:-1], cuts[1:]):
            sh = ct.DOCP(prob, N, sch, device=0, value_order="csr", pattern="structural", steps=(a, b))
            v2 = torch.full_like(vf, SENT)
            c2 = torch.full_like(cf, SENT)
            sh.cons_jac(x, c2, v2)
            lo, hi = sh.shard.vals_main_begin, sh.shard.vals_main_end
            assert lo == covered
            covered = hi
            wrote = v2 != SENT
            assert bool(wrote[lo:hi].all()) and not bool(wrote[:lo].any()) and not bool(wrote[hi:].any())
            assert torch.equal(c2[sh.shard.c_row_begin:sh.shard.c_row_end], cf[sh.shard.c_row_begin:sh.shard.c_row_end])
            v[lo:hi] = v2[lo:hi]
            sh.close()
        assert covered == full.nnzj and torch.equal(v, vf)
        full.close()


def test_csr_multi_device_handle_and_sharded_iterate(torch_cuda):
    """ctd_create_sharded with value_order = CSR: three shards on the one GPU, iterate sharded (NaN outside ownership) and read in
    place; the pieces are the shards' single ranges."""
    torch = torch_cuda
    for prob, sch, N in (("goddard", "gauss_legendre_2", 1000), ("double_integrator_path", "midpoint", 999), ("goddard_all", "trapeze", 500)):
        full = ct.DOCP(prob, N, sch, device=0, value_order="csr")
        x = bench_inputs(describe(full, prob, sch), perturb=1e-3)
        cf, vf = full.cons_jac(torch.from_numpy(x).cuda())
        md = ct.MultiDeviceDOCP(prob, N, sch, [0, 0, 0], value_order="csr")
        blk, nv = full.discretization._step_variables_block, full.dims.NLP_v
        xs = []
        for k, s in enumerate(md.shards):
            t = np.full_like(x, np.nan)
            end = s.step_end * blk if k < 2 else x.size - nv
            t[s.step_begin * blk:end] = x[s.step_begin * blk:end]
            if nv:
                t[-nv:] = x[-nv:]
            xs.append(torch.from_numpy(t).cuda())
        cs = [torch.full_like(cf, SENT) for _ in range(3)]
        vs = [torch.full_like(vf, SENT) for _ in range(3)]
        md.cons_jac(xs, cs, vs, x_mode=md.X_SHARDED_IN_PLACE, stitch=True, sync=True)
        v = torch.full_like(vf, SENT)
        for k, s in enumerate(md.shards):
            assert torch.equal(cs[k], cf)
            assert bool((vs[k][:s.vals_main_begin] == SENT).all()) and bool((vs[k][s.vals_main_end:] == SENT).all())
            v[s.vals_main_begin:s.vals_main_end] = vs[k][s.vals_main_begin:s.vals_main_end]
        assert torch.equal(v, vf)
        assert md.shards[0].vals_main_begin == 0 and md.shards[2].vals_main_end == full.nnzj
        md.close(); full.close()


def test_csr_for_runtime_ocps_and_several_controls_per_step(torch_cuda):
    import jit_defs
    rt = jit_defs.twin("goddard")
    for sch in ("gauss_legendre_2", "trapeze", "midpoint"):
        for pattern in ("manual", "optimized"):
            check_pair(torch_cuda, rt, 57, sch, pattern)
    for cs in (2, 3):
        for prob in ("goddard", "quadrotor"):
            check_pair(torch_cuda, prob, 41, "midpoint", "structural", control_steps=cs)


def test_csr_feeds_a_sparse_matvec(torch_cuda):
    """what the order is for: rowptr / colind + the device value array ARE a CSR matrix -- J d by a CSR matvec on the GPU equals the
    directional derivative of c (central differences of the same handle's constraints)"""
    torch = torch_cuda
    prob, sch, N = "goddard", "gauss_legendre_2", 2000
    d = ct.DOCP(prob, N, sch, device=0, value_order="csr")
    x = torch.from_numpy(bench_inputs(describe(d, prob, sch), perturb=1e-3)).cuda()
    c, v = d.cons_jac(x)
    rp, ci = ct.DOCP_Jacobian_csr(d)
    J = torch.sparse_csr_tensor(torch.from_numpy(rp).cuda(), torch.from_numpy(ci).cuda(), v, size=(d.dim_NLP_constraints, d.dim_NLP_variables))
    dirv = torch.from_numpy(np.cos(0.01 * np.arange(d.dim_NLP_variables))).cuda()
    jd = (J @ dirv.unsqueeze(1)).squeeze(1)
    h = 1e-6
    fd = (d.cons(x + h * dirv) - d.cons(x - h * dirv)) / (2 * h)
    assert float((jd - fd).abs().max()) <= 1e-6 * max(1.0, float(fd.abs().max()))
    d.close()


@pytest.mark.parametrize("prob,sch,N", [("goddard", "gauss_legendre_2", 10000), ("goddard_all", "trapeze", 1001),
                                        ("double_integrator_path", "midpoint", 3000), ("quadrotor", "gauss_legendre_3", 300),
                                        ("double_integrator_freet0tf", "euler_implicit", 5)])
def test_whole_iteration_in_csr_order(torch_cuda, prob, sch, N):
    """ctd_eval_all_dev_async (the horizontally fused iteration kernel) on a handle with value_order = CSR: its Jacobian values are
    bit-identical to the single-purpose kernel's CSR values, which are the CSC handle's values under the host permutation; objective,
    gradient, c and the Hessian values do not depend on the order."""
    torch = torch_cuda
    d, dc = ct.DOCP(prob, N, sch, pattern="structural", device=0, value_order="csr"), ct.DOCP(prob, N, sch, pattern="structural", device=0)
    x = bench_inputs(describe(d, prob, sch), perturb=1e-2)
    y = np.cos(0.41 * np.arange(d.dim_NLP_constraints))
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    new = lambda n: torch.full((n,), 777.25, dtype=torch.float64, device="cuda")      # noqa: E731
    f, g, c, v, h = new(1), new(d.dim_NLP_variables), new(d.dim_NLP_constraints), new(d.nnzj), new(d.nnzh)
    d.eval_all(xd, yd, 0.9, f, g, c, v, h, sync=True)
    c1, v1 = d.cons_jac(xd)
    assert torch.equal(c, c1) and torch.equal(v, v1)
    fc, gc, cc, vc, hc = new(1), new(d.dim_NLP_variables), new(d.dim_NLP_constraints), new(d.nnzj), new(d.nnzh)
    dc.eval_all(xd, yd, 0.9, fc, gc, cc, vc, hc, sync=True)
    assert torch.equal(f, fc) and torch.equal(g, gc) and torch.equal(c, cc) and torch.equal(h, hc)
    rows, cols = dc.jac_structure()
    perm = torch.from_numpy(np.lexsort((cols, rows))).cuda()
    assert torch.equal(v, vc[perm])
    d.close()
    dc.close()
