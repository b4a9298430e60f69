"""CPU tests of the multi-GPU plumbing (world_size 2, gloo): shard ranges and the in-place stitching of the constraint
vector.  The per-shard values come from the oracle here (there is no GPU and the engine has no CPU path); on the GPU
box the same plumbing is driven by bench.py with the HIP engine."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ctdirect_jl_amd as ct
from ctdirect_jl_amd import dist as ctdist


def test_shard_steps_partition():
    for N in (1, 7, 10, 100, 10001):
        for world in (1, 2, 3, 8):
            if world > N:
                continue
            blocks = [ctdist.shard_steps(N, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == N
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, prob, sch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        here = os.path.dirname(os.path.abspath(__file__))
        sys.path.insert(0, here)
        sys.path.insert(0, os.path.dirname(here))
        from helpers import bench_inputs, describe
        from oracle.oracle import OracleDOCP
        o = OracleDOCP(prob, sch, N)
        x = bench_inputs(describe(o, prob, sch), perturb=1e-3)
        full = o.constraints(x)
        d = ct.DOCP(prob, N, sch, steps=ctdist.shard_steps(N, world, rank), device=-1)
        cb = d.discretization._state_stage_eqs_block + d.discretization._step_pathcons_block
        # this rank "computes" only its rows (taken from the oracle), everything else is a sentinel
        c = torch.full((d.dim_NLP_constraints,), 666.666, dtype=torch.float64)
        a, b = d.shard.c_row_begin, d.shard.c_row_end
        c[a:b] = torch.from_numpy(full[a:b])
        if rank == world - 1:
            c[N * cb:] = torch.from_numpy(full[N * cb:])     # the tail rows (final path + boundary) come from the last rank
        ctdist.stitch_constraints(c, N, cb, world, rank)
        ok = bool(np.array_equal(c.numpy(), full))
        tot = ctdist.reduce_objective(float(rank + 1))
        # Hessian: the shards' partial V x V sums (taken from the kernel-logic emulator, tests only) add up to the oracle's
        from emu import emu
        y = np.cos(0.3 * np.arange(o.dim_NLP_constraints))
        hfull = o.hess_coord(x, y, 0.5)
        sb, se = ctdist.shard_steps(N, world, rank)
        part = emu.hess(ct.PROBLEMS[prob], ct.SCHEMES[sch], 0, N, x, y, 0.5, step_begin=sb, step_end=se)
        hv = torch.from_numpy(part.copy())
        vv = d.hess_shard_info()[2]
        ctdist.reduce_hessian_vv(hv, vv)
        lo, hi, _ = d.hess_shard_info()
        ok = ok and bool(np.allclose(hv.numpy()[vv], hfull[vv], rtol=1e-12, atol=1e-12))
        ok = ok and bool(np.allclose(hv.numpy()[lo:hi], hfull[lo:hi], rtol=1e-10, atol=1e-10))
        # sharded iterate: the halo exchange fills exactly the foreign entries this rank's rows / columns read
        sh = ctdist.ShardedDOCP(lambda steps=None: ct.DOCP(prob, N, sch, steps=steps, device=-1), N, world=world, rank=rank)
        xs = np.full_like(x, np.nan)
        oa, ob = sh.owned_variables()
        xs[oa:ob] = x[oa:ob]
        nv = d.dims.NLP_v
        if nv:
            xs[-nv:] = x[-nv:]
        xt = torch.from_numpy(xs)
        sh.exchange_halo(xt)
        blk, n = sh.blk, sh.n
        need = [(0, n), (N * blk, N * blk + n)]
        if rank + 1 < world:
            need.append((se * blk, se * blk + sh.halo_w))
        if rank > 0 and sh.halo_lo:
            need.append(((sb - 1) * blk, sb * blk))
        for lo_, hi_ in need:
            ok = ok and bool(np.array_equal(xt.numpy()[lo_:hi_], x[lo_:hi_]))
        untouched = np.ones(x.size, dtype=bool)
        untouched[oa:ob] = False
        if nv:
            untouched[-nv:] = False
        for lo_, hi_ in need:
            untouched[lo_:hi_] = False
        ok = ok and bool(np.isnan(xt.numpy()[untouched]).all())
        # sharded multipliers: own rows + what exchange_multipliers fetches is everything this rank's Hessian entries read
        ys = np.full_like(y, np.nan)
        ca, cz = sh.owned_constraints()
        ys[ca:cz] = y[ca:cz]
        yt = torch.from_numpy(ys)
        sh.exchange_multipliers(yt)
        ok = ok and (cz == (o.dim_NLP_constraints if rank == world - 1 else se * cb)) and ca == sb * cb
        ok = ok and bool(np.array_equal(yt.numpy()[N * cb:], y[N * cb:]))                     # tail rows everywhere
        part_s = emu.hess(ct.PROBLEMS[prob], ct.SCHEMES[sch], 0, N, xt.numpy(), yt.numpy(), 0.5, step_begin=sb, step_end=se)
        wr = part != 666.666
        ok = ok and bool(np.array_equal(part_s != 666.666, wr)) and bool(np.array_equal(part_s[wr], part[wr]))     # bit for bit, no NaN
        q.put((rank, ok, tot))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N,prob,sch", [(10, "goddard_all", "gauss_legendre_2"), (7, "goddard", "trapeze"),
                                        (64, "double_integrator_path", "midpoint"), (9, "double_integrator_path", "midpoint"),
                                        (11, "goddard_all", "midpoint"), (8, "quadrotor", "trapeze"), (6, "goddard_all", "euler_implicit")])
def test_stitch_constraints_world2_gloo(N, prob, sch):
    _run_world(2, N, prob, sch)


@pytest.mark.parametrize("N,prob,sch", [(11, "goddard_all", "midpoint"), (10, "quadrotor", "trapeze"), (13, "goddard_all", "gauss_legendre_2")])
def test_stitch_constraints_world3_gloo(N, prob, sch):
    """three ranks: the middle one has a neighbour on both sides (halo of x from both, multipliers of the previous rank's last step), ragged split"""
    _run_world(3, N, prob, sch)


def _run_world(world, N, prob, sch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, prob, sch, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert all(tot == world * (world + 1) / 2 for _, _, tot in res)


def test_library_stitch_index_map_equals_the_python_stitcher():
    """ctd_stitch_c (the RCCL all-gather inside the library) unpacks the gathered padded blocks with the index map
    stitch_src (csrc/ctd_layout.hpp); the Python route (`_Stitcher`) builds the same map as a tensor.  They must agree for
    ragged and equal splits, with and without tail rows -- and the library's split rule must be dist.shard_steps."""
    import ctypes as C
    from emu import emu
    import ctdirect_jl_amd as ct
    L = emu.lib()
    for N, cb, tail, G in ((10, 3, 4, 3), (7, 2, 0, 7), (12, 5, 6, 4), (1001, 9, 7, 8), (9, 4, 0, 2)):
        ncon = N * cb + tail
        blocks = [ctdist.shard_steps(N, G, r) for r in range(G)]
        for r, (b, e) in enumerate(blocks):
            bb, ee = C.c_int64(), C.c_int64()
            assert ct._lib.lib().ctd_shard_steps(N, G, r, C.byref(bb), C.byref(ee)) == 0 and (bb.value, ee.value) == (b, e)
        smax = max(e - b for b, e in blocks) * cb + tail
        out = np.zeros(ncon, dtype=np.int64)
        L.emu_stitch_src(C.c_int64(N), cb, G, C.c_int64(smax), C.c_int64(ncon), out.ctypes.data_as(C.c_void_p))
        want = np.empty(ncon, dtype=np.int64)
        for r, (b, e) in enumerate(blocks):
            want[b * cb:e * cb] = r * smax + np.arange((e - b) * cb)
        want[N * cb:] = (G - 1) * smax + (blocks[-1][1] - blocks[-1][0]) * cb + np.arange(tail)
        assert np.array_equal(out, want), (N, cb, tail, G)
