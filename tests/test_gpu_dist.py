"""Two ranks on ONE GPU through the HIP engine (gloo carries the collectives: RCCL refuses two ranks on one device): the
time-step sharded callbacks with a SHARDED iterate -- every entry of x a rank does not own is NaN until `exchange_halo`
fetches the few it needs -- against the oracle.  The driver's 8-GPU run takes the same code path with one rank per GPU
over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, prob, sch, q, order="csc"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        here = os.path.dirname(os.path.abspath(__file__))
        sys.path.insert(0, here)
        sys.path.insert(0, os.path.dirname(here))
        import ctdirect_jl_amd as ct
        from ctdirect_jl_amd import dist as ctdist
        from helpers import TOL, bench_inputs, describe, hess_err, relerr
        from oracle.oracle import OracleDOCP
        torch.cuda.set_device(0)
        o = OracleDOCP(prob, sch, N)
        x = bench_inputs(describe(o, prob, sch), perturb=1e-3)
        sh = ctdist.ShardedDOCP(lambda steps=None: ct.DOCP(prob, N, sch, device=0, steps=steps, pattern="structural", value_order=order), N, world=world, rank=rank)
        d = sh.docp
        # sharded iterate: own entries + the replicated variables, NaN everywhere else
        xs = np.full_like(x, np.nan)
        a, b = sh.owned_variables()
        xs[a:b] = x[a:b]
        nv = d.dims.NLP_v
        if nv:
            xs[-nv:] = x[-nv:]
        xd = torch.from_numpy(xs).cuda()
        c = torch.full((d.dim_NLP_constraints,), 777.0, dtype=torch.float64, device="cuda")
        v = torch.full((d.nnzj,), 777.0, dtype=torch.float64, device="cuda")
        step = sh.bind_cons_jac(xd, c, v, stitch=True, x_mode="halo")
        step(); step()
        torch.cuda.synchronize()
        o.set_pattern_mode(1)
        cref, vref = o.constraints(x), o.jac_coord(x)
        if order == "csr":                  # the oracle's values are in the reference's CSC order
            cp_, rv_ = o.jac_pattern()
            vref = vref[np.lexsort((np.repeat(np.arange(len(cp_) - 1), np.diff(cp_)), rv_))]
        chk = {}
        chk['c'] = relerr(c.cpu().numpy(), cref) <= TOL                   # whole residual on every rank after stitching
        lo, hi = d.shard.vals_main_begin, d.shard.vals_main_end          # the rank's contiguous CSC range
        vh = v.cpu().numpy()
        chk['J own range'] = bool(np.all(np.isfinite(vh[lo:hi]))) and relerr(vh[lo:hi], vref[lo:hi]) <= TOL
        # the pieces composed over the ranks give the whole Jacobian: every entry written by exactly the ranks that own it
        mine = torch.from_numpy((vh != 777.0).astype(np.float64))
        tot = mine.clone()
        dist.all_reduce(tot)
        chk['J coverage'] = bool((tot >= 1).all())
        vz = torch.from_numpy(np.where(vh != 777.0, vh, 0.0) / np.maximum(tot.numpy(), 1.0))
        dist.all_reduce(vz)
        chk['J composed'] = relerr(vz.numpy(), vref) <= TOL
        # objective (one all-reduce of one double on the device) and the Hessian's all-reduced V x V entries
        f = sh.obj(xd)
        chk['obj'] = abs(f - o.objective(x)) <= TOL * max(1.0, abs(o.objective(x)))
        y = np.cos(0.3 * np.arange(o.dim_NLP_constraints))
        hv = torch.full((d.nnzh,), 777.0, dtype=torch.float64, device="cuda")
        sh.hess_coord(xd, torch.from_numpy(y).cuda(), 0.5, hv)
        torch.cuda.synchronize()
        href = o.hess_coord(x, y, 0.5)
        hlo, hhi, vv = d.hess_shard_info()
        hh = hv.cpu().numpy()
        chk['H own range'] = hess_err(o, x, y, 0.5, hh, ref=href, idx=slice(hlo, hhi)) <= TOL      # (sign-changing y: backward-error scale)
        chk['H vv'] = hess_err(o, x, y, 0.5, hh, ref=href, idx=np.asarray(vv, dtype=int)) <= TOL
        # replicated iterate instead: rank 0 broadcasts all of x
        xb = torch.from_numpy(x if rank == 0 else np.full_like(x, np.nan)).cuda()
        c2 = torch.full_like(c, 777.0)
        v2 = torch.full_like(v, 777.0)
        sh.bind_cons_jac(xb, c2, v2, stitch=True, x_mode="broadcast")()
        torch.cuda.synchronize()
        chk['broadcast c'] = relerr(c2.cpu().numpy(), cref) <= TOL
        # sharded iterate read IN PLACE: no exchange at all -- the kernels load the neighbour's entries from the other process'
        # buffer (IPC mapping; over xGMI on a multi-GPU node).  The local copy of everything a rank does not own stays NaN
        xp = torch.from_numpy(xs).cuda()
        c3 = torch.full_like(c, 777.0)
        v3 = torch.full_like(v, 777.0)
        step = sh.bind_cons_jac(xp, c3, v3, stitch=False, x_mode="peer")
        dist.barrier()
        step(); step()
        torch.cuda.synchronize()
        dist.barrier()                      # nobody frees its buffer while the other rank's kernel may still read it
        c3h, v3h = c3.cpu().numpy(), v3.cpu().numpy()
        r0, r1 = d.shard.c_row_begin, d.shard.c_row_end
        chk['peer c rows'] = bool(np.array_equal(c3h[r0:r1], c.cpu().numpy()[r0:r1]))          # bit-identical to the exchanged run
        chk['peer c tail'] = bool(np.array_equal(c3h[N * sh.cb:], c.cpu().numpy()[N * sh.cb:]))
        chk['peer J'] = bool(np.array_equal(v3h, vh))
        chk['peer x untouched'] = bool(np.array_equal(np.isnan(xp.cpu().numpy()), np.isnan(xs)))
        # the objective and the Hessian read the same sharded iterate in place (the shard table stays on the handle); the gradient
        # pass is not restated for it and refuses instead of evaluating on NaNs
        f3 = sh.obj(xp)
        chk['peer obj'] = bool(f3 == f)
        hv3 = torch.full((d.nnzh,), 777.0, dtype=torch.float64, device="cuda")
        sh.hess_coord(xp, torch.from_numpy(y).cuda(), 0.5, hv3)
        torch.cuda.synchronize()
        dist.barrier()
        chk['peer H'] = bool(np.array_equal(hv3.cpu().numpy(), hh))
        # the gradient is the whole objective's on every rank and reads only the x it is given: a whole iterate, shard table or not
        gw = d.grad(torch.from_numpy(x).cuda()).cpu().numpy()
        gref = o.gradient(x)
        chk['grad on the whole x'] = relerr(gw, gref) <= TOL
        # ... and the SHARDED gradient (round 4): own entries from the NaN-padded iterate read in place, d/dv all-reduced; the entries
        # other ranks own stay untouched, the ranks' pieces compose the whole gradient
        gs = torch.full((d.dim_NLP_variables,), 777.0, dtype=torch.float64, device="cuda")
        sh.grad(xp, gs)
        torch.cuda.synchronize()
        dist.barrier()
        gh = gs.cpu().numpy()
        own = np.zeros(len(gh), dtype=bool)
        own[a:b] = True
        if nv:
            own[-nv:] = True
        chk['sharded grad own'] = bool(np.all(np.isfinite(gh[own]))) and relerr(gh[own], gref[own]) <= TOL
        chk['sharded grad elsewhere untouched'] = bool(np.all(gh[~own] == 777.0))
        # with copied halos instead of the shard table (x_mode "halo" left them in xd)
        sh.disable_peer_x()
        # sharded MULTIPLIERS (round 4): own rows of y + what exchange_multipliers fetches (previous rank's last step, tail rows), NaN elsewhere
        ys = np.full_like(y, np.nan)
        ca, cz = sh.owned_constraints()
        ys[ca:cz] = y[ca:cz]
        yd = torch.from_numpy(ys).cuda()
        sh.exchange_multipliers(yd)
        hv4 = torch.full((d.nnzh,), 777.0, dtype=torch.float64, device="cuda")
        sh.hess_coord(xd, yd, 0.5, hv4)
        torch.cuda.synchronize()
        chk['H from sharded multipliers'] = bool(np.array_equal(hv4.cpu().numpy(), hh))
        gs2 = torch.full_like(gs, 777.0)
        sh.grad(xd, gs2)
        torch.cuda.synchronize()
        chk['sharded grad (halo copies)'] = bool(np.array_equal(gs2.cpu().numpy(), gh))
        sh.close()
        bad = [k for k, v_ in chk.items() if not v_]
        q.put((rank, True if not bad else bad))
    except Exception as e:      # noqa: BLE001 -- the parent reports it
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N,prob,sch", [(2000, "goddard", "gauss_legendre_2"), (1001, "goddard_all", "trapeze"),
                                        (3000, "double_integrator_path", "midpoint"), (501, "quadrotor", "gauss_legendre_3"),
                                        (777, "goddard_all", "euler_implicit"),
                                        (20000, "goddard", "gauss_legendre_2")])      # (10 000 steps per rank: the lane-per-step Hessian kernel)
def test_two_ranks_one_gpu_sharded_iterate(N, prob, sch):
    _two_ranks(N, prob, sch, "csc")


@pytest.mark.parametrize("N,prob,sch", [(1001, "goddard_all", "trapeze"), (3000, "double_integrator_path", "midpoint"), (2000, "goddard", "gauss_legendre_2")])
def test_two_ranks_one_gpu_sharded_iterate_csr_order(N, prob, sch):
    """the same with the Jacobian values in CSR order (one value range per rank)"""
    _two_ranks(N, prob, sch, "csr")


def _two_ranks(N, prob, sch, order):
    assert torch.cuda.is_available()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, prob, sch, q, order)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok is True for _, ok in res), res


@pytest.mark.parametrize("N,prob,sch", [(1000, "goddard", "gauss_legendre_2"), (1000, "goddard_all", "trapeze"),
                                        (999, "double_integrator_path", "midpoint"), (400, "quadrotor12", "gauss_legendre_3"),
                                        (500, "goddard_all", "euler_implicit")])
@pytest.mark.parametrize("order", ["csc", "csr"])
def test_multi_device_handle_three_shards_one_gpu(N, prob, sch, order):
    """ctd_create_sharded / ctd_cons_jac_sharded_dev_async (the single-process multi-GPU entry point of the C ABI) with the
    one GPU of this box named three times: sharded iterate (NaN outside what a shard owns until the engine's peer copies
    fetch the halos, or the kernels read them in place through ctd_set_x_shards), stitched c on every shard, Jacobian pieces composed -- bit-identical to the unsharded handle."""
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    import ctdirect_jl_amd as ct
    from helpers import bench_inputs, describe
    full = ct.DOCP(prob, N, sch, device=0, pattern="structural", value_order=order)
    x = bench_inputs(describe(full, prob, sch), perturb=1e-3)
    xd = torch.from_numpy(x).cuda()
    cf, vf = full.cons_jac(xd)
    md = ct.MultiDeviceDOCP(prob, N, sch, [0, 0, 0], pattern="structural", value_order=order)
    assert (md.dim_NLP_variables, md.dim_NLP_constraints, md.nnzj) == (full.dim_NLP_variables, full.dim_NLP_constraints, full.nnzj)
    assert [s.step_begin for s in md.shards] == [0, md.shards[0].step_end, md.shards[1].step_end] and md.shards[2].step_end == N
    blk, nv = full.discretization._step_variables_block, full.dims.NLP_v
    for mode in (md.X_SHARDED_IN_PLACE, md.X_SHARDED, md.X_SHARDED_COPY, md.X_FROM_DEVICE0, md.X_IN_PLACE):
        xs = []
        for k, s in enumerate(md.shards):
            if mode == md.X_IN_PLACE or (mode == md.X_FROM_DEVICE0 and k == 0):
                xs.append(xd.clone())
                continue
            t = np.full_like(x, np.nan)
            if mode in (md.X_SHARDED_IN_PLACE, md.X_SHARDED, md.X_SHARDED_COPY):
                end = s.step_end * blk if k < 2 else x.size - nv
                t[s.step_begin * blk:end] = x[s.step_begin * blk:end]
                if nv:
                    t[-nv:] = x[-nv:]
            xs.append(torch.from_numpy(t).cuda())
        cs = [torch.full_like(cf, 777.0) for _ in range(3)]
        vs = [torch.full_like(vf, 777.0) for _ in range(3)]
        for _ in range(2):            # twice: the second call must not overwrite rows another shard is still pulling
            md.cons_jac(xs, cs, vs, x_mode=mode, stitch=True, sync=False)
        md.sync()
        if mode == md.X_SHARDED_IN_PLACE:          # read in place: nothing was copied into the shards' buffers
            for k in range(3):
                assert torch.isnan(xs[k]).any()
                assert torch.isnan(xs[k][:md.shards[k].step_begin * blk]).all()
        if mode in (md.X_SHARDED, md.X_SHARDED_COPY) and not sch.startswith("gauss"):
            # the copying protocol leaves in every shard's buffer what ANY callback of that shard reads (the gradient and the Hessian of a
            # one-point scheme read the previous block whatever the Jacobian's value order needs)
            for k in (1, 2):
                b = md.shards[k].step_begin
                assert not torch.isnan(xs[k][(b - 1) * blk:b * blk]).any(), (mode, k)
        v = torch.full_like(vf, 777.0)
        for k in range(3):
            assert torch.equal(cs[k], cf), (mode, k)                 # whole residual on every shard, bit for bit
            v = torch.where(vs[k] != 777.0, vs[k], v)
        assert torch.equal(v, vf), mode
    md.close()
    full.close()


@pytest.mark.parametrize("prob,sch,N", [("double_integrator_path", "midpoint", 999), ("goddard_all", "trapeze", 500), ("quadrotor", "gauss_legendre_2", 301),
                                        ("double_integrator_path", "euler_implicit", 77), ("goddard", "gauss_legendre_3", 1000),
                                        ("double_integrator_freet0tf", "euler", 300), ("least_squares_with_constraint", "midpoint", 200)])
def test_sharded_gradient_three_shards_through_the_c_abi(prob, sch, N):
    """ctd_grad_shard_dev_async on three shard handles of one process: each buffer holds its shard's variables + v, NaN elsewhere,
    the neighbours' entries come through ctd_set_x_shards; the shards' pieces compose the whole gradient bit for bit (the d/dv
    entries are the sum of the three partials, the one all-reduce a distributed caller makes) -- Lagrange, Mayer and Bolza costs,
    free times, every scheme class."""
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    import ctdirect_jl_amd as ct
    from helpers import TOL, bench_inputs, describe, relerr
    full = ct.DOCP(prob, N, sch, device=0)
    x = bench_inputs(describe(full, prob, sch), perturb=1e-3)
    gref = full.grad(torch.from_numpy(x).cuda())
    blk, nv = full.discretization._step_variables_block, full.dims.NLP_v
    cuts = [0, N // 3, (2 * N) // 3 + 1, N]
    hs = [ct.DOCP(prob, N, sch, device=0, steps=(cuts[k], cuts[k + 1])) for k in range(3)]
    xs = []
    for k in range(3):
        t = np.full_like(x, np.nan)
        end = cuts[k + 1] * blk if k < 2 else x.size - nv
        t[cuts[k] * blk:end] = x[cuts[k] * blk:end]
        if nv:
            t[-nv:] = x[-nv:]
        xs.append(torch.from_numpy(t).cuda())
    g = torch.full_like(gref, 777.0)
    tail = torch.zeros(nv, dtype=torch.float64, device="cuda")
    for k in range(3):
        hs[k].set_x_shards(cuts, [t.data_ptr() for t in xs], k)
        gk = torch.full_like(gref, 777.0)
        hs[k].grad_shard(xs[k], gk, sync=True)
        end = cuts[k + 1] * blk if k < 2 else x.size - nv
        assert bool((gk[:cuts[k] * blk] == 777.0).all()) and bool((gk[end:x.size - nv] == 777.0).all()), k      # own entries only
        g[cuts[k] * blk:end] = gk[cuts[k] * blk:end]
        if nv:
            tail += gk[-nv:]
    if nv:
        g[-nv:] = tail
    assert not bool((g == 777.0).any())
    assert torch.equal(g[:x.size - nv], gref[:x.size - nv])                                  # bit for bit
    assert relerr(g.cpu().numpy(), gref.cpu().numpy()) <= TOL                               # (d/dv: three partial sums instead of one)
    for h in hs:
        h.close()
    full.close()


def test_in_place_mode_without_peer_access_falls_back_to_copies(monkeypatch):
    """CTD_X_SHARDED_IN_PLACE dereferences the other shards' buffers inside the kernels; on a topology without peer
    access (simulated: CTD_TEST_NO_PEER makes ctd_create_sharded record every pair of distinct shards as unreachable) the call
    must take the copying protocol instead of faulting, say so, and give the same bits."""
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    import ctdirect_jl_amd as ct
    from helpers import bench_inputs, describe
    prob, sch, N = "double_integrator_path", "midpoint", 999
    full = ct.DOCP(prob, N, sch, device=0)
    x = bench_inputs(describe(full, prob, sch), perturb=1e-3)
    cf, vf = full.cons_jac(torch.from_numpy(x).cuda())
    monkeypatch.setenv("CTD_TEST_NO_PEER", "1")
    md = ct.MultiDeviceDOCP(prob, N, sch, [0, 0])
    monkeypatch.delenv("CTD_TEST_NO_PEER")
    blk = full.discretization._step_variables_block
    xs = []
    for k, s in enumerate(md.shards):
        t = np.full_like(x, np.nan)
        end = s.step_end * blk if k == 0 else x.size
        t[s.step_begin * blk:end] = x[s.step_begin * blk:end]
        xs.append(torch.from_numpy(t).cuda())
    cs = [torch.full_like(cf, 777.0) for _ in range(2)]
    vs = [torch.full_like(vf, 777.0) for _ in range(2)]
    md.cons_jac(xs, cs, vs, x_mode=md.X_SHARDED_IN_PLACE, stitch=True, sync=True)
    assert "no peer access" in md.last_error()
    assert not torch.isnan(xs[0][md.shards[0].step_end * blk:md.shards[0].step_end * blk + 2]).any()     # the halo was COPIED
    v = torch.where(vs[0] != 777.0, vs[0], vs[1])
    assert torch.equal(cs[0], cf) and torch.equal(cs[1], cf) and torch.equal(v, vf)
    md.close()
    full.close()


def test_library_stitch_over_a_one_rank_rccl_communicator():
    """`ctd_stitch_c`: the all-gather of the constraint row blocks INSIDE the library over an ncclComm_t the host created (no
    torch.distributed in the path -- what a one-process-per-GPU Julia host uses).  RCCL refuses two ranks on one device, so
    the one-GPU box runs a ONE-rank communicator: the padded pack / ncclAllGather / index kernel (goddard_all: p + bc tail
    rows) and the in-place form (a problem without tail rows does not exist, so: ragged vs equal only differ in padding) both
    execute for real and must leave c exactly as the kernel wrote it."""
    import ctypes as C
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    import ctdirect_jl_amd as ct
    from helpers import bench_inputs, describe
    cand = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")) + ["/opt/rocm/lib/librccl.so.1"]
    rccl = C.CDLL([p for p in cand if os.path.exists(p)][0], mode=C.RTLD_GLOBAL)

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    torch.cuda.set_device(0)
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        for prob, sch, N in (("goddard_all", "trapeze", 1001), ("goddard", "gauss_legendre_2", 2000)):
            d = ct.DOCP(prob, N, sch, device=0, pattern="structural")
            x = torch.from_numpy(bench_inputs(describe(d, prob, sch), perturb=1e-3)).cuda()
            c, _ = d.cons_jac(x)
            want = c.clone()
            d.stitch_c(comm.value, 1, 0, c)
            d.sync(); torch.cuda.synchronize()
            assert torch.equal(c, want)
            # a shard that is not block `rank` of the balanced split is refused, never gathered wrongly
            with pytest.raises(Exception):
                d.stitch_c(comm.value, 2, 0, c)
            d.close()
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
