#!/usr/bin/env python3
"""bench.py -- NLP-callback throughput of the MI355X collocation engine (BASELINE.json metric).

One "step" = one fused evaluation (constraints c(x) + sparse Jacobian values, `ctd_cons_jac_dev_async`) of the workload
on inputs already resident in HBM.  Workloads (`--config`):

    cfg2_weak    Goddard, Gauss-Legendre 2 (stagewise), 10 000 time steps PER GPU (default): BASELINE.json configs[1]
                 at N = 1; at N > 1 the global grid has 10 000 x N steps (weak scaling)
    cfg2_strong  the same 10 000-step transcription sharded over the N GPUs (strong scaling, north_star's 1/2/4/8 line)
    cfg4         Goddard, Gauss-Legendre 3, 80 000 steps sharded over the N GPUs            (BASELINE.json configs[3])
    cfg5         12-state quadrotor, Gauss-Legendre 3, 20 000 steps sharded over the N GPUs (BASELINE.json configs[4])

The grid is sharded by time step; each rank's rows of c and its Jacobian values stay on the rank (row-sharded outputs).
At N > 1 the ITERATE IS SHARDED like the steps: a rank's x buffer holds its own steps' variables and the replicated v, and
NaN everywhere else.  The few entries of other ranks its rows read (next rank's first node, previous rank's last block for
one-point schemes, first / final state) are loaded by the evaluation kernel IN PLACE from the owner's HBM over xGMI (IPC-
mapped buffers, `ShardedDOCP.enable_peer_x` / `ctd_set_x_shards`): the timed step contains no collective, no copy and no
extra kernel.  The same line carries secondary figures: `halo_allgather` (the entries fetched with one RCCL all-gather
instead), `stitched_c` (+ the all-gather that hands every rank the whole constraint vector), `broadcast_x` (a replicated
iterate sent whole from rank 0) and `no_exchange` (x in place), and the per-rank kernel times.
`value`: weak scaling = shard evaluations all ranks completed per second; strong scaling = evaluations of the whole
transcription per second; both over the max-over-ranks wall time of the K timed steps.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C] [--x-mode peer|halo]

N > 1: one process per GPU.  Under torch.distributed.run (RANK / WORLD_SIZE in the environment) this process is one rank;
started plainly (`python bench.py --gpus N`) it spawns the N rank processes itself -- fresh interpreters, nothing in the
launching process ever touches the GPU (it does not even import torch) -- watches ALL of them, and relays rank 0's line: when
any rank exits non-zero the others are terminated and the launcher exits non-zero within seconds (no rank is left waiting
in a collective for a peer that died).  On a box with fewer than N GPUs the ranks find that out themselves, share the
devices (gloo carries the barrier: RCCL refuses two ranks on one device) and the line says `"rehearsal": true`.

At N > 1 the default line (`--config cfg2_weak`) also carries a `strong` block: the strong-scaling workloads north_star names
(cfg2_strong = the 10 000-step Goddard transcription sharded over the N GPUs, cfg4, cfg5), each with evals/s, per-rank kernel
time and roofline fraction, and the `stitched_c` variant (+ the RCCL all-gather of the constraint vector).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy)
CONFIGS = {
    "cfg2_weak": dict(problem="goddard", scheme="gauss_legendre_2", steps=10000, per_gpu=True, scaling="weak",
                      what="BASELINE.json configs[1] per GPU"),
    "cfg2_strong": dict(problem="goddard", scheme="gauss_legendre_2", steps=10000, per_gpu=False, scaling="strong",
                        what="BASELINE.json configs[1] sharded over the GPUs"),
    "cfg4": dict(problem="goddard", scheme="gauss_legendre_3", steps=80000, per_gpu=False, scaling="strong",
                 what="BASELINE.json configs[3]"),
    "cfg5": dict(problem="quadrotor12", scheme="gauss_legendre_3", steps=20000, per_gpu=False, scaling="strong",
                 what="BASELINE.json configs[4]"),
}


def cpu_baseline(problem, scheme, N, x, budget_s=10.0):
    """CPU figures on the host cores of this box, from the oracle (a C++ restatement of the Julia reference, kind = "port"):
    `value` = reference-faithful mode: cons! once + the Jacobian the way ADNLPModels obtains it (one pass of the constraints
    on one-partial duals per colour of the pattern), single-threaded as the reference is.  `best_effort` = what a CPU
    implementation written for speed reaches: per-step block Jacobians, OpenMP over the time steps, all cores."""
    from oracle.oracle import OracleDOCP
    o = OracleDOCP(problem, scheme, N)
    o.jac_pattern()                      # build pattern + colouring outside the timed region (build-time in the reference)
    o.constraints(x); o.jac_coord(x)     # warm

    def rate(fn, budget):
        n, t0 = 0, time.perf_counter()
        while True:
            fn()
            n += 1
            el = time.perf_counter() - t0
            if el >= budget:
                return n, el

    n, el = rate(lambda: (o.constraints(x), o.jac_coord(x)), budget_s)
    out = {"value": n / el, "unit": "evals/s", "cores": 1, "kind": "port",
           "sample": f"{n} fused evaluations (cons! + {o.jac_ncolors()}-colour forward-dual jac_coord!) of the same "
                     f"{problem}/{scheme} N={N} workload in {el:.1f} s, oracle/ctd_oracle.cpp, 1 thread"}
    # host threads: what the process may use, at most 16 (a one-GPU box shares its host: more threads than its CPU share only
    # oversubscribe -- 256 OpenMP threads on such a box ran 100x slower than 16)
    avail = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    cores = max(1, min(o.jac_ncolors(), avail))
    o.jac_coord_mt(x, cores)
    n, el = rate(lambda: (o.constraints(x), o.jac_coord_mt(x, cores)), budget_s / 3)
    out["multithreaded"] = {"value": n / el, "unit": "evals/s", "cores": cores,
                            "sample": f"{n} evaluations in {el:.1f} s, one colour pass per thread"}
    if hasattr(o, "cons_jac_block"):
        o.cons_jac_block(x, avail)
        n, el = rate(lambda: o.cons_jac_block(x, avail), budget_s / 2)
        out["best_effort"] = {"value": n / el, "unit": "evals/s", "cores": avail, "kind": "port",
                              "sample": f"{n} fused evaluations in {el:.1f} s: per-step block Jacobians (dense local forward "
                                        f"duals), OpenMP over the time steps, {avail} threads (oracle/ctd_oracle.cpp "
                                        f"orc_cons_jac_block)"}
    return out


def self_launch(argv, gpus, script=None):
    """`python bench.py --gpus N` without a launcher: start the N rank processes (fresh interpreters; this process never touches
    the GPU -- it does not import torch and counts no devices: a rank that finds fewer GPUs than ranks switches to the
    rehearsal mode by itself), watch ALL of them, relay rank 0's JSON line.  The first rank that exits non-zero ends the run:
    the others are terminated (then killed) and the launcher returns that code within seconds, instead of leaving rank 0 in a
    collective until some outer timeout.  CTD_BENCH_DEADLINE_S (default 1500) bounds the whole run the same way."""
    import socket
    import subprocess
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out0 = tempfile.TemporaryFile()
    procs = []
    for r in range(gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(gpus), LOCAL_WORLD_SIZE=str(gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    deadline = time.monotonic() + float(os.environ.get("CTD_BENCH_DEADLINE_S", "1500"))
    rc, why = 0, None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            rc, why = bad[0][1], f"rank {bad[0][0]} exited with code {bad[0][1]}"
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > deadline:
            rc, why = 124, "deadline (CTD_BENCH_DEADLINE_S) reached"
            break
        time.sleep(0.2)
    if why is not None:          # end exactly the processes started here, by pid
        sys.stderr.write(f"bench.py launcher: {why}; terminating the other ranks\n")
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    return rc if rc >= 0 else 128 - rc       # (a rank killed by a signal: 128 + signal number, like a shell)


def main():
    if "WORLD_SIZE" not in os.environ:
        ap = argparse.ArgumentParser(add_help=False)
        ap.add_argument("--gpus", type=int, default=1)
        gpus = ap.parse_known_args()[0].gpus
        if gpus > 1:
            sys.exit(self_launch(sys.argv[1:], gpus))
    # dmabuf IPC (the only mode the host driver supports): needed by hipIpcGetMemHandle / RCCL, whoever launched this rank
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # exactly ONE line on stdout: libraries that print banners there (RCCL prints its version block on communicator creation)
    # write to stderr for the duration of the run; the JSON line goes to the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        _main(real_stdout)
    finally:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        os.close(real_stdout)


def _main(real_stdout):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg2_weak")
    ap.add_argument("--x-mode", choices=("peer", "halo"), default="peer",
                    help="N > 1: how a rank obtains the few entries of x its neighbours own: 'peer' = read in place by the kernel from "
                         "the owner's HBM (IPC-mapped buffers over xGMI; falls back to 'halo' when the mapping is refused), 'halo' = "
                         "one all-gather before the evaluation.  The line records the other one as a secondary figure")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the kernel-only figures of the other configs")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the `strong` block (cfg2_strong / cfg4 / cfg5) of the default line")
    ap.add_argument("--two-streams", action="store_true",
                    help="also time independent evaluations on two streams (overlapping kernels: keeps it out of the default run, whose rocprofv3 "
                         "kernel statistics are the timed region's)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    PROBLEM, SCHEME = cfg["problem"], cfg["scheme"]
    t_start = time.monotonic()
    budget_s = float(os.environ.get("CTD_BENCH_BUDGET_S", "420"))      # optional blocks are skipped once this much wall time is gone

    import datetime
    import numpy as np
    import torch
    import torch.distributed as dist

    import ctdirect_jl_amd as ct
    from ctdirect_jl_amd import dist as ctdist
    from helpers import bench_inputs, describe

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU: the engine has no CPU path"
    # rehearsal knobs (one-GPU box): CTD_BENCH_DEVICE pins every rank to one device, CTD_BENCH_BACKEND=gloo replaces RCCL
    # (which refuses two ranks on one device).  The driver's multi-GPU run uses neither: one rank per GPU over RCCL/xGMI.
    rehearsal = bool(os.environ.get("CTD_BENCH_DEVICE"))
    if rehearsal:
        local_rank = int(os.environ["CTD_BENCH_DEVICE"])
    elif world > 1 and torch.cuda.device_count() < int(os.environ.get("LOCAL_WORLD_SIZE", world)):
        # a box with fewer GPUs than ranks (whoever launched them): the ranks share the devices (a rehearsal of the code path,
        # not a scaling measurement; RCCL refuses two ranks on one device, gloo carries the barrier)
        rehearsal = True
        local_rank = local_rank % max(1, torch.cuda.device_count())
        os.environ.setdefault("CTD_BENCH_BACKEND", "gloo")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = os.environ.get("CTD_BENCH_BACKEND", "nccl")
    # CTD_DIST_FORCE=1 (rehearsal on the one-GPU box): a ONE-rank RCCL group, every collective of the multi-GPU step issued
    dist_on = world > 1 or os.environ.get("CTD_DIST_FORCE") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # a collective whose peer never arrives (a rank that died, a mapping that hangs) ends this rank after the timeout instead
        # of holding the line until an outer limit; the launcher / torch.distributed.run then ends the others
        tmo = datetime.timedelta(seconds=float(os.environ.get("CTD_BENCH_COLLECTIVE_TIMEOUT_S", "180")))
        if backend == "nccl":     # bind the communicator to this rank's GPU up front (barriers need no device guess)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
    if os.environ.get("CTD_BENCH_TEST_EXIT_RANK") == str(rank):      # test hook: this rank dies before the first collective of the run
        os._exit(17)

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def within_budget():
        """rank 0 decides whether an optional block still runs; every rank follows (one tiny collective)"""
        ok = torch.tensor([1.0 if time.monotonic() - t_start < budget_s else 0.0], dtype=torch.float64,
                          device=dev if backend == "nccl" else "cpu")
        if dist_on:
            dist.broadcast(ok, src=0)
        return bool(ok.item() == 1.0)

    region = {}
    # the timing machinery itself is warmed before the timed region (first creation / recording of HIP events costs tens of
    # microseconds once per process)
    _w0, _w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _w0.record(); _w1.record(); torch.cuda.synchronize(dev); _w0.elapsed_time(_w1)

    def timed(step, warmup, steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(warmup):
            step()
        e0.record(); e1.record()
        sync_all()
        # HIP events on the launch stream (the handle launches on torch's current stream) bracket the timed region: GPU-side
        # duration of the K launches, the figure the roofline uses
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            step()
        e1.record()
        # (polling the closing event before the synchronisation was tried against a once-seen 10 ms wake-up stall and costs
        # 12 us per region -- hipEventQuery in a loop, then a synchronisation that still takes its 16 us: profiles/r03_experiments.md)
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0       # this rank's K steps are done; the MAX over the ranks (below) is when the last one was
        if dist_on:                         # the closing barrier of the bracket: its own latency (a collective) is not a step
            dist.barrier()
            torch.cuda.synchronize(dev)
        region["ms_per_launch"] = e0.elapsed_time(e1) / steps
        if dist_on:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t[0])
        return el

    def measure(name, steps, warmup, ksec):
        """One workload of CONFIGS on this job's ranks: the timed step + its secondary figures + per-rank kernel times.
        N = 1: one fused evaluation, pointers pre-bound.  N > 1: the rank's evaluation on the SHARDED iterate -- x_mode "peer":
        the neighbours' entries read in place by the kernel, outputs row-sharded, no collective before or
        after the kernel; x_mode "halo": one all-gather of the few entries first."""
        from types import SimpleNamespace
        cf = CONFIGS[name]
        prob, sch = cf["problem"], cf["scheme"]
        N = cf["steps"] * world if cf["per_gpu"] else cf["steps"]

        def make(steps=None):      # the handle launches on torch's current stream: ordered with the RCCL collectives
            return ct.DOCP(prob, N, sch, device=local_rank, steps=steps, stream="torch")

        sh = ctdist.ShardedDOCP(make, N, world=world, rank=rank)
        docp = sh.docp
        x_host = bench_inputs(describe(docp, prob, sch), perturb=1e-3)
        x_full = torch.from_numpy(x_host).to(dev)
        if dist_on:           # sharded iterate: own variables + the replicated v, NaN everywhere else
            xs = np.full_like(x_host, np.nan)
            a_, b_ = sh.owned_variables()
            xs[a_:b_] = x_host[a_:b_]
            nv_ = docp.dims.NLP_v
            if nv_:
                xs[-nv_:] = x_host[-nv_:]
            x = torch.from_numpy(xs).to(dev)
        else:
            x = x_full
        c = torch.zeros(docp.dim_NLP_constraints, dtype=torch.float64, device=dev)
        vals = torch.zeros(docp.nnzj, dtype=torch.float64, device=dev)
        x_mode = args.x_mode if dist_on else None
        peer_error = None
        if dist_on and world > 1 and x_mode == "peer":
            try:        # (raises on every rank together or on none: the ranks agree inside)
                sh.enable_peer_x(x)
            except RuntimeError as e:      # IPC mapping refused on this node: fall back to fetching the entries with one all-gather
                peer_error = repr(e)[:300]
                x_mode = "halo"
        el = timed(sh.bind_cons_jac(x, c, vals, stitch=False, x_mode=x_mode), warmup, steps)
        out = SimpleNamespace(name=name, cfg=cf, N=N, sh=sh, docp=docp, x_host=x_host, x_full=x_full, x=x, c=c, vals=vals, el=el,
                              steps=steps, x_mode=x_mode, peer_error=peer_error, region_ms=region["ms_per_launch"],
                              per_step_value=(world if cf["scaling"] == "weak" else 1), secondary={})
        secondary = out.secondary
        if dist_on:
            # the rank's outputs from the sharded iterate (NaN outside what it owns) equal those from the whole iterate, bit for bit
            c_chk, v_chk = torch.zeros_like(c), torch.zeros_like(vals)
            sh.bind_cons_jac(x_full, c_chk, v_chk, stitch=False, x_mode=None)()
            sh.bind_cons_jac(x, c, vals, stitch=False, x_mode=x_mode)()
            torch.cuda.synchronize(dev)
            r0, r1 = docp.shard.c_row_begin, docp.shard.c_row_end
            lo_, hi_ = docp.shard.vals_main_begin, docp.shard.vals_main_end
            same = torch.tensor([float(torch.equal(c[r0:r1], c_chk[r0:r1]) and torch.equal(vals[lo_:hi_], v_chk[lo_:hi_])
                                       and (world == 1 or bool(torch.isnan(x).any())))], dtype=torch.float64, device=dev)
            dist.all_reduce(same, op=dist.ReduceOp.MIN)
            del c_chk, v_chk
            secondary["sharded_iterate_check"] = {"bit_identical_to_whole_iterate_on_every_rank": bool(same.item() == 1.0),
                                                  "what": "each rank's c rows and CSC range from its NaN-padded shard of x vs from the whole x"}
            ks = max(1, min(steps, ksec))
            if peer_error:
                secondary["peer_error"] = peer_error
            other = "halo" if x_mode == "peer" else "peer"
            a_, b_ = sh.owned_variables()
            flag = torch.zeros(1, dtype=torch.float64, device=dev)

            def ordered(launch):
                # what a solver iteration adds around the evaluation: the rank updates ITS entries of x (here: multiplied by one),
                # then a collective every rank takes part in (the step-acceptance all-reduce; gloo moves host memory: a barrier
                # stands in) orders all updates before any rank's next evaluation reads its neighbours' entries in place
                def call():
                    x[a_:b_].mul_(1.0)
                    if backend == "nccl":
                        dist.all_reduce(flag)
                    else:
                        torch.cuda.synchronize(dev)
                        dist.barrier()
                    launch()
                return call
            figures = [
                ("other_x_mode", dict(stitch=False, x_mode=other), None,
                 "the timed step with the other way of obtaining the neighbours' entries: 'halo' = one all-gather (pack / all-gather / unpack) "
                 "before the evaluation, 'peer' = read in place"),
                ("stitched_c", dict(stitch=True, x_mode=x_mode), None, "the timed step + all-gather of the row blocks of c (every rank ends with the whole c)"),
                ("ordered_step", dict(stitch=False, x_mode=x_mode), ordered,
                 "own-shard update of x + one all-reduce of one double (a solver's step acceptance, which orders every rank's update before "
                 "its neighbours' reads) + the timed step: what the ordering contract of the in-place reads costs per iteration"),
                ("broadcast_x", dict(stitch=False, x_mode="broadcast"), None, "replicated iterate: rank 0 broadcasts all of x before the evaluation"),
                ("no_exchange", dict(stitch=False, x_mode=None), None, "evaluation only, the whole x in place on every rank: what reading the neighbours' entries in place costs is the difference to the timed step")]
            for key, kw, wrap, what in figures:
                if key == "other_x_mode" and (peer_error or world == 1):
                    continue
                try:      # a secondary figure must never cost the line
                    xin = x if kw["x_mode"] in ("peer", "halo") else x_full
                    call = sh.bind_cons_jac(xin, c, vals, **kw)
                    els = timed(wrap(call) if wrap else call, min(warmup, 50), ks)
                    secondary[key] = {"value": ks * out.per_step_value / els, "unit": "evals/s", "ms_per_step": els / ks * 1e3,
                                      "steps": ks, "what": what}
                    if key == "other_x_mode":
                        secondary[key]["x_mode"] = other
                except Exception as e:
                    secondary[key] = {"error": repr(e)[:300]}
            if "other_x_mode" in secondary and "x_mode" in secondary["other_x_mode"]:      # (the key earlier lines used)
                secondary["halo_allgather" if other == "halo" else "peer_in_place"] = secondary["other_x_mode"]
            # one whole solver iteration on the SHARDED iterate and SHARDED multipliers (round 4): objective (all-reduce of one double),
            # gradient (own entries; all-reduce of the nv entries of d/dv), constraints + Jacobian (the timed step), multipliers of the
            # previous rank's last step + tail rows (one small all-gather), Hessian values (all-reduce of the V x V entries) -- nothing
            # replicated, nothing of length nvar or ncon moved
            try:
                ys = np.full(docp.dim_NLP_constraints, np.nan)
                ca_, cz_ = sh.owned_constraints()
                ys[ca_:cz_] = 0.6 + 0.4 * np.sin(0.7 * np.arange(ca_, cz_) + 0.3)
                y_sh = torch.from_numpy(ys).to(dev)
                g_sh = torch.zeros(docp.dim_NLP_variables, dtype=torch.float64, device=dev)
                h_sh = torch.zeros(docp.nnzh, dtype=torch.float64, device=dev)
                cj = sh.bind_cons_jac(x, c, vals, stitch=False, x_mode=x_mode)

                def iteration():
                    sh.obj(x, as_tensor=True)
                    cj()                                  # (x_mode "halo": its all-gather also serves the gradient and the Hessian below)
                    sh.grad(x, g_sh)
                    sh.exchange_multipliers(y_sh)
                    sh.hess_coord(x, y_sh, 1.0, h_sh)
                ki = max(20, min(ks, 200))
                els = timed(iteration, min(warmup, 20), ki)
                secondary["sharded_iteration"] = {
                    "ms_per_iteration": els / ki * 1e3, "iterations": ki,
                    "finite_own_outputs": bool(torch.isfinite(g_sh[a_:b_]).all()) and bool(torch.isfinite(h_sh[docp.hess_shard_info()[0]:docp.hess_shard_info()[1]]).all()),
                    "what": "objective + gradient + constraints + Jacobian values + Hessian values of the rank's shard on the sharded x and sharded y "
                            "(collectives: 1 double, nv doubles, cb + tail doubles, the V x V entries); never `value`"}
                del y_sh, g_sh, h_sh
            except Exception as e:
                secondary["sharded_iteration"] = {"error": repr(e)[:300]}
        # kernel time per rank: per-dispatch start / stop events (hipExtLaunchKernelGGL), median of five batches of 200
        out.per_dispatch = sorted(docp.time_cons_jac(x_full, c, vals, iters=200) for _ in range(5))[2]
        # algorithmic bytes of one launch: read the shard's x, write its c rows and Jacobian values:
        # B = 8 (nvar + ncon + nnzj) of the per-rank sub-problem
        b, e = sh.steps
        out.one = ct.DOCP(prob, e - b, sch, device=-1)
        out.alg_bytes = 8 * (out.one.dim_NLP_variables + out.one.dim_NLP_constraints + out.one.nnzj)
        out.per_rank = None
        if dist_on:
            t = torch.zeros(world, dtype=torch.float64, device=dev)
            t[rank] = out.per_dispatch
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            out.per_rank = [{"rank": r, "kernel_ms": float(t[r]), "frac": out.alg_bytes / (float(t[r]) * 1e-3) / 1e9 / HBM_PEAK_GBS}
                            for r in range(world)]
        return out

    m = measure(args.config, args.steps, args.warmup, 500)
    sh, docp, x_host, x_full, x, c, vals = m.sh, m.docp, m.x_host, m.x_full, m.x, m.c, m.vals
    N, el, x_mode, one, alg_bytes = m.N, m.el, m.x_mode, m.one, m.alg_bytes
    region_ms, per_step_value, secondary, per_dispatch, per_rank = m.region_ms, m.per_step_value, m.secondary, m.per_dispatch, m.per_rank
    b, e = sh.steps

    # N > 1, default line: the strong-scaling workloads north_star names ("a 10k-step Goddard transcription at 1/2/4/8" =
    # cfg2_strong; BASELINE configs[3] / [4] sharded), each with its stitched-c variant -- `value` stays the weak-scaling figure
    strong = {}
    if world > 1 and args.config == "cfg2_weak" and not args.no_strong:
        for name in ("cfg2_strong", "cfg4", "cfg5"):
            if not within_budget():
                strong[name] = {"skipped": f"wall-time budget of {budget_s:.0f} s (CTD_BENCH_BUDGET_S) used up"}
                continue
            try:
                ks = max(20, min(args.steps, 500 if name == "cfg2_strong" else 200))
                ms_ = measure(name, ks, min(args.warmup, 50), 200)
                blk = {"value": ks / ms_.el, "unit": "evals/s", "scaling": "strong", "steps": ks, "ms_per_step": ms_.el / ks * 1e3,
                       "workload": f"{ms_.cfg['problem']} / {ms_.cfg['scheme']}, {ms_.N} time steps sharded over {world} GPUs: "
                                   f"{ms_.cfg['what']}; value = evaluations of the whole transcription per second",
                       "x_mode": ms_.x_mode, "per_rank": ms_.per_rank, "algorithmic_bytes_per_rank_launch": ms_.alg_bytes}
                blk.update(ms_.secondary)
                strong[name] = blk
                ms_.sh.close()
                del ms_
                torch.cuda.empty_cache()
            except Exception as ex:      # (a failure on one rank only leaves the others in a collective: its timeout ends the run)
                strong[name] = {"error": repr(ex)[:300]}

    # roofline of the dominant (only) kernel, per rank.  kernel_ms: N = 1: HIP events on the launch stream around the K
    # timed launches / K (back-to-back launches: the kernel's average duration including the dispatch gap); N > 1 (the
    # region also holds collectives): per-dispatch start / stop events (hipExtLaunchKernelGGL), median of five batches of 200
    kernel_ms = region_ms if not dist_on else per_dispatch
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    traffic, traffic_src = None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if world == 1 and args.config == "cfg2_weak" and os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("bench_kernel", {}).get("hbm_bytes_per_launch")
            traffic_src = ("IMPORTED from profiles/pmc_traffic.json (separate rocprofv3 --pmc passes of this command, "
                           "profiles/collect.sh); not measured in this run")
        except Exception:
            traffic = None

    # the other single-GPU BASELINE configs (parity-test cases, not the bench line): kernel time and roofline fraction
    others, hessian, optimized, csr_rows, large = [], [], [], [], []
    if world == 1 and not args.no_extras:
        for prob, sch, n in (("double_integrator_path", "midpoint", 100000), ("goddard", "gauss_legendre_3", 80000),
                             ("quadrotor", "gauss_legendre_3", 20000), ("quadrotor12", "gauss_legendre_3", 20000),
                             ("quadrotor12", "midpoint", 20000)):
            d2 = ct.DOCP(prob, n, sch, device=local_rank, stream="torch")
            x2 = torch.from_numpy(bench_inputs(describe(d2, prob, sch), perturb=1e-3)).to(dev)
            c2 = torch.zeros(d2.dim_NLP_constraints, dtype=torch.float64, device=dev)
            v2 = torch.zeros(d2.nnzj, dtype=torch.float64, device=dev)
            ms2 = d2.time_cons_jac(x2, c2, v2, iters=50)
            b2 = 8 * (d2.dim_NLP_variables + d2.dim_NLP_constraints + d2.nnzj)
            others.append({"workload": f"{prob}/{sch} N={n}", "kernel_ms": ms2, "algorithmic_bytes": b2,
                           "achieved_GBs": b2 / (ms2 * 1e-3) / 1e9, "frac_of_8TBs": b2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS})
            d2.close()
            del x2, c2, v2
        # CTD_PATTERN_OPTIMIZED (the sparsity the reference's default backend detects): fewer entries = fewer bytes written
        for prob, sch, n in (("goddard", "gauss_legendre_3", 80000), ("quadrotor12", "gauss_legendre_3", 20000)):
            d2 = ct.DOCP(prob, n, sch, device=local_rank, stream="torch", pattern="optimized")
            d0 = ct.DOCP(prob, n, sch, device=-1)
            x2 = torch.from_numpy(bench_inputs(describe(d2, prob, sch), perturb=1e-3)).to(dev)
            c2 = torch.zeros(d2.dim_NLP_constraints, dtype=torch.float64, device=dev)
            v2 = torch.zeros(d2.nnzj, dtype=torch.float64, device=dev)
            ms2 = d2.time_cons_jac(x2, c2, v2, iters=50)
            b2 = 8 * (d2.dim_NLP_variables + d2.dim_NLP_constraints + d2.nnzj)
            b0 = 8 * (d0.dim_NLP_variables + d0.dim_NLP_constraints + d0.nnzj)
            optimized.append({"workload": f"{prob}/{sch} N={n}, pattern=optimized", "nnzj": d2.nnzj, "nnzj_manual": d0.nnzj,
                              "kernel_ms": ms2, "algorithmic_bytes": b2, "bytes_saved_vs_manual": b0 - b2,
                              "achieved_GBs": b2 / (ms2 * 1e-3) / 1e9, "frac_of_8TBs": b2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS})
            d2.close()
            del x2, c2, v2
        # CTD_ORDER_CSR (north_star: "assembled ... in CSR on device"): the same kernel with a row-order emit template; the same bytes
        # are written, so the time should stay within a few per cent of the CSC order's (never `value`: the metric's order is the
        # reference's CSC)
        for prob, sch, n, pat in (("goddard", "gauss_legendre_2", 10000, "manual"), ("goddard", "gauss_legendre_3", 80000, "manual"),
                                  ("quadrotor12", "gauss_legendre_3", 20000, "manual"), ("quadrotor12", "gauss_legendre_3", 20000, "optimized"),
                                  ("double_integrator_path", "midpoint", 100000, "manual")):
            try:
                row = {"workload": f"{prob}/{sch} N={n}, pattern={pat}"}
                for order in ("csc", "csr"):
                    d2 = ct.DOCP(prob, n, sch, device=local_rank, stream="torch", pattern=pat, value_order=order)
                    x2 = torch.from_numpy(bench_inputs(describe(d2, prob, sch), perturb=1e-3)).to(dev)
                    c2 = torch.zeros(d2.dim_NLP_constraints, dtype=torch.float64, device=dev)
                    v2 = torch.zeros(d2.nnzj, dtype=torch.float64, device=dev)
                    ms2 = sorted(d2.time_cons_jac(x2, c2, v2, iters=50) for _ in range(3))[1]
                    b2 = 8 * (d2.dim_NLP_variables + d2.dim_NLP_constraints + d2.nnzj)
                    row[order] = {"kernel_ms": ms2, "frac_of_8TBs": b2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, "launch": d2.launch_info()}
                    row["algorithmic_bytes"] = b2
                    d2.close()
                    del x2, c2, v2
                row["csr_over_csc"] = row["csr"]["kernel_ms"] / row["csc"]["kernel_ms"]
                csr_rows.append(row)
            except Exception as ex:
                csr_rows.append({"workload": f"{prob}/{sch} N={n}, pattern={pat}", "error": repr(ex)[:300]})
        # the SAME kernels on grids large enough to be bandwidth-bound (the bench workload's 10 000 steps are one round of workgroups:
        # launch + one dependent chain; at 4 million steps the launch is 0.1 % of the kernel): what fraction of the 8 TB/s the
        # emission sustains.  (tests/test_gpu_max_sizes.py checks 2^24 steps -- 3.2e9 values -- bit for bit.)
        for prob, sch, n, order in (("goddard", "gauss_legendre_2", 1 << 22, "csc"), ("goddard", "gauss_legendre_2", 1 << 22, "csr"),
                                    ("goddard", "gauss_legendre_3", 1 << 22, "csc"), ("double_integrator_path", "midpoint", 1 << 23, "csc")):
            try:
                d2 = ct.DOCP(prob, n, sch, device=local_rank, stream="torch", value_order=order)
                x2 = torch.from_numpy(bench_inputs(describe(d2, prob, sch), perturb=1e-3)).to(dev)
                c2 = torch.zeros(d2.dim_NLP_constraints, dtype=torch.float64, device=dev)
                v2 = torch.zeros(d2.nnzj, dtype=torch.float64, device=dev)
                ms2 = sorted(d2.time_cons_jac(x2, c2, v2, iters=12) for _ in range(3))[1]      # (the first pass over fresh gigabytes is a cold one)
                b2 = 8 * (d2.dim_NLP_variables + d2.dim_NLP_constraints + d2.nnzj)
                large.append({"workload": f"{prob}/{sch} N={n}, order={order}", "kernel_ms": ms2, "algorithmic_bytes": b2,
                              "achieved_GBs": b2 / (ms2 * 1e-3) / 1e9, "frac_of_8TBs": b2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "launch": d2.launch_info()})
                d2.close()
                del x2, c2, v2
                torch.cuda.empty_cache()
            except Exception as ex:
                large.append({"workload": f"{prob}/{sch} N={n}, order={order}", "error": repr(ex)[:300]})
        # the Hessian-of-the-Lagrangian row (hess_coord!): kernel-only figures, not part of `value`.
        # Algorithmic bytes: read x and y, write the lower-triangular values: 8 (nvar + ncon + nnzh).
        for prob, sch, n, pat in (("goddard", "gauss_legendre_2", 10000, "manual"), ("goddard", "gauss_legendre_3", 80000, "manual"),
                                  ("quadrotor", "gauss_legendre_3", 20000, "manual"), ("quadrotor12", "gauss_legendre_3", 20000, "manual"),
                                  # the pattern the reference's default backend uses (`:optimized`): a fifth of the entries, the same work
                                  ("goddard", "gauss_legendre_3", 80000, "optimized"), ("quadrotor", "gauss_legendre_3", 20000, "optimized"),
                                  ("quadrotor12", "gauss_legendre_3", 20000, "optimized"), ("quadrotor12", "midpoint", 20000, "optimized")):
            d2 = ct.DOCP(prob, n, sch, device=local_rank, stream="torch", pattern=pat)
            x2 = torch.from_numpy(bench_inputs(describe(d2, prob, sch), perturb=1e-3)).to(dev)
            y2 = torch.from_numpy(0.6 + 0.4 * np.sin(0.7 * np.arange(d2.dim_NLP_constraints) + 0.3)).to(dev)
            h2 = torch.zeros(d2.nnzh, dtype=torch.float64, device=dev)
            ms2 = d2.time_hess(x2, y2, h2, 1.0, iters=50)
            b2 = 8 * (d2.dim_NLP_variables + d2.dim_NLP_constraints + d2.nnzh)
            hessian.append({"workload": f"{prob}/{sch} N={n}" + (", pattern=optimized" if pat == "optimized" else ""), "nnzh": d2.nnzh,
                            "kernel_ms": ms2, "algorithmic_bytes": b2,
                            "achieved_GBs": b2 / (ms2 * 1e-3) / 1e9, "frac_of_8TBs": b2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS})
            d2.close()
            del x2, y2, h2

    # "objective reported separately", "also report host-pointer mode": the other callbacks of the SAME
    # workload (never `value`): objective, gradient, one whole solver iteration (obj + grad + cons + Jacobian + Hessian in two
    # launches) with everything resident in HBM, and the fused evaluation through the host-pointer entry point (PCIe inclusive)
    separately = {}
    if world == 1 and not args.no_extras:
        try:
            import time as _t
            K2 = 200
            g_ = torch.zeros(docp.dim_NLP_variables, dtype=torch.float64, device=dev)
            f_ = torch.zeros(1, dtype=torch.float64, device=dev)
            y_ = torch.from_numpy(0.6 + 0.4 * np.sin(0.7 * np.arange(docp.dim_NLP_constraints) + 0.3)).to(dev)
            h_ = torch.zeros(docp.nnzh, dtype=torch.float64, device=dev)

            def rate(fn, k=K2):
                for _ in range(20):
                    fn()
                torch.cuda.synchronize(dev)
                t0 = _t.perf_counter()
                for _ in range(k):
                    fn()
                torch.cuda.synchronize(dev)
                return (_t.perf_counter() - t0) / k
            t_obj = rate(lambda: docp.eval_all(x_full, None, 1.0, f_, None, None, None, None))
            t_grad = rate(lambda: docp.grad(x_full, g_, sync=False))
            t_iter = rate(lambda: docp.eval_all(x_full, y_, 1.0, f_, g_, c, vals, h_))
            xh, ch, vh = ct.pinned_empty(docp.dim_NLP_variables), ct.pinned_empty(docp.dim_NLP_constraints), ct.pinned_empty(docp.nnzj)
            xh[:] = x_host
            t_host = rate(lambda: docp.cons_jac(xh, ch, vh), 50)
            # two handles on two streams, launches alternating: independent evaluations (line-search candidates, multiple starts)
            # overlap one kernel's launch / drain with the other's evaluation.  NOT `value`: a solver's evaluations depend on each other.
            t_two, same2 = float("nan"), None
            s2 = torch.cuda.Stream(device=dev) if args.two_streams else None
            if s2 is not None:
              with torch.cuda.stream(s2):
                d_b = ct.DOCP(PROBLEM, N, SCHEME, device=local_rank, stream="torch")
                c_b, v_b = torch.zeros_like(c), torch.zeros_like(vals)
                l_b = d_b.bind_cons_jac(x_full, c_b, v_b, sync=False)
              l_a = docp.bind_cons_jac(x_full, c, vals, sync=False)
              t_two = rate(lambda: (l_a(), l_b()), 500) / 2.0
              same2 = bool(torch.equal(c_b, c) and torch.equal(v_b, vals))
              d_b.close()
            # the SAME workload with the values in CSR order (north_star: "in CSR on device"; NLPModels' jac_structure! / jac_coord! are
            # coordinate lists in any order, so a shim may serve either): same kernel, row-order emit template.  Never `value`, which
            # stays on the reference's own order (SparseArrays.sparse = CSC)
            d_r = ct.DOCP(PROBLEM, N, SCHEME, device=local_rank, stream="torch", value_order="csr")
            c_r, v_r = torch.zeros_like(c), torch.zeros_like(vals)
            l_r = d_r.bind_cons_jac(x_full, c_r, v_r, sync=False)
            t_csr = rate(l_r, min(args.steps, 2000))
            t_csc = rate(docp.bind_cons_jac(x_full, c, vals, sync=False), min(args.steps, 2000))
            d_r.close()
            separately = {"csr_order_same_workload": {
                "evals_per_s": 1.0 / t_csr, "ms_per_step": t_csr * 1e3, "csc_order_same_loop_ms_per_step": t_csc * 1e3,
                "what": "ctd_desc.value_order = CTD_ORDER_CSR: K back-to-back fused evaluations of the bench workload, wall clock around the loop "
                        "(the CSC figure of the same loop beside it)"}}
            separately.update({"same_workload_other_callbacks": {
                "objective_device": {"ms_per_call": t_obj * 1e3, "calls_per_s": 1.0 / t_obj},
                "gradient_device": {"ms_per_call": t_grad * 1e3, "calls_per_s": 1.0 / t_grad},
                "whole_iteration_device": {"ms_per_call": t_iter * 1e3, "calls_per_s": 1.0 / t_iter,
                                           "what": "ctd_eval_all_dev_async: objective + gradient + constraints + Jacobian values + Hessian values at one (x, y), two launches"},
                "fused_cons_jac_host_pointers_pinned": {"ms_per_call": t_host * 1e3, "calls_per_s": 1.0 / t_host,
                                                         "bytes_over_pcie": 8 * (docp.dim_NLP_variables + docp.dim_NLP_constraints + docp.nnzj),
                                                         "what": "ctd_cons_jac on page-locked host arrays: H2D x + kernel + D2H c, values (PCIe inclusive; never `value`)"}}})
            if s2 is not None:
                separately["two_streams_alternating"] = {
                    "ms_per_eval": t_two * 1e3, "evals_per_s": 1.0 / t_two, "outputs_identical": same2,
                    "what": "independent evaluations of the same workload on two handles / two streams, launched alternately (never `value`; --two-streams)"}
            del g_, f_, y_, h_
        except Exception as e:      # a secondary figure must never cost the line
            separately = {"same_workload_other_callbacks": {"error": repr(e)[:300]}}

    # per-workload rocprofv3 rows (profiles/collect_workloads.sh: one process per workload, so a row is ONE workload): IMPORTED
    # from the committed summary, not measured in this run
    wl_rows = {}
    try:
        wl_file = next(f for f in ("r04_workloads.json", "r03_workloads.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
        for r in json.load(open(os.path.join(ROOT, "profiles", wl_file)))["workloads"]:
            if r.get("value_order", "csc") != "csc":
                continue
            wl_rows[(r["problem"], r["scheme"], r["N"], r["pattern"], r["kernel"])] = {
                k: r[k] for k in ("rocprof_avg_ns", "frac_of_8TBs", "hbm_bytes_per_launch", "traffic_over_algorithmic")}
    except Exception:
        pass
    for lst, pat, kern in ((others, "manual", "cons_jac"), (optimized, "optimized", "cons_jac"), (hessian, "manual", "hess")):
        for e in lst:
            prob_sch, n_ = e["workload"].split(",")[0].split(" N=")
            prob_, sch_ = prob_sch.split("/")
            row = wl_rows.get((prob_, sch_, int(n_), "optimized" if "pattern=optimized" in e["workload"] else pat, kern))
            if row:
                e["rocprof_imported_from_profiles_workloads"] = row

    if rank == 0:
        shard_txt = (f"{cfg['steps']} time steps per GPU (global grid {N} steps, time-step sharded)" if cfg["per_gpu"]
                     else f"{N} time steps" + (f" sharded over {world} GPUs ({e - b} per GPU)" if world > 1 else ""))
        out = {
            "metric": "NLP callback evals/s (constraints+sparse Jac), N-step Goddard, 1/2/4/8 GPU",
            "value": args.steps * per_step_value / el,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": cfg["scaling"],
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.config}: {PROBLEM} / {SCHEME}, {shard_txt}; fused cons!+jac_coord! on HBM-resident x; "
                            + cfg["what"] + ("" if world == 1 else
                                             ("; timed step = shard evaluation on the sharded iterate, neighbours' entries read "
                                              "in place by the kernel (no collective in the step), outputs row-sharded; value counts "
                                              if x_mode == "peer" else
                                              "; timed step = halo exchange of the sharded iterate (one all-gather; mapping the "
                                              "neighbours' buffers failed on this node) + shard evaluation; value counts ")
                                             + ("one shard evaluation per GPU per step" if cfg["scaling"] == "weak"
                                                else "one evaluation of the whole transcription per step")),
                "nvar_per_gpu": one.dim_NLP_variables, "ncon_per_gpu": one.dim_NLP_constraints, "nnzj_per_gpu": one.nnzj,
                "launch": docp.launch_info(),
            },
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": f"ctd::cons_jac_kernel<{PROBLEM}, {SCHEME}>", "kernel_ms": kernel_ms,
                         "kernel_ms_per_dispatch_events": per_dispatch,
                         "timing": ("HIP events on the launch stream around the K timed launches / K" if world == 1 else
                                    "per-dispatch HIP events of rank 0's kernel (the timed region also holds collectives)"),
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        if per_rank is not None:
            out["roofline"]["per_rank"] = per_rank
        if rehearsal:
            out["rehearsal"] = True
            out["config"]["rehearsal"] = f"{world} ranks share device {local_rank} of a box with fewer GPUs; backend {backend}"
        if others:
            out["other_configs_kernel_only"] = others
        if optimized:
            out["optimized_pattern_kernel_only"] = optimized
        if csr_rows:
            out["csr_order_kernel_only"] = csr_rows
        if large:
            out["large_grid_kernel_only"] = large
        if hessian:
            out["hessian_kernel_only"] = hessian
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(PROBLEM, SCHEME, N, x_host)
        out.update(secondary)
        if strong:
            out["strong"] = strong
        if dist_on:
            out["config"]["x_mode"] = x_mode
        out.update(separately)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
