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

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C]

N > 1: one process per GPU.  Under torch.distributed.run (RANK / WORLD_SIZE in the environment) this process is one rank;
started plainly (`python bench.py --gpus N`) it spawns the N rank processes itself -- before anything touches the GPU --
and relays rank 0's line.  On a box with fewer than N GPUs the ranks share the devices (gloo carries the barrier: RCCL
refuses two ranks on one device) and the line says `"rehearsal": true`.
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


def self_launch(argv, gpus):
    """`python bench.py --gpus N` without a launcher: start the N rank processes (fresh interpreters; nothing in this
    process has touched the GPU), relay rank 0's JSON line, exit with the worst return code."""
    import socket
    import subprocess
    import torch
    ndev = torch.cuda.device_count()          # (counting devices does not initialise the GPU)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(gpus), LOCAL_WORLD_SIZE=str(gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if ndev < gpus:                       # rehearsal: the ranks share the box's device(s)
            env["CTD_BENCH_DEVICE"] = str(r % max(ndev, 1))
            env.setdefault("CTD_BENCH_BACKEND", "gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        p.wait()
        rc = rc or p.returncode
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return rc


def main():
    if "WORLD_SIZE" not in os.environ:
        ap = argparse.ArgumentParser(add_help=False)
        ap.add_argument("--gpus", type=int, default=1)
        gpus = ap.parse_known_args()[0].gpus
        if gpus > 1:
            sys.exit(self_launch(sys.argv[1:], gpus))
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the kernel-only figures of the other configs")
    ap.add_argument("--two-streams", action="store_true",
                    help="also time independent evaluations on two streams (overlapping kernels: keeps it out of the default run, whose rocprofv3 "
                         "kernel statistics are the timed region's)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    PROBLEM, SCHEME = cfg["problem"], cfg["scheme"]

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
        # launched by torch.distributed.run on a box with fewer GPUs than ranks: the ranks share the devices (a rehearsal of the
        # code path, not a scaling measurement; RCCL refuses two ranks on one device, gloo carries the barrier)
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
        if backend == "nccl":     # bind the communicator to this rank's GPU up front (barriers need no device guess)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    N = cfg["steps"] * world if cfg["per_gpu"] else cfg["steps"]

    def make(steps=None):      # the handle launches on torch's current stream: ordered with the RCCL collectives
        return ct.DOCP(PROBLEM, N, SCHEME, device=local_rank, steps=steps, stream="torch")

    sh = ctdist.ShardedDOCP(make, N, world=world, rank=rank)
    docp = sh.docp
    x_host = bench_inputs(describe(docp, PROBLEM, SCHEME), perturb=1e-3)
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

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize(dev)

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

    # The timed step.  N = 1: one fused evaluation, pointers pre-bound.  N > 1: the rank's evaluation on the SHARDED iterate,
    # the neighbours' entries read in place by the kernel (x_mode "peer"); outputs stay row-sharded: no
    # collective before or after the kernel.
    x_mode = "peer" if dist_on else None
    peer_error = None
    if dist_on and world > 1:
        try:        # (raises on every rank together or on none: the ranks agree inside)
            sh.enable_peer_x(x)
        except RuntimeError as e:      # IPC mapping refused on this node: fall back to fetching the entries with one all-gather
            peer_error = repr(e)[:300]
            x_mode = "halo"
    el = timed(sh.bind_cons_jac(x, c, vals, stitch=False, x_mode=x_mode), args.warmup, args.steps)
    region_ms = region["ms_per_launch"]
    per_step_value = (world if cfg["scaling"] == "weak" else 1)
    secondary = {}
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
        secondary["sharded_iterate_check"] = {"bit_identical_to_whole_iterate_on_every_rank": bool(same.item() == 1.0),
                                              "what": "each rank's c rows and CSC range from its NaN-padded shard of x vs from the whole x"}
        ks = max(1, min(args.steps, 500))
        if peer_error:
            secondary["peer_error"] = peer_error
        for key, kw, what in (
                ("halo_allgather", dict(stitch=False, x_mode="halo"), "the neighbours' entries fetched with one all-gather (pack / all-gather / unpack) before the evaluation instead of read in place"),
                ("stitched_c", dict(stitch=True, x_mode=x_mode), "the timed step + all-gather of the row blocks of c (every rank ends with the whole c)"),
                ("broadcast_x", dict(stitch=False, x_mode="broadcast"), "replicated iterate: rank 0 broadcasts all of x before the evaluation"),
                ("no_exchange", dict(stitch=False, x_mode=None), "evaluation only, the whole x in place on every rank: what reading the neighbours' entries in place costs is the difference to the timed step")):
            try:      # a secondary figure must never cost the line
                xin = x if kw["x_mode"] in ("peer", "halo") else x_full
                els = timed(sh.bind_cons_jac(xin, c, vals, **kw), min(args.warmup, 50), ks)
                secondary[key] = {"value": ks * per_step_value / els, "unit": "evals/s", "ms_per_step": els / ks * 1e3,
                                  "steps": ks, "what": what}
            except Exception as e:
                secondary[key] = {"error": repr(e)[:300]}

    # roofline of the dominant (only) kernel, per rank.  kernel_ms: N = 1: HIP events on the launch stream around the K
    # timed launches / K (back-to-back launches: the kernel's average duration including the dispatch gap); N > 1 (the
    # region also holds collectives): per-dispatch start / stop events (hipExtLaunchKernelGGL), median of five batches of 200
    per_dispatch = sorted(docp.time_cons_jac(x_full, c, vals, iters=200) for _ in range(5))[2]
    kernel_ms = region_ms if not dist_on else per_dispatch
    # algorithmic bytes of one launch: read the shard's x, write its c rows and Jacobian values:
    # B = 8 (nvar + ncon + nnzj) of the per-rank sub-problem
    b, e = sh.steps
    one = ct.DOCP(PROBLEM, e - b, SCHEME, device=-1)
    alg_bytes = 8 * (one.dim_NLP_variables + one.dim_NLP_constraints + one.nnzj)
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    per_rank = None
    if dist_on:
        t = torch.zeros(world, dtype=torch.float64, device=dev)
        t[rank] = per_dispatch
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per_rank = [{"rank": r, "kernel_ms": float(t[r]), "frac": alg_bytes / (float(t[r]) * 1e-3) / 1e9 / HBM_PEAK_GBS}
                    for r in range(world)]
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
    others, hessian, optimized = [], [], []
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
        # the Hessian-of-the-Lagrangian row (hess_coord!): kernel-only figures, not part of `value`.
        # Algorithmic bytes: read x and y, write the lower-triangular values: 8 (nvar + ncon + nnzh).
        for prob, sch, n in (("goddard", "gauss_legendre_2", 10000), ("goddard", "gauss_legendre_3", 80000),
                             ("quadrotor", "gauss_legendre_3", 20000), ("quadrotor12", "gauss_legendre_3", 20000)):
            d2 = ct.DOCP(prob, n, sch, device=local_rank, stream="torch")
            x2 = torch.from_numpy(bench_inputs(describe(d2, prob, sch), perturb=1e-3)).to(dev)
            y2 = torch.from_numpy(0.6 + 0.4 * np.sin(0.7 * np.arange(d2.dim_NLP_constraints) + 0.3)).to(dev)
            h2 = torch.zeros(d2.nnzh, dtype=torch.float64, device=dev)
            ms2 = d2.time_hess(x2, y2, h2, 1.0, iters=50)
            b2 = 8 * (d2.dim_NLP_variables + d2.dim_NLP_constraints + d2.nnzh)
            hessian.append({"workload": f"{prob}/{sch} N={n}", "nnzh": d2.nnzh, "kernel_ms": ms2, "algorithmic_bytes": b2,
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
            separately = {"same_workload_other_callbacks": {
                "objective_device": {"ms_per_call": t_obj * 1e3, "calls_per_s": 1.0 / t_obj},
                "gradient_device": {"ms_per_call": t_grad * 1e3, "calls_per_s": 1.0 / t_grad},
                "whole_iteration_device": {"ms_per_call": t_iter * 1e3, "calls_per_s": 1.0 / t_iter,
                                           "what": "ctd_eval_all_dev_async: objective + gradient + constraints + Jacobian values + Hessian values at one (x, y), two launches"},
                "fused_cons_jac_host_pointers_pinned": {"ms_per_call": t_host * 1e3, "calls_per_s": 1.0 / t_host,
                                                         "bytes_over_pcie": 8 * (docp.dim_NLP_variables + docp.dim_NLP_constraints + docp.nnzj),
                                                         "what": "ctd_cons_jac on page-locked host arrays: H2D x + kernel + D2H c, values (PCIe inclusive; never `value`)"}}}
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
        for r in json.load(open(os.path.join(ROOT, "profiles", "r03_workloads.json")))["workloads"]:
            wl_rows[(r["problem"], r["scheme"], r["N"], r["pattern"], r["kernel"])] = {
                k: r[k] for k in ("rocprof_avg_ns", "frac_of_8TBs", "hbm_bytes_per_launch", "traffic_over_algorithmic")}
    except Exception:
        pass
    for lst, pat, kern in ((others, "manual", "cons_jac"), (optimized, "optimized", "cons_jac"), (hessian, "manual", "hess")):
        for e in lst:
            prob_sch, n_ = e["workload"].split(",")[0].split(" N=")
            prob_, sch_ = prob_sch.split("/")
            row = wl_rows.get((prob_, sch_, int(n_), pat, kern))
            if row:
                e["rocprof_imported_from_profiles_r03_workloads"] = row

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
        if hessian:
            out["hessian_kernel_only"] = hessian
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(PROBLEM, SCHEME, N, x_host)
        out.update(secondary)
        out.update(separately)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
