#!/usr/bin/env python3
"""bench.py -- NLP-callback throughput of the MI355X collocation engine (BASELINE.json metric).

One "step" = one fused evaluation (constraints c(x) + sparse Jacobian values, `ctd_cons_jac_dev_async`) of the workload
on inputs already resident in HBM.  Workload at N GPUs: Goddard rocket, Gauss-Legendre 2 (stagewise), 10 000 time steps
PER GPU -- BASELINE.json configs[1] at N = 1; at N > 1 the global grid has 10 000 x N steps, sharded by time step (weak
scaling); each rank's rows of c and its Jacobian values stay on the rank (row-sharded outputs, no collective on the path).
`value` = (shard evaluations all ranks completed) / (max-over-ranks wall time of the K timed steps).  At N > 1 the line
also carries `stitched_c`: the same step followed by the RCCL all-gather that hands every rank the whole constraint vector.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: launched by torch.distributed.run)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PROBLEM, SCHEME, STEPS_PER_GPU = "goddard", "gauss_legendre_2", 10000
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy)


def cpu_baseline(x, budget_s=12.0):
    """Reference-faithful CPU path on the host cores of this box: the oracle (a C++ restatement of the Julia reference,
    kind = "port") evaluates cons! once and the Jacobian the way ADNLPModels does -- one pass of the constraints on
    one-partial duals per colour of the pattern -- single-threaded, as the reference is."""
    from oracle.oracle import OracleDOCP
    o = OracleDOCP(PROBLEM, SCHEME, STEPS_PER_GPU)
    o.jac_pattern()                      # build pattern + colouring outside the timed region (build-time in the reference)
    o.constraints(x); o.jac_coord(x)     # warm
    n, t0 = 0, time.perf_counter()
    while True:
        o.constraints(x)
        o.jac_coord(x)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    out = {"value": n / el, "unit": "evals/s", "cores": 1, "kind": "port",
           "sample": f"{n} fused evaluations (cons! + {o.jac_ncolors()}-colour forward-dual jac_coord!) of the same "
                     f"{PROBLEM}/{SCHEME} N={STEPS_PER_GPU} workload in {el:.1f} s, oracle/ctd_oracle.cpp, 1 thread"}
    # the same passes with the colours spread over the host cores this process may use (the reference itself is
    # single-threaded; this is the multi-core figure a CPU implementation could reach with the same algorithm)
    cores = max(1, min(o.jac_ncolors(), len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    o.jac_coord_mt(x, cores)
    n, t0 = 0, time.perf_counter()
    while True:
        o.constraints(x)
        o.jac_coord_mt(x, cores)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s / 2:
            break
    out["multithreaded"] = {"value": n / el, "unit": "evals/s", "cores": cores,
                            "sample": f"{n} evaluations in {el:.1f} s, one colour pass per thread"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

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
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
    assert torch.cuda.is_available(), "bench.py needs a GPU: the engine has no CPU path"
    # rehearsal knobs (one-GPU box): CTD_BENCH_DEVICE pins every rank to one device, CTD_BENCH_BACKEND=gloo replaces RCCL
    # (which refuses two ranks on one device).  The driver's multi-GPU run uses neither: one rank per GPU over RCCL/xGMI.
    if os.environ.get("CTD_BENCH_DEVICE"):
        local_rank = int(os.environ["CTD_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("CTD_BENCH_BACKEND", "nccl")
        if backend == "nccl":     # bind the communicator to this rank's GPU up front (barriers need no device guess)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    N = STEPS_PER_GPU * world
    def make(steps=None):      # the handle launches on torch's current stream: ordered with the RCCL collectives
        return ct.DOCP(PROBLEM, N, SCHEME, device=local_rank, steps=steps, stream="torch")

    sh = ctdist.ShardedDOCP(make, N, world=world, rank=rank)
    docp = sh.docp
    x_host = bench_inputs(describe(docp, PROBLEM, SCHEME), perturb=1e-3)
    x = torch.from_numpy(x_host).to(dev)
    c = torch.zeros(docp.dim_NLP_constraints, dtype=torch.float64, device=dev)
    vals = torch.zeros(docp.nnzj, dtype=torch.float64, device=dev)

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    region = {}

    def timed(step, warmup, steps):
        for _ in range(warmup):
            step()
        sync_all()
        # HIP events on the launch stream (the handle launches on torch's current stream) bracket the timed region: GPU-side
        # duration of the K launches, the figure the roofline uses
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            step()
        e1.record()
        sync_all()
        el = time.perf_counter() - t0
        region["ms_per_launch"] = e0.elapsed_time(e1) / steps
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t[0])
        return el

    # One fused evaluation of the rank's shard, pointers pre-bound.  The path has no exchange step: c rows and Jacobian
    # values are left row-sharded for a distributed consumer, so the timed step holds no collective.
    el = timed(sh.bind_cons_jac(x, c, vals, stitch=False), args.warmup, args.steps)
    region_ms = region["ms_per_launch"]
    stitched = None
    if world > 1:
        # the optional service for a consumer that wants the whole residual on every rank: + one in-place RCCL all-gather
        # of the row blocks of c per evaluation (reported beside `value`, never instead of it)
        ks = max(1, min(args.steps, 500))
        try:
            els = timed(sh.bind_cons_jac(x, c, vals, stitch=True), min(args.warmup, 50), ks)
            stitched = {"value": ks * world / els, "unit": "evals/s", "ms_per_step": els / ks * 1e3, "steps": ks,
                        "what": "same step + in-place all-gather of the row blocks of c (every rank ends with the whole c)"}
        except Exception as e:            # the secondary figure must never cost the line
            stitched = {"error": repr(e)[:300]}

    # roofline of the dominant (only) kernel: per-dispatch HIP events on the stream it is launched on
    # average launch duration of the kernel: (a) HIP events around the K timed launches on the launch stream (above) -- the
    # launches run back to back, so elapsed / K is the kernel's average duration including the dispatch gap; (b) per-dispatch
    # start / stop events (hipExtLaunchKernelGGL), median of five batches of 200 -- noisier (clock state), reported beside it
    per_dispatch = sorted(docp.time_cons_jac(x, c, vals, iters=200) for _ in range(5))[2]
    kernel_ms = region_ms
    sh_nnz = docp.nnzj if world == 1 else None
    # algorithmic bytes of one launch: read x, write c rows and Jacobian values of the shard
    # B = 8 (nvar + ncon + nnzj) for the per-GPU 10 000-step problem
    one = ct.DOCP(PROBLEM, STEPS_PER_GPU, SCHEME, device=-1)
    alg_bytes = 8 * (one.dim_NLP_variables + one.dim_NLP_constraints + one.nnzj)
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("bench_kernel", {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    # the other single-GPU BASELINE configs (parity-test cases, not the bench line): kernel time and roofline fraction
    others = []
    if world == 1:
        for prob, sch, n in (("double_integrator_path", "midpoint", 100000), ("goddard", "gauss_legendre_3", 80000),
                             ("quadrotor", "gauss_legendre_3", 20000), ("quadrotor12", "gauss_legendre_3", 20000)):
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

    # the Hessian-of-the-Lagrangian row (hess_coord!): kernel-only figures, not part of `value`.
    # Algorithmic bytes: read x and y, write the lower-triangular values: 8 (nvar + ncon + nnzh).
    hessian = []
    if world == 1:
        import numpy as np
        for prob, sch, n in ((PROBLEM, SCHEME, STEPS_PER_GPU), ("goddard", "gauss_legendre_3", 80000),
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

    if rank == 0:
        out = {
            "metric": "NLP callback evals/s (constraints+sparse Jac), N-step Goddard, 1/2/4/8 GPU",
            "value": args.steps * world / el,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{PROBLEM} / {SCHEME} (stagewise), {STEPS_PER_GPU} time steps per GPU "
                            f"(global grid {N} steps, time-step sharded); fused cons!+jac_coord! on HBM-resident x; "
                            + ("single GPU = BASELINE.json configs[1]" if world == 1 else
                               "outputs row-sharded (no collective on the path); value counts one 10000-step shard "
                               "evaluation per GPU per step; `stitched_c` = the same with c all-gathered"),
                "nvar_per_gpu": one.dim_NLP_variables, "ncon_per_gpu": one.dim_NLP_constraints, "nnzj_per_gpu": one.nnzj,
                "launch": docp.launch_info(),
            },
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "ctd::cons_jac_kernel<GoddardOCP, SC_IRK>", "kernel_ms": kernel_ms,
                         "kernel_ms_per_dispatch_events": per_dispatch,
                         "timing": "HIP events on the launch stream around the K timed launches / K",
                         "algorithmic_bytes_per_launch": alg_bytes},
            "other_configs_kernel_only": others,
            "hessian_kernel_only": hessian,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(x_host)
        if stitched is not None:
            out["stitched_c"] = stitched
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
