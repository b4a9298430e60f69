#!/usr/bin/env python3
"""Per-step time of the SINGLE-PROCESS multi-device entry point of the C ABI (`ctd_cons_jac_sharded_dev_async`, the path a
Julia host owning a whole node would use) for every iterate mode.  On a one-GPU box the device is listed G times (the
shards' kernels then run one after the other on that device, so a step costs G shard kernels): what the figures show is what
each mode ADDS to the plain shard kernels -- peer copies + events (CTD_X_SHARDED_COPY, CTD_X_FROM_DEVICE0, stitch) or
nothing but two event operations per neighbour (CTD_X_SHARDED_IN_PLACE: neighbours' entries read in place by the kernels).

    python bench/sharded_cabi.py [G] [cfg ...]        -> one JSON line per (cfg, stream mode)"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctdirect_jl_amd as ct  # noqa: E402
from helpers import bench_inputs, describe  # noqa: E402

CFGS = {"cfg2x": ("goddard", "gauss_legendre_2", 10000), "cfg3": ("double_integrator_path", "midpoint", 100000),
        "cfg4": ("goddard", "gauss_legendre_3", 80000)}


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    ndev = torch.cuda.device_count()
    devices = [k % ndev for k in range(G)]
    for name in sys.argv[2:] or ["cfg2x", "cfg3", "cfg4"]:
        prob, sch, n = CFGS[name]
        N = n * G if name == "cfg2x" else n          # cfg2x: 10 000 steps per shard (the weak-scaling bench workload)
        for stream in ("own", "torch"):
            md = ct.MultiDeviceDOCP(prob, N, sch, devices, stream=stream)
            full = ct.DOCP(prob, N, sch, device=-1)
            x = bench_inputs(describe(full, prob, sch), perturb=1e-3)
            blk, nv = full.discretization._step_variables_block, full.dims.NLP_v
            xs_whole = [torch.from_numpy(x).to(f"cuda:{d}") for d in devices]
            xs_shard = []
            for k, s in enumerate(md.shards):
                t = np.full_like(x, np.nan)
                end = s.step_end * blk if k < G - 1 else x.size - nv
                t[s.step_begin * blk:end] = x[s.step_begin * blk:end]
                if nv:
                    t[-nv:] = x[-nv:]
                xs_shard.append(torch.from_numpy(t).to(f"cuda:{devices[k]}"))
            cs = [torch.zeros(md.dim_NLP_constraints, dtype=torch.float64, device=f"cuda:{d}") for d in devices]
            vs = [torch.zeros(md.nnzj, dtype=torch.float64, device=f"cuda:{d}") for d in devices]

            def rate(xs, mode, stitch, iters=500):
                for _ in range(50):
                    md.cons_jac(xs, cs, vs, x_mode=mode, stitch=stitch, sync=False)
                md.sync(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(iters):
                    md.cons_jac(xs, cs, vs, x_mode=mode, stitch=stitch, sync=False)
                md.sync(); torch.cuda.synchronize()
                return (time.perf_counter() - t0) / iters * 1e6

            out = {"workload": f"{prob}/{sch} N={N}", "shards": G, "devices": devices, "stream": stream,
                   "us_per_step": {
                       "in_place": rate(xs_whole, md.X_IN_PLACE, False),
                       "sharded_read_in_place": rate([t.clone() for t in xs_shard], md.X_SHARDED_IN_PLACE, False),
                       "sharded_peer_copies": rate([t.clone() for t in xs_shard], md.X_SHARDED_COPY, False),
                       "from_device0": rate(xs_whole, md.X_FROM_DEVICE0, False),
                       "sharded_read_in_place+stitch": rate([t.clone() for t in xs_shard], md.X_SHARDED_IN_PLACE, True)}}
            print(json.dumps(out), flush=True)
            md.close()
            full.close()


if __name__ == "__main__":
    main()
