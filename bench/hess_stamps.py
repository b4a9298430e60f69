#!/usr/bin/env python3
"""In-kernel phase stamps of the Hessian-of-the-Lagrangian kernel (ctd_hess_debug_stamps): python bench/hess_stamps.py [cfg ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctdirect_jl_amd as ct  # noqa: E402
from helpers import bench_inputs, describe  # noqa: E402
from stamps import CFGS  # noqa: E402


def main():
    for spec in sys.argv[1:] or ["cfg2", "cfg4", "cfg5p", "cfg5"]:
        name, _, pattern = spec.partition(":")            # "cfg5:optimized"
        prob, sch, N = CFGS[name]
        d = ct.DOCP(prob, N, sch, device=0, pattern=pattern or "manual")
        x = torch.from_numpy(bench_inputs(describe(d, prob, sch), perturb=1e-3)).cuda()
        y = torch.from_numpy(0.6 + 0.4 * np.sin(0.7 * np.arange(d.dim_NLP_constraints) + 0.3)).cuda()
        v = torch.zeros(d.nnzh, dtype=torch.float64, device="cuda")
        ms = sorted(d.time_hess(x, y, v, 1.0, iters=100) for _ in range(3))[1]
        st = d.hess_debug_stamps(x, y, v, 1.0).astype(np.int64)
        rt = st[:, :, 0] * 10.0 / 1000.0
        cy = st[:, :, 1]
        t0 = rt[:, 0].min()
        b = 8 * (d.dim_NLP_variables + d.dim_NLP_constraints + d.nnzh)
        print(f"== {spec} {prob} {sch} N={N} nnzh {d.nnzh} {d.hess_launch_info()} kernel {ms * 1e3:.2f} us = {b / ms / 1e6 / 8000:.3f} of 8 TB/s; "
              f"last block start {rt[:, 0].max() - t0:.2f}, end {rt[:, 4].max() - t0:.2f}")
        for i, nm in enumerate(["load", "eval", "emit-issue", "drain"]):
            dt = rt[1:, i + 1] - rt[1:, i]
            dc = cy[1:, i + 1] - cy[1:, i]
            print(f"   {nm:10s} mean {dt.mean():6.2f} us  p50 {np.median(dt):6.2f}  p95 {np.percentile(dt, 95):6.2f}   cycles p50 {int(np.median(dc))}")
        print("   tile total mean %.2f; starts pctl 10/50/90: %s; edge block %s" % (
            (rt[1:, 4] - rt[1:, 0]).mean(), [round(float(np.percentile(rt[:, 0] - t0, p)), 1) for p in (10, 50, 90)],
            [round(float(rt[0, i + 1] - rt[0, i]), 2) for i in range(4)]))
        d.close()


if __name__ == "__main__":
    main()
