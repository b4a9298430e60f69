#!/usr/bin/env python3
"""One device-resident solver iteration: objective, gradient, constraints + Jacobian values and Hessian values of the
Lagrangian at one (x, y).  Wall time per iteration (1) enqueued one after the other on one stream (`*_dev_async`), and
(2) through `ctd_eval_all_dev_async` (callbacks side by side).      python bench/iteration.py [cfg ...]"""
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
from stamps import CFGS  # noqa: E402


def main():
    for name in sys.argv[1:] or ["cfg2", "cfg3", "cfg4", "cfg5p", "cfg5"]:
        prob, sch, N = CFGS[name]
        d = ct.DOCP(prob, N, sch, device=0)
        x = torch.from_numpy(bench_inputs(describe(d, prob, sch), perturb=1e-3)).cuda()
        y = torch.from_numpy(0.6 + 0.4 * np.sin(0.7 * np.arange(d.dim_NLP_constraints) + 0.3)).cuda()
        f = torch.zeros(1, dtype=torch.float64, device="cuda")
        g = torch.zeros(d.dim_NLP_variables, dtype=torch.float64, device="cuda")
        c = torch.zeros(d.dim_NLP_constraints, dtype=torch.float64, device="cuda")
        v = torch.zeros(d.nnzj, dtype=torch.float64, device="cuda")
        h = torch.zeros(d.nnzh, dtype=torch.float64, device="cuda")

        def serial():
            d.obj_async(x, f)
            d.grad(x, g, sync=False)
            d.cons_jac(x, c, v, sync=False)
            d.hess_coord(x, y, 1.0, h, sync=False)

        def fused():
            d.eval_all(x, y, 1.0, f, g, c, v, h)

        res = {}
        for nm, fn in (("one stream", serial), ("eval_all", fused)):
            for _ in range(50):
                fn()
            d.sync()
            K = 500
            t0 = time.perf_counter()
            for _ in range(K):
                fn()
            d.sync()
            res[nm] = (time.perf_counter() - t0) / K * 1e6
        ref = [t.clone() for t in (f, g, c, v, h)]
        serial()
        d.sync()
        same = all(torch.equal(a, b) for a, b in zip(ref[:4], (f, g, c, v)))       # same bodies, same rounding mode
        hd = float(((ref[4] - h).abs() / h.abs().clamp(min=1.0)).max())                 # (the single-purpose Hessian build contracts multiply-adds)
        print(f"{name} {prob}/{sch} N={N}: one stream {res['one stream']:.1f} us   eval_all {res['eval_all']:.1f} us   "
              f"f, g, c, J bit-identical: {same}; Hessian max rel. difference {hd:.1e}", flush=True)
        d.close()


if __name__ == "__main__":
    main()
