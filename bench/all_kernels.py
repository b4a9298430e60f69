#!/usr/bin/env python3
"""Kernel times of the fused constraint / Jacobian kernel and of the Hessian kernel for every registry problem x scheme at one grid
size (per-dispatch events, median of three batches), with the algorithmic bytes and the fraction of 8 TB/s:
    python bench/all_kernels.py [N]  > profiles/rNN_all_kernels.md"""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctdirect_jl_amd as ct  # noqa: E402
from helpers import bench_inputs, describe  # noqa: E402


def main():
    warnings.simplefilter("ignore")
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    print(f"# Every registry problem x scheme at N = {N} (MI355X, `bench/all_kernels.py {N}`)\n")
    print("kernel time in us (per-dispatch events, median of 3 x 50 launches); bytes = 8 (nvar + ncon + nnz); frac = bytes / time / 8 TB/s\n")
    print("| problem | scheme | nnzj | cons+Jac us | frac | nnzh | Hessian us | frac | Hessian kernel |")
    print("|---|---|---|---|---|---|---|---|---|")
    for prob in ct.PROBLEMS:
        for sch in ct.SCHEMES:
            d = ct.DOCP(prob, N, sch, device=0)
            x = torch.from_numpy(bench_inputs(describe(d, prob, sch), perturb=1e-3)).cuda()
            y = torch.from_numpy(0.6 + 0.4 * np.sin(0.7 * np.arange(d.dim_NLP_constraints) + 0.3)).cuda()
            c = torch.zeros(d.dim_NLP_constraints, dtype=torch.float64, device="cuda")
            v = torch.zeros(d.nnzj, dtype=torch.float64, device="cuda")
            h = torch.zeros(d.nnzh, dtype=torch.float64, device="cuda")
            tj = sorted(d.time_cons_jac(x, c, v, iters=50) for _ in range(3))[1] * 1e3
            th = sorted(d.time_hess(x, y, h, 1.0, iters=50) for _ in range(3))[1] * 1e3
            bj = 8.0 * (d.dim_NLP_variables + d.dim_NLP_constraints + d.nnzj)
            bh = 8.0 * (d.dim_NLP_variables + d.dim_NLP_constraints + d.nnzh)
            print(f"| {prob} | {sch} | {d.nnzj} | {tj:.1f} | {bj / tj / 8e6:.2f} | {d.nnzh} | {th:.1f} | {bh / th / 8e6:.2f} | "
                  f"{d.hess_kernel_info()['kernel']} |", flush=True)
            d.close()


if __name__ == "__main__":
    main()
