#!/usr/bin/env python3
"""Direct-shooting layout (`control_steps` controls per step, midpoint scheme): durations of the constraint / Jacobian kernel and of the
Hessian kernel per launch (per-dispatch HIP events on the handle's stream, median of 7 runs), with the algorithmic bytes of
SURVEY.md section 8d (8 (nvar + ncon + nnz)) against 8 TB/s.  `python3 bench/control_steps.py [N]` -> a markdown table."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctdirect_jl_amd as ct  # noqa: E402
from helpers import bench_inputs, describe  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    print(f"| problem | controls per step | pattern | nnzj | cons/Jac µs | frac of 8 TB/s | nnzh | Hessian µs | frac of 8 TB/s |")
    print("|---|---|---|---|---|---|---|---|---|")
    for prob in ("goddard", "double_integrator_path", "quadrotor", "quadrotor12"):
        for cs in (1, 2, 3):
            for pattern in ("manual", "optimized"):
                d = ct.DOCP(prob, N, "midpoint", device=0, pattern=pattern, control_steps=cs)
                x = torch.from_numpy(bench_inputs(describe(d, prob, "midpoint"), perturb=1e-3)).cuda()
                c = torch.zeros(d.dim_NLP_constraints, dtype=torch.float64, device="cuda")
                v = torch.zeros(d.nnzj, dtype=torch.float64, device="cuda")
                y = torch.from_numpy(0.6 + 0.4 * np.sin(0.7 * np.arange(d.dim_NLP_constraints) + 0.3)).cuda()
                h = torch.zeros(d.nnzh, dtype=torch.float64, device="cuda")
                tcj = sorted(d.time_cons_jac(x, c, v, iters=200) for _ in range(7))[3] * 1e3
                th = sorted(d.time_hess(x, y, h, 1.0, iters=100) for _ in range(7))[3] * 1e3
                bcj = 8 * (d.dim_NLP_variables + d.dim_NLP_constraints + d.nnzj)
                bh = 8 * (d.dim_NLP_variables + d.dim_NLP_constraints + d.nnzh)
                print(f"| {prob} | {cs} | {pattern} | {d.nnzj} | {tcj:.2f} | {bcj / tcj / 8e6:.3f} | {d.nnzh} | {th:.2f} | {bh / th / 8e6:.3f} |", flush=True)
                d.close()


if __name__ == "__main__":
    main()
