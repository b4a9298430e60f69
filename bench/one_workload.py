#!/usr/bin/env python3
"""ONE workload, ONE kernel, nothing else on the GPU: the process rocprofv3 wraps to get per-workload evidence
(profiles/collect_workloads.sh): `python3 bench/one_workload.py cfg5:optimized 300` launches the constraint / Jacobian kernel of
that workload 300 times back to back; `cfg4:hess` the Hessian kernel.  Prints one JSON line with the sizes and the algorithmic
bytes per launch (8 (nvar + ncon + nnzj); Hessian: 8 (nvar + ncon + nnzh))."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "bench"))
import ctdirect_jl_amd as ct  # noqa: E402
from helpers import bench_inputs, describe  # noqa: E402
from stamps import CFGS  # noqa: E402


def main():
    spec, iters = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 300
    parts = spec.split(":")
    base, tail = parts[0], parts[1:]
    kind = "hess" if "hess" in tail else "cons_jac"
    pattern = "optimized" if "optimized" in tail else "manual"
    order = "csr" if "csr" in tail else "csc"              # (ctd_desc.value_order: the constraint / Jacobian kernel's value order)
    prob, sch, N = CFGS[base]
    d = ct.DOCP(prob, N, sch, device=0, pattern=pattern, value_order=order)
    x = torch.from_numpy(bench_inputs(describe(d, prob, sch), perturb=1e-3)).cuda()
    if kind == "cons_jac":
        c = torch.zeros(d.dim_NLP_constraints, dtype=torch.float64, device="cuda")
        v = torch.zeros(d.nnzj, dtype=torch.float64, device="cuda")
        launch = d.bind_cons_jac(x, c, v, sync=False)       # pointers pre-bound: back-to-back launches, as in bench.py's timed loop
        for _ in range(iters):
            launch()
        alg = 8 * (d.dim_NLP_variables + d.dim_NLP_constraints + d.nnzj)
    else:
        y = torch.from_numpy(0.6 + 0.4 * np.sin(0.7 * np.arange(d.dim_NLP_constraints) + 0.3)).cuda()
        h = torch.zeros(d.nnzh, dtype=torch.float64, device="cuda")
        for _ in range(iters):
            d.hess_coord(x, y, 1.0, h, sync=False)
        alg = 8 * (d.dim_NLP_variables + d.dim_NLP_constraints + d.nnzh)
    d.sync()
    torch.cuda.synchronize()
    print(json.dumps({"workload": spec, "problem": prob, "scheme": sch, "N": N, "pattern": pattern, "value_order": order, "kernel": kind, "launches": iters,
                      "nvar": d.dim_NLP_variables, "ncon": d.dim_NLP_constraints, "nnzj": d.nnzj, "nnzh": d.nnzh,
                      "algorithmic_bytes_per_launch": alg}))
    d.close()


if __name__ == "__main__":
    main()
