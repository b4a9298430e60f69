#!/usr/bin/env python3
"""Same-box A/B of the long-grid launch geometry (CTD_LONG_GRID=0 against the default): python bench/long_grid_ab.py cfg3_8M g_mid_4M ..."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "bench"))
import ctdirect_jl_amd as ct  # noqa: E402
from helpers import bench_inputs, describe  # noqa: E402
from stamps import CFGS  # noqa: E402

for name in sys.argv[1:]:
    prob, sch, N = CFGS[name]
    if os.environ.get("AB_LOG2N"):                       # crossover search: AB_LOG2N=18 CTD_LONG_GRID_ROUNDS=1 python bench/long_grid_ab.py ...
        N = 1 << int(os.environ["AB_LOG2N"])
    row = []
    for flag in ("0", "1", "0", "1"):
        os.environ["CTD_LONG_GRID"] = flag
        d = ct.DOCP(prob, N, sch, device=0)
        x = torch.from_numpy(bench_inputs(describe(d, prob, sch), perturb=1e-3)).cuda()
        c = torch.zeros(d.dim_NLP_constraints, dtype=torch.float64, device="cuda")
        v = torch.zeros(d.nnzj, dtype=torch.float64, device="cuda")
        ms = sorted(d.time_cons_jac(x, c, v, iters=(12 if N > (1 << 21) else 60)) for _ in range(3))[1]
        li = d.launch_info()
        b = 8 * (d.dim_NLP_variables + d.dim_NLP_constraints + d.nnzj)
        row.append(f"long_grid={flag} T={li['steps_per_tile']} block={li['block']} lds={li['lds_bytes'] // 1024}K: {ms * 1e3:.1f} us ({b / ms / 1e9 / 8:.3f} of 8 TB/s)")
        d.close()
        del x, c, v
        torch.cuda.empty_cache()
    print(name, prob, sch, N, " | ".join(row), flush=True)
