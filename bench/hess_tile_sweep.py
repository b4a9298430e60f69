#!/usr/bin/env python3
"""Kernel time of the Hessian kernel over the steps per tile (CTD_HESS_TILE): python bench/hess_tile_sweep.py cfg4 cfg5:optimized [...]"""
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

TILES = {"cfg2": (6, 8, 10, 12, 14, 17, 20, 24, 32, 40), "cfg3": (32, 64, 128, 196, 256), "cfg4": (8, 10, 12, 16, 20, 24, 26, 28, 32, 40, 48),
         "cfg5p": (3, 4, 5, 6, 8, 10, 12, 14, 16, 20), "cfg5": (4, 5, 6, 7, 8, 9, 10, 12, 14, 16), "q12_mid": (8, 12, 16, 20, 24, 32, 40, 48),
         "quad_mid": (8, 12, 16, 24, 32, 40, 48), "cfg3_8M": (32, 64, 96, 128, 196, 256), "g_mid_4M": (16, 32, 48, 64, 96, 128),
         "gall_mid_4M": (16, 32, 48, 64, 96)}


def main():
    for spec in sys.argv[1:] or ["cfg4"]:
        name, _, pattern = spec.partition(":")
        prob, sch, N = CFGS[name]
        row = []
        for T in (0,) + TILES[name]:                       # 0: the engine's own choice (default_hess_tile)
            os.environ["CTD_HESS_TILE"] = str(T)
            d = ct.DOCP(prob, N, sch, device=0, pattern=pattern or "manual")
            x = torch.from_numpy(bench_inputs(describe(d, prob, sch), perturb=1e-3)).cuda()
            y = torch.from_numpy(0.6 + 0.4 * np.sin(0.7 * np.arange(d.dim_NLP_constraints) + 0.3)).cuda()
            v = torch.zeros(d.nnzh, dtype=torch.float64, device="cuda")
            ms = sorted(d.time_hess(x, y, v, 1.0, iters=(50 if N < 1000000 else 8)) for _ in range(3))[1]
            li = d.hess_launch_info()
            gb = 8e-9 * (d.dim_NLP_variables + d.dim_NLP_constraints + d.nnzh)
            row.append(f"{'default ' if T == 0 else ''}T={li['steps_per_tile']}(lds {li['lds_bytes'] // 1024}K, {d.hess_kernel_info()['kernel']}):{ms * 1e3:.1f} ({gb / ms / 8:.2f})")
            d.close()
        print(f"{spec}  " + "  ".join(row), flush=True)
    os.environ.pop("CTD_HESS_TILE", None)


if __name__ == "__main__":
    main()
