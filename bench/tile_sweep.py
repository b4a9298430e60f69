#!/usr/bin/env python3
"""Kernel time of the constraint / Jacobian kernel over (steps per tile, workgroup size): python bench/tile_sweep.py cfg2 [...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctdirect_jl_amd as ct  # noqa: E402
from helpers import bench_inputs, describe  # noqa: E402
from stamps import CFGS  # noqa: E402

TILES = {"cfg2": (8, 10, 14, 16, 20, 21, 24, 32, 40), "cfg3": (32, 64, 96, 128, 196, 256, 391), "cfg4": (16, 24, 32, 48, 64),
         "cfg5p": (6, 8, 10, 12, 14, 16, 20), "cfg5": (4, 5, 6, 7, 8, 9, 10),
         "cfg2_4M": (0, 48, 64, 72, 80, 96, 112, 128), "cfg4_4M": (0, 40, 48, 56, 64, 80),
         "gall_trap": (8, 10, 12, 16, 20, 24, 32, 40, 64), "gall_gl2": (8, 10, 12, 16, 20, 24, 32, 40), "quad_trap": (8, 10, 12, 16, 20, 24, 32)}
BLOCKS = tuple(int(b) for b in os.environ.get("SWEEP_BLOCKS", "256,320,384,512").split(","))


def main():
    for name in sys.argv[1:] or ["cfg2"]:
        prob, sch, N = CFGS[name]
        if os.environ.get("SWEEP_TILES"):
            TILES[name] = tuple(int(t) for t in os.environ["SWEEP_TILES"].split(","))
        for blk in BLOCKS:
            row = []
            for T in TILES.get(name, (16, 32)):
                os.environ["CTD_TILE"], os.environ["CTD_BLOCK"] = (str(T) if T else ""), (str(blk) if T else "")       # T = 0: the defaults
                d = ct.DOCP(prob, N, sch, device=0, value_order=os.environ.get("SWEEP_ORDER", "csc"))
                x = torch.from_numpy(bench_inputs(describe(d, prob, sch), perturb=1e-3)).cuda()
                c = torch.zeros(d.dim_NLP_constraints, dtype=torch.float64, device="cuda")
                v = torch.zeros(d.nnzj, dtype=torch.float64, device="cuda")
                ms = sorted(d.time_cons_jac(x, c, v, iters=(200 if N < 1000000 else 12)) for _ in range(3))[1]
                row.append(f"T={d.launch_info()['steps_per_tile']}(lds {d.launch_info()['lds_bytes'] // 1024}K):{ms * 1e3:.2f}")
                d.close()
            print(f"{name} block={blk}  " + "  ".join(row), flush=True)
    os.environ.pop("CTD_TILE", None)
    os.environ.pop("CTD_BLOCK", None)


if __name__ == "__main__":
    main()
