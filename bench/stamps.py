#!/usr/bin/env python3
"""In-kernel phase stamps of the constraint / Jacobian kernel (diagnostics instantiation, ctd_debug_stamps).

    python bench/stamps.py [cfg ...]      cfg in {cfg2, cfg3, cfg4, cfg5p, cfg5}; default: cfg2 cfg3 cfg4

Per workgroup lane 0 stamps: start | after prologue (direct tiles: code prefetch issued; staged: load + barrier) | after
eval + barrier | after fin | after the emit loops were issued | after the workgroup's stores drained.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctdirect_jl_amd as ct  # noqa: E402
from helpers import bench_inputs, describe  # noqa: E402

CFGS = {"cfg2": ("goddard", "gauss_legendre_2", 10000), "cfg3": ("double_integrator_path", "midpoint", 100000),
        "cfg4": ("goddard", "gauss_legendre_3", 80000), "cfg5p": ("quadrotor", "gauss_legendre_3", 20000),
        "cfg5": ("quadrotor12", "gauss_legendre_3", 20000),
        # extra: OCPs whose kernels run the staged driver (path constraints in every scheme)
        "gall_trap": ("goddard_all", "trapeze", 20000), "gall_gl2": ("goddard_all", "gauss_legendre_2", 20000),
        "quad_trap": ("quadrotor", "trapeze", 20000), "q12_mid": ("quadrotor12", "midpoint", 20000), "quad_mid": ("quadrotor", "midpoint", 20000),
        "q12_trap": ("quadrotor12", "trapeze", 20000),
        # one tile + the edge block: the kernel's serial latency chain without any contention
        "cfg2_tiny": ("goddard", "gauss_legendre_2", 21), "cfg2_small": ("goddard", "gauss_legendre_2", 2100),
        # grids large enough to be bandwidth-bound (round 4): 4 194 304 steps
        "cfg2_4M": ("goddard", "gauss_legendre_2", 1 << 22), "cfg4_4M": ("goddard", "gauss_legendre_3", 1 << 22),
        "di_gl2_4M": ("double_integrator_path", "gauss_legendre_2", 1 << 22), "gall_gl2_2M": ("goddard_all", "gauss_legendre_2", 1 << 21),
        "cfg3_8M": ("double_integrator_path", "midpoint", 1 << 23), "g_trap_4M": ("goddard", "trapeze", 1 << 22), "g_mid_4M": ("goddard", "midpoint", 1 << 22),
        "g_eul_4M": ("goddard", "euler", 1 << 22), "g_euli_4M": ("goddard", "euler_implicit", 1 << 22), "di_eul_8M": ("double_integrator_path", "euler", 1 << 23),
        "di_euli_8M": ("double_integrator_path", "euler_implicit", 1 << 23),
        "gall_mid_4M": ("goddard_all", "midpoint", 1 << 22), "di_mid_8M": ("double_integrator_freet0tf", "midpoint", 1 << 23), "ls_mid_8M": ("least_squares_with_constraint", "midpoint", 1 << 23),
        "di_gl3_4M": ("double_integrator_freet0tf", "gauss_legendre_3", 1 << 22)}


def main():
    names = sys.argv[1:] or ["cfg2", "cfg3", "cfg4"]
    for name in names:
        name, _, pattern = name.partition(":")          # "cfg5:optimized" -> CTD_PATTERN_OPTIMIZED
        prob, sch, N = CFGS[name]
        cs = int(os.environ.get("CTD_STAMPS_CS", "1"))        # controls per step (midpoint workloads: the direct-shooting layout)
        d = ct.DOCP(prob, N, sch, device=0, pattern=pattern or "manual", control_steps=cs)
        name = name + (":" + pattern if pattern else "") + (f" cs={cs}" if cs > 1 else "")
        x = torch.from_numpy(bench_inputs(describe(d, prob, sch), perturb=1e-3)).cuda()
        c = torch.zeros(d.dim_NLP_constraints, dtype=torch.float64, device="cuda")
        v = torch.zeros(d.nnzj, dtype=torch.float64, device="cuda")
        ms = sorted(d.time_cons_jac(x, c, v, iters=200) for _ in range(5))[2]
        sub = None
        if os.environ.get("CTD_SUBSTAMPS"):
            st, sub = d.debug_stamps(x, c, v, sub=True)
            st, sub = st.astype(np.int64), sub.astype(np.int64)
        else:
            st = d.debug_stamps(x, c, v).astype(np.int64)
        rt = st[:, :, 0] * 10.0 / 1000.0
        cy = st[:, :, 1]
        t0 = rt[:, 0].min()
        li = d.launch_info()
        print(f"== {name} {prob} {sch} N={N} {li} kernel {ms * 1e3:.2f} us; last block start {rt[:, 0].max() - t0:.2f}, "
              f"end {rt[:, 5].max() - t0:.2f}")
        for i, nm in enumerate(["prologue", "eval", "fin", "emit-issue", "drain"]):
            dt = rt[1:, i + 1] - rt[1:, i]
            dc = cy[1:, i + 1] - cy[1:, i]
            print(f"   {nm:10s} mean {dt.mean():6.2f} us  p50 {np.median(dt):6.2f}  p95 {np.percentile(dt, 95):6.2f}   cycles p50 {int(np.median(dc))}")
        print("   tile total mean %.2f; starts pctl 10/50/90: %s; edge block phases %s total %.2f" % (
            (rt[1:, 5] - rt[1:, 0]).mean(), [round(float(np.percentile(rt[:, 0] - t0, p)), 1) for p in (10, 50, 90)],
            [round(float(rt[0, i + 1] - rt[0, i]), 2) for i in range(5)], float(rt[0, 5] - rt[0, 0])))
        if sub is not None:
            # wave 0 = dynamics lanes, wave 1 = lead lanes; ids: 0 start of phase_eval, 1 task decoded, 6 inputs + times arrived,
            # 2 dynamics + partials done, 3 chain rule done, 4 record stored, 5 end of phase_eval; cycles relative to the kernel-start stamp
            base = st[1:, 0, 1][:, None]
            for w, nm in ((0, "dyn wave"), (1, "lead wave")):
                rel = sub[1:, w, :] - base
                med = [int(np.median(rel[:, i])) if (sub[1:, w, i] > 0).any() else -1 for i in range(8)]
                print(f"   sub-stamps {nm}: cycles since kernel-start stamp, median: " + ", ".join(f"{i}:{m}" for i, m in enumerate(med)))
        d.close()


if __name__ == "__main__":
    main()
