#!/usr/bin/env python3
"""A/B timing of two builds of libctdirect_hip.so on the SAME box (boxes differ by ~10 %): the kernels of the named workloads,
each build in its own process, alternating, median of the per-dispatch event timings.

    python bench/ab.py scratch/base/libctdirect_hip.so ctdirect.jl_amd/libctdirect_hip.so cfg2 cfg5 cfg5:optimized ...
    (child mode: python bench/ab.py --child cfg ...  with CTD_LIB_PATH set)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "bench"))


def child(names):
    import numpy as np
    import torch
    import ctdirect_jl_amd as ct
    from helpers import bench_inputs, describe
    from stamps import CFGS
    out = {}
    for name in names:
        base, *tail = name.split(":")                     # "cfg5:optimized:hess", "cfg2:csr"
        kind = "hess" if "hess" in tail else "cj"
        pattern = "optimized" if "optimized" in tail else ("structural" if "structural" in tail else "manual")
        prob, sch, N = CFGS[base]
        d = ct.DOCP(prob, N, sch, device=0, pattern=pattern, value_order="csr" if "csr" in tail else "csc")
        x = torch.from_numpy(bench_inputs(describe(d, prob, sch), perturb=1e-3)).cuda()
        if kind == "cj":
            c = torch.zeros(d.dim_NLP_constraints, dtype=torch.float64, device="cuda")
            v = torch.zeros(d.nnzj, dtype=torch.float64, device="cuda")
            ms = sorted(d.time_cons_jac(x, c, v, iters=200) for _ in range(7))[3]
        else:
            y = torch.from_numpy(0.6 + 0.4 * np.sin(0.7 * np.arange(d.dim_NLP_constraints) + 0.3)).cuda()
            h = torch.zeros(d.nnzh, dtype=torch.float64, device="cuda")
            ms = sorted(d.time_hess(x, y, h, 1.0, iters=100) for _ in range(7))[3]
        out[name] = ms * 1e3
        d.close()
    print(json.dumps(out))


def main():
    if sys.argv[1] == "--child":
        return child(sys.argv[2:])
    libs, names = sys.argv[1:3], sys.argv[3:]
    res = {lib: {n: [] for n in names} for lib in libs}
    for rep in range(3):
        for lib in libs:
            path, *kv = lib.split(",")               # "path/to/lib.so,CTD_EARLY=0": environment knobs of that side
            env = dict(os.environ, CTD_LIB_PATH=os.path.abspath(path), **dict(x.split("=", 1) for x in kv))
            o = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + names, env=env, capture_output=True, text=True)
            line = [l for l in o.stdout.splitlines() if l.startswith("{")]
            if not line:
                print("FAILED", lib, o.stderr[-2000:])
                return 1
            for n, v in json.loads(line[-1]).items():
                res[lib][n].append(v)
    print(f"{'workload':28s} {'A us':>9s} {'B us':>9s}   B/A      (A = {libs[0]}, B = {libs[1]})")
    for n in names:
        a, b = sorted(res[libs[0]][n])[1], sorted(res[libs[1]][n])[1]
        print(f"{n:28s} {a:9.2f} {b:9.2f}   {b / a:5.3f}")


if __name__ == "__main__":
    sys.exit(main())
