#!/usr/bin/env python3
"""verbose trace of the interior-point loop on one catalogue problem:  python -u bench/explore_ipm_one.py moonlander midpoint 250 [filter|merit] [every]"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import ctdirect_jl_amd as ct
import ipm, jit_defs, problem_folder_defs as pf
np.seterr(all='ignore')
name, sch, N = sys.argv[1], sys.argv[2], int(sys.argv[3])
ls = sys.argv[4] if len(sys.argv) > 4 else "filter"
every = int(sys.argv[5]) if len(sys.argv) > 5 else 20
prob, want, init = jit_defs.catalogue(name) if name in jit_defs.CATALOGUE else pf.folder(name)
d = ct.DOCP(prob, N, sch, pattern="structural", device=0)
lv, uv = ct.variables_bounds(d)
x0 = np.clip(ct.initial_guess(d, init), lv, uv)
base = ipm.NLP.from_docp(d, x0, ct)
if len(sys.argv) > 6:                      # elastic mode with this penalty
    el = ipm.elastic(base, float(sys.argv[6]))
    r = ipm.solve(el, max_iter=int(sys.argv[7]) if len(sys.argv) > 7 else 600, time_limit=100, verbose=every, linesearch=ls)
    nx = base.n
    c = d.cons(r.x[:nx])
    lc, uc = ct.constraints_bounds(d)
    print("elastic: obj of the original", base.sgn * base.obj(r.x[:nx]), "want", want, "status", r.status, "iters", r.iters, "sum p+n", r.x[nx:].sum(),
          "violation of the original", max(np.max(np.maximum(lc - c, 0)), np.max(np.maximum(c - uc, 0))), "kkt", r.kkt)
else:
    r = ipm.solve(base, max_iter=400, time_limit=60, verbose=every, linesearch=ls)
    print("obj", r.obj, "want", want, "status", r.status, "iters", r.iters, "violation", r.violation, "kkt", r.kkt)
