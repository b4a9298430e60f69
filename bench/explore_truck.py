#!/usr/bin/env python3
"""truck_trailer: scipy's trust-constr on 50 steps (the solution tests/test_gpu_solve_catalogue.py accepts within 8 %), then the in-repo
interior-point loop on the reference's 250-step grid warm-started from it (small initial barrier parameter)."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import ctdirect_jl_amd as ct
import ipm, jit_defs
from test_gpu_solve_catalogue import _solve
np.seterr(all='ignore')
t0 = time.time()
prob, want, init = jit_defs.catalogue("truck_trailer")
obj, want, viol, res = _solve("truck_trailer", "trapeze", 50, maxiter=3000)
print(f"scipy N=50: {obj:.6f} viol {viol:.1e} nit {res.nit} {time.time()-t0:.0f}s", flush=True)
d50 = ct.DOCP(prob, 50, "trapeze", pattern="structural", device=0)
sol = ct.unpack_solution(d50, res.x)
warm = dict(time=sol["T"], state=sol["X"], control=sol["U"], variable=sol["v"])
for N in (100, 250):
    d = ct.DOCP(prob, N, "trapeze", pattern="structural", device=0)
    lv, uv = ct.variables_bounds(d)
    x0 = np.clip(ct.initial_guess(d, warm), lv, uv)
    for mu0 in (1e-2, 1e-4):
        for ls in ("elastic1e3", "elastic1e5"):
            t1 = time.time()
            r = ipm.solve_elastic(ipm.NLP.from_docp(d, x0, ct), rhos=(float(ls[7:]),), max_iter=800, time_limit=40, mu0=mu0, linesearch="filter")
            print(f"ipm N={N} mu0={mu0:g} {ls}: obj {r.obj:.6f} (cat {want}) status {r.status} iters {r.iters} violation {r.violation:.1e} kkt {r.kkt:.1e} {time.time()-t1:.1f}s", flush=True)
            if r.status == 0:
                sol = ct.unpack_solution(d, r.x)
                warm = dict(time=sol["T"], state=sol["X"], control=sol["U"], variable=sol["v"])
    d.close()
