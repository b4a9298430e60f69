import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import problem_folder_defs as pf
from test_gpu_problem_folder import _solve
runs = [("parametric", "midpoint", 60, 800), ("swimmer", "midpoint", 100, 1500), ("swimmer", "gauss_legendre_2", 50, 1500),
        ("algal_bacterial", "gauss_legendre_2", 100, 4000), ("algal_bacterial", "midpoint", 200, 4000), ("algal_bacterial", "trapeze", 200, 4000),
        ("bioreactor_Ndays", "midpoint", 300, 3000)]
for name, sch, N, it in runs:
    t0 = time.time()
    try:
        obj, want, viol, res = _solve(name, sch, N, maxiter=it)
        print(f"{name}/{sch} N={N}: objective {obj:.6f} (catalogue {want}), violation {viol:.1e}, status {res.status}, nit {res.nit}, {time.time()-t0:.0f} s", flush=True)
    except Exception as e:
        print(name, sch, N, "FAILED", repr(e)[:200], flush=True)
