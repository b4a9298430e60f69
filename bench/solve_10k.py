#!/usr/bin/env python3
"""north_star's 10 000-step Goddard transcription SOLVED end to end: the in-repo interior-point loop (tests/ipm.py) through the GPU
callbacks (host-pointer entry points of the C ABI), with the time spent inside the callbacks beside the time of the host's sparse
linear algebra.   python -u bench/solve_10k.py [N]"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import ctdirect_jl_amd as ct
import ipm
np.seterr(all='ignore')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
for sch in ("trapeze", "midpoint", "gauss_legendre_2"):          # gauss_legendre_2 at 10 000 steps = BASELINE.json configs[1], the bench workload
    d = ct.DOCP("goddard", N, sch, pattern="structural", device=0)
    lv, uv = ct.variables_bounds(d)
    x0 = np.clip(ct.initial_guess(d, "problem"), lv, uv)
    nlp = ipm.NLP.from_docp(d, x0, ct)
    t0 = time.time()
    r = ipm.solve_auto(nlp, max_iter=300, time_limit=150)
    el = time.time() - t0
    cb = sum(nlp.seconds.values())
    print(f"goddard/{sch} N={N}: nvar {d.dim_NLP_variables} ncon {d.dim_NLP_constraints} nnzj {d.nnzj} nnzh {d.nnzh} | objective {r.obj:.7f} (catalogue 1.01257) status {r.status} "
          f"iterations {r.iters} violation {r.violation:.1e} KKT {r.kkt:.1e} | total {el:.1f} s, inside the callbacks {cb:.2f} s "
          f"({ {k: (nlp.calls[k], round(v, 3)) for k, v in nlp.seconds.items()} }), host linear algebra and bookkeeping {el - cb:.1f} s", flush=True)
    d.close()
