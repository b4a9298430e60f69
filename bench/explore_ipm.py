#!/usr/bin/env python3
"""The in-repo interior-point loop (tests/ipm.py) on the reference's solve catalogue and problem folder at the reference's DEFAULT grid
(250 steps), through the GPU callbacks:  python bench/explore_ipm.py [N]"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import ctdirect_jl_amd as ct
import ipm
import jit_defs
import problem_folder_defs as pf
np.seterr(all='ignore')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 250
jobs = [("cat", n, s) for n, s in (("beam", "midpoint"), ("fuller", "midpoint"), ("jackson", "midpoint"), ("vanderpol", "trapeze"), ("simple_integrator", "midpoint"),
                                   ("bolza_freetf", "midpoint"), ("robbins", "midpoint"), ("double_integrator_tf", "trapeze"), ("moonlander", "midpoint"),
                                   ("double_integrator_nobounds", "midpoint"), ("double_integrator_freet0tf", "midpoint"), ("electric_vehicle", "midpoint"),
                                   ("insurance", "trapeze"), ("space_shuttle", "trapeze"), ("goddard_all", "midpoint"), ("glider", "midpoint"), ("truck_trailer", "trapeze"))]
jobs += [("pf", n, "midpoint") for n in ("algal_bacterial", "bioreactor_1day", "bioreactor_Ndays", "parametric", "swimmer", "goddard_all_f0f1")]
if len(sys.argv) > 2 and sys.argv[2] == "gl2":        # every problem on the stagewise Gauss-Legendre 2 grid
    jobs = [(k, n, "gauss_legendre_2") for k, n, _ in jobs]
if len(sys.argv) > 2 and sys.argv[2] == "hard":      # the ones the loop does not settle on the midpoint / trapeze grid: other schemes
    jobs = [("cat", "moonlander", "gauss_legendre_2"), ("cat", "moonlander", "trapeze"), ("cat", "insurance", "midpoint"), ("cat", "insurance", "gauss_legendre_2"),
            ("cat", "space_shuttle", "midpoint"), ("cat", "space_shuttle", "gauss_legendre_2"), ("pf", "bioreactor_1day", "trapeze"), ("pf", "bioreactor_1day", "gauss_legendre_2"),
            ("pf", "swimmer", "trapeze"), ("pf", "swimmer", "gauss_legendre_2")]
for kind, name, sch in jobs:
    try:
        prob, want, init = jit_defs.catalogue(name) if kind == "cat" else pf.folder(name)
        d = ct.DOCP(prob, N, sch, pattern="structural", device=0)
        lv, uv = ct.variables_bounds(d)
        x0 = np.clip(ct.initial_guess(d, init), lv, uv)
        t0 = time.time()
        r = ipm.solve_auto(ipm.NLP.from_docp(d, x0, ct), max_iter=600, time_limit=30)
        rel = abs(r.obj - want) / abs(want) if want else float("nan")
        print(f"{name:28s} {sch:9s} N={N} obj {r.obj:.6f} catalogued {want} rel {rel:.1e} status {r.status} iters {r.iters} violation {r.violation:.1e} kkt {r.kkt:.1e} {time.time() - t0:.1f} s", flush=True)
        d.close()
    except Exception as e:
        print(name, sch, "EXC", repr(e)[:200], flush=True)
