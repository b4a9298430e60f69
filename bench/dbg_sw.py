import os, sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import ctdirect_jl_amd as ct
import problem_folder_defs as pf
from test_gpu_jit import _mp_reference
from helpers import relerr
rt, _, _ = pf.folder("swimmer")
for sch in ("gauss_legendre_2", "midpoint", "trapeze"):
  for N in (1, 3):
    d = ct.DOCP(rt, N, sch, pattern="structural", device=0)
    rng = np.random.default_rng(1)
    x = 0.35 + 0.6 * rng.random(d.dim_NLP_variables)
    y = rng.standard_normal(d.dim_NLP_constraints)
    md, cref, Jref, fref, gref, Href = _mp_reference(pf.mp_problem("swimmer"), sch, N, x, y, 0.6)
    xd = torch.from_numpy(x).cuda()
    c = torch.full((d.dim_NLP_constraints,), 777.0, dtype=torch.float64, device="cuda")
    v = torch.full((d.nnzj,), 777.0, dtype=torch.float64, device="cuda")
    d.cons_jac(xd, c, v)
    c = c.cpu().numpy()
    rows, cols = d.jac_structure()
    print(sch, N, os.environ.get("CTD_DYN_SYM"), "c err", relerr(c, cref), "J err", relerr(v.cpu().numpy(), Jref[rows - 1, cols - 1]), d.launch_info())
    bad = np.nonzero(np.abs(c - cref) > 1e-8)[0]
    print("   bad rows", bad[:20], "cb", d.discretization._state_stage_eqs_block)
