"""Differential fuzz campaign (a tool, not collected by pytest): the engine on the GPU against the CPU oracle on random
(problem, scheme, grid size 1..2500, uniform / user time grid, pattern mode, tile overrides, shard of the grid): constraints,
Jacobian values, objective, gradient, Hessian values; unwritten outputs are detected with a sentinel.

    python tests/fuzz_gpu.py <seconds> <seed>

Round 1: two campaigns of ~12 000 cases each; the first one found the tiny-grid shard bug fixed in
tests/test_kernel_logic_emu.py::test_every_split_of_a_tiny_grid_composes, the second ran clean (worst relative
difference: Hessian 1.3e-11, everything else <= 4e-15)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import ctdirect_jl_amd as ct
from oracle.oracle import OracleDOCP
from helpers import relerr, TOL, bench_inputs, describe
budget, seed = float(sys.argv[1]), int(sys.argv[2])
JIT = len(sys.argv) > 3 and sys.argv[3] in ("jit", "jitcs")      # also draw the expression twins of the registry problems (hiprtc)
CS_ONLY = len(sys.argv) > 3 and sys.argv[3] in ("cs", "jitcs")    # midpoint scheme only, mostly with several controls per step
if JIT:
    import jit_defs
rng = np.random.default_rng(seed)
probs, schemes = list(ct.PROBLEMS), list(ct.SCHEMES)
t_end = time.time() + budget
ncase = nfail = 0
worst = {}
t_note = time.time() + 60.0
while time.time() < t_end:
    if time.time() >= t_note:          # progress line (long runs behind a quiet pipe look hung to gpurun)
        print(f"... {ncase} cases, {nfail} failures", flush=True)
        t_note += 60.0
    prob, sch = probs[rng.integers(len(probs))], ("midpoint" if CS_ONLY else schemes[rng.integers(len(schemes))])
    big = rng.random() < 0.25
    N = int(rng.integers(1, 2500 if big else 90))
    if prob in ("quadrotor", "quadrotor12") and big: N = int(rng.integers(1, 600))
    tg = None
    if rng.random() < 0.4:
        tg = np.concatenate([[0.0], np.cumsum(0.05 + rng.random(N))]); tg = tg / tg[-1] * (0.5 + rng.random()) + rng.random() * 0.3
    r_ = rng.random()
    mode = "structural" if r_ < 0.25 else ("optimized" if r_ < 0.45 and sch != "euler_implicit" else "manual")
    # round 2: the Hessian kernels' variants -- lane-per-step kernel forced / off, walk modes, edge workgroups
    os.environ["CTD_HESS_STEP"] = str(rng.choice(["0", "1", "2", "2"]))
    os.environ["CTD_HESS_COMPACT"] = str(rng.choice(["", "", "0", "1", "2"]))
    os.environ["CTD_HESS_EDGE_BLOCKS"] = str(rng.choice(["", "", "1", "3", "7"]))
    # round 4: the value order of the Jacobian (CSR: row-order emit template; checked against the oracle's CSC values under the host
    # permutation) and the number of edge workgroups of the constraint / Jacobian kernel
    order = "csr" if rng.random() < 0.4 else "csc"
    os.environ["CTD_EDGE_BLOCKS"] = str(rng.choice(["", "", "1", "2", "5", "16"]))
    # the long-grid launch geometry (eight waves per workgroup, the largest tile whose records fit 64 / 74 KiB) forced on every grid
    os.environ["CTD_LONG_GRID_ROUNDS"] = str(rng.choice(["", "", "0"]))
    tile = int(rng.integers(1, 70)) if rng.random() < 0.5 else 0
    htile = int(rng.integers(1, 70)) if rng.random() < 0.5 else 0
    steps = None
    if rng.random() < 0.35 and N >= 2:
        a = int(rng.integers(0, N)); b = int(rng.integers(a + 1, N + 1)); steps = (a, b)
    os.environ["CTD_TILE"] = str(tile) if tile else ""; os.environ["CTD_HESS_TILE"] = str(htile) if htile else ""
    use_twin = JIT and prob in jit_defs.TWINS and rng.random() < 0.5
    # (operator-level dependence is a property of the expressions: Goddard's twin writes r' = x2 where the reference -- and the
    # registry functor and the oracle -- write F0 + u F1 with a zero component, so their optimized patterns differ by design)
    if use_twin and mode == "optimized": mode = "structural"
    # round 3: the direct-shooting layout (midpoint scheme): 2 - 3 controls per step from the registry, up to 6 for the twins
    cs = 1
    if sch == "midpoint" and rng.random() < (0.9 if CS_ONLY else 0.5):
        cs = int(rng.integers(2, 7 if use_twin else 4))
    api = int(rng.integers(0, 3))          # 0: fused device call, 1: cons + jac_coord separately, 2: host-pointer (numpy) call
    desc = (f"{prob} {sch} N={N} grid={'user' if tg is not None else 'uniform'} mode={mode} tile={tile} htile={htile} steps={steps} "
            f"twin={int(use_twin)} api={api} step={os.environ['CTD_HESS_STEP']} compact={os.environ['CTD_HESS_COMPACT']} "
            f"eb={os.environ['CTD_HESS_EDGE_BLOCKS']} cs={cs} order={order} ceb={os.environ['CTD_EDGE_BLOCKS']} lg={os.environ['CTD_LONG_GRID_ROUNDS']}")
    try:
        d = ct.DOCP(jit_defs.twin(prob) if use_twin else prob, N, sch, time_grid=tg, pattern=mode, device=0, steps=steps, control_steps=cs, value_order=order)
        o = OracleDOCP(prob, sch, N, time_grid=tg, control_steps=cs) if tg is not None else OracleDOCP(prob, sch, N, control_steps=cs)
        if mode == "structural": o.set_pattern_mode(1)
        if mode == "optimized": o.set_pattern_mode(2)
        nvar, ncon = d.dim_NLP_variables, d.dim_NLP_constraints
        assert (nvar, ncon) == (o.dim_NLP_variables, o.dim_NLP_constraints)
        if rng.random() < 0.5:
            x = 0.35 + 0.25 * rng.random(nvar)
            if d.dims.NLP_v: x[-d.dims.NLP_v:] = np.sort(0.3 + rng.random(d.dims.NLP_v))      # t0 < tf when both are free
            sane = not prob.startswith("goddard")      # Goddard's drag term exp(-500 (r - 1)) is ~1e100 there: no 2nd-order check
        else:
            x = bench_inputs(describe(o, prob, sch), perturb=0.0) * (1 + 0.02 * (rng.random(nvar) - 0.5))
            sane = True
        y = 0.6 + 0.4 * rng.random(ncon)
        xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
        SENT = 777.25
        c = torch.full((ncon,), SENT, dtype=torch.float64, device="cuda"); v = torch.full((d.nnzj,), SENT, dtype=torch.float64, device="cuda")
        if api == 0 or steps is not None:
            d.cons_jac(xd, c, v); torch.cuda.synchronize()
            c, v = c.cpu().numpy(), v.cpu().numpy()
        elif api == 1:
            d.cons(xd, c); d.jac_coord(xd, v); torch.cuda.synchronize()
            c, v = c.cpu().numpy(), v.cpu().numpy()
        else:
            c, v = d.cons_jac(x)
        cref, vref = o.constraints(x), o.jac_coord(x)
        if order == "csr":                                      # the oracle's values are in the reference's CSC order
            cp_, rv_ = o.jac_pattern()
            cols_ = np.repeat(np.arange(len(cp_) - 1), np.diff(cp_))
            vref = vref[np.lexsort((cols_, rv_))]
        errs = {}
        if steps is None:
            errs["c"], errs["jac"] = relerr(c, cref), relerr(v, vref)
            assert not np.any(c == SENT) and not np.any(v == SENT), "unwritten outputs"
        else:
            wc, wv = c != SENT, v != SENT                       # a shard writes its rows / CSC ranges only
            errs["c"], errs["jac"] = relerr(c[wc], cref[wc]), relerr(v[wv], vref[wv])
            sh = d.shard
            assert wc[sh.c_row_begin:sh.c_row_end].all() and wv[sh.vals_main_begin:sh.vals_main_end].all(), "shard range not written"
        errs["obj"] = abs(d.obj(xd) - o.objective(x)) / max(1.0, abs(o.objective(x))) if steps is None else 0.0
        errs["grad"] = relerr(d.grad(xd).cpu().numpy(), o.gradient(x))
        if sane:
            hs = torch.full((d.nnzh,), SENT, dtype=torch.float64, device="cuda")
            d.hess_coord(xd, yd, 0.8, hs); torch.cuda.synchronize()
            h = hs.cpu().numpy(); href = o.hess_coord(x, y, 0.8)
            if steps is None:
                assert not np.any(h == SENT), "unwritten Hessian entries"
                errs["hess"] = relerr(h, href)
            else:
                w = h != SENT
                w[np.array(d.hess_shard_info()[2], dtype=int)] = False          # V x V: partial sums on a shard
                b0, b1 = d.hess_shard_info()[0], d.hess_shard_info()[1]
                assert w[b0:b1].all(), "Hessian shard range not written"
                errs["hess"] = relerr(h[w], href[w])
        ncase += 1
        for k, e in errs.items():
            if not (e <= TOL):
                nfail += 1; print("FAIL", desc, k, e, flush=True)
            if e > worst.get(k, (0, ""))[0]: worst[k] = (e, desc)
        d.close()
    except Exception as ex:
        nfail += 1; print("EXC ", desc, repr(ex)[:300], flush=True)
print(f"cases {ncase} failures {nfail}")
for k, (e, dsc) in worst.items(): print(f"worst {k}: {e:.3e}  at {dsc}")
