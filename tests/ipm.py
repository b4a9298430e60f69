"""TEST INFRASTRUCTURE ONLY ("drive Ipopt ... or a small in-repo interior-point / SQP loop through the ABI").

A compact primal-dual interior-point method of Ipopt's class (Waechter & Biegler 2006, restated from the paper, not from any code):
log-barrier on the simple bounds, slacks for the two-sided constraint rows, Newton steps on the perturbed KKT conditions through
ONE sparse symmetric-indefinite system per iteration, inertia-free regularisation by a curvature test (Chiang & Zavala 2016),
fraction-to-the-boundary rule, an l1 merit function with Armijo backtracking, monotone barrier update, gradient-based scaling.

It only ever sees an NLP through the five callbacks the engine serves -- obj, grad, cons, jac_coord (on jac_structure), hess_coord
(on hess_structure, lower triangle, L = sigma f + y'c) -- plus bounds and an initial point, i.e. exactly what the reference hands
to Ipopt through NLPModels (src/solve.jl).  `NLP.from_docp(d)` wraps a ctdirect DOCP (GPU callbacks through the C ABI, host
pointers), `NLP.from_oracle(o)` the CPU oracle (for developing this file without a GPU)."""
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


class NLP:
    def __init__(self, n, m, obj, grad, cons, jac, hess, xl, xu, cl, cu, x0, maximize=False):
        self.n, self.m = n, m
        sgn = -1.0 if maximize else 1.0
        self.sgn = sgn
        self.calls, self.seconds = {}, {}          # per callback: number of calls, seconds spent inside (incl. the sparse-matrix assembly)

        def timed(name, fn):
            def w(*a):
                t0 = time.perf_counter()
                out = fn(*a)
                self.calls[name] = self.calls.get(name, 0) + 1
                self.seconds[name] = self.seconds.get(name, 0.0) + time.perf_counter() - t0
                return out
            return w
        obj, grad, cons, jac, hess = timed("obj", obj), timed("grad", grad), timed("cons", cons), timed("jac", jac), timed("hess", hess)
        self.obj = lambda x: sgn * obj(x)
        self.grad = lambda x: sgn * grad(x)
        self.cons, self.jac = cons, jac
        self.hess = lambda x, y, s: hess(x, y, sgn * s)
        self.xl, self.xu, self.cl, self.cu, self.x0 = xl, xu, cl, cu, x0

    @staticmethod
    def from_docp(d, x0, ct, fused=False):
        n, m = d.dim_NLP_variables, d.dim_NLP_constraints
        cl, cu = ct.constraints_bounds(d)
        xl, xu = ct.variables_bounds(d)
        jr, jc = d.jac_structure()
        hr, hc = d.hess_structure()
        offd = hr != hc

        # the sparse structures are assembled ONCE (the patterns are fixed); a callback then only gathers the returned values into place
        Jt = sp.csr_matrix((np.arange(1, len(jr) + 1, dtype=np.float64), (jr - 1, jc - 1)), shape=(m, n))
        jperm = Jt.data.astype(np.int64) - 1
        Ht = sp.coo_matrix((np.concatenate([np.arange(1, len(hr) + 1), np.arange(1, len(hr) + 1)[offd]]).astype(np.float64),
                            (np.concatenate([hr - 1, hc[offd] - 1]), np.concatenate([hc - 1, hr[offd] - 1]))), shape=(n, n)).tocsr()
        hperm = Ht.data.astype(np.int64) - 1

        def jac_of(vals):
            return sp.csr_matrix((vals[jperm], Jt.indices, Jt.indptr), shape=(m, n))

        def hess_of(vals):
            return sp.csr_matrix((vals[hperm], Ht.indices, Ht.indptr), shape=(n, n))

        def jac(x):
            return jac_of(d.jac_coord(x))

        def hess(x, y, s):
            return hess_of(d.hess_coord(x, y, s))
        nlp = NLP(n, m, d.obj, d.grad, d.cons, jac, hess, xl, xu, cl, cu, x0, maximize=bool(d.flags.max))
        if fused:
            # ONE call of ctd_eval_all_dev_async per iteration (objective + gradient + constraints + Jacobian values + Hessian values at the
            # iterate and its multipliers, two launches) instead of five host-pointer callbacks: `solve` announces the point with
            # prepare(x, y, sigma); the callbacks at exactly that point are served from the device results, any other point (the line
            # search's trial points) goes through the single callbacks as before
            import torch
            dev = torch.device("cuda", d.device)
            buf = dict(x=torch.zeros(n, dtype=torch.float64, device=dev), y=torch.zeros(m, dtype=torch.float64, device=dev), f=torch.zeros(1, dtype=torch.float64, device=dev),
                       g=torch.zeros(n, dtype=torch.float64, device=dev), c=torch.zeros(m, dtype=torch.float64, device=dev),
                       v=torch.zeros(d.nnzj, dtype=torch.float64, device=dev), h=torch.zeros(d.nnzh, dtype=torch.float64, device=dev))
            cache = {}
            sgn = nlp.sgn
            base = dict(obj=nlp.obj, grad=nlp.grad, cons=nlp.cons, jac=nlp.jac, hess=nlp.hess)

            def prepare(x, y, sigma):
                t0 = time.perf_counter()
                buf["x"].copy_(torch.from_numpy(np.ascontiguousarray(x)))
                buf["y"].copy_(torch.from_numpy(np.ascontiguousarray(y)))
                d.eval_all(buf["x"], buf["y"], sgn * sigma, buf["f"], buf["g"], buf["c"], buf["v"], buf["h"], sync=True)
                hv = buf["h"].cpu().numpy()
                cache.clear()
                cache.update(x=x.copy(), y=y.copy(), sigma=sigma, f=sgn * float(buf["f"][0]), g=sgn * buf["g"].cpu().numpy(), c=buf["c"].cpu().numpy(),
                             J=jac_of(buf["v"].cpu().numpy()), H=hess_of(hv))
                nlp.calls["eval_all"] = nlp.calls.get("eval_all", 0) + 1
                nlp.seconds["eval_all"] = nlp.seconds.get("eval_all", 0.0) + time.perf_counter() - t0

            def at(x):
                return "x" in cache and np.array_equal(x, cache["x"])
            nlp.prepare = prepare
            nlp.obj = lambda x: cache["f"] if at(x) else base["obj"](x)
            nlp.grad = lambda x: cache["g"] if at(x) else base["grad"](x)
            nlp.cons = lambda x: cache["c"] if at(x) else base["cons"](x)
            nlp.jac = lambda x: cache["J"] if at(x) else base["jac"](x)
            nlp.hess = lambda x, y, s: cache["H"] if (at(x) and s == cache["sigma"] and np.array_equal(y, cache["y"])) else base["hess"](x, y, s)
        return nlp

    @staticmethod
    def from_oracle(o, x0, maximize=False):
        n, m = o.dim_NLP_variables, o.dim_NLP_constraints
        xl, xu, cl, cu = o.bounds()
        cp, rv = o.jac_pattern()
        jcols = np.repeat(np.arange(n), np.diff(cp))
        hp, hrv = o.hess_pattern()
        hcols = np.repeat(np.arange(n), np.diff(hp))
        offd = hrv != hcols

        def jac(x):
            return sp.csr_matrix((o.jac_coord(x), (rv, jcols)), shape=(m, n))

        def hess(x, y, s):
            v = o.hess_coord(x, y, s)
            return sp.coo_matrix((np.concatenate([v, v[offd]]), (np.concatenate([hrv, hcols[offd]]), np.concatenate([hcols, hrv[offd]]))),
                                 shape=(n, n)).tocsr()
        return NLP(n, m, o.objective, o.gradient, o.constraints, jac, hess, xl, xu, cl, cu, x0, maximize=maximize)


def elastic(nlp, rho=1e4):
    """l1-elastic form of an NLP (what SNOPT calls elastic mode, Ipopt's restoration problem without the proximity term):
        min  f(x) + rho sum(p + n)    s.t.  cl <= c(x) - p + n <= cu,   p, n >= 0
    Every point has a feasible completion (p, n absorb the residual), so the interior-point loop never has to find a feasible point of
    nonlinear equations from far away; for rho above the multipliers the solution has p = n = 0 and is the original problem's."""
    n, m = nlp.n, nlp.m
    E = sp.hstack([-sp.identity(m), sp.identity(m)], format="csr")
    c0 = nlp.cons(np.clip(nlp.x0, nlp.xl, nlp.xu))
    gap = np.where(c0 < nlp.cl, nlp.cl - c0, np.where(c0 > nlp.cu, c0 - nlp.cu, 0.0))
    p0 = np.where(c0 > nlp.cu, gap, 0.0) + 1e-2
    n0 = np.where(c0 < nlp.cl, gap, 0.0) + 1e-2
    el = NLP(n + 2 * m, m,
             obj=lambda z: nlp.obj(z[:n]) + rho * z[n:].sum(),
             grad=lambda z: np.concatenate([nlp.grad(z[:n]), np.full(2 * m, rho)]),
             cons=lambda z: nlp.cons(z[:n]) - z[n:n + m] + z[n + m:],
             jac=lambda z: sp.hstack([nlp.jac(z[:n]), E], format="csr"),
             hess=lambda z, y, s: sp.block_diag([nlp.hess(z[:n], y, s), sp.csr_matrix((2 * m, 2 * m))], format="csr"),
             xl=np.concatenate([nlp.xl, np.zeros(2 * m)]), xu=np.concatenate([nlp.xu, np.full(2 * m, np.inf)]), cl=nlp.cl, cu=nlp.cu,
             x0=np.concatenate([nlp.x0, p0, n0]))
    el.sgn, el.inner = 1.0, nlp            # (nlp.obj / grad / hess already carry the sign of a maximisation)
    return el


def solve_elastic(nlp, rhos=(1e2, 1e4, 1e6), **kw):
    """the interior-point loop on the elastic form, the penalty raised until the elastic variables vanish: returns the Result of the
    ORIGINAL problem (x, y, objective, violation of the original constraints) + .rho, .elastic_sum"""
    total = 0
    last = None
    for rho in rhos:
        el = elastic(nlp, rho)
        r = solve(el, **kw)
        total += r.iters
        x = r.x[:nlp.n]
        c = nlp.cons(x)
        viol = max(float(np.max(np.maximum(nlp.cl - c, 0.0), initial=0.0)), float(np.max(np.maximum(c - nlp.cu, 0.0), initial=0.0)))
        out = Result()
        out.x, out.y, out.zl, out.zu = x, r.y, r.zl[:nlp.n], r.zu[:nlp.n]
        out.obj, out.violation, out.kkt, out.iters, out.rho = nlp.sgn * nlp.obj(x), viol, r.kkt, total, rho
        out.elastic_sum = float(np.abs(r.x[nlp.n:]).sum())
        out.status = 0 if (r.status == 0 and viol <= 1e-6) else 2
        last = out
        if out.status == 0:
            break
    return last


class Result:
    pass


class _BorderedLU:
    """Factorisation of the KKT matrix of a transcription: banded after a reverse Cuthill-McKee ordering EXCEPT for the few dense rows /
    columns of the optimisation variables v (free final time: every step depends on it).  General-purpose fill-reducing orderings
    drown in the fill of that arrow (10 000-step Goddard: 18 s per factorisation with SuperLU's default ordering); eliminating the
    dense border by a Schur complement and factorising the banded rest in its natural order takes 0.08 s.  The ordering is computed once
    (the pattern does not change between iterations)."""

    def __init__(self):
        self.keep = None

    def factor(self, K):
        from scipy.sparse.csgraph import reverse_cuthill_mckee
        K = K.tocsr()
        if getattr(self, "plain", False):          # (decided at the first factorisation: SuperLU's own ordering was faster on this pattern)
            self.K, self.size, self.dense = K, K.shape[0], np.zeros(0, dtype=int)
            self.rows = np.arange(self.size)
            self.lu = spla.splu(K.tocsc())
            return self
        first = self.keep is None
        t0 = time.perf_counter()
        self._factor_bordered(K, reverse_cuthill_mckee)
        if first:
            tb = time.perf_counter() - t0
            if tb > 0.3 and K.shape[0] <= 30000:   # wide step blocks: threshold pivoting in the natural order breaks the band
                t0 = time.perf_counter()
                lu = spla.splu(K.tocsc())
                if time.perf_counter() - t0 < tb:
                    self.plain, self.lu, self.dense, self.rows = True, lu, np.zeros(0, dtype=int), np.arange(self.size)
        return self

    def _factor_bordered(self, K, reverse_cuthill_mckee):
        if self.keep is None or self.size != K.shape[0]:
            cnt = np.diff(K.indptr)
            self.size = K.shape[0]
            self.dense = np.where(cnt > max(200, 20 * np.median(cnt)))[0]
            self.keep = np.setdiff1d(np.arange(self.size), self.dense)
            K0 = K[self.keep][:, self.keep]
            self.perm = np.asarray(reverse_cuthill_mckee(K0.tocsr(), symmetric_mode=True))
            self.rows = self.keep[self.perm]
        self.K = K
        K0p = K[self.rows][:, self.rows].tocsc()
        self.lu = spla.splu(K0p)        # (COLAMD: its fill bound holds whatever rows the partial pivoting picks; the RCM order only helps locality)
        if len(self.dense):
            self.B = K[self.rows][:, self.dense].toarray()
            self.Bt = K[self.dense][:, self.rows].toarray()
            self.Y = self.lu.solve(self.B)
            self.S = K[self.dense][:, self.dense].toarray() - self.Bt @ self.Y
        return self

    def _solve_once(self, rhs):
        r0, x = rhs[self.rows], np.zeros(self.size)
        y0 = self.lu.solve(r0)
        if len(self.dense):
            x1 = np.linalg.solve(self.S, rhs[self.dense] - self.Bt @ y0)
            x[self.dense] = x1
            y0 = y0 - self.Y @ x1
        x[self.rows] = y0
        return x

    def solve(self, rhs):
        x = self._solve_once(rhs)
        for _ in range(2):              # iterative refinement (threshold pivoting on the band)
            x = x + self._solve_once(rhs - self.K @ x)
        return x


def solve_auto(nlp, **kw):
    """the filter line search first (it does not stall on the negative curvature of singular arcs: the Gauss-Legendre transcriptions of
    Goddard), the l1 merit function when that does not converge (the 8-state quadrotor)"""
    r = solve(nlp, linesearch="filter", **kw)
    if r.status != 0:
        r2 = solve(nlp, linesearch="merit", **kw)
        if r2.status == 0 or (r2.status == 1 and r.status != 1):
            r2.iters += r.iters
            return r2
    return r


def solve(nlp, tol=1e-8, max_iter=500, mu0=0.1, verbose=False, acceptable_tol=1e-6, time_limit=None, kappa_eps=10.0, dc0=0.0, mu_lin=0.2, linesearch="merit"):
    """returns Result(x, y, obj, status, iters, violation, kkt); status 0 = converged to tol, 1 = acceptable, 2 = iteration / time limit"""
    t_start = time.time()
    n, m = nlp.n, nlp.m
    eq = nlp.cl == nlp.cu
    ineq = ~eq
    mi = int(ineq.sum())
    # --- scaling (Ipopt's gradient-based rule, section 3.8): objective and every constraint row so that gradients at x0 are <= 100
    x = np.clip(nlp.x0.astype(float).copy(), nlp.xl, nlp.xu)
    g0 = nlp.grad(x)
    sf = min(1.0, 100.0 / max(1e-300, np.max(np.abs(g0)))) if g0.size else 1.0
    J0 = nlp.jac(x)
    rown = np.asarray(abs(J0).max(axis=1).todense()).ravel() if m else np.zeros(0)
    sc = np.minimum(1.0, 100.0 / np.maximum(rown, 1e-300))
    Sc = sp.diags(sc)
    # --- variables z = (x, s): s = slacks of the inequality rows; bounds, slightly relaxed (section 3.5)
    zl = np.concatenate([nlp.xl, (sc * nlp.cl)[ineq]])
    zu = np.concatenate([nlp.xu, (sc * nlp.cu)[ineq]])
    hasl, hasu = np.isfinite(zl), np.isfinite(zu)
    zl = np.where(hasl, zl - 1e-8 * np.maximum(1.0, np.abs(zl)), -np.inf)
    zu = np.where(hasu, zu + 1e-8 * np.maximum(1.0, np.abs(zu)), np.inf)
    nz = n + mi
    ceq_target = np.where(eq, sc * nlp.cl, 0.0)

    def push(z):      # initial point strictly inside the bounds (section 3.6)
        k1 = k2 = 1e-2
        pl = np.where(hasl, np.minimum(k1 * np.maximum(1.0, np.abs(zl)), k2 * np.where(hasu, zu - zl, np.inf)), 0.0)
        pu = np.where(hasu, np.minimum(k1 * np.maximum(1.0, np.abs(zu)), k2 * np.where(hasl, zu - zl, np.inf)), 0.0)
        z = np.where(hasl, np.maximum(z, zl + pl), z)
        return np.where(hasu, np.minimum(z, zu - pu), z)

    def cfun(xv):
        return sc * nlp.cons(xv)

    c_raw = cfun(x)
    z = push(np.concatenate([x, c_raw[ineq]]))
    lam = np.zeros(m)
    vl = np.where(hasl, 1.0, 0.0)
    vu = np.where(hasu, 1.0, 0.0)
    mu = mu0
    Aslack = sp.csr_matrix((-np.ones(mi), (np.where(ineq)[0], np.arange(mi))), shape=(m, mi)) if mi else sp.csr_matrix((m, 0))

    def resid(z):
        c = cfun(z[:n])
        r = c - ceq_target
        r[ineq] -= z[n:]
        return r

    def barrier(z, mu):
        b = 0.0
        if hasl.any():
            b -= mu * np.sum(np.log(z[hasl] - zl[hasl]))
        if hasu.any():
            b -= mu * np.sum(np.log(zu[hasu] - z[hasu]))
        return b

    def phi(z, mu):
        return sf * nlp.obj(z[:n]) + barrier(z, mu)

    dw_last = 0.0
    nu = 1.0
    res = Result()
    status = 2
    it = 0
    err0 = float("inf")
    last_alpha = None
    kkt_solver = _BorderedLU()
    filt, th_ref = [], None
    for it in range(max_iter):
        xv = z[:n]
        if hasattr(nlp, "prepare"):
            nlp.prepare(xv, lam * sc, sf)
        g = np.concatenate([sf * nlp.grad(xv), np.zeros(mi)])
        J = Sc @ nlp.jac(xv)
        A = sp.hstack([J, Aslack], format="csr") if mi else J
        r = resid(z)
        dl = np.where(hasl, z - zl, 1.0)
        du = np.where(hasu, zu - z, 1.0)
        dual_inf = g + A.T @ lam - vl + vu
        sd = max(1.0, (np.abs(lam).sum() + np.abs(vl).sum() + np.abs(vu).sum()) / max(1, m + 2 * nz) / 100.0)
        compl = lambda mu_: max(np.max(np.abs(dl * vl - mu_)[hasl], initial=0.0), np.max(np.abs(du * vu - mu_)[hasu], initial=0.0))
        err0 = max(np.max(np.abs(dual_inf), initial=0.0) / sd, np.max(np.abs(r), initial=0.0), compl(0.0) / sd)
        errmu = max(np.max(np.abs(dual_inf), initial=0.0) / sd, np.max(np.abs(r), initial=0.0), compl(mu) / sd)
        if verbose and (it % verbose == 0):
            print(f"it {it:4d} obj {nlp.sgn * nlp.obj(xv):.8f} inf_pr {np.max(np.abs(r), initial=0.0):.2e} inf_du {np.max(np.abs(dual_inf), initial=0.0):.2e} mu {mu:.1e} dw {dw_last:.1e} "
                  f"sd {sd:.1e} |lam| {np.max(np.abs(lam), initial=0.0):.1e} |v| {max(np.max(vl, initial=0.0), np.max(vu, initial=0.0)):.1e} last alpha {last_alpha}")
        if err0 <= tol:
            status = 0
            break
        if time_limit is not None and time.time() - t_start > time_limit:
            break
        # barrier update (monotone, section 3.1, eq. 7)
        if errmu <= kappa_eps * mu and mu > tol / 10.0:          # (one reduction per iteration: a Newton step at every barrier level)
            mu = max(tol / 10.0, min(mu_lin * mu, mu ** 1.5))
            nu = 1.0
            filt = []
        tau = max(0.99, 1.0 - mu)
        # Newton system on (dz, dlam) with the bound multipliers eliminated
        Wx = nlp.hess(xv, lam * sc, sf)
        W = sp.block_diag([Wx, sp.csr_matrix((mi, mi))], format="csr") if mi else Wx
        sig = np.where(hasl, vl / dl, 0.0) + np.where(hasu, vu / du, 0.0)
        gphi = g - np.where(hasl, mu / dl, 0.0) + np.where(hasu, mu / du, 0.0)
        rhs = -np.concatenate([gphi + A.T @ lam, r])
        dw = 0.0
        dc = dc0
        dz = dlam = None
        lu = None
        for attempt in range(40):
            kkt = kkt_solver
            K = sp.bmat([[W + sp.diags(sig + dw), A.T], [A, sp.diags(np.full(m, -dc))]], format="csr")
            try:
                lu = kkt.factor(K)
                sol = lu.solve(rhs)
                ok = bool(np.all(np.isfinite(sol)))
                if ok and np.max(np.abs(K @ sol - rhs)) > 1e-6 * max(1.0, np.max(np.abs(rhs))):
                    ok = False                       # (numerically) singular: a rank-deficient Jacobian -- regularise the constraint block (section 3.1)
                    if dc == 0.0:
                        dc = 1e-8 * mu ** 0.25
                        continue
            except (RuntimeError, np.linalg.LinAlgError):
                ok = False
                if dc == 0.0:
                    dc = 1e-8 * mu ** 0.25
            if ok:
                dz, dlam = sol[:nz], sol[nz:]
                curv = dz @ (W @ dz) + dz @ ((sig + dw) * dz)
                # inertia-free test: positive curvature of the barrier Lagrangian along the step (on the null space of A up to the residual)
                if curv >= 1e-10 * (dz @ dz):
                    break
            if dw == 0.0:
                dw = 1e-4 if dw_last == 0.0 else max(1e-20, dw_last / 3.0)
            else:
                dw *= 100.0 if dw_last == 0.0 else 8.0
            if dw > 1e40:
                break
        if dz is None:
            status = 3
            break
        if dw > 0:
            dw_last = dw
        dvl = np.where(hasl, (mu - vl * dz) / dl - vl, 0.0)
        dvu = np.where(hasu, (mu + vu * dz) / du - vu, 0.0)
        # fraction to the boundary
        def amax(v, dv, lim):
            neg = dv < 0
            return min(1.0, np.min(-lim * v[neg] / dv[neg], initial=1.0)) if neg.any() else 1.0
        a_pri = min(amax(np.where(hasl, dl, 1.0), np.where(hasl, dz, 0.0), tau), amax(np.where(hasu, du, 1.0), np.where(hasu, -dz, 0.0), tau))
        a_du = min(amax(np.where(hasl, vl, 1.0), dvl, tau), amax(np.where(hasu, vu, 1.0), dvu, tau))
        # l1 merit function phi_mu + nu ||c||_1, penalty from the model decrease (Byrd, Hribar, Nocedal)
        th = np.abs(r).sum()
        curv = max(0.0, dz @ (W @ dz) + dz @ (sig * dz))
        gd = gphi @ dz
        if th > 1e-14:
            nu_need = (gd + 0.5 * curv) / (0.9 * th)
            if nu < nu_need:
                nu = nu_need + 1.0
        D = gd - nu * th
        phi0 = phi(z, mu)
        m0 = phi0 + nu * th
        if th_ref is None:
            th_ref = max(1.0, th)
        augment = [False]

        def acceptable(zq, rq, aq):
            """merit: Armijo on phi + nu ||c||_1.  filter (section 2.3): sufficient progress in the violation OR in the barrier function against
            the current point, not dominated by a filter entry; Armijo on the barrier function when the step is a descent step at an
            (almost) feasible point"""
            pq, tq = phi(zq, mu), np.abs(rq).sum()
            if not (np.isfinite(pq) and np.isfinite(tq)):
                return False
            if linesearch != "filter":
                return pq + nu * tq <= m0 + 1e-8 * aq * D + 10 * np.finfo(float).eps * abs(m0)
            if tq > 1e4 * th_ref or any(tq >= tf_ and pq >= pf_ for tf_, pf_ in filt):
                return False
            if gd < 0 and aq * (-gd) ** 2.3 > th ** 1.1 and th <= 1e-4 * th_ref:
                augment[0] = False
                return pq <= phi0 + 1e-8 * aq * gd + 10 * np.finfo(float).eps * abs(phi0)
            augment[0] = True
            return tq <= (1 - 1e-5) * th or pq <= phi0 - 1e-5 * th
        a = a_pri
        accepted = False
        for ls in range(40):
            zt = z + a * dz
            rt = resid(zt)
            if acceptable(zt, rt, a):
                accepted = True
                break
            if ls == 0 and np.abs(rt).sum() >= th and lu is not None:
                # second-order correction (section 2.4): the full step was refused because the constraint violation grew (Maratos effect)
                csoc = a * r + rt
                for p_ in range(4):
                    sol2 = lu.solve(-np.concatenate([gphi + A.T @ lam, csoc]))
                    if not np.all(np.isfinite(sol2)):
                        break
                    d2 = sol2[:nz]
                    a2 = min(amax(np.where(hasl, dl, 1.0), np.where(hasl, d2, 0.0), tau), amax(np.where(hasu, du, 1.0), np.where(hasu, -d2, 0.0), tau))
                    z2 = z + a2 * d2
                    r2 = resid(z2)
                    if acceptable(z2, r2, a2):
                        zt, a, dlam, accepted = z2, a2, sol2[nz:], True
                        break
                    if np.abs(r2).sum() > 0.99 * np.abs(rt).sum():
                        break
                    csoc = a2 * csoc + r2
                    rt = r2
                if accepted:
                    break
            a *= 0.5
        if not accepted:
            # no decrease along the Newton direction: take the tiny step anyway once, with more regularisation next time
            dw_last = min(1e8, max(1e-4, dw_last * 10.0))
            zt = z + a * dz
        if accepted and linesearch == "filter" and augment[0]:
            filt.append(((1 - 1e-5) * th, phi0 - 1e-5 * th))
        last_alpha = (round(a_pri, 4), round(a_du, 4), a, accepted)
        z = zt
        lam = lam + a * dlam
        vl = vl + a_du * dvl
        vu = vu + a_du * dvu
        # keep the bound multipliers near mu / slack (section 3.2, eq. 16)
        ks = 1e10
        dl = np.where(hasl, z - zl, 1.0)
        du = np.where(hasu, zu - z, 1.0)
        vl = np.where(hasl, np.clip(vl, mu / (ks * dl), ks * mu / dl), 0.0)
        vu = np.where(hasu, np.clip(vu, mu / (ks * du), ks * mu / du), 0.0)
    xv = z[:n]
    c = nlp.cons(xv)
    res.x, res.y, res.iters = xv, lam * sc / sf, it
    res.zl, res.zu = vl[:n] / sf, vu[:n] / sf          # bound multipliers of min sgn f:  grad + J'y - zl + zu = 0
    res.obj = nlp.sgn * nlp.obj(xv)
    res.violation = max(float(np.max(np.maximum(nlp.cl - c, 0.0), initial=0.0)), float(np.max(np.maximum(c - nlp.cu, 0.0), initial=0.0)),
                        float(np.max(np.maximum(nlp.xl - xv, 0.0), initial=0.0)), float(np.max(np.maximum(xv - nlp.xu, 0.0), initial=0.0)))
    if status == 2 and err0 <= acceptable_tol:
        status = 1
    res.status = status
    res.kkt = float(err0)
    return res
