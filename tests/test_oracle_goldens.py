"""CPU tests: the oracle (oracle/ctd_oracle.cpp) against every known-answer value the reference's own tests hold
for this path and against the 50-digit mpmath fixtures in tests/golden/."""
import os

import numpy as np
import pytest

from helpers import (TOL, csc_to_set, dense_on_pattern, golden_files, hess_golden_files, hess_on_pattern, load_golden,
                     load_hess_golden, relerr)


def _exact_xu(d):
    """x = t^2, u = 2t, K = 2 t_ij  (reference: test/ci/test_discretization_stagewise.jl:20-42)."""
    _, T = d.grids()
    _, _, c = d.butcher()
    blk, s = d.step_variables_block, d.stage
    xu = np.zeros(d.dim_NLP_variables)
    for i in range(d.steps + 1):
        xu[i * blk] = T[i] ** 2
    for i in range(d.steps):
        ti, hi = T[i], T[i + 1] - T[i]
        for j in range(s):
            tij = ti + c[j] * hi
            xu[i * blk + 1 + j] = 2 * tij
            xu[i * blk + 1 + s + j] = 2 * tij
    return xu


@pytest.mark.parametrize("scheme,stage,blk,eqs,nvar,ncon", [
    ("gauss_legendre_2", 2, 5, 3, 16, 11),
    ("gauss_legendre_3", 3, 7, 4, 22, 14),
])
def test_G1_stagewise_exact_feasible_trajectory(oracle_lib, scheme, stage, blk, eqs, nvar, ncon):
    # reference: test/ci/test_discretization_stagewise.jl:53-100 and :176-198
    grid = [0.0, 0.2, 0.6, 1.0]
    d = oracle_lib.OracleDOCP("stagewise_scalar", scheme, time_grid=grid)
    assert (d.n, d.m, d.nv, d.boundary_cons, d.path_cons) == (1, 1, 0, 2, 0)
    assert d.stage == stage
    assert d.step_variables_block == blk
    assert d.state_stage_eqs_block == eqs
    assert d.dim_NLP_variables == nvar
    assert d.dim_NLP_constraints == ncon
    _, fixed = d.grids()
    assert np.allclose(fixed, grid)
    lv, uv, lc, uc = d.bounds()
    for j in range(stage):                       # bounds on stage controls :72-76
        assert lv[1 + j] == 0.0 and uv[1 + j] == 2.0
    xu = _exact_xu(d)
    assert xu[1] != xu[stage]                    # u11 != u1s :87
    c = d.constraints(xu)
    assert np.max(np.abs(c - lc)) <= 1e-12      # :96-97
    assert np.max(np.abs(c - uc)) <= 1e-12
    assert abs(d.objective(xu) - 4.0 / 3.0) <= 1e-12   # :98


def test_G2_goddard_midpoint_manual_nnz(oracle_lib):
    # reference: test/ci/test_modeler_solver.jl:37  (default Collocation: midpoint, grid_size 250)
    d = oracle_lib.OracleDOCP("goddard", "midpoint", 250)
    assert d.jac_nnz() == 6028
    full, lower = d.hess_nnz()
    assert lower == 6519


def test_G3_zero_control_dims(oracle_lib):
    # reference: test/ci/test_zero_control_allocations.jl:31 and :138
    d = oracle_lib.OracleDOCP("estimate_initial_condition", "midpoint", 10)
    assert d.m == 0 and d.dim_NLP_variables == 10 * 2 + 2 + 2 and d.dim_NLP_constraints > 0
    d = oracle_lib.OracleDOCP("estimate_rotation_rate", "midpoint", 10)
    assert d.m == 0 and d.nv == 1 and d.dim_NLP_variables == 10 * 2 + 2 + 1
    colptr, rowval = d.jac_pattern()
    assert rowval.min() >= 0 and rowval.max() < d.dim_NLP_constraints      # :141-150


@pytest.mark.parametrize("N,nvar,ncon,nnzh", [(1000, 4005, 6007, 30024), (10000, 40005, 60007, 300024)])
def test_G4_goddard_all_trapeze_sizes(oracle_lib, N, nvar, ncon, nnzh):
    # reference: test/archives/AD_backend.md:59-61 (sizes and manual Hessian nnz); the archived manual Jacobian
    # nnz 42043 / 420043 (:63) is what STRUCTURAL mode gives; the current trapeze.jl:203 gives 39043 (hazard H1).
    d = oracle_lib.OracleDOCP("goddard_all", "trapeze", N)
    assert (d.dim_NLP_variables, d.dim_NLP_constraints) == (nvar, ncon)
    assert d.hess_nnz()[1] == nnzh
    assert d.jac_nnz() == 39 * N + 43
    d.set_pattern_mode(1)
    assert d.jac_nnz() == 42 * N + 43


@pytest.mark.parametrize("sw,cc", [("gauss_legendre_2", "gauss_legendre_2_constant_control"),
                                    ("gauss_legendre_3", "gauss_legendre_3_constant_control")])
def test_G5_stagewise_vs_constant_control_dims(oracle_lib, sw, cc):
    # reference: test/ci/test_discretization_stagewise.jl:131-139
    grid = np.linspace(0.0, 1.0, 21)
    a = oracle_lib.OracleDOCP("stagewise_scalar", cc, time_grid=grid)
    b = oracle_lib.OracleDOCP("stagewise_scalar", sw, time_grid=grid)
    assert a.dim_NLP_constraints == b.dim_NLP_constraints
    assert b.dim_NLP_variables == a.dim_NLP_variables + b.steps * (b.stage - 1) * b.m


def test_G6_default_initial_guess(oracle_lib):
    # reference: src/DOCP_variables.jl:126, test/ci/test_initial_guess.jl:32-38
    d = oracle_lib.OracleDOCP("double_integrator_path", "midpoint", 10)
    assert np.all(d.initial_guess(use_problem_init=False) == 0.1)
    g = oracle_lib.OracleDOCP("goddard", "gauss_legendre_2", 5)
    x0 = g.initial_guess(use_problem_init=True)
    blk = g.step_variables_block
    assert np.allclose(x0[0:3], [1.01, 0.05, 0.8]) and np.allclose(x0[5 * blk:5 * blk + 3], [1.01, 0.05, 0.8])
    assert np.all(x0[3:blk] == 0.1) and x0[-1] == 0.1


def test_survey_table_sizes(oracle_lib):
    # SURVEY.md section 8 head table (derived from the reference's size formulas)
    rows = [("goddard", "trapeze", 100, 405, 304, 2428),
            ("goddard_all", "trapeze", 100, 405, 607, 3943),
            ("goddard", "gauss_legendre_2", 10000, 110004, 90004, 1110028),
            ("double_integrator_path", "midpoint", 100000, 300002, 300005, 1300019)]
    for prob, sch, N, nvar, ncon, nnzj in rows:
        d = oracle_lib.OracleDOCP(prob, sch, N)
        assert (d.dim_NLP_variables, d.dim_NLP_constraints, d.jac_nnz()) == (nvar, ncon, nnzj)
    # formula-only rows (patterns too large to materialise in a unit test)
    for prob, sch, N, nvar, ncon in [("goddard", "gauss_legendre_3", 80000, 1200004, 960004),
                                     ("quadrotor12", "gauss_legendre_3", 20000, 1200013, 980024),
                                     ("quadrotor", "gauss_legendre_3", 20000, 880009, 660015)]:
        d = oracle_lib.OracleDOCP(prob, sch, N)
        assert (d.dim_NLP_variables, d.dim_NLP_constraints) == (nvar, ncon)


def test_time_grid_errors(oracle_lib):
    # reference: src/DOCP_data.jl:186-189 (ArgumentError), :342-349 (unknown scheme)
    with pytest.raises(ValueError):
        oracle_lib.OracleDOCP("goddard", "midpoint", time_grid=[0.0, 0.5, 0.5, 1.0])
    with pytest.raises(RuntimeError):
        oracle_lib.OracleDOCP("goddard", 99, 10)
    # non-normalised grids are normalised (:191-196)
    d = oracle_lib.OracleDOCP("double_integrator_path", "midpoint", time_grid=[1.0, 2.0, 4.0, 5.0])
    nrm, fixed = d.grids()
    assert np.allclose(nrm, [0.0, 0.25, 0.75, 1.0]) and np.allclose(fixed, [0.0, 0.5, 1.5, 2.0])


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: p.split("/")[-1][:-5])
def test_oracle_matches_mpmath_fixture(oracle_lib, path):
    g = load_golden(path)
    d = oracle_lib.OracleDOCP(g["problem"], g["scheme"], g["grid_size"], time_grid=g["time_grid"], control_steps=g.get("control_steps", 1))
    assert d.dim_NLP_variables == g["dims"]["nvar"] and d.dim_NLP_constraints == g["dims"]["ncon"]
    assert d.step_variables_block == g["dims"]["step_variables_block"]
    xu = g["xu"]
    c = d.constraints(xu)
    assert not np.any(c == 666.666)                       # every row written (sentinel, test/benchmark.jl:106,126)
    assert relerr(c, g["c"]) <= TOL
    assert abs(d.objective(xu) - g["objective"]) / max(1.0, abs(g["objective"])) <= TOL
    assert relerr(d.gradient(xu), g["gradient"]) <= TOL
    Jd = d.jac_dense(xu)
    assert relerr(Jd, g["J"]) <= TOL
    # STRUCTURAL pattern covers every true nonzero; coloured values on it equal the exact derivative
    d.set_pattern_mode(1)
    colptr, rowval = d.jac_pattern()
    pat = csc_to_set(colptr, rowval)
    true_nz = {(int(r), int(cc)) for r, cc, _ in g["jac_nonzeros"]}
    # implicit Euler evaluates the path constraints of node i with u(t_i) = U_{i-1} (euler.jl:59-72) while its pattern lists
    # (path_i, U_i) (euler.jl:231): those true nonzeros are outside the reference's pattern in either mode
    blk, cb, n, m = d.step_variables_block, d.state_stage_eqs_block + d.path_cons, d.n, d.m
    euler_shift = set()
    if g["scheme"] == "euler_implicit":
        euler_shift = {(r, cc) for r, cc in true_nz if r < d.steps * cb and r % cb >= d.state_stage_eqs_block and r // cb >= 1
                       and (r // cb - 1) * blk + n <= cc < (r // cb - 1) * blk + n + m}
    assert true_nz - euler_shift <= pat
    if not euler_shift:
        assert relerr(d.jac_coord(xu), dense_on_pattern(g["J"], colptr, rowval)) <= TOL
    # REFERENCE_MANUAL pattern: only trapeze with free times drops true nonzeros (dyn rows x v, hazard H1)
    d.set_pattern_mode(0)
    colptr0, rowval0 = d.jac_pattern()
    missing = true_nz - csc_to_set(colptr0, rowval0)
    if g["scheme"] == "trapeze" and (d.freet0 or d.freetf or g["problem"] == "estimate_rotation_rate"):
        nv0 = d.dim_NLP_variables - d.nv
        assert all(cc >= nv0 for _, cc in missing)
    else:
        assert missing == euler_shift


@pytest.mark.parametrize("path", hess_golden_files(), ids=lambda p: p.split("/")[-1][:-5])
def test_oracle_hessian_matches_mpmath_fixture(oracle_lib, path):
    """hess_coord! of the oracle (sparse second-order sweep in double) against the 50-digit Hessian of the Lagrangian,
    on the lower triangle of DOCP_Hessian_pattern; the reference's pattern holds every true nonzero of these cases."""
    g = load_hess_golden(path)
    d = oracle_lib.OracleDOCP(g["problem"], g["scheme"], g["grid_size"], time_grid=g["time_grid"], control_steps=g.get("control_steps", 1))
    if os.path.exists(path.replace("hess_", "")):
        first = load_golden(path.replace("hess_", ""))
        assert np.array_equal(first["xu"], g["xu"])       # same input vector as the first-order fixture of the tag
    colptr, rowval = d.hess_pattern()
    assert len(rowval) == d.hess_nnz()[1]
    for j in range(d.dim_NLP_variables):                  # lower triangle, rows sorted
        rows = rowval[colptr[j]:colptr[j + 1]]
        assert np.all(rows >= j) and np.all(np.diff(rows) > 0)
    want, outside = hess_on_pattern(g["H"], colptr, rowval)
    vals, dropped = d.hess_coord(g["xu"], g["y"], g["obj_weight"], return_dropped=True)
    if g["scheme"].startswith("euler"):
        # DOCP_Hessian_pattern of the Euler schemes (euler.jl:270-355) holds no (xf, xf) / (xf, v) entries for the explicit
        # variant and no (x_i, u_{i-1}) coupling for the implicit one: true nonzeros there are lost by the reference too
        assert dropped[1] == len(outside)
    else:
        assert not outside and dropped == (0, 0)
    assert relerr(vals, want) <= TOL
    # linear in (obj_weight, y): obj_weight = 0 and y = 0 separate the two parts
    a = d.hess_coord(g["xu"], 0 * g["y"], g["obj_weight"])
    b = d.hess_coord(g["xu"], g["y"], 0.0)
    assert relerr(a + b, want) <= TOL


def test_G2_hessian_pattern_nnzh(oracle_lib):
    """nnzh = 6519 for Goddard / midpoint / N = 250 (test/archives/AD_backend.md:86)."""
    d = oracle_lib.OracleDOCP("goddard", "midpoint", 250)
    colptr, rowval = d.hess_pattern()
    assert len(rowval) == 6519 and colptr[-1] == 6519


@pytest.mark.parametrize("problem, scheme, N", [("goddard", "gauss_legendre_2", 200), ("goddard_all", "trapeze", 64),
                                                ("quadrotor", "gauss_legendre_3", 9)])
def test_threaded_coloured_jacobian_equals_serial(oracle_lib, problem, scheme, N):
    """the multi-core CPU baseline of bench.py runs the same coloured passes: bit-identical values for any thread count"""
    o = oracle_lib.OracleDOCP(problem, scheme, N)
    x = 0.1 + 0.05 * np.sin(1.3 * np.arange(o.dim_NLP_variables))
    want = o.jac_coord(x)
    for nt in (1, 3, 64):
        assert np.array_equal(o.jac_coord_mt(x, nt), want)


@pytest.mark.parametrize("prob", ["goddard", "goddard_all", "double_integrator_path", "quadrotor", "quadrotor12",
                                  "double_integrator_freet0tf", "least_squares_with_constraint"])
def test_block_mode_equals_coloured_jacobian(oracle_lib, prob):
    """The oracle's block mode (one time step at a time on dense local duals; the checker of the full-size Jacobians and
    the best-effort CPU baseline) gives the constraints and, at every pattern position, the exact partial derivative: equal
    to the coloured passes where the pattern is complete, to dense Jacobian columns where it is not (hazard H1)."""
    from helpers import bench_inputs, describe
    for sch in ("trapeze", "midpoint", "euler", "gauss_legendre_2", "gauss_legendre_3", "gauss_legendre_2_constant_control"):
        for mode in (0, 1):
            for N in (1, 2, 7):
                o = oracle_lib.OracleDOCP(prob, sch, N)
                o.set_pattern_mode(mode)
                x = bench_inputs(describe(o, prob, sch), perturb=1e-3)
                c, v = o.cons_jac_block(x, 3)
                assert np.array_equal(c, o.constraints(x))
                ref = o.jac_coord(x)
                if not np.array_equal(v, ref):
                    J = o.jac_dense(x)
                    cp, rv = o.jac_pattern()
                    ref = np.array([J[rv[k], j] for j in range(len(cp) - 1) for k in range(cp[j], cp[j + 1])])
                assert np.array_equal(v, ref), (prob, sch, mode, N)
    oi = oracle_lib.OracleDOCP(prob, "euler_implicit", 3)      # its path rows read the previous step's control: no block mode
    assert oi.cons_jac_block(np.full(oi.dim_NLP_variables, 0.3), 1) is None


def test_G2_optimized_backend_counts(oracle_lib):
    """The `:optimized` half of the reference's CI golden: Goddard, midpoint, N = 250 -> nnzj 4504, nnzh 5259
    (test/ci/test_modeler_solver.jl:32: ADNLPModels detects the sparsity itself with its operator-overloading tracer).  The
    oracle restates that detection with the sparse second-order number it pushes through `constraints` / `objective`
    (pattern mode 2); goddard_all, trapeze, N = 1000 gives the archived Hessian count 11011 (test/archives/AD_backend.md:61)."""
    o = oracle_lib.OracleDOCP("goddard", "midpoint", 250)
    o.set_pattern_mode(2)
    assert o.jac_nnz() == 4504
    assert len(o.hess_pattern()[1]) == 5259
    o2 = oracle_lib.OracleDOCP("goddard_all", "trapeze", 1000)
    o2.set_pattern_mode(2)
    assert len(o2.hess_pattern()[1]) == 11011
    # the traced pattern holds every true nonzero: coloured values on it == dense Jacobian
    o3 = oracle_lib.OracleDOCP("goddard_all", "trapeze", 6)
    o3.set_pattern_mode(2)
    from helpers import bench_inputs, describe
    x = bench_inputs(describe(o3, "goddard_all", "trapeze"), perturb=1e-3)
    J = o3.jac_dense(x)
    cp, rv = o3.jac_pattern()
    mask = np.zeros_like(J, dtype=bool)
    for j in range(len(cp) - 1):
        mask[rv[cp[j]:cp[j + 1]], j] = True
    assert not np.any(J[~mask] != 0.0)
    assert np.array_equal(o3.jac_coord(x), dense_on_pattern(J, cp, rv))


@pytest.mark.parametrize("prob", ["goddard", "goddard_all", "double_integrator_path", "quadrotor", "least_squares_with_constraint",
                                  "double_integrator_freet0tf"])
def test_block_mode_equals_full_hessian_sweep(oracle_lib, prob):
    """The oracle's block mode of hess_coord (one time step at a time, the checker of the full-size Hessians) against its full
    sparse second-order sweep, all three pattern modes; the same entries are dropped by the reference's incomplete patterns."""
    from helpers import bench_inputs, describe, relerr
    for sch in ("trapeze", "midpoint", "euler", "gauss_legendre_2", "gauss_legendre_3_constant_control"):
        for mode in (0, 1, 2):
            for N in (1, 2, 7):
                o = oracle_lib.OracleDOCP(prob, sch, N)
                o.set_pattern_mode(mode)
                x = bench_inputs(describe(o, prob, sch), perturb=1e-2)
                y = np.cos(0.3 * np.arange(o.dim_NLP_constraints))
                a, da = o.hess_coord(x, y, 0.7, return_dropped=True)
                b, db = o.hess_coord_block(x, y, 0.7, 3, return_dropped=True)
                assert relerr(b, a) <= 1e-12 and da[1] == db[1], (prob, sch, mode, N)
    oi = oracle_lib.OracleDOCP(prob, "euler_implicit", 3)
    assert oi.hess_coord_block(np.full(oi.dim_NLP_variables, 0.3), np.ones(oi.dim_NLP_constraints), 1.0, 1) is None


# ---- reference-held sparsity counts of the `:optimized` backend that need another FORM of a problem, or a problem the registry
# does not hold (VERDICT r03 "missing" 2): the oracle restates them as oracle-only problems (ids 10, 11), the engine takes the same
# problems as run-time OCPs (tests/problem_folder_defs.py); both must reproduce what the reference's archives publish.
def _engine_counts(name, scheme, N, pattern):
    import problem_folder_defs as pf
    import ctdirect_jl_amd as ct
    rt, _, _ = pf.folder(name)
    d = ct.DOCP(rt, N, scheme, device=-1, pattern=pattern)
    return d.dim_NLP_variables, d.dim_NLP_constraints, d.nnzj, d.nnzh


def test_goddard_all_archived_optimized_counts_28011(oracle_lib):
    """test/archives/AD_backend.md:56-64: goddard_all, trapeze, `:optimized`: Jac nnz 28011 (N = 1000) / 280011 (N = 10000), Hess nnz
    11011 / 110011, vars 4005 / 40005, cons 6007 / 60007.  28 entries per step is the traced pattern of the dynamics written
    F0(x) + u F1(x) (test/problems/goddard.jl:7-15,44): `u * 0` in the r-row counts as a dependence on U_i and U_{i+1}.  The f! of the
    CURRENT goddard_all (goddard.jl:127-132: r[1] = x[2]) has no such term: 26 per step = 26011, which is what both the oracle and the
    engine give for the registry problem -- the archive predates that f!.  Both forms are pinned here; the Hessian count is the same
    for both (u * 0 is linear)."""
    for N, nnzj, nnzh, nvar, ncon in ((1000, 28011, 11011, 4005, 6007), (10000, 280011, 110011, 40005, 60007)):
        o = oracle_lib.OracleDOCP("goddard_all_f0f1", "trapeze", N)
        o.set_pattern_mode(2)
        assert (o.dim_NLP_variables, o.dim_NLP_constraints) == (nvar, ncon)
        assert o.jac_nnz() == nnzj
        if N == 1000:
            assert len(o.hess_pattern()[1]) == nnzh
        assert _engine_counts("goddard_all_f0f1", "trapeze", N, "optimized") == (nvar, ncon, nnzj, nnzh)
    # the current file's form: 26 per step, on both sides
    o = oracle_lib.OracleDOCP("goddard_all", "trapeze", 1000)
    o.set_pattern_mode(2)
    assert o.jac_nnz() == 26011
    import ctdirect_jl_amd as ct
    d = ct.DOCP("goddard_all", 1000, "trapeze", device=-1, pattern="optimized")
    assert (d.nnzj, d.nnzh) == (26011, 11011)
    # the `:manual` column of the same table (42043 / 420043, 30024 / 300024) is the STRUCTURAL pattern (hazard H1: the current
    # trapeze.jl:203 gives 39043)
    ds = ct.DOCP("goddard_all", 1000, "trapeze", device=-1, pattern="structural")
    dm = ct.DOCP("goddard_all", 1000, "trapeze", device=-1, pattern="manual")
    assert (ds.nnzj, dm.nnzj, dm.nnzh) == (42043, 39043, 30024)


def test_algal_bacterial_published_tables(oracle_lib):
    """test/archives/jump_ctdirect.md:41-67, the only per-configuration tables the reference publishes (columns "CT" = `:optimized`,
    "Manual" = `:manual`).  Trapeze 1000 / 5000: variables 8008 / 40008, constraints 6006 / 30006, nnz jacobian 42006 / 210006
    (optimized) and 96076 / 480072 (manual: the archive's code listed the dynamics rows x U_{i+1} block twice less... see below),
    nnz hessian 12012 / 60012 (optimized).  Gauss-Legendre 2 with a piecewise constant control (the archive's GL2) 1000 / 5000:
    constraints 18006 / 90006, nnz jacobian 118006 / 590006, nnz hessian 63000 / 315000 (optimized)."""
    for N, nvar, ncon, nnzj, nnzh in ((1000, 8008, 6006, 42006, 12012), (5000, 40008, 30006, 210006, 60012)):
        o = oracle_lib.OracleDOCP("algal_bacterial", "trapeze", N)
        o.set_pattern_mode(2)
        assert (o.dim_NLP_variables, o.dim_NLP_constraints, o.jac_nnz()) == (nvar, ncon, nnzj)
        if N == 1000:
            assert len(o.hess_pattern()[1]) == nnzh
        assert _engine_counts("algal_bacterial", "trapeze", N, "optimized") == (nvar, ncon, nnzj, nnzh)
    for N, ncon, nnzj, nnzh in ((1000, 18006, 118006, 63000), (5000, 90006, 590006, 315000)):
        o = oracle_lib.OracleDOCP("algal_bacterial", "gauss_legendre_2_constant_control", N)
        o.set_pattern_mode(2)
        assert (o.dim_NLP_constraints, o.jac_nnz()) == (ncon, nnzj)
        if N == 1000:
            assert len(o.hess_pattern()[1]) == nnzh
        nv_, nc_, nj_, nh_ = _engine_counts("algal_bacterial", "gauss_legendre_2_constant_control", N, "optimized")
        assert (nc_, nj_, nh_) == (ncon, nnzj, nnzh)
        # variables: the archive says 20008 / 100008 -- its IRK layout still carried a final control U_{N+1} (as trapeze does);
        # the current irk.jl:138-160 has none: N (n + m + s n) + n = 20006 / 100006
        assert nv_ == N * 20 + 6
    # the "Manual" columns (`:manual` = DOCP_Jacobian_pattern / DOCP_Hessian_pattern as written), oracle and engine:
    #   trapeze   nnzj 96076 / 480072 in the archive: 96 per step + 72 = 96072 / 480072 -- the archive's own 5000-step figure fits the
    #             formula, its 1000-step figure is 4 off (a typo: the same code cannot give +76 at one size and +72 at another);
    #             nnzh 100072 / 500072 reproduced
    #   GL2 (cc)  nnzj 384072 / 1920072 reproduced; nnzh 210072 / 1050072 in the archive, 210069 / 1050069 here: the 3 = m (m + 1) / 2
    #             entries of the final-control block of the archive's layout (the same U_{N+1} that makes its variable count 20008)
    for sch, N, nnzj, nnzh in (("trapeze", 1000, 96072, 100072), ("trapeze", 5000, 480072, 500072),
                               ("gauss_legendre_2_constant_control", 1000, 384072, 210072 - 3),
                               ("gauss_legendre_2_constant_control", 5000, 1920072, 1050072 - 3)):
        _, _, nj_, nh_ = _engine_counts("algal_bacterial", sch, N, "manual")
        assert (nj_, nh_) == (nnzj, nnzh), (sch, N, nj_, nh_)
        if N == 1000:
            o = oracle_lib.OracleDOCP("algal_bacterial", sch, N)
            o.set_pattern_mode(0)
            assert (o.jac_nnz(), len(o.hess_pattern()[1])) == (nnzj, nnzh)
