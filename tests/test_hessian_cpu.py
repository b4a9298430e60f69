"""CPU tests of the Hessian-of-the-Lagrangian row: the pattern bookkeeping of the C-ABI library and the
kernel logic (ctdirect.jl_amd/csrc/ctd_hess_body.hpp compiled with g++ and stepped serially, tests/emu/ -- test
infrastructure only) against the oracle and the 50-digit mpmath fixtures.  The HIP build of the same code is checked on
hardware by tests/test_gpu_hessian.py."""
import numpy as np
import pytest

import ctdirect_jl_amd as ct
from emu import emu
from helpers import TOL, bench_inputs, describe, hess_golden_files, hess_on_pattern, load_hess_golden, relerr

PAIRS = [(p, s) for p in ct.PROBLEMS for s in ct.SCHEMES]


def test_reference_nnzh_through_the_abi():
    # nnzh 6519: test/archives/AD_backend.md:86 (Goddard, midpoint, 250 steps)
    d = ct.DOCP("goddard", device=-1)
    assert d.nnzh == 6519
    rows, cols = d.hess_structure()
    assert len(rows) == 6519 and np.all(rows >= cols) and rows.min() == 1 and rows.max() == d.dim_NLP_variables
    # goddard_all trapeze: nnzh 30024 / 300024 (test/archives/AD_backend.md:63)
    assert ct.DOCP("goddard_all", 1000, "trapeze", device=-1).nnzh == 30024
    assert ct.DOCP("goddard_all", 10000, "trapeze", device=-1).nnzh == 300024


def test_hessian_on_host_only_handle_fails_loudly():
    d = ct.DOCP("goddard", 10, "midpoint", device=-1)
    with pytest.raises(ct.CTDirectError) as e:
        d.hess_coord(np.full(d.dim_NLP_variables, 0.1), np.zeros(d.dim_NLP_constraints))
    assert e.value.status == ct._lib.CTD_ENODEVICE


@pytest.mark.parametrize("prob,cs", [("goddard", 2), ("quadrotor", 3), ("quadrotor", 5), ("double_integrator_freet0tf", 5)])
def test_emulated_hessian_kernel_with_several_controls_per_step(oracle_lib, prob, cs):
    """control_steps > 1 (midpoint): one stage-type point per control of the step; 5 controls: the points of a step summed
    before the emission (hess_sums_stages).  Edge-only sizes, one and several tiles, ragged grid, the three patterns."""
    rng = np.random.default_rng(11)
    pid, sid = ct.PROBLEMS[prob], ct.SCHEMES["midpoint"]
    for N, tg, tile, nthr in ((1, None, 0, 64), (4, None, 2, 5), (7, None, 3, 64), (9, np.cumsum(rng.uniform(0.5, 1.5, 10)), 0, 96)):
        o = oracle_lib.OracleDOCP(prob, "midpoint", N, time_grid=tg, control_steps=cs)
        x = o.initial_guess() + 0.05 * rng.standard_normal(o.dim_NLP_variables)
        y = rng.standard_normal(o.dim_NLP_constraints)
        for mode in (0, 1, 2):
            o.set_pattern_mode(mode)
            want, dropped = o.hess_coord(x, y, 0.7, return_dropped=True)
            assert dropped == (0, 0)
            with emu.control_steps(cs):
                cp, rv = emu.hess_csc(pid, sid, mode, N, tg)
                vals = emu.hess(pid, sid, mode, N, x, y, 0.7, tg, tile=tile, nthr=nthr)
            ocp, orv = o.hess_pattern()
            assert np.array_equal(cp, ocp) and np.array_equal(rv, orv)
            assert not np.any(vals == 666.666) and relerr(vals, want) <= TOL, (N, mode)


@pytest.mark.parametrize("prob,sch", PAIRS, ids=[f"{p}-{s}" for p, s in PAIRS])
def test_hessian_pattern_matches_oracle(oracle_lib, prob, sch):
    """lower triangle of DOCP_Hessian_pattern: library (periodic column starts + explicit edges) == oracle (literal
    add_nonzero_block! pushes + SparseArrays.sparse), bit-exact, incl. all-edge sizes N < 5"""
    rng = np.random.default_rng(5)
    for N, tg in ((1, None), (2, None), (4, None), (5, None), (6, None), (41, None), (9, np.cumsum(rng.uniform(0.5, 1.5, 10)))):
        o = oracle_lib.OracleDOCP(prob, sch, N, time_grid=tg)
        cp, rv = o.hess_pattern()
        d = ct.DOCP(prob, N, sch, time_grid=tg, device=-1)
        cp2, rv2 = ct.DOCP_Hessian_pattern(d)
        assert d.nnzh == len(rv) and np.array_equal(cp, cp2) and np.array_equal(rv, rv2)
        rows, cols = d.hess_structure()
        assert np.array_equal(rows - 1, rv) and np.array_equal(np.repeat(np.arange(len(cp) - 1), np.diff(cp)), cols - 1)
        cp3, rv3 = emu.hess_csc(ct.PROBLEMS[prob], ct.SCHEMES[sch], 0, N, tg)     # also checks hess_column_start
        assert np.array_equal(cp, cp3) and np.array_equal(rv, rv3)


def test_structural_hessian_pattern_adds_final_state_x_variable():
    """STRUCTURAL mode: the (xf, v) block whose add_nonzero_block! call has an empty range in the reference
    (irk.jl:483, irk_stagewise.jl:625); nothing else changes"""
    for sch in ("gauss_legendre_2", "gauss_legendre_3_constant_control", "midpoint", "trapeze"):
        a = ct.DOCP("goddard", 12, sch, device=-1)
        b = ct.DOCP("goddard", 12, sch, pattern="structural", device=-1)
        extra = 3 * 1 if sch.startswith("gauss") else 0
        assert b.nnzh == a.nnzh + extra
        ra, ca = a.hess_structure()
        rb, cb = b.hess_structure()
        sa, sb = set(zip(ra, ca)), set(zip(rb, cb))
        assert sa <= sb
        if extra:
            nv0 = a.dim_NLP_variables
            blk = a.discretization._step_variables_block
            assert sb - sa == {(nv0, 12 * blk + k + 1) for k in range(3)}


@pytest.mark.parametrize("path", hess_golden_files(), ids=lambda p: p.split("/")[-1][:-5])
def test_emulated_hessian_kernel_matches_fixture(path):
    g = load_hess_golden(path)
    pid, sid = ct.PROBLEMS[g["problem"]], ct.SCHEMES[g["scheme"]]
    with emu.control_steps(g.get("control_steps", 1)):
        for mode in (0, 2) if "control_steps" in g else (0,):    # (several controls per step: also on the optimized pattern)
            cp, rv = emu.hess_csc(pid, sid, mode, g["grid_size"], g["time_grid"])
            want, outside = hess_on_pattern(g["H"], cp, rv)
            # (the Euler patterns of the reference, euler.jl:270-355, leave some true nonzeros out: test_oracle_goldens.py)
            assert not outside or g["scheme"].startswith("euler")
            for tile, nthr in ((0, 64), (1, 3), (3, 17)):
                vals = emu.hess(pid, sid, mode, g["grid_size"], g["xu"], g["y"], g["obj_weight"], g["time_grid"], tile=tile, nthr=nthr)
                assert not np.any(vals == 666.666)               # every entry of the pattern written
                assert relerr(vals, want) <= TOL


@pytest.mark.parametrize("prob,sch", PAIRS, ids=[f"{p}-{s}" for p, s in PAIRS])
def test_emulated_hessian_kernel_matches_oracle_tiled(oracle_lib, prob, sch):
    rng = np.random.default_rng(7)
    pid, sid = ct.PROBLEMS[prob], ct.SCHEMES[sch]
    for N, tg, tile, nthr in ((1, None, 0, 64), (3, None, 2, 5), (5, None, 0, 64), (6, None, 4, 64), (37, None, 8, 96),
                              (13, np.cumsum(rng.uniform(0.5, 1.5, 14)), 5, 32)):
        o = oracle_lib.OracleDOCP(prob, sch, N, time_grid=tg)
        x = bench_inputs(describe(o, prob, sch), perturb=1e-2)
        y = rng.standard_normal(o.dim_NLP_constraints) * rng.choice([1e-2, 1.0, 10.0], o.dim_NLP_constraints)
        for sigma in (1.0, -0.3):
            want, dropped = o.hess_coord(x, y, sigma, return_dropped=True)
            assert dropped == (0, 0) or sch.startswith("euler")       # euler.jl:270-355 leaves true nonzeros out
            vals = emu.hess(pid, sid, 0, N, x, y, sigma, tg, tile=tile, nthr=nthr)
            assert not np.any(vals == 666.666)
            assert relerr(vals, want) <= TOL
        # the OPTIMIZED pattern (pattern mode 2): same values at its (fewer) positions, nothing of the exact Hessian left out
        try:
            ct.DOCP(prob, N, sch, time_grid=tg, pattern="optimized", device=-1)
        except ct.CTDirectError:
            assert sch == "euler_implicit"
            continue
        o.set_pattern_mode(2)
        want, dropped = o.hess_coord(x, y, 0.7, return_dropped=True)
        assert dropped == (0, 0)
        vals = emu.hess(pid, sid, 2, N, x, y, 0.7, tg, tile=tile, nthr=nthr)
        assert vals.size == want.size and not np.any(vals == 666.666)
        assert relerr(vals, want) <= TOL
    # linearity in (obj_weight, y) and zero multipliers: H(0, 0) = 0
    o = oracle_lib.OracleDOCP(prob, sch, 7)
    x = bench_inputs(describe(o, prob, sch), perturb=1e-2)
    assert not np.any(emu.hess(pid, sid, 0, 7, x, np.zeros(o.dim_NLP_constraints), 0.0))


@pytest.mark.parametrize("compact", ["0", "1", "2"])
@pytest.mark.parametrize("edge_blocks", ["1", "3"])
def test_emulated_hessian_walk_modes(oracle_lib, monkeypatch, compact, edge_blocks):
    """The tiles' walk over the step-periodic segment: all entries (0), entries with terms after a zero fill of the tile's
    part of vals (1), entries with terms + explicit zero stores (2) -- and the edge entries shared by several edge
    workgroups: same values, every entry written."""
    monkeypatch.setenv("CTD_HESS_COMPACT", compact)
    monkeypatch.setenv("CTD_HESS_EDGE_BLOCKS", edge_blocks)
    rng = np.random.default_rng(21)
    for prob, sch in (("goddard", "gauss_legendre_3"), ("quadrotor", "gauss_legendre_2"), ("goddard_all", "midpoint"),
                      ("double_integrator_freet0tf", "gauss_legendre_3")):
        pid, sid = ct.PROBLEMS[prob], ct.SCHEMES[sch]
        for N, tile, nthr in ((19, 4, 256), (9, 3, 64)):
            o = oracle_lib.OracleDOCP(prob, sch, N)
            x = bench_inputs(describe(o, prob, sch), perturb=1e-2)
            y = rng.standard_normal(o.dim_NLP_constraints)
            vals = emu.hess(pid, sid, 0, N, x, y, 0.9, None, tile=tile, nthr=nthr)
            assert not np.any(vals == 666.666)
            assert relerr(vals, o.hess_coord(x, y, 0.9)) <= TOL


@pytest.mark.parametrize("prob", ["goddard", "goddard_all", "quadrotor", "double_integrator_freet0tf", "least_squares_with_constraint"])
def test_emulated_hessian_shards_compose(prob):
    """time-step shards (multi-GPU): every entry outside the V x V block is written by exactly one shard with the value of
    the full evaluation, and the V x V entries of the shards add up to the full ones"""
    N = 23
    rng = np.random.default_rng(9)
    for sch in ct.SCHEMES:
        d = ct.DOCP(prob, N, sch, device=-1)
        x = bench_inputs(describe(d, prob, sch), perturb=1e-2)
        y = rng.standard_normal(d.dim_NLP_constraints)
        pid, sid = ct.PROBLEMS[prob], ct.SCHEMES[sch]
        full = emu.hess(pid, sid, 0, N, x, y, 0.8, tile=4, nthr=64)
        _, _, vv = d.hess_shard_info()
        notvv = np.ones(d.nnzh, dtype=bool)
        notvv[vv] = False
        for cuts in ([0, 11, 23], [0, 1, 22, 23], [0, 7, 8, 23]):
            acc = np.full(d.nnzh, 666.666)
            vvsum = np.zeros(len(vv))
            for a, b in zip(cuts[:-1], cuts[1:]):
                part = np.full(d.nnzh, 666.666)
                emu.hess(pid, sid, 0, N, x, y, 0.8, tile=3, nthr=32, step_begin=a, step_end=b, vals=part)
                wrote = (part != 666.666) & notvv
                assert not np.any(wrote & (acc != 666.666))              # disjoint outside the V x V block
                acc[wrote] = part[wrote]
                sh = ct.DOCP(prob, N, sch, device=-1, steps=(a, b))
                lo, hi, _ = sh.hess_shard_info()
                assert np.all(part[lo:hi] != 666.666)                    # the shard's contiguous CSC range is complete
                vvsum += part[vv]
            assert np.array_equal(acc[notvv], full[notvv])
            assert relerr(vvsum, full[vv]) <= 1e-13
