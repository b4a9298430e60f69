"""CPU tests of the KERNEL LOGIC: the engine's phase functions (ctdirect.jl_amd/csrc/ctd_kernel_body.hpp) are compiled
with g++ and stepped serially (tests/emu/, test infrastructure only) and must reproduce the mpmath fixtures and the
oracle.  This exercises the emit tables, tile/edge indexing and the chain-rule algebra without a GPU; the HIP build of
the same code is checked on hardware by tests/test_gpu_parity.py."""
import numpy as np
import pytest

import ctdirect_jl_amd as ct
from emu import emu
from helpers import TOL, bench_inputs, dense_on_pattern, golden_files, load_golden, relerr, describe


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: p.split("/")[-1][:-5])
def test_emulated_kernel_matches_fixture(oracle_lib, path):
    g = load_golden(path)
    pid, sid = ct.PROBLEMS[g["problem"]], ct.SCHEMES[g["scheme"]]
    cs = g.get("control_steps", 1)
    o = oracle_lib.OracleDOCP(g["problem"], g["scheme"], g["grid_size"], time_grid=g["time_grid"], control_steps=cs)
    with emu.control_steps(cs):
        for mode in (0, 1, 2) if cs > 1 else (0, 1):
            o.set_pattern_mode(mode)
            cp, rv = o.jac_pattern()
            cp2, rv2 = emu.csc(pid, sid, mode, g["grid_size"], g["time_grid"])
            assert np.array_equal(cp, cp2) and np.array_equal(rv, rv2)
            ref = dense_on_pattern(g["J"], cp, rv)
            for tile, nthr in ((0, 64), (1, 3), (3, 17)):
                c, vals = emu.cons_jac(pid, sid, mode, g["grid_size"], g["xu"], g["time_grid"], tile=tile, nthr=nthr)
                assert relerr(c, g["c"]) <= TOL
                assert relerr(vals, ref) <= TOL


PAIRS = [(p, s) for p in ct.PROBLEMS for s in ct.SCHEMES]


@pytest.mark.parametrize("prob,sch", PAIRS, ids=[f"{p}-{s}" for p, s in PAIRS])
def test_emulated_kernel_matches_oracle_tiled(oracle_lib, prob, sch):
    rng = np.random.default_rng(3)
    for N, tg in ((5, None), (6, None), (37, None), (13, np.cumsum(rng.uniform(0.5, 1.5, 14)))):
        o = oracle_lib.OracleDOCP(prob, sch, N, time_grid=tg)
        x = bench_inputs(describe(o, prob, sch), perturb=1e-3)
        cref = o.constraints(x)
        Jd = o.jac_dense(x) if N <= 13 else None
        for mode in (0, 1, 2):
            if mode == 2:      # OPTIMIZED: refused (never silently wrong) for implicit Euler with path constraints and controls
                try:
                    ct.DOCP(prob, N, sch, time_grid=tg, pattern="optimized", device=-1)
                except ct.CTDirectError:
                    assert sch == "euler_implicit"
                    continue
            o.set_pattern_mode(mode)
            cp, rv = o.jac_pattern()
            if Jd is not None:
                vref = dense_on_pattern(Jd, cp, rv)
            elif mode >= 1:
                vref = o.jac_coord(x)
            else:
                continue
            cp2, rv2 = emu.csc(ct.PROBLEMS[prob], ct.SCHEMES[sch], mode, N, tg)
            assert np.array_equal(cp, cp2) and np.array_equal(rv, rv2)
            for tile, nthr in ((0, 64), (1, 5), (4, 33), (7, 256)):
                c, vals = emu.cons_jac(ct.PROBLEMS[prob], ct.SCHEMES[sch], mode, N, x, tg, tile=tile, nthr=nthr)
                assert not np.any(c == 666.666) and not np.any(vals == 666.666)      # every output written
                assert relerr(c, cref) <= TOL and relerr(vals, vref) <= TOL


@pytest.mark.parametrize("prob", ["goddard", "goddard_all", "quadrotor", "double_integrator_freet0tf"])
def test_emulated_shards_compose_exactly(prob):
    N = 23
    for sch in ct.SCHEMES:
        d = ct.DOCP(prob, N, sch, device=-1)
        x = bench_inputs(describe(d, prob, sch), perturb=1e-3)
        pid, sid = ct.PROBLEMS[prob], ct.SCHEMES[sch]
        cf, vf = emu.cons_jac(pid, sid, 0, N, x, tile=4, nthr=64)
        for cuts in ([0, 11, 23], [0, 1, 22, 23]):
            c = np.full_like(cf, 666.666)
            v = np.full_like(vf, 666.666)
            for a, b in zip(cuts[:-1], cuts[1:]):
                c2 = np.full_like(cf, 666.666)
                v2 = np.full_like(vf, 666.666)
                emu.cons_jac(pid, sid, 0, N, x, tile=3, nthr=32, step_begin=a, step_end=b, c=c2, vals=v2)
                both = (c != 666.666) & (c2 != 666.666)                  # shards write disjoint step rows / CSC ranges;
                ncb = N * (d.discretization._state_stage_eqs_block + d.discretization._step_pathcons_block)
                assert not np.any(both[:ncb])                            # only the tail rows (final path + boundary) are
                assert np.array_equal(c[both], c2[both])                 # computed by every shard, with identical values
                assert np.array_equal(c2[ncb:], cf[ncb:])                # ... by EVERY shard (single all-gather stitching)
                assert not np.any((v != 666.666) & (v2 != 666.666))
                c = np.where(c2 != 666.666, c2, c)
                v = np.where(v2 != 666.666, v2, v)
            assert np.array_equal(c, cf) and np.array_equal(v, vf)


@pytest.mark.parametrize("prob", ["goddard", "goddard_all", "quadrotor", "double_integrator_freet0tf", "double_integrator_path"])
def test_emulated_sharded_iterate_read_in_place(prob):
    """ctd_set_x_shards: every shard evaluates from a buffer that holds ONLY its own variables (NaN elsewhere); the next
    shard's first node, the previous shard's last block, X_1 and X_{N+1} are read in place from the owners' buffers through
    the XHalo table.  Bit-identical to the unsharded evaluation for every scheme, 2 / 3 / 5 shards (also one-step shards and
    grids without a periodic part), both drivers."""
    for sch in ct.SCHEMES:
        for N, Gs in ((23, (2, 3, 5)), (4, (2, 4)), (7, (7,))):
            d = ct.DOCP(prob, N, sch, device=-1, pattern="structural")
            x = bench_inputs(describe(d, prob, sch), perturb=1e-3)
            pid, sid = ct.PROBLEMS[prob], ct.SCHEMES[sch]
            cf, vf = emu.cons_jac(pid, sid, 1, N, x, tile=4, nthr=64)
            for G in Gs:
                c, v = emu.cons_jac_sharded(pid, sid, 1, N, x, G, tile=3, nthr=64)
                assert np.array_equal(c, cf) and np.array_equal(v, vf), (sch, N, G)


@pytest.mark.parametrize("prob, sch", [("goddard", "gauss_legendre_3"), ("goddard_all", "trapeze"),
                                       ("double_integrator_freet0tf", "euler_implicit"),
                                       ("least_squares_with_constraint", "gauss_legendre_2_constant_control")])
def test_every_split_of_a_tiny_grid_composes(oracle_lib, prob, sch):
    """Grids of fewer than 5 steps have no step-periodic part: every step column is an explicit edge entry.  Each of them
    must still be written by exactly the shard that owns its step, for every way of cutting [0, N) into up to three
    shards -- constraints, Jacobian values and Hessian values (V x V entries: partial sums that add up)."""
    SENT = 777.25
    pid, sid = ct.PROBLEMS[prob], ct.SCHEMES[sch]
    for N in (2, 3, 4, 5):
        o = oracle_lib.OracleDOCP(prob, sch, N)
        x = 0.35 + 0.25 * np.random.default_rng(0).random(o.dim_NLP_variables)
        if o.nv:
            x[-o.nv:] = np.sort(x[-o.nv:])
        y = 0.5 + np.random.default_rng(1).random(o.dim_NLP_constraints)
        cref, vref, href = o.constraints(x), o.jac_coord(x), o.hess_coord(x, y, 0.8)
        cuts = [(0, a, N) for a in range(1, N)] + [(0, a, b, N) for a in range(1, N) for b in range(a + 1, N)]
        for cut in cuts:
            c, v = np.full(len(cref), SENT), np.full(len(vref), SENT)
            h, hw = np.zeros(len(href)), np.zeros(len(href), bool)
            for a, b in zip(cut[:-1], cut[1:]):
                emu.cons_jac(pid, sid, 0, N, x, step_begin=a, step_end=b, c=c, vals=v)
                hv = np.full(len(href), SENT)
                emu.hess(pid, sid, 0, N, x, y, 0.8, step_begin=a, step_end=b, vals=hv)
                d = ct.DOCP(prob, N, sch, device=-1, steps=(a, b))
                isvv = np.zeros(len(href), bool)
                isvv[np.array(d.hess_shard_info()[2], dtype=int)] = True
                d.close()
                w = hv != SENT
                assert not (hw & w & ~isvv).any(), (N, cut, "Hessian entry written by two shards")
                h[w & ~isvv] = hv[w & ~isvv]
                h[isvv] += hv[isvv]
                hw |= w
            assert hw.all() and not np.any(c == SENT) and not np.any(v == SENT), (N, cut)
            assert relerr(c, cref) <= TOL and relerr(v, vref) <= TOL and relerr(h, href) <= 1e-9, (N, cut)


def test_xcd_tile_order_is_a_pure_permutation(monkeypatch):
    """CTD_XCD=1 hands every XCD one contiguous run of tiles (ctd_layout.hpp xcd_tile): a bijection of the tile indices, so
    every output is still written once and the values do not change (the V x V Hessian entries: same terms, other order)"""
    pid, sid = ct.PROBLEMS["goddard_all"], ct.SCHEMES["gauss_legendre_2"]
    for N, tile in ((37, 2), (64, 3), (7, 1)):
        d = ct.DOCP("goddard_all", N, "gauss_legendre_2", device=-1)
        rng = np.random.default_rng(2)
        x = 0.4 + 0.2 * rng.random(d.dim_NLP_variables)
        y = rng.standard_normal(d.dim_NLP_constraints)
        monkeypatch.setenv("CTD_XCD", "0")
        c0, v0 = emu.cons_jac(pid, sid, 0, N, x, tile=tile)
        h0 = emu.hess(pid, sid, 0, N, x, y, 0.6, tile=tile)
        monkeypatch.setenv("CTD_XCD", "1")
        c1, v1 = emu.cons_jac(pid, sid, 0, N, x, tile=tile)
        h1 = emu.hess(pid, sid, 0, N, x, y, 0.6, tile=tile)
        assert np.array_equal(c0, c1) and np.array_equal(v0, v1)
        vv = np.zeros(len(h0), bool)
        vv[np.array(d.hess_shard_info()[2], dtype=int)] = True
        assert np.array_equal(h0[~vv], h1[~vv]) and np.allclose(h0[vv], h1[vv], rtol=1e-13, atol=0)
        d.close()


@pytest.mark.parametrize("prob", ["goddard_all", "quadrotor", "quadrotor12", "goddard", "double_integrator_path"])
def test_multi_tile_workgroups_write_the_same_outputs(monkeypatch, prob):
    """KParams::wg_stride (staged driver): a workgroup walks the blocks w, w + stride, ... on one LDS image -- templates, v and the
    lane's codes fetched once, the x slice of the next tile loaded before the current one is emitted (load_issue / load_commit).
    Bit-identical to one block per workgroup for every scheme and pattern, strides that leave full, ragged and single rounds,
    also on a sharded iterate (first / last tiles read through the XHalo table)."""
    N = 29
    pid = ct.PROBLEMS[prob]
    for sch in ct.SCHEMES:
        sid = ct.SCHEMES[sch]
        d = ct.DOCP(prob, N, sch, device=-1, pattern="structural")
        x = bench_inputs(describe(d, prob, sch), perturb=1e-3)
        d.close()
        for mode in (0, 1):
            monkeypatch.delenv("CTD_EMU_WG_STRIDE", raising=False)
            c0, v0 = emu.cons_jac(pid, sid, mode, N, x, tile=3, nthr=64)
            for stride, tile, nthr in ((1, 3, 64), (3, 3, 64), (4, 2, 33), (7, 1, 5), (50, 3, 64)):
                monkeypatch.setenv("CTD_EMU_WG_STRIDE", str(stride))
                c1, v1 = emu.cons_jac(pid, sid, mode, N, x, tile=tile, nthr=nthr)
                assert np.array_equal(c0, c1) and np.array_equal(v0, v1), (sch, mode, stride)
        monkeypatch.setenv("CTD_EMU_WG_STRIDE", "2")
        c2, v2 = emu.cons_jac_sharded(pid, sid, 1, N, x, 3, tile=2, nthr=64)
        monkeypatch.delenv("CTD_EMU_WG_STRIDE")
        c3, v3 = emu.cons_jac(pid, sid, 1, N, x, tile=3, nthr=64)
        assert np.array_equal(c2, c3) and np.array_equal(v2, v3), sch


@pytest.mark.parametrize("prob", ["quadrotor12", "quadrotor"])
def test_full_lane_tiles_of_the_wide_ocps(oracle_lib, prob):
    """Sparse eval blocks (DynNZ: the records hold the structural nonzeros of df/dx, df/du only) let a tile of a wide OCP hold as
    many steps as the evaluating waves have lanes: 16 steps on Gauss-Legendre 3 (48 stage points + 16 path points per wave), 21 on
    GL2, 24 - 31 on the one-point schemes; the lead tasks run behind the dynamics on the same lanes.  Against the oracle, all three
    patterns, 256- and 320-lane workgroups, grids that end in a short tile."""
    pid = ct.PROBLEMS[prob]
    for sch, tiles in (("gauss_legendre_3", (16,)), ("gauss_legendre_2", (21,)), ("gauss_legendre_2_constant_control", (16,)),
                       ("midpoint", (24, 31)), ("trapeze", (30,)), ("euler_implicit", (24,))):
        sid = ct.SCHEMES[sch]
        N = 37
        o = oracle_lib.OracleDOCP(prob, sch, N)
        x = bench_inputs(describe(o, prob, sch), perturb=1e-3)
        cref = o.constraints(x)
        for mode in (0, 1, 2):
            o.set_pattern_mode(mode)
            vref = o.jac_coord(x) if mode >= 1 else None
            for tile in tiles:
                for nthr in (256, 320):
                    c, vals = emu.cons_jac(pid, sid, mode, N, x, tile=tile, nthr=nthr)
                    assert not np.any(c == 666.666) and not np.any(vals == 666.666)
                    assert relerr(c, cref) <= TOL
                    if vref is not None:
                        assert relerr(vals, vref) <= TOL, (sch, mode, tile)
                    else:
                        c7, v7 = emu.cons_jac(pid, sid, mode, N, x, tile=7, nthr=256)
                        assert np.array_equal(vals, v7) and np.array_equal(c, c7)


def test_early_emission_knob_stays_bit_identical(monkeypatch):
    """CTD_EARLY=1 (off by default: measured slower, profiles/r03_experiments.md) lets the lead wave of a Gauss-Legendre tile store
    the outputs that only read its own records while the dynamics are still evaluated; the emit phase then walks the late positions
    only.  Emulated (CTD_EMU_EARLY): the same bits, all three patterns, also on the sparse eval blocks."""
    for prob in ("goddard", "double_integrator_freet0tf", "stagewise_scalar"):
        for sch in ("gauss_legendre_2", "gauss_legendre_3", "gauss_legendre_2_constant_control", "gauss_legendre_1"):
            d = ct.DOCP(prob, 41, sch, device=-1, pattern="structural")
            x = bench_inputs(describe(d, prob, sch), perturb=1e-3)
            d.close()
            for mode in (0, 1, 2):
                monkeypatch.delenv("CTD_EMU_EARLY", raising=False)
                c0, v0 = emu.cons_jac(ct.PROBLEMS[prob], ct.SCHEMES[sch], mode, 41, x, tile=8, nthr=320)
                monkeypatch.setenv("CTD_EMU_EARLY", "1")
                c1, v1 = emu.cons_jac(ct.PROBLEMS[prob], ct.SCHEMES[sch], mode, 41, x, tile=8, nthr=320)
                assert np.array_equal(c0, c1) and np.array_equal(v0, v1), (prob, sch, mode)
    monkeypatch.delenv("CTD_EMU_EARLY", raising=False)


@pytest.mark.parametrize("prob,sch,tile", [("goddard", "gauss_legendre_2", 111), ("goddard", "gauss_legendre_3", 84), ("double_integrator_path", "midpoint", 254),
                                           ("goddard", "midpoint", 199), ("goddard_all", "gauss_legendre_2", 85), ("goddard", "euler_implicit", 199),
                                           ("double_integrator_freet0tf", "gauss_legendre_3", 112)])
def test_emulated_long_grid_geometry(prob, sch, tile):
    """The launch geometry `ctd_create` takes on grids of 8 rounds of resident workgroups and more (round 4: eight waves per workgroup, the
    largest tile whose records fit 64 / 74 KiB -- the tile sizes of profiles/r04_tiles_long_grids.log), here in the serial emulator of the
    same phase templates: 512 lanes, 84 - 254 steps per tile, a grid of several tiles with a ragged last one, both value orders, against
    the oracle (GPU: tests/test_gpu_max_sizes.py)."""
    from oracle.oracle import OracleDOCP
    N = 2 * tile + 37
    o = OracleDOCP(prob, sch, N)
    o.set_pattern_mode(1)
    x = bench_inputs(describe(o, prob, sch), perturb=1e-3)
    cref, vref = o.constraints(x), o.jac_coord(x)
    pid, sid = ct.PROBLEMS[prob], ct.SCHEMES[sch]
    c, vals = emu.cons_jac(pid, sid, 1, N, x, tile=tile, nthr=512)
    assert not np.any(c == 666.666) and not np.any(vals == 666.666)
    assert relerr(c, cref) <= TOL and relerr(vals, vref) <= TOL
    with emu.value_order(1):
        c2, v2 = emu.cons_jac(pid, sid, 1, N, x, tile=tile, nthr=512)
    cp, rv = o.jac_pattern()
    cols = np.repeat(np.arange(len(cp) - 1), np.diff(cp))
    assert np.array_equal(c2, c) and np.array_equal(v2, vals[np.lexsort((cols, rv))])
