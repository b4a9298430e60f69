"""CSR value order (ctd_desc.value_order = CTD_ORDER_CSR, north_star: "assembled ... in CSR on device") on the CPU: the host
model's row-order tables drive the SAME phase functions through the serial emulator (tests/emu/, test infrastructure only) and
must give, bit for bit, the CSC values under the host permutation CSC -> CSR -- every problem, every scheme, all three
patterns, tiny grids (all-edge mode), ragged grids, shards (ONE value range per shard, the ranges partition the array) and the
sharded iterate read in place.  The reference order being replaced: SparseArrays.sparse(Is, Js, ...) at
src/ode/irk_stagewise.jl:555-558 / midpoint.jl:229-232.  The HIP build of the same code: tests/test_gpu_csr.py."""
import numpy as np
import pytest
import scipy.sparse as sp

import ctdirect_jl_amd as ct
from emu import emu
from helpers import bench_inputs, describe


def csc_to_csr_perm(colptr, rowval, ncon):
    """perm[k_csr] = k_csc, and (rowptr, colind) of the same pattern by rows (scipy sorts the column indices of every row)"""
    nvar = len(colptr) - 1
    A = sp.csc_matrix((np.arange(1, len(rowval) + 1, dtype=np.float64), rowval, colptr), shape=(ncon, nvar)).tocsr()
    A.sort_indices()
    return (A.data - 1).astype(np.int64), A.indptr.astype(np.int64), A.indices.astype(np.int64)


PAIRS = [(p, s) for p in ct.PROBLEMS for s in ct.SCHEMES]


@pytest.mark.parametrize("prob,sch", PAIRS, ids=[f"{p}-{s}" for p, s in PAIRS])
def test_csr_values_equal_csc_values_permuted(prob, sch):
    rng = np.random.default_rng(5)
    pid, sid = ct.PROBLEMS[prob], ct.SCHEMES[sch]
    for N, tg in ((1, None), (3, None), (5, None), (6, None), (37, None), (13, np.cumsum(rng.uniform(0.5, 1.5, 14)))):
        d = ct.DOCP(prob, N, sch, time_grid=tg, device=-1)
        x = bench_inputs(describe(d, prob, sch), perturb=1e-3)
        for mode in (0, 1, 2):
            try:
                cp, rv = emu.csc(pid, sid, mode, N, tg)
            except RuntimeError:
                assert mode == 2 and sch == "euler_implicit"      # (refused, never silently wrong)
                continue
            ncon = d.dim_NLP_constraints
            perm, rowptr, colind = csc_to_csr_perm(cp, rv, ncon)
            c0, v0 = emu.cons_jac(pid, sid, mode, N, x, tg, tile=4, nthr=64)
            with emu.value_order(1):
                rp, ci, info = emu.csr(pid, sid, mode, N, tg)
                assert np.array_equal(rp, rowptr) and np.array_equal(ci, colind)
                assert info["vr"] == 0                                   # V entries are inline: no separate streams
                # every step is regular by rows, whatever N; only implicit Euler's step 0 can differ (its path rows see U_0 where the
                # later ones see U_{i-1})
                assert info["reg_last"] == N and info["reg_first"] in ((0, 1) if sch == "euler_implicit" else (0,))
                for tile, nthr in ((0, 64), (1, 5), (4, 33), (7, 256)):
                    c1, v1 = emu.cons_jac(pid, sid, mode, N, x, tg, tile=tile, nthr=nthr)
                    assert not np.any(v1 == 666.666) and not np.any(c1 == 666.666)
                    assert np.array_equal(c1, c0)
                    assert np.array_equal(v1, v0[perm]), (N, mode, tile, nthr)


@pytest.mark.parametrize("prob", ["goddard", "goddard_all", "quadrotor", "double_integrator_freet0tf", "double_integrator_path"])
def test_csr_shards_own_one_range_each_and_compose_exactly(prob):
    N = 23
    for sch in ct.SCHEMES:
        d = ct.DOCP(prob, N, sch, device=-1)
        x = bench_inputs(describe(d, prob, sch), perturb=1e-3)
        pid, sid = ct.PROBLEMS[prob], ct.SCHEMES[sch]
        for mode in (0, 2):
            if mode == 2 and sch == "euler_implicit" and d.dims.path_cons > 0 and d.dims.NLP_u > 0:
                continue
            with emu.value_order(1):
                cf, vf = emu.cons_jac(pid, sid, mode, N, x, tile=4, nthr=64)
                for cuts in ([0, 11, 23], [0, 1, 22, 23], [0, 5, 6, 7, 23]):
                    v = np.full_like(vf, 666.666)
                    c = np.full_like(cf, 666.666)
                    covered = 0
                    for a, b in zip(cuts[:-1], cuts[1:]):
                        c2 = np.full_like(cf, 666.666)
                        v2 = np.full_like(vf, 666.666)
                        emu.cons_jac(pid, sid, mode, N, x, tile=3, nthr=32, step_begin=a, step_end=b, c=c2, vals=v2)
                        lo, hi = emu.shard_range(pid, sid, mode, N, a, b)
                        assert lo == covered                                  # the ranges of consecutive shards partition the array
                        covered = hi
                        wrote = v2 != 666.666
                        assert wrote[lo:hi].all() and not wrote[:lo].any() and not wrote[hi:].any()      # ONE range, nothing else
                        v[lo:hi] = v2[lo:hi]
                        cb = d.discretization._state_stage_eqs_block + d.discretization._step_pathcons_block
                        c[a * cb:b * cb] = c2[a * cb:b * cb]
                        if b == N:
                            c[N * cb:] = c2[N * cb:]
                    assert covered == len(vf)
                    assert np.array_equal(v, vf) and np.array_equal(c, cf)


@pytest.mark.parametrize("prob,sch", [("goddard", "gauss_legendre_2"), ("goddard_all", "trapeze"), ("double_integrator_path", "midpoint"),
                                      ("quadrotor", "gauss_legendre_3"), ("goddard_all", "euler_implicit"), ("quadrotor12", "midpoint"),
                                      ("double_integrator_freet0tf", "euler")])
def test_csr_sharded_iterate_read_in_place(prob, sch):
    pid, sid = ct.PROBLEMS[prob], ct.SCHEMES[sch]
    for N in (7, 40):
        d = ct.DOCP(prob, N, sch, device=-1)
        x = bench_inputs(describe(d, prob, sch), perturb=1e-3)
        with emu.value_order(1):
            cf, vf = emu.cons_jac(pid, sid, 1, N, x, tile=4, nthr=64)
            for G in (2, 3, 7):
                c, v = emu.cons_jac_sharded(pid, sid, 1, N, x, G, tile=3, nthr=64)
                assert np.array_equal(c, cf) and np.array_equal(v, vf), (N, G)


def test_csr_with_several_controls_per_step():
    for prob in ("goddard", "double_integrator_path", "quadrotor"):
        pid, sid = ct.PROBLEMS[prob], ct.SCHEMES["midpoint"]
        for cs in (2, 3):
            with emu.control_steps(cs):
                for N in (3, 9):
                    d = ct.DOCP(prob, N, "midpoint", device=-1, control_steps=cs)
                    x = 0.3 + 0.05 * np.sin(np.arange(d.dim_NLP_variables))
                    for mode in (0, 1, 2):
                        cp, rv = emu.csc(pid, sid, mode, N)
                        perm, rowptr, colind = csc_to_csr_perm(cp, rv, d.dim_NLP_constraints)
                        c0, v0 = emu.cons_jac(pid, sid, mode, N, x, tile=2, nthr=64)
                        with emu.value_order(1):
                            c1, v1 = emu.cons_jac(pid, sid, mode, N, x, tile=2, nthr=64)
                        assert np.array_equal(c1, c0) and np.array_equal(v1, v0[perm])


def test_csr_structure_through_the_c_abi():
    """host-only handles: ctd_jac_csr on any handle, ctd_jac_structure following the handle's value order, shard info = one range"""
    for prob, sch, N in (("goddard", "gauss_legendre_2", 50), ("goddard_all", "trapeze", 9), ("double_integrator_path", "midpoint", 4)):
        for pattern in ("manual", "structural", "optimized"):
            d0 = ct.DOCP(prob, N, sch, device=-1, pattern=pattern)
            d1 = ct.DOCP(prob, N, sch, device=-1, pattern=pattern, value_order="csr")
            cp, rv = ct.DOCP_Jacobian_pattern(d0)
            perm, rowptr, colind = csc_to_csr_perm(cp, rv, d0.dim_NLP_constraints)
            for d in (d0, d1):
                rp, ci = ct.DOCP_Jacobian_csr(d)
                assert np.array_equal(rp, rowptr) and np.array_equal(ci, colind)
            r0, c0 = d0.jac_structure()
            r1, c1 = d1.jac_structure()
            assert np.array_equal(r1, r0[perm]) and np.array_equal(c1, c0[perm])
            assert np.all(np.diff(r1) >= 0)                                   # by rows
            assert d1.nnzj == d0.nnzj and d1.value_order == "csr" and d0.value_order == "csc"
            a, b = N // 3, N - 1
            s1 = ct.DOCP(prob, N, sch, device=-1, pattern=pattern, value_order="csr", steps=(a, b))
            cb = d0.discretization._state_stage_eqs_block + d0.discretization._step_pathcons_block
            assert (s1.shard.vals_main_begin, s1.shard.vals_main_end) == (rowptr[a * cb], rowptr[b * cb])
            sl = ct.DOCP(prob, N, sch, device=-1, pattern=pattern, value_order="csr", steps=(b, N))
            assert (sl.shard.vals_main_begin, sl.shard.vals_main_end) == (rowptr[b * cb], d0.nnzj)
    # the Hessian "in CSR": the lower triangle by columns is the upper triangle by rows -- same arrays, same value order
    d = ct.DOCP("goddard", 20, "midpoint", device=-1)
    cp, rv = ct.DOCP_Hessian_pattern(d)
    rp, ci = ct.DOCP_Hessian_csr(d)
    assert np.array_equal(cp, rp) and np.array_equal(rv, ci)
    for j in range(len(rp) - 1):
        assert np.all(ci[rp[j]:rp[j + 1]] >= j)
    with pytest.raises(ct.CTDirectError):
        ct.DOCP("goddard", 20, "midpoint", device=-1, value_order=7)
