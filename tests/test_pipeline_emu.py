"""CPU test of the optional pipelined kernel driver (cons_jac_pipe_kernel, CTD_PIPE=1): its producer/consumer loop is
stepped serially by the emulator (tests/emu, test infrastructure only) and must match the oracle like the classic driver."""
import numpy as np
import pytest

import ctdirect_jl_amd as ct
from emu import emu
from helpers import TOL, bench_inputs, dense_on_pattern, describe, relerr

PAIRS = [(p, s) for p in ("goddard", "goddard_all", "double_integrator_path", "quadrotor", "least_squares_with_constraint",
                          "double_integrator_freet0tf") for s in ct.SCHEMES]


@pytest.mark.parametrize("prob,sch", PAIRS, ids=[f"{p}-{s}" for p, s in PAIRS])
def test_pipelined_driver_matches_oracle(oracle_lib, prob, sch):
    for N in (5, 23, 64):
        o = oracle_lib.OracleDOCP(prob, sch, N)
        o.set_pattern_mode(1)
        x = bench_inputs(describe(o, prob, sch), perturb=1e-3)
        cp, rv = o.jac_pattern()
        vref = o.jac_coord(x) if N > 13 else dense_on_pattern(o.jac_dense(x), cp, rv)
        cref = o.constraints(x)
        for Ts, chunk, nthr in ((-1, 0, 256), (1, 3, 128), (2, 7, 256), (8, 40, 256)):
            c, v = emu.cons_jac(ct.PROBLEMS[prob], ct.SCHEMES[sch], 1, N, x, pipe_Ts=Ts, pipe_chunk=chunk, nthr=nthr)
            assert not np.any(c == 666.666) and not np.any(v == 666.666)
            assert relerr(c, cref) <= TOL and relerr(v, vref) <= TOL
