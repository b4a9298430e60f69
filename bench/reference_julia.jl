# bench/reference_julia.jl -- times the REFERENCE (CTDirect.jl + ADNLPModels, CPU) on bench.py's workload.
#
# Not run by this repository: the build image and the GPU box have no Julia (SURVEY.md section 8c/8d), so the
# `cpu_baseline` of bench.py is the C++ restatement under oracle/ ("kind": "port").  A maintainer with a Julia that
# has CTDirect and its test dependencies installed can run this file to put the real reference number beside it:
#
#     julia --project=<CTDirect checkout> bench/reference_julia.jl <CTDirect checkout> [grid_size] [scheme]
#
# It measures what the metric measures: one cons!(c, x) plus one jac_coord!(vals, x) at a fixed x, with the manual
# sparsity pattern (src/collocation.jl:116-120), for Goddard (test/problems/goddard.jl) with gauss_legendre_2 at
# 10000 steps by default.  UNTESTED here for the reason above.
using CTDirect, CTModels, CTParser, ADNLPModels, NLPModels, BenchmarkTools
import CTParser: @def

root = length(ARGS) >= 1 ? ARGS[1] : error("usage: reference_julia.jl <CTDirect checkout> [grid_size] [scheme]")
grid_size = length(ARGS) >= 2 ? parse(Int, ARGS[2]) : 10_000
scheme = length(ARGS) >= 3 ? Symbol(ARGS[3]) : :gauss_legendre_2
include(joinpath(root, "test", "problems", "goddard.jl"))

prob = goddard()
docp = CTDirect.DOCP(prob.ocp, grid_size, 1, scheme, nothing)      # src/DOCP_data.jl:293
nvar, ncon = docp.dim_NLP_variables, docp.dim_NLP_constraints
f = x -> CTDirect.__objective(x, docp)                              # src/collocation.jl:101-102
c! = (c, x) -> CTDirect.__constraints!(c, x, docp)
# The iterate bench.py evaluates at (tests/helpers.py `bench_inputs(..., perturb = 1e-3)`, SURVEY.md section 8d): r = 1 + 0.01 tau,
# v = 0.1 sin(pi tau), m = 1 - 0.4 tau, stage controls 0.5 + 0.5 cos(7 t_ij + j), stage variables 0.3 sin(3 tau + k + 0.5 j) + 0.05 (k + 1)
# (0-based k, j), tf = 0.2, plus 1e-3 x the xorshift64* stream.  (A 0.1 fill would put Goddard's exp(-500 (r - 1)) at 1e195.)
function bench_inputs(docp, grid_size, scheme)
    N, n, m = grid_size, 3, 1
    disc = CTDirect.disc_model(docp)
    blk = disc._step_variables_block
    s = hasproperty(disc, :stage) ? disc.stage : 0
    stagewise = scheme in (:gauss_legendre_2, :gauss_legendre_3)
    x = fill(0.1, docp.dim_NLP_variables)
    x[end] = 0.2
    tau = collect(0:N) ./ N
    for i in 1:(N + 1)
        o = (i - 1) * blk
        t = tau[i]
        x[o + 1] = 1 + 0.01t; x[o + 2] = 0.1 * sin(pi * t); x[o + 3] = 1 - 0.4t
        i == N + 1 && scheme != :trapeze && continue
        if stagewise
            for j in 1:s
                tj = t + disc.butcher_c[j] * (tau[i + 1] - t)
                x[o + n + (j - 1) * m + 1] = 0.5 + 0.5 * cos(7tj + j)
            end
        else
            x[o + n + 1] = 0.5 + 0.5 * cos(7t)
        end
        i == N + 1 && continue
        cu = stagewise ? s * m : m
        for j in 1:s, k in 1:n
            x[o + n + cu + (j - 1) * n + k] = 0.3 * sin(3t + (k - 1) + 0.5 * (j - 1)) + 0.05k
        end
    end
    state = 0x9E3779B97F4A7C15
    for i in eachindex(x)
        state ⊻= state >> 12; state ⊻= state << 25; state ⊻= state >> 27
        r = state * 0x2545F4914F6CDD1D
        x[i] += 1e-3 * (Float64(r >> 11) / 2.0^53 * 2.0 - 1.0)
    end
    return x
end
x = bench_inputs(docp, grid_size, scheme)

jac_backend = ADNLPModels.SparseADJacobian(nvar, f, ncon, c!, CTDirect.DOCP_Jacobian_pattern(docp))
nlp = ADNLPModels.ADNLPModel!(
    f, x, docp.bounds.var_l, docp.bounds.var_u, c!, docp.bounds.con_l, docp.bounds.con_u;
    minimize=(!docp.flags.max),
    gradient_backend=ADNLPModels.ReverseDiffADGradient,
    jacobian_backend=jac_backend,
    hessian_backend=ADNLPModels.EmptyADbackend,
    hprod_backend=ADNLPModels.EmptyADbackend,
    jtprod_backend=ADNLPModels.EmptyADbackend,
    jprod_backend=ADNLPModels.EmptyADbackend,
    ghjvprod_backend=ADNLPModels.EmptyADbackend,
)
c = zeros(ncon)
vals = zeros(NLPModels.get_nnzj(nlp))
t_cons = @belapsed NLPModels.cons!($nlp, $x, $c)
t_jac = @belapsed NLPModels.jac_coord!($nlp, $x, $vals)
println("{\"reference\": \"CTDirect.jl\", \"scheme\": \"$(scheme)\", \"grid_size\": $(grid_size), \"nvar\": $(nvar), ",
        "\"ncon\": $(ncon), \"nnzj\": $(length(vals)), \"cons_s\": $(t_cons), \"jac_coord_s\": $(t_jac), ",
        "\"evals_per_s\": $(1 / (t_cons + t_jac)), \"threads\": 1}")
