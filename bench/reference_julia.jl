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
x = fill(0.1, nvar)                                                 # default initial guess, src/DOCP_variables.jl:126

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
