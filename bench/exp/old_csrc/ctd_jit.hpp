// ctd_jit.hpp -- OCPs defined at run time.
//
// In the reference the OCP functions are Julia closures produced by CTModels / CTParser (called at src/ode/trapeze.jl:66,
// midpoint.jl:64, irk.jl:291, irk_stagewise.jl:441, src/DOCP_functions.jl:35-48,108-110,136-138).  Closures cannot cross a
// C ABI, so besides the compiled registry (ctd_problems.hpp) the engine accepts an OCP as TEXT: one arithmetic expression
// per output of dynamics / Lagrange cost / Mayer cost / path constraints / boundary constraints (ctd_ocp_def in
// include/ctdirect_hip.h).  The expressions are parsed here (no C++ is accepted from the caller), turned into a functor with
// the same shape as the registry's, and the SAME kernel templates are compiled for it with hiprtc for gfx950 when a handle is
// created -- a run-time defined problem runs the same code path at the same speed as a built-in one.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "ctd_layout.hpp"
#include "ctd_problems.hpp"

struct ctd_ocp_def;

namespace ctd {

constexpr int kRuntimeIdBase = 1000;

// postfix form of one expression, kept for analyses that need the functions on the HOST (the structural-sparsity probe
// of the Hessian tables, ctd_hess_host.cpp); the numbers themselves are only ever computed by the compiled kernels
// RT_ZERO: a function whose derivative vanishes identically (floor): no dependence; RT_MAX: max / min, first order in both operands
enum RtOpKind : uint8_t { RT_CONST, RT_T, RT_X, RT_U, RT_V, RT_X0, RT_XF, RT_ADD, RT_SUB, RT_MUL, RT_DIV, RT_NEG, RT_NONLIN, RT_POW, RT_ZERO, RT_MAX };
struct RtOp { uint8_t kind; int16_t k; };
using RtProgram = std::vector<RtOp>;

// structural nonzeros of the dynamics' first partials as the code generator found them (gen_sym_dyn): row-major slots, -1 = zero
struct DynNZMap {
    bool sparse = false;
    int n_f = 0, n_g = 0;
    std::vector<int> map_f, map_g;       // [r n + c], [r m + c]
};
// text of the DynNZ<type> specialisation (ctd_kernel_body.hpp) for a generated functor / a registry problem
std::string dyn_nz_source(const std::string& type, int n, int m, const DynNZMap& nz);

struct RtOcp {
    std::string name;
    DynNZMap dyn_nz;                   // sparse eval blocks of the generated dynamics code (sparse = false: forward duals, dense)
    ProblemInfo info;                  // info.name points into `name`
    bool dyn_t, dyn_v, path_t, path_v, lag_t, lag_v;
    int dc, hk, maxb;
    bool has_sym = false;              // the functor carries symbolically differentiated stage functions (ctd_sym.hpp)
    std::string functor_src;           // namespace ctd { struct UserOCP { ... }; }
    std::vector<RtProgram> p_dynamics, p_path, p_boundary;
    RtProgram p_lagrange, p_mayer;     // empty: absent
};

inline bool is_runtime_problem(int id) { return id >= kRuntimeIdBase; }
const RtOcp* runtime_ocp(int id);
// parses and registers; returns a status code of include/ctdirect_hip.h (0 = ok) and the new problem id
int register_runtime_ocp(const ctd_ocp_def* def, int* id, std::string& err);

// Expression -> C++ (exposed for tests).  `kind`: 0 dynamics / Lagrange / path (t, x, u, v), 1 Mayer / boundary (x0, xf, v).
// uses_t / uses_v report explicit dependence.  Returns false and sets err on a syntax / name error.
struct ExprCtx {
    int n, m, nv;
    int kind;
    std::map<std::string, double> constants;
    // "name = expression" entries of ctd_ocp_def.constants: sub-expressions with a name (the `aux = ...` lines of a CTParser @def
    // block, e.g. test/problems/swimmer.jl:39-53), substituted where they are used; an alias may use the aliases declared before it
    std::map<std::string, std::string> aliases;
};
bool expr_to_cpp(const std::string& expr, const ExprCtx& cx, std::string& out, bool& is_const, bool& uses_t, bool& uses_v,
                 std::string& err, RtProgram* prog = nullptr);

}  // namespace ctd
