// =================================================================================================
// TEST INFRASTRUCTURE ONLY.  CPU oracle for the CTDirect.jl collocation hot path.
//
// This file is a CPU restatement (C++17, single-threaded, no GPU code) of the reference algorithm in
// /root/reference (CTDirect.jl v1.0.12, 100 % Julia).  It exists so the HIP engine in
// ctdirect.jl_amd/csrc/ can be checked against an independent statement of the same arithmetic.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
// never links, imports or calls anything under oracle/.
//
// What each part follows (reference file:line):
//   DOCP sizes / scheme structs     src/DOCP_data.jl:293-365, src/ode/trapeze.jl:14-42, midpoint.jl:17-39,
//                                   irk.jl:138-160, irk_stagewise.jl:136-163 ; Butcher tables irk.jl:41-43,77-79,
//                                   111-119, irk_stagewise.jl:61-64,103-109
//   time grid                       src/DOCP_data.jl:176-214 (DOCPtime), :437-458 (get_time_grid)
//   getters                         src/ode/common.jl:113-170, irk_stagewise.jl:173-224
//   __constraints!, path, boundary  src/DOCP_functions.jl:80-140
//   setWorkArray / step constraints trapeze.jl:50-71,118-142 ; midpoint.jl:47-72,124-156 ; irk.jl:167-172,236-308 ;
//                                   irk_stagewise.jl:235-239,394-460
//   __objective / integral          src/DOCP_functions.jl:23-54 ; trapeze.jl:78-110 ; midpoint.jl:79-116 ;
//                                   irk.jl:179-228 ; irk_stagewise.jl:344-384
//   bounds / initial guess          src/DOCP_functions.jl:163-191 ; src/DOCP_variables.jl:21-145 ; irk_stagewise.jl:250-335
//   Jacobian / Hessian patterns     trapeze.jl:149-303 ; midpoint.jl:163-300 ; irk.jl:315-496 ; irk_stagewise.jl:468-638 ;
//                                   add_nonzero_block! src/ode/common.jl:285-312
//   Euler explicit / implicit       src/ode/euler.jl:10-50 (struct), :59-72 (control getter), :79-159 (work array, integral,
//                                   step constraints), :166-355 (patterns)
//   Hessian values                  third-party in the reference (ADNLPModels sparse Hessian backend over the closures,
//                                   src/collocation.jl:121-125): restated as a sparse second-order forward sweep over the
//                                   same objective / constraints templates (dual2.hpp), on the lower triangle of
//                                   DOCP_Hessian_pattern
//   Jacobian values                 third-party in the reference (ADNLPModels.SparseADJacobian, not under
//                                   /root/reference): column-colour the pattern, one pass of c!(Dual) per colour,
//                                   decompress into CSC order.  Restated here from the reference's own description
//                                   (test/archives/AD_backend.md:3-5,18-19) and call site (src/collocation.jl:116-120).
//
// Pinning status: Julia is not installed here, so the reference cannot be executed.  The oracle is pinned by the
// reference's own known-answer tests (tests/test_oracle_goldens.py): exact-feasible stagewise trajectory c == 0 and
// objective 4/3 (test/ci/test_discretization_stagewise.jl:53-100,176-198), nnzj 6028 / nnzh 6519
// (test/ci/test_modeler_solver.jl:37), zero-control dims 24 / 23 (test/ci/test_zero_control_allocations.jl:31,138),
// goddard_all trapeze 4005/6007, 40005/60007 (test/archives/AD_backend.md:59-60), and by 50-digit mpmath fixtures
// (tests/golden/).  Jacobian and Hessian VALUES have no golden in the reference ("parity unpinned" for values; pinned by
// the build's own first- and second-order mpmath derivative fixtures and finite differences).
// =================================================================================================
#include <malloc.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "dual.hpp"
#include "dual2.hpp"
#include "dualn.hpp"
#include "problems.hpp"

namespace orc {

// Scheme ids (shared numbering with include/ctdirect_hip.h, restated here on purpose).
enum Scheme {
    TRAPEZE = 0,
    MIDPOINT = 1,
    GL1_CC = 2,   // :gauss_legendre_1 (test only in the reference), constant control
    GL2_CC = 3,   // :gauss_legendre_2_constant_control
    GL3_CC = 4,   // :gauss_legendre_3_constant_control
    GL2_SW = 5,   // :gauss_legendre_2  (stagewise controls)
    GL3_SW = 6,   // :gauss_legendre_3
    EULER_EXPLICIT = 7,   // :euler | :euler_explicit | :euler_forward      (src/DOCP_data.jl:315-317)
    EULER_IMPLICIT = 8,   // :euler_implicit | :euler_backward               (src/DOCP_data.jl:318-320)
};

struct Dims { int NLP_x, NLP_u, NLP_v, path_cons, boundary_cons; };               // DOCP_data.jl:88-94
struct Flags { bool freet0, freetf, lagrange, mayer, max; };                      // DOCP_data.jl:24-30

struct Disc {                                                                      // per-scheme struct fields
    int scheme;
    int stage = 0;
    double a[3][3] = {{0}};
    double b[3] = {0};
    double c[3] = {0};
    bool irk = false, stagewise = false;
    bool euler = false, euler_explicit = true;   // euler.jl:10-50
    int control_block = 0;            // stagewise only
    int step_variables_block = 0;
    int state_stage_eqs_block = 0;
    int step_pathcons_block = 0;
    bool final_control = false;
};

struct Docp {
    int problem;
    Dims dims;
    Flags flags;
    int steps;
    int control_steps = 1;                                                        // DOCPtime.control_steps, DOCP_data.jl:149 (direct shooting: >= 1 controls per step)
    std::vector<double> normalized_grid, fixed_grid;                              // DOCP_data.jl:147-152
    Disc disc;
    int64_t dim_NLP_variables = 0, dim_NLP_constraints = 0;
    std::vector<double> var_l, var_u, con_l, con_u;
    // pattern mode: 0 = REFERENCE_MANUAL (bug-compatible with DOCP_Jacobian_pattern), 1 = STRUCTURAL
    // (= manual + the dynamics-row x v block that trapeze.jl:203 leaves out, hazard H1)
    int pattern_mode = 0;
    // cached pattern + colouring
    bool have_pattern = false;
    std::vector<int64_t> colptr, rowval;
    bool have_hess_pattern = false;
    std::vector<int64_t> hcolptr, hrowval;      // lower triangle of DOCP_Hessian_pattern, CSC, 0-based
    std::vector<int> color;
    int ncolors = 0;
    std::string err;
};

// ---------------------------------------------------------------------------------------------
// scheme constructors = size formulas
// ---------------------------------------------------------------------------------------------
static void set_butcher(Disc& d, int s) {
    d.stage = s;
    if (s == 1) {                                  // irk.jl:41-43
        d.a[0][0] = 0.5; d.b[0] = 1; d.c[0] = 0.5;
    } else if (s == 2) {                           // irk.jl:77-79, irk_stagewise.jl:61-64
        d.a[0][0] = 0.25;                   d.a[0][1] = (0.25 - std::sqrt(3.0) / 6);
        d.a[1][0] = (0.25 + std::sqrt(3.0) / 6); d.a[1][1] = 0.25;
        d.b[0] = 0.5; d.b[1] = 0.5;
        d.c[0] = 0.5 - std::sqrt(3.0) / 6; d.c[1] = 0.5 + std::sqrt(3.0) / 6;
    } else {                                       // irk.jl:111-119, irk_stagewise.jl:103-109
        d.a[0][0] = (5.0 / 36.0);                      d.a[0][1] = (2.0 / 9 - std::sqrt(15.0) / 15);  d.a[0][2] = (5.0 / 36 - std::sqrt(15.0) / 30);
        d.a[1][0] = (5.0 / 36.0 + std::sqrt(15.0) / 24); d.a[1][1] = (2.0 / 9.0);                      d.a[1][2] = (5.0 / 36.0 - std::sqrt(15.0) / 24);
        d.a[2][0] = (5.0 / 36 + std::sqrt(15.0) / 30);   d.a[2][1] = (2.0 / 9 + std::sqrt(15.0) / 15);  d.a[2][2] = (5.0 / 36.0);
        d.b[0] = 5.0 / 18.0; d.b[1] = 4.0 / 9.0; d.b[2] = 5.0 / 18.0;
        d.c[0] = 0.5 - 0.1 * std::sqrt(15.0); d.c[1] = 0.5; d.c[2] = 0.5 + 0.1 * std::sqrt(15.0);
    }
}

static void build_scheme(Docp& p, int scheme) {
    Disc& d = p.disc;
    const Dims& m = p.dims;
    const int64_t N = p.steps;
    d.scheme = scheme;
    switch (scheme) {
        case TRAPEZE:                                             // trapeze.jl:14-42
            d.final_control = true;
            d.step_variables_block = m.NLP_x + m.NLP_u;
            d.state_stage_eqs_block = m.NLP_x;
            d.step_pathcons_block = m.path_cons;
            p.dim_NLP_variables = N * d.step_variables_block + m.NLP_x + m.NLP_v + m.NLP_u;
            break;
        case MIDPOINT:                                            // midpoint.jl:17-39
            d.step_variables_block = m.NLP_x + m.NLP_u * p.control_steps;      // :20
            d.state_stage_eqs_block = m.NLP_x;
            d.step_pathcons_block = m.path_cons;
            p.dim_NLP_variables = N * d.step_variables_block + m.NLP_x + m.NLP_v;
            break;
        case EULER_EXPLICIT: case EULER_IMPLICIT:                 // euler.jl:19-49 (same layout as midpoint)
            d.euler = true;
            d.euler_explicit = scheme == EULER_EXPLICIT;
            d.step_variables_block = m.NLP_x + m.NLP_u;
            d.state_stage_eqs_block = m.NLP_x;
            d.step_pathcons_block = m.path_cons;
            p.dim_NLP_variables = N * d.step_variables_block + m.NLP_x + m.NLP_v;
            break;
        case GL1_CC: case GL2_CC: case GL3_CC: {                  // irk.jl:138-160
            int s = scheme - GL1_CC + 1;
            set_butcher(d, s);
            d.irk = true;
            d.step_variables_block = m.NLP_x + m.NLP_u + m.NLP_x * s;
            d.state_stage_eqs_block = m.NLP_x * (1 + s);
            d.step_pathcons_block = m.path_cons;
            p.dim_NLP_variables = N * d.step_variables_block + m.NLP_x + m.NLP_v;
            break;
        }
        case GL2_SW: case GL3_SW: {                               // irk_stagewise.jl:136-163
            int s = scheme - GL2_SW + 2;
            set_butcher(d, s);
            d.irk = true; d.stagewise = true;
            d.control_block = m.NLP_u * s;
            d.step_variables_block = m.NLP_x + d.control_block + s * m.NLP_x;
            d.state_stage_eqs_block = m.NLP_x * (1 + s);
            d.step_pathcons_block = m.path_cons;
            p.dim_NLP_variables = N * d.step_variables_block + m.NLP_x + m.NLP_v;
            break;
        }
        default:
            throw std::runtime_error("Unknown discretization method");   // DOCP_data.jl:342-349
    }
    p.dim_NLP_constraints = N * (d.state_stage_eqs_block + d.step_pathcons_block) + d.step_pathcons_block + m.boundary_cons;
    // control_steps > 1 (src/direct_shooting.jl:55-71): the other scheme structs size their blocks with it too (trapeze.jl:20,
    // euler.jl:22, irk.jl:141) but only midpoint.jl:47-72,99-116,137-155 has sub-step dynamics -- there the extra controls would
    // be variables nothing reads.  Restated for the midpoint scheme only.
    if (p.control_steps > 1 && scheme != MIDPOINT) throw std::runtime_error("control_steps > 1 is restated for :midpoint only");
}

// ---------------------------------------------------------------------------------------------
// getters (0-based offsets of the reference's 1-based views)
// ---------------------------------------------------------------------------------------------
// get_OCP_variable: common.jl:113-115
template <class T> static const T* get_OCP_variable(const T* xu, const Docp& p) { return xu + (p.dim_NLP_variables - p.dims.NLP_v); }
// get_OCP_state_at_time_step: common.jl:124-128  (i is 1-based as in the reference)
template <class T> static const T* get_state(const T* xu, const Docp& p, int64_t i) { return xu + (i - 1) * p.disc.step_variables_block; }
// get_OCP_control_at_time_step (generic): common.jl:140-155
template <class T> static const T* get_control_generic(const T* xu, const Docp& p, int64_t i, int j = 1) {
    if (!p.disc.final_control && i == p.steps + 1) i = p.steps;
    return xu + (i - 1) * p.disc.step_variables_block + p.dims.NLP_x + (j - 1) * p.dims.NLP_u;     // :150-152 (j-th control of the step)
}
// get_stagecontrol_at_time_step: irk_stagewise.jl:173-188
template <class T> static const T* get_stagecontrol(const T* xu, const Docp& p, int64_t i, int j) {
    return xu + (i - 1) * p.disc.step_variables_block + p.dims.NLP_x + (j - 1) * p.dims.NLP_u;
}
// get_stagevars_at_time_step: common.jl:166-170 (constant control) / irk_stagewise.jl:212-224 (stagewise)
template <class T> static const T* get_stagevars(const T* xu, const Docp& p, int64_t i, int j) {
    int cb = p.disc.stagewise ? p.disc.control_block : p.dims.NLP_u;
    return xu + (i - 1) * p.disc.step_variables_block + p.dims.NLP_x + cb + (j - 1) * p.dims.NLP_x;
}
// control used by path constraints / solution: generic view, or the b-weighted stage average
// for stagewise schemes (irk_stagewise.jl:197-205; returns a new vector, hazard H5)
template <class T> static void get_OCP_control(const T* xu, const Docp& p, int64_t i, T* ui) {
    const int m = p.dims.NLP_u;
    if (p.disc.stagewise) {
        if (i == p.steps + 1) i = p.steps;
        const T* u1 = get_stagecontrol(xu, p, i, 1);
        for (int k = 0; k < m; ++k) ui[k] = p.disc.b[0] * u1[k];
        for (int j = 2; j <= p.disc.stage; ++j) {
            const T* uj = get_stagecontrol(xu, p, i, j);
            for (int k = 0; k < m; ++k) ui[k] = ui[k] + p.disc.b[j - 1] * uj[k];
        }
    } else if (p.disc.euler) {                     // euler.jl:59-72: explicit u(t_N+1) = U_N, implicit u(t_1) = U_1 and u(t_i) = U_{i-1}
        int64_t blk;
        if (p.disc.euler_explicit) blk = (i == p.steps + 1 ? p.steps : i) - 1;
        else blk = (i == 1 ? 2 : i) - 2;
        const T* u = xu + blk * p.disc.step_variables_block + p.dims.NLP_x;
        for (int k = 0; k < m; ++k) ui[k] = u[k];
    } else {
        const T* u = get_control_generic(xu, p, i);
        for (int k = 0; k < m; ++k) ui[k] = u[k];
    }
}

// get_time_grid: DOCP_data.jl:437-458   grid = t0 + normalized_grid * (tf - t0)
template <class P, class T> static void get_time_grid(const T* xu, const Docp& p, std::vector<T>& grid) {
    grid.resize(p.steps + 1);
    const T* v = get_OCP_variable(xu, p);
    T t0 = P::template t0<T>(v);
    T tf = P::template tf<T>(v);
    for (int64_t i = 0; i <= p.steps; ++i) grid[i] = t0 + p.normalized_grid[i] * (tf - t0);
}
template <class P, class T> static void time_grid_for(const T* xu, const Docp& p, std::vector<T>& grid) {
    if (p.flags.freet0 || p.flags.freetf) {
        get_time_grid<P, T>(xu, p, grid);
    } else {
        grid.resize(p.steps + 1);
        for (int64_t i = 0; i <= p.steps; ++i) grid[i] = T(p.fixed_grid[i]);
    }
}

// ---------------------------------------------------------------------------------------------
// __constraints!  (src/DOCP_functions.jl:80-115)
// ---------------------------------------------------------------------------------------------
template <class P, class T> static void constraints(const Docp& p, const T* xu, T* c) {
    const int n = p.dims.NLP_x, m = p.dims.NLP_u, np = p.dims.path_cons;
    const int64_t N = p.steps;
    const Disc& d = p.disc;
    const int cblk = d.state_stage_eqs_block + d.step_pathcons_block;
    std::vector<T> grid;
    time_grid_for<P, T>(xu, p, grid);
    const T* v = get_OCP_variable(xu, p);

    // ---- setWorkArray
    std::vector<T> work;
    if (d.scheme == TRAPEZE) {                    // trapeze.jl:50-71: f at all N+1 nodes
        work.resize((size_t)n * (N + 1));
        for (int64_t i = 1; i <= N + 1; ++i) {
            P::template dynamics<T>(&work[(i - 1) * n], grid[i - 1], get_state(xu, p, i), get_control_generic(xu, p, i), v);
        }
    } else if (d.scheme == MIDPOINT) {            // midpoint.jl:47-72: f at the N midpoints, once per control step
        const int cs = p.control_steps;
        work.resize((size_t)n * N * cs);
        std::vector<T> xs(n);
        size_t offset = 0;
        for (int64_t i = 1; i <= N; ++i) {
            T ts = 0.5 * (grid[i - 1] + grid[i]);
            const T* xi = get_state(xu, p, i);
            const T* xip1 = get_state(xu, p, i + 1);
            for (int k = 0; k < n; ++k) xs[k] = 0.5 * (xi[k] + xip1[k]);
            for (int j = 1; j <= cs; ++j) {       // :61-69 (the same t_s, x_s for every control of the step)
                P::template dynamics<T>(&work[offset], ts, xs.data(), get_control_generic(xu, p, i, j), v);
                offset += n;
            }
        }
    } else if (d.euler) {                         // euler.jl:79-105: f at (t_i, x_i, u_i) or (t_i+1, x_i+1, u(t_i+1) = U_i)
        work.resize((size_t)n * N);
        std::vector<T> uw(m > 0 ? m : 1);
        for (int64_t i = 1; i <= N; ++i) {
            const int64_t index = d.euler_explicit ? i : i + 1;
            get_OCP_control(xu, p, index, uw.data());
            P::template dynamics<T>(&work[(i - 1) * n], grid[index - 1], get_state(xu, p, index), uw.data(), v);
        }
    } else {                                      // irk.jl:167-172, irk_stagewise.jl:235-239: [x_ij ; sum_bk]
        work.resize(2 * (size_t)n);
    }

    std::vector<T> ui(m > 0 ? m : 1);
    // ---- main loop on time steps  (DOCP_functions.jl:92-98)
    for (int64_t i = 1; i <= N; ++i) {
        const int64_t offset = (i - 1) * cblk;
        const T ti = grid[i - 1];
        const T tip1 = grid[i];
        const T* xi = get_state(xu, p, i);
        const T* xip1 = get_state(xu, p, i + 1);
        if (d.scheme == TRAPEZE) {                // trapeze.jl:118-142
            T half_hi = 0.5 * (tip1 - ti);
            const T* fi = &work[(i - 1) * n];
            const T* fip1 = &work[i * n];
            for (int k = 0; k < n; ++k) {
                T x_next = xi[k] + half_hi * (fi[k] + fip1[k]);
                c[offset + k] = xip1[k] - x_next;
            }
        } else if (d.scheme == MIDPOINT) {        // midpoint.jl:124-156
            const int cs = p.control_steps;
            T hi = (tip1 - ti) / (double)cs;      // :134
            if (cs == 1) {                        // :138-140
                const T* fi = &work[(i - 1) * n];
                for (int k = 0; k < n; ++k) c[offset + k] = xip1[k] - (xi[k] + hi * fi[k]);
            } else {                              // :146-153: x_next = x_i; x_next += h_i work_j, j = 1..control_steps
                std::vector<T> x_next(xi, xi + n);
                size_t offset_dyn = (size_t)(i - 1) * n * cs;
                for (int j = 1; j <= cs; ++j) {
                    for (int k = 0; k < n; ++k) x_next[k] = x_next[k] + hi * work[offset_dyn + k];
                    offset_dyn += n;
                }
                for (int k = 0; k < n; ++k) c[offset + k] = xip1[k] - x_next[k];
            }
        } else if (d.euler) {                     // euler.jl:141-159
            T hi = tip1 - ti;
            const T* fi = &work[(i - 1) * n];
            for (int k = 0; k < n; ++k) c[offset + k] = xip1[k] - (xi[k] + hi * fi[k]);
        } else {                                  // irk.jl:236-308 / irk_stagewise.jl:394-460
            T hi = tip1 - ti;
            T* work_xij = work.data();
            T* work_sumbk = work.data() + n;
            int offset_stage_eqs = n;
            for (int j = 1; j <= d.stage; ++j) {
                T tij = ti + d.c[j - 1] * hi;
                const T* kij = get_stagevars(xu, p, i, j);
                const T* uij = d.stagewise ? get_stagecontrol(xu, p, i, j) : get_control_generic(xu, p, i);
                if (j == 1) {
                    for (int k = 0; k < n; ++k) work_sumbk[k] = d.b[j - 1] * kij[k];
                } else {
                    for (int k = 0; k < n; ++k) work_sumbk[k] = work_sumbk[k] + d.b[j - 1] * kij[k];
                }
                for (int k = 0; k < n; ++k) work_xij[k] = xi[k];
                for (int l = 1; l <= d.stage; ++l) {
                    const T* kil = get_stagevars(xu, p, i, l);
                    for (int k = 0; k < n; ++k) work_xij[k] = work_xij[k] + hi * d.a[j - 1][l - 1] * kil[k];
                }
                T* cs = c + offset + offset_stage_eqs;
                P::template dynamics<T>(cs, tij, work_xij, uij, v);
                for (int k = 0; k < n; ++k) cs[k] = kij[k] - cs[k];
                offset_stage_eqs += n;
            }
            for (int k = 0; k < n; ++k) c[offset + k] = xip1[k] - (xi[k] + hi * work_sumbk[k]);
        }
        // path constraints  (stepPathConstraints!, DOCP_functions.jl:122-140)
        if (np > 0) {
            get_OCP_control(xu, p, i, ui.data());
            P::template path<T>(c + offset + d.state_stage_eqs_block, ti, xi, ui.data(), v);
        }
    }
    // path constraints at final time  (DOCP_functions.jl:100)
    if (np > 0) {
        const int64_t offset = N * cblk;
        get_OCP_control(xu, p, N + 1, ui.data());
        P::template path<T>(c + offset, grid[N], get_state(xu, p, N + 1), ui.data(), v);
    }
    // boundary constraints  (DOCP_functions.jl:103-111)
    if (p.dims.boundary_cons > 0) {
        const int64_t offset = p.dim_NLP_constraints - p.dims.boundary_cons;
        P::template boundary<T>(c + offset, get_state(xu, p, 1), get_state(xu, p, N + 1), v);
    }
}

// ---------------------------------------------------------------------------------------------
// __objective  (src/DOCP_functions.jl:23-54) with per-scheme integral()
// ---------------------------------------------------------------------------------------------
template <class P, class T> static T objective(const Docp& p, const T* xu) {
    const int n = p.dims.NLP_x, m = p.dims.NLP_u;
    const int64_t N = p.steps;
    const Disc& d = p.disc;
    std::vector<T> grid;
    time_grid_for<P, T>(xu, p, grid);
    const T* v = get_OCP_variable(xu, p);
    T obj_mayer(0.0), obj_lagrange(0.0);
    if (p.flags.mayer) obj_mayer = P::template mayer<T>(get_state(xu, p, 1), get_state(xu, p, N + 1), v);
    if (p.flags.lagrange) {
        T value(0.0);
        if (d.scheme == TRAPEZE) {                // trapeze.jl:78-110
            {
                T hi = grid[1] - grid[0];
                value = value + hi / 2.0 * P::template lagrange<T>(grid[0], get_state(xu, p, 1), get_control_generic(xu, p, 1), v);
            }
            for (int64_t i = 2; i <= N; ++i) {
                T hi2 = grid[i] - grid[i - 2];
                value = value + hi2 / 2.0 * P::template lagrange<T>(grid[i - 1], get_state(xu, p, i), get_control_generic(xu, p, i), v);
            }
            {
                T hi = grid[N] - grid[N - 1];
                value = value + hi / 2.0 * P::template lagrange<T>(grid[N], get_state(xu, p, N + 1), get_control_generic(xu, p, N + 1), v);
            }
        } else if (d.scheme == MIDPOINT && p.control_steps > 1) {     // midpoint.jl:99-116
            const int cs = p.control_steps;
            std::vector<T> xs(n);
            for (int64_t i = 1; i <= N; ++i) {
                T hi = (grid[i] - grid[i - 1]) / (double)cs;
                const T* xi = get_state(xu, p, i);
                const T* xip1 = get_state(xu, p, i + 1);
                for (int k = 0; k < n; ++k) xs[k] = 0.5 * (xi[k] + xip1[k]);
                for (int j = 1; j <= cs; ++j) {
                    T tij = grid[i - 1] + (j - 0.5) * hi;          // :110
                    value = value + hi * P::template lagrange<T>(tij, xs.data(), get_control_generic(xu, p, i, j), v);
                }
            }
        } else if (d.scheme == MIDPOINT) {        // midpoint.jl:79-97
            std::vector<T> xs(n);
            for (int64_t i = 1; i <= N; ++i) {
                T hi = grid[i] - grid[i - 1];
                T ts = 0.5 * (grid[i - 1] + grid[i]);
                const T* xi = get_state(xu, p, i);
                const T* xip1 = get_state(xu, p, i + 1);
                for (int k = 0; k < n; ++k) xs[k] = 0.5 * (xi[k] + xip1[k]);
                value = value + hi * P::template lagrange<T>(ts, xs.data(), get_control_generic(xu, p, i), v);
            }
        } else if (d.euler) {                     // euler.jl:112-134
            std::vector<T> uw(m > 0 ? m : 1);
            for (int64_t i = 1; i <= N; ++i) {
                const int64_t index = d.euler_explicit ? i : i + 1;
                get_OCP_control(xu, p, index, uw.data());
                T hi = grid[i] - grid[i - 1];
                value = value + hi * P::template lagrange<T>(grid[index - 1], get_state(xu, p, index), uw.data(), v);
            }
        } else {                                  // irk.jl:179-228 / irk_stagewise.jl:344-384
            std::vector<T> work_xij(n);
            for (int64_t i = 1; i <= N; ++i) {
                T ti = grid[i - 1];
                const T* xi = get_state(xu, p, i);
                T hi = grid[i] - ti;
                T local_sum(0.0);
                for (int j = 1; j <= d.stage; ++j) {
                    T tij = ti + d.c[j - 1] * hi;
                    const T* uij = d.stagewise ? get_stagecontrol(xu, p, i, j) : get_control_generic(xu, p, i);
                    for (int k = 0; k < n; ++k) work_xij[k] = xi[k];
                    for (int l = 1; l <= d.stage; ++l) {
                        const T* kil = get_stagevars(xu, p, i, l);
                        for (int k = 0; k < n; ++k) work_xij[k] = work_xij[k] + hi * d.a[j - 1][l - 1] * kil[k];
                    }
                    T term = d.b[j - 1] * P::template lagrange<T>(tij, work_xij.data(), uij, v);
                    if (!d.stagewise && j == 1) local_sum = term;      // irk.jl:214-221 (assign on j==1)
                    else local_sum = local_sum + term;                 // irk_stagewise.jl:376 (0.0 + ...)
                }
                value = value + hi * local_sum;
            }
        }
        obj_lagrange = value;
        (void)m;
    }
    return obj_mayer + obj_lagrange;
}

// ---------------------------------------------------------------------------------------------
// bounds  (src/DOCP_functions.jl:163-191, src/DOCP_variables.jl:21-98, irk_stagewise.jl:250-300)
// ---------------------------------------------------------------------------------------------
static void build_bounds_block(int dim, const std::vector<BoxEntry>& box, std::vector<double>& lb, std::vector<double>& ub) {
    lb.assign(dim, -INF);
    ub.assign(dim, INF);
    for (const BoxEntry& e : box) { lb[e.index] = e.lb; ub[e.index] = e.ub; }
}

template <class P> static void variables_bounds(Docp& p) {
    const int n = p.dims.NLP_x, m = p.dims.NLP_u, nv = p.dims.NLP_v;
    const int64_t N = p.steps;
    const Disc& d = p.disc;
    p.var_l.assign(p.dim_NLP_variables, -INF);
    p.var_u.assign(p.dim_NLP_variables, INF);
    std::vector<double> x_lb, x_ub, u_lb, u_ub, v_lb, v_ub;
    build_bounds_block(n, P::state_box(), x_lb, x_ub);
    build_bounds_block(m, P::control_box(), u_lb, u_ub);
    for (int64_t i = 1; i <= N + 1; ++i) {
        int64_t off = (i - 1) * d.step_variables_block;          // set_state_at_time_step!
        for (int k = 0; k < n; ++k) { p.var_l[off + k] = x_lb[k]; p.var_u[off + k] = x_ub[k]; }
    }
    if (m > 0) {
        if (d.stagewise) {                                       // irk_stagewise.jl:275-286
            for (int64_t i = 1; i <= N; ++i)
                for (int j = 1; j <= d.stage; ++j) {
                    int64_t off = (i - 1) * d.step_variables_block + n + (j - 1) * m;
                    for (int k = 0; k < m; ++k) { p.var_l[off + k] = u_lb[k]; p.var_u[off + k] = u_ub[k]; }
                }
        } else {                                                 // DOCP_variables.jl:40-49 + setter common.jl:209-223
            for (int64_t i = 1; i <= N + 1; ++i) {
                if (i <= N || (d.final_control && i <= N + 1)) {
                    for (int j = 1; j <= p.control_steps; ++j) {     // DOCP_variables.jl:44-47
                        int64_t off = (i - 1) * d.step_variables_block + n + (j - 1) * m;
                        for (int k = 0; k < m; ++k) { p.var_l[off + k] = u_lb[k]; p.var_u[off + k] = u_ub[k]; }
                    }
                }
            }
        }
    }
    if (nv > 0) {
        build_bounds_block(nv, P::variable_box(), v_lb, v_ub);
        for (int k = 0; k < nv; ++k) {
            p.var_l[p.dim_NLP_variables - nv + k] = v_lb[k];
            p.var_u[p.dim_NLP_variables - nv + k] = v_ub[k];
        }
    }
}

template <class P> static void constraints_bounds(Docp& p) {
    const int np = p.dims.path_cons, nb = p.dims.boundary_cons;
    p.con_l.assign(p.dim_NLP_constraints, 0.0);
    p.con_u.assign(p.dim_NLP_constraints, 0.0);
    std::vector<double> plb(np > 0 ? np : 1), pub(np > 0 ? np : 1), blb(nb > 0 ? nb : 1), bub(nb > 0 ? nb : 1);
    P::path_bounds(plb.data(), pub.data());
    P::boundary_bounds(blb.data(), bub.data());
    int64_t offset = 0;
    for (int64_t i = 1; i <= p.steps + 1; ++i) {
        if (i <= p.steps) offset += p.disc.state_stage_eqs_block;
        if (np > 0) {
            for (int k = 0; k < np; ++k) { p.con_l[offset + k] = plb[k]; p.con_u[offset + k] = pub[k]; }
            offset += np;
        }
    }
    if (nb > 0) {
        for (int k = 0; k < nb; ++k) { p.con_l[offset + k] = blb[k]; p.con_u[offset + k] = bub[k]; }
    }
}

// __initial_guess  (src/DOCP_variables.jl:122-145, irk_stagewise.jl:302-335).  use_problem_init selects the
// problem file's own `init` tuple; otherwise the CTModels default (everything left at 0.1).
template <class P> static void initial_guess(const Docp& p, bool use_problem_init, double* X) {
    const int n = p.dims.NLP_x, m = p.dims.NLP_u, nv = p.dims.NLP_v;
    const int64_t N = p.steps;
    const Disc& d = p.disc;
    for (int64_t k = 0; k < p.dim_NLP_variables; ++k) X[k] = 0.1;
    if (!use_problem_init) return;
    std::vector<double> tmp(std::max(std::max(n, m), std::max(nv, 1)));
    if (nv > 0 && P::init_variable(tmp.data()))
        for (int k = 0; k < nv; ++k) X[p.dim_NLP_variables - nv + k] = tmp[k];
    std::vector<double> grid;
    get_time_grid<P, double>(X, p, grid);
    for (int64_t i = 1; i <= N + 1; ++i) {
        double ti = grid[i - 1];
        if (P::init_state(ti, tmp.data())) {
            int64_t off = (i - 1) * d.step_variables_block;
            for (int k = 0; k < n; ++k) X[off + k] = tmp[k];
        }
        if (m > 0 && !d.stagewise) {
            if (i <= N || d.final_control) {
                if (P::init_control(ti, tmp.data())) {
                    for (int j = 1; j <= p.control_steps; ++j) {     // DOCP_variables.jl:138-140: init.control(t_i) for every control of the step
                        int64_t off = (i - 1) * d.step_variables_block + n + (j - 1) * m;
                        for (int k = 0; k < m; ++k) X[off + k] = tmp[k];
                    }
                }
            }
        }
    }
    if (m > 0 && d.stagewise) {
        for (int64_t i = 1; i <= N; ++i) {
            double ti = grid[i - 1];
            double hi = grid[i] - ti;
            for (int j = 1; j <= d.stage; ++j) {
                double tij = ti + d.c[j - 1] * hi;
                if (P::init_control(tij, tmp.data())) {
                    int64_t off = (i - 1) * d.step_variables_block + n + (j - 1) * m;
                    for (int k = 0; k < m; ++k) X[off + k] = tmp[k];
                }
            }
        }
    }
}

// __initial_guess with time-dependent init.state(t) / init.control(t) given as sampled trajectories (the interpolated and
// warm-start guesses of test/ci/test_initial_guess.jl): same loops as above (src/DOCP_variables.jl:122-145,
// irk_stagewise.jl:302-335), the init functions being piecewise-linear interpolants with held end values.
static void interp_row(int64_t K, const double* T, const double* data, int dim, double t, double* out) {
    int64_t lo = 0;
    if (K == 1 || t <= T[0]) { for (int k = 0; k < dim; ++k) out[k] = data[k]; return; }
    if (t >= T[K - 1]) { for (int k = 0; k < dim; ++k) out[k] = data[(K - 1) * dim + k]; return; }
    while (lo + 1 < K && T[lo + 1] <= t) ++lo;
    const double w = (t - T[lo]) / (T[lo + 1] - T[lo]);
    for (int k = 0; k < dim; ++k) out[k] = data[lo * dim + k] + w * (data[(lo + 1) * dim + k] - data[lo * dim + k]);
}
template <class P> static void initial_guess_sampled(const Docp& p, int64_t K, const double* T, const double* Xs, const double* Us,
                                                      const double* v, double* X) {
    const int n = p.dims.NLP_x, m = p.dims.NLP_u, nv = p.dims.NLP_v;
    const int64_t N = p.steps;
    const Disc& d = p.disc;
    for (int64_t k = 0; k < p.dim_NLP_variables; ++k) X[k] = 0.1;
    if (v) for (int k = 0; k < nv; ++k) X[p.dim_NLP_variables - nv + k] = v[k];
    std::vector<double> grid, tmp(std::max(std::max(n, m), 1));
    get_time_grid<P, double>(X, p, grid);
    for (int64_t i = 1; i <= N + 1; ++i) {
        const double ti = grid[i - 1];
        const int64_t off = (i - 1) * d.step_variables_block;
        if (Xs) { interp_row(K, T, Xs, n, ti, tmp.data()); for (int k = 0; k < n; ++k) X[off + k] = tmp[k]; }
        if (Us && m > 0 && !d.stagewise && (i <= N || d.final_control)) {
            interp_row(K, T, Us, m, ti, tmp.data());
            for (int k = 0; k < m; ++k) X[off + n + k] = tmp[k];
        }
    }
    if (Us && m > 0 && d.stagewise)
        for (int64_t i = 1; i <= N; ++i) {
            const double ti = grid[i - 1], hi = grid[i] - ti;
            for (int j = 1; j <= d.stage; ++j) {
                interp_row(K, T, Us, m, ti + d.c[j - 1] * hi, tmp.data());
                for (int k = 0; k < m; ++k) X[(i - 1) * d.step_variables_block + n + (j - 1) * m + k] = tmp[k];
            }
        }
}

// ---------------------------------------------------------------------------------------------
// sparsity patterns: literal (Is, Js) pushes + SparseArrays.sparse semantics
// ---------------------------------------------------------------------------------------------
struct IJ {
    std::vector<int64_t> Is, Js;
    // add_nonzero_block!(Is, Js, i_start, i_end, j_start, j_end; sym)   common.jl:297-306   (1-based, inclusive)
    void block(int64_t i0, int64_t i1, int64_t j0, int64_t j1, bool sym = false) {
        for (int64_t i = i0; i <= i1; ++i)
            for (int64_t j = j0; j <= j1; ++j) {
                Is.push_back(i); Js.push_back(j);
                if (sym) { Is.push_back(j); Js.push_back(i); }
            }
    }
    // add_nonzero_block!(Is, Js, i, j; sym)   common.jl:307-312
    void single(int64_t i, int64_t j, bool sym = false) {
        Is.push_back(i); Js.push_back(j);
        if (sym) { Is.push_back(j); Js.push_back(i); }
    }
};

// SparseArrays.sparse(Is, Js, ones(Bool), nrow, ncol): CSC, rows sorted within a column, duplicates merged.
static void to_csc(const IJ& ij, int64_t ncol, std::vector<int64_t>& colptr, std::vector<int64_t>& rowval) {
    const size_t nn = ij.Is.size();
    std::vector<int64_t> cnt(ncol + 1, 0);
    for (size_t k = 0; k < nn; ++k) cnt[ij.Js[k]]++;               // Js are 1-based -> bucket j
    std::vector<int64_t> start(ncol + 2, 0);
    for (int64_t j = 1; j <= ncol; ++j) start[j + 1] = start[j] + cnt[j];
    std::vector<int64_t> rows(nn);
    std::vector<int64_t> fill(start.begin(), start.end());
    for (size_t k = 0; k < nn; ++k) rows[fill[ij.Js[k]]++] = ij.Is[k];
    colptr.assign(ncol + 1, 0);
    rowval.clear();
    rowval.reserve(nn);
    for (int64_t j = 1; j <= ncol; ++j) {
        auto b = rows.begin() + start[j], e = rows.begin() + start[j + 1];
        std::sort(b, e);
        auto u = std::unique(b, e);
        for (auto it = b; it != u; ++it) rowval.push_back(*it - 1);   // store 0-based
        colptr[j] = (int64_t)rowval.size();
    }
}

static void jacobian_pattern_ij(const Docp& p, IJ& ij) {
    const Disc& d = p.disc;
    const Dims& dm = p.dims;
    const int64_t N = p.steps;
    const int64_t v_start = p.dim_NLP_variables - dm.NLP_v + 1, v_end = p.dim_NLP_variables;
    const int64_t c_block_step = d.state_stage_eqs_block + d.step_pathcons_block;
    const int64_t blk = d.step_variables_block;
    if (d.scheme == TRAPEZE) {                                         // trapeze.jl:149-233
        for (int64_t i = 1; i <= N; ++i) {
            int64_t c_offset = (i - 1) * c_block_step;
            int64_t dyn_start = c_offset + 1, dyn_end = c_offset + dm.NLP_x;
            int64_t path_start = c_offset + dm.NLP_x + 1, path_end = c_offset + c_block_step;
            int64_t var_offset = (i - 1) * blk;
            int64_t xi_start = var_offset + 1, xi_end = var_offset + dm.NLP_x;
            int64_t ui_start = var_offset + dm.NLP_x + 1, ui_end = var_offset + dm.NLP_x + dm.NLP_u;
            int64_t xip1_end = var_offset + dm.NLP_x + dm.NLP_u + dm.NLP_x;
            int64_t uip1_start = var_offset + dm.NLP_x * 2 + dm.NLP_u + 1, uip1_end = var_offset + dm.NLP_x * 2 + dm.NLP_u * 2;
            ij.block(dyn_start, dyn_end, xi_start, xi_end);
            ij.block(dyn_start, dyn_end, ui_start, xip1_end);
            ij.block(dyn_start, dyn_end, uip1_start, uip1_end);
            ij.block(path_start, path_end, xi_start, xi_end);
            ij.block(path_start, path_end, ui_start, ui_end);
            ij.block(path_start, path_end, v_start, v_end);             // :203  (path rows only -- hazard H1)
            if (p.pattern_mode == 1) ij.block(dyn_start, dyn_end, v_start, v_end);   // STRUCTURAL: what the comment at :202 intends
        }
        int64_t c_offset = N * c_block_step, c_block = d.step_pathcons_block;
        int64_t var_offset = N * blk;
        int64_t xf_start = var_offset + 1, xf_end = var_offset + dm.NLP_x;
        int64_t uf_start = var_offset + dm.NLP_x + 1, uf_end = var_offset + dm.NLP_x + dm.NLP_u;
        ij.block(c_offset + 1, c_offset + c_block, xf_start, xf_end);
        ij.block(c_offset + 1, c_offset + c_block, uf_start, uf_end);
        ij.block(c_offset + 1, c_offset + c_block, v_start, v_end);
        c_offset = N * c_block_step + d.step_pathcons_block;
        c_block = dm.boundary_cons;
        ij.block(c_offset + 1, c_offset + c_block, 1, dm.NLP_x);
        ij.block(c_offset + 1, c_offset + c_block, xf_start, xf_end);
        ij.block(c_offset + 1, c_offset + c_block, v_start, v_end);
        return;
    }
    if (d.euler) {                                                     // euler.jl:166-263
        for (int64_t i = 1; i <= N; ++i) {
            int64_t c_block = c_block_step, c_offset = (i - 1) * c_block;
            int64_t dyn_start = c_offset + 1, dyn_end = c_offset + dm.NLP_x, dyn_lag = c_offset + dm.NLP_x;
            int64_t path_start = c_offset + dm.NLP_x + 1, path_end = c_offset + c_block;
            int64_t var_offset = (i - 1) * blk;
            int64_t xi_start = var_offset + 1, xi_end = var_offset + dm.NLP_x, li = var_offset + dm.NLP_x;
            int64_t ui_start = var_offset + dm.NLP_x + 1, ui_end = var_offset + dm.NLP_x + dm.NLP_u;
            int64_t xip1_end = var_offset + blk + dm.NLP_x, lip1 = var_offset + blk + dm.NLP_x;
            ij.block(dyn_start, dyn_end, xi_start, xi_end);                                    // :208-210
            ij.block(dyn_start, dyn_end, ui_start, xip1_end);
            ij.block(dyn_start, dyn_end, v_start, v_end);
            if (p.flags.lagrange) {                                                            // :214-226 (stale "lagrange state" entries)
                if (d.euler_explicit) {
                    ij.block(dyn_lag, dyn_lag, xi_start, ui_end);
                    ij.single(dyn_lag, lip1);
                } else {
                    ij.block(dyn_lag, dyn_lag, li, lip1);
                }
                ij.block(dyn_lag, dyn_lag, v_start, v_end);
            }
            ij.block(path_start, path_end, xi_start, xi_end);                                  // :230-232
            ij.block(path_start, path_end, ui_start, ui_end);
            ij.block(path_start, path_end, v_start, v_end);
            // STRUCTURAL: implicit Euler evaluates the path constraints of node i >= 2 with u(t_i) = U_{i-1} (euler.jl:59-72),
            // which :231 does not list
            if (p.pattern_mode == 1 && !d.euler_explicit && i >= 2)
                ij.block(path_start, path_end, ui_start - blk, ui_end - blk);
        }
    } else if (d.scheme == MIDPOINT) {                                 // midpoint.jl:163-233
        for (int64_t i = 1; i <= N; ++i) {
            int64_t c_block = c_block_step, c_offset = (i - 1) * c_block;
            int64_t var_offset = (i - 1) * blk;
            int64_t xi_start = var_offset + 1;
            int64_t ui_end = var_offset + dm.NLP_x + dm.NLP_u;
            int64_t xip1_end = var_offset + blk + dm.NLP_x;
            ij.block(c_offset + 1, c_offset + dm.NLP_x, xi_start, xip1_end);
            ij.block(c_offset + dm.NLP_x + 1, c_offset + c_block, xi_start, ui_end);
            ij.block(c_offset + 1, c_offset + c_block, v_start, v_end);
        }
    } else {                                                           // irk.jl:315-416 / irk_stagewise.jl:468-558
        const int s = d.stage;
        const int cu = d.stagewise ? dm.NLP_u * s : dm.NLP_u;
        for (int64_t i = 1; i <= N; ++i) {
            int64_t c_block = c_block_step, c_offset = (i - 1) * c_block;
            int64_t dyn_start = c_offset + 1, dyn_end = c_offset + dm.NLP_x;
            int64_t stage_start = c_offset + dm.NLP_x + 1, stage_end = c_offset + (s + 1) * dm.NLP_x;
            int64_t path_start = c_offset + (s + 1) * dm.NLP_x + 1, path_end = c_offset + c_block;
            int64_t var_offset = (i - 1) * blk;
            int64_t xi_start = var_offset + 1, xi_end = var_offset + dm.NLP_x;
            int64_t ui_end = var_offset + dm.NLP_x + cu;
            int64_t ki_start = var_offset + dm.NLP_x + cu + 1, ki_end = var_offset + blk;
            int64_t xip1_end = var_offset + blk + dm.NLP_x;
            ij.block(dyn_start, dyn_end, xi_start, xi_end);
            ij.block(dyn_start, dyn_end, ki_start, xip1_end);
            ij.block(dyn_start, dyn_end, v_start, v_end);
            ij.block(stage_start, stage_end, xi_start, ki_end);
            ij.block(stage_start, stage_end, v_start, v_end);
            ij.block(path_start, path_end, xi_start, ui_end);
            ij.block(path_start, path_end, v_start, v_end);
        }
    }
    // 2. final path constraints, 3. boundary constraints: identical text in midpoint.jl:206-227,
    // irk.jl:383-402, irk_stagewise.jl:526-548 (u(tf) = U_N convention)
    const int cu = d.stagewise ? dm.NLP_u * d.stage : dm.NLP_u;
    int64_t c_offset = N * c_block_step, c_block = d.step_pathcons_block;
    int64_t var_offset = N * blk;
    int64_t xf_start = var_offset + 1, xf_end = var_offset + dm.NLP_x;
    int64_t uf_start = var_offset - blk + dm.NLP_x + 1, uf_end = var_offset - blk + dm.NLP_x + cu;
    ij.block(c_offset + 1, c_offset + c_block, xf_start, xf_end);
    ij.block(c_offset + 1, c_offset + c_block, uf_start, uf_end);
    ij.block(c_offset + 1, c_offset + c_block, v_start, v_end);
    c_offset = N * c_block_step + d.step_pathcons_block;
    c_block = dm.boundary_cons;
    ij.block(c_offset + 1, c_offset + c_block, 1, dm.NLP_x);
    ij.block(c_offset + 1, c_offset + c_block, xf_start, xf_end);
    ij.block(c_offset + 1, c_offset + c_block, v_start, v_end);
    if ((d.stagewise || d.euler) && p.flags.lagrange)                  // irk_stagewise.jl:550-552, euler.jl:257-259 (hazard H2)
        ij.single(p.dim_NLP_constraints, dm.NLP_x);
}

static void hessian_pattern_ij(const Docp& p, IJ& ij) {
    const Disc& d = p.disc;
    const Dims& dm = p.dims;
    const int64_t N = p.steps;
    const int64_t v_start = p.dim_NLP_variables - dm.NLP_v + 1, v_end = p.dim_NLP_variables;
    const int64_t blk = d.step_variables_block;
    ij.block(v_start, v_end, v_start, v_end);
    if (d.scheme == TRAPEZE) {                                         // trapeze.jl:240-303
        for (int64_t i = 1; i <= N; ++i) {
            int64_t var_block = blk * 2, var_offset = (i - 1) * blk;
            ij.block(var_offset + 1, var_offset + var_block, var_offset + 1, var_offset + var_block);
            ij.block(var_offset + 1, var_offset + var_block, v_start, v_end, true);
        }
        if (p.flags.mayer || dm.boundary_cons > 0) {
            int64_t var_offset = N * blk;
            ij.block(1, dm.NLP_x, var_offset + 1, var_offset + dm.NLP_x, true);
        }
        return;
    }
    if (d.euler) {                                                     // euler.jl:270-355
        for (int64_t i = 1; i <= N; ++i) {
            int64_t var_offset = (i - 1) * blk;
            int64_t xi_start = var_offset + 1, xi_end = var_offset + dm.NLP_x;
            int64_t xip1_start = var_offset + blk + 1, xip1_end = var_offset + blk + dm.NLP_x;
            int64_t ui_start = var_offset + dm.NLP_x + 1, ui_end = var_offset + dm.NLP_x + dm.NLP_u;
            ij.block(ui_start, ui_end, ui_start, ui_end);
            ij.block(ui_start, ui_end, v_start, v_end, true);
            if (d.euler_explicit) {
                ij.block(xi_start, xi_end, xi_start, xi_end);
                ij.block(xi_start, xi_end, ui_start, ui_end, true);
                ij.block(xi_start, xi_end, v_start, v_end, true);
            } else {
                ij.block(xip1_start, xip1_end, xip1_start, xip1_end);
                ij.block(xip1_start, xip1_end, ui_start, ui_end, true);
                ij.block(xip1_start, xip1_end, v_start, v_end, true);
                ij.block(xi_start, xi_end, xi_start, xi_end);
                ij.block(xi_start, xi_end, ui_start, ui_end, true);
                ij.block(xi_start, xi_end, v_start, v_end, true);
            }
        }
        int64_t var_offset = N * blk;
        ij.block(1, dm.NLP_x, var_offset + 1, var_offset + dm.NLP_x, true);
        return;
    }
    if (d.scheme == MIDPOINT) {                                        // midpoint.jl:240-300
        for (int64_t i = 1; i <= N; ++i) {
            int64_t var_offset = (i - 1) * blk;
            int64_t xi_start = var_offset + 1, xip1_end = var_offset + blk + dm.NLP_x;
            ij.block(xi_start, xip1_end, xi_start, xip1_end);
            ij.block(xi_start, xip1_end, v_start, v_end, true);
        }
        int64_t var_offset = N * blk;
        ij.block(1, dm.NLP_x, var_offset + 1, var_offset + dm.NLP_x, true);
        return;
    }
    // irk.jl:423-496 / irk_stagewise.jl:565-638
    const int cu = d.stagewise ? dm.NLP_u * d.stage : dm.NLP_u;
    for (int64_t i = 1; i <= N; ++i) {
        int64_t var_offset = (i - 1) * blk;
        int64_t xi_start = var_offset + 1, ki_end = var_offset + blk;
        ij.block(xi_start, ki_end, xi_start, ki_end);
        ij.block(xi_start, ki_end, v_start, v_end, true);
    }
    int64_t var_offset = N * blk;
    int64_t xf_start = var_offset + 1, xf_end = var_offset + dm.NLP_x;
    int64_t uf_start = var_offset - blk + dm.NLP_x + 1, uf_end = var_offset - blk + dm.NLP_x + cu;
    ij.block(xf_start, xf_end, xf_start, xf_end);
    ij.block(uf_start, uf_end, uf_start, uf_end);
    ij.block(xf_start, xf_end, uf_start, uf_end, true);
    ij.block(xf_start, uf_end, v_start, v_end, true);                  // empty range (hazard H3)
    if (p.pattern_mode == 1) ij.block(xf_start, xf_end, v_start, v_end, true);    // STRUCTURAL: the block the comment there intends
    ij.block(uf_start, uf_end, v_start, v_end, true);
    ij.block(1, dm.NLP_x, xf_start, xf_end, true);
}

// Greedy distance-1 column colouring in natural order (stand-in for SparseMatrixColorings' column colouring
// used by ADNLPModels.SparseADJacobian; any valid colouring decompresses to the same values).
static void color_columns(Docp& p, const std::vector<int64_t>& colptr, const std::vector<int64_t>& rowval) {
    const int64_t ncol = p.dim_NLP_variables, nrow = p.dim_NLP_constraints;
    std::vector<std::vector<int>> rowcolors(nrow);
    p.color.assign(ncol, -1);
    p.ncolors = 0;
    std::vector<char> forbidden;
    for (int64_t j = 0; j < ncol; ++j) {
        forbidden.assign(p.ncolors + 1, 0);
        for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k)
            for (int cidx : rowcolors[rowval[k]]) forbidden[cidx] = 1;
        int cidx = 0;
        while (cidx < p.ncolors && forbidden[cidx]) ++cidx;
        if (cidx == p.ncolors) ++p.ncolors;
        p.color[j] = cidx;
        for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k) rowcolors[rowval[k]].push_back(cidx);
    }
}

// pattern mode 2 (":optimized"): the sparsity ADNLPModels detects itself when no pattern is handed over
// (src/collocation.jl:131-134: the `backend` preset -> SparseConnectivityTracer's global operator-overloading tracer over
// c!(c, x) and the Lagrangian).  Restated with the sparse second-order number S2: its gradient / Hessian containers hold an
// entry for every variable (pair) an operation chain touches, whatever the value -- the same conservative, operator-level
// rule (`u * 0` still depends on u; test/ci/test_modeler_solver.jl:32 pins nnzj 4504 / nnzh 5259 for Goddard, midpoint, N = 250).
static void traced_patterns(Docp& p, IJ& jac, std::vector<std::pair<int64_t, int64_t>>& hess_lower);

static void ensure_pattern(Docp& p) {
    if (p.have_pattern) return;
    IJ ij;
    if (p.pattern_mode == 2) {
        std::vector<std::pair<int64_t, int64_t>> hl;
        traced_patterns(p, ij, hl);
        to_csc(ij, p.dim_NLP_variables, p.colptr, p.rowval);
        color_columns(p, p.colptr, p.rowval);      // every structural nonzero is in the traced pattern: colour on it
        p.have_pattern = true;
        return;
    }
    jacobian_pattern_ij(p, ij);
    to_csc(ij, p.dim_NLP_variables, p.colptr, p.rowval);
    // The colouring must see every TRUE structural nonzero or the compressed passes mix columns.  The reference's manual
    // patterns miss some (hazards: trapeze dynamics x v, trapeze.jl:203; implicit Euler path_i x U_{i-1}, euler.jl:59-72 vs
    // :231), so the oracle colours on the pattern plus those blocks and reports the exact partials at the pattern's positions.
    const Disc& d = p.disc;
    const Dims& dm = p.dims;
    const int64_t cb = d.state_stage_eqs_block + d.step_pathcons_block, blk = d.step_variables_block;
    if (d.scheme == TRAPEZE)
        for (int64_t i = 1; i <= p.steps; ++i)
            ij.block((i - 1) * cb + 1, (i - 1) * cb + dm.NLP_x, p.dim_NLP_variables - dm.NLP_v + 1, p.dim_NLP_variables);
    if (d.euler && !d.euler_explicit && dm.path_cons > 0 && dm.NLP_u > 0)
        for (int64_t i = 2; i <= p.steps; ++i)
            ij.block((i - 1) * cb + dm.NLP_x + 1, i * cb, (i - 2) * blk + dm.NLP_x + 1, (i - 2) * blk + dm.NLP_x + dm.NLP_u);
    std::vector<int64_t> ccp, crv;
    to_csc(ij, p.dim_NLP_variables, ccp, crv);
    color_columns(p, ccp, crv);
    p.have_pattern = true;
}

// jac_coord!: one pass of c!(Dual) per colour, decompressed into the pattern's CSC order
template <class P> static void jacobian_colored(Docp& p, const double* xu, double* vals) {
    ensure_pattern(p);
    const int64_t nvar = p.dim_NLP_variables, ncon = p.dim_NLP_constraints;
    std::vector<D1> z(nvar), cz(ncon);
    for (int col = 0; col < p.ncolors; ++col) {
        for (int64_t j = 0; j < nvar; ++j) z[j] = D1(xu[j], p.color[j] == col ? 1.0 : 0.0);
        constraints<P, D1>(p, z.data(), cz.data());
        for (int64_t j = 0; j < nvar; ++j)
            if (p.color[j] == col)
                for (int64_t k = p.colptr[j]; k < p.colptr[j + 1]; ++k) vals[k] = cz[p.rowval[k]].d;
    }
}

// The same coloured passes spread over host threads (OpenMP, one colour at a time per thread; every pass is independent and
// writes its own columns' entries): the multi-core CPU baseline bench.py reports beside the single-thread one.
template <class P> static void jacobian_colored_mt(Docp& p, const double* xu, double* vals, int nthreads) {
    ensure_pattern(p);
    const int64_t nvar = p.dim_NLP_variables, ncon = p.dim_NLP_constraints;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > p.ncolors) nthreads = p.ncolors;
    // every pass allocates its time grid / work arrays (as the reference does); keep those in the per-thread malloc arenas:
    // one mmap + munmap per pass would serialise the threads on the address-space lock and its TLB shoot-downs
    static const int arena_ok = mallopt(M_MMAP_THRESHOLD, 32 << 20) + mallopt(M_TRIM_THRESHOLD, 512 << 20);
    (void)arena_ok;
    const Docp& cp = p;
    // OpenMP keeps its worker threads (and their malloc arenas and warm buffers) between calls
#pragma omp parallel num_threads(nthreads)
    {
        static thread_local std::vector<D1> z, cz;
        z.resize(nvar); cz.resize(ncon);
#pragma omp for schedule(static, 1)
        for (int col = 0; col < cp.ncolors; ++col) {
            for (int64_t j = 0; j < nvar; ++j) z[j] = D1(xu[j], cp.color[j] == col ? 1.0 : 0.0);
            constraints<P, D1>(cp, z.data(), cz.data());
            for (int64_t j = 0; j < nvar; ++j)
                if (cp.color[j] == col)
                    for (int64_t k = cp.colptr[j]; k < cp.colptr[j + 1]; ++k) vals[k] = cz[cp.rowval[k]].d;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// BLOCK mode: fused constraints + Jacobian values, one time step at a time.
// NOT how the reference obtains its Jacobian (that is jacobian_colored above).  Two uses: (1) the "best-effort CPU"
// figure of bench.py (OpenMP over the time steps, direct per-step block Jacobian); (2) an
// independent, fast checker of every Jacobian entry at the full BASELINE sizes, where ~100 coloured passes over 10^6
// variables are too slow for a test.
// Step i is differentiated as a ONE-STEP transcription of the same OCP on the grid {tau_i, tau_i+1}: its variables
// [X_i, U_i.., K_i.., X_i+1, (U_i+1), V] are contiguous in xu apart from V, so the same `constraints` template evaluates
// the step's rows on dense local duals (DN, one partial per local variable).  The final-time path rows come out of the
// last step's sub-problem, the boundary rows are evaluated directly on duals of (X_1, X_N+1, V).
// Not available for implicit Euler (its path rows read the previous step's control): returns false.
// ---------------------------------------------------------------------------------------------
template <class P> static void make_docp(Docp& p, int scheme, int64_t grid_size, const double* time_grid, int64_t time_grid_len);

template <class P, int CAP> static void cons_jac_block_cap(const Docp& cp, const double* xu, double* c, double* vals, int nthreads) {
    using DN = DNT<CAP>;
    const Disc& d = cp.disc;
    const int n = cp.dims.NLP_x, nv = cp.dims.NLP_v, np = cp.dims.path_cons, nb = cp.dims.boundary_cons;
    const int64_t N = cp.steps, nvar = cp.dim_NLP_variables, ncon = cp.dim_NLP_constraints;
    const int blk = d.step_variables_block, cb = d.state_stage_eqs_block + d.step_pathcons_block;
#pragma omp parallel num_threads(nthreads)
    {
        Docp p1;
        p1.problem = cp.problem;
        p1.control_steps = cp.control_steps;
        make_docp<P>(p1, d.scheme, 1, nullptr, 0);
        const int W = (int)p1.dim_NLP_variables;            // local variables of one step
        const int nloc = W - nv;                             // contiguous in xu from (i-1) * blk
        std::vector<DN> z(W), cz(p1.dim_NLP_constraints);
        // scatter the partials of local rows [0, nr) (global rows row0 ...) into the pattern's CSC positions
        auto scatter = [&](int64_t col, int l, int64_t row0, int nr, const DN* rows) {
            const int64_t* b = cp.rowval.data() + cp.colptr[col];
            const int64_t* e = cp.rowval.data() + cp.colptr[col + 1];
            for (const int64_t* it = std::lower_bound(b, e, row0); it != e && *it < row0 + nr; ++it)
                vals[it - cp.rowval.data()] = rows[*it - row0].d[l];
        };
#pragma omp for schedule(static)
        for (int64_t i = 1; i <= N; ++i) {
            p1.normalized_grid[0] = cp.normalized_grid[i - 1]; p1.normalized_grid[1] = cp.normalized_grid[i];
            p1.fixed_grid[0] = cp.fixed_grid[i - 1]; p1.fixed_grid[1] = cp.fixed_grid[i];
            const double* xs = xu + (i - 1) * (int64_t)blk;
            for (int l = 0; l < nloc; ++l) z[l] = DN::variable(xs[l], l);
            for (int k = 0; k < nv; ++k) z[nloc + k] = DN::variable(xu[nvar - nv + k], nloc + k);
            constraints<P, DN>(p1, z.data(), cz.data());
            const int64_t row0 = (i - 1) * (int64_t)cb;
            if (c) for (int r = 0; r < cb; ++r) c[row0 + r] = cz[r].v;
            const bool last = (i == N);
            if (c && last) for (int r = 0; r < np; ++r) c[N * (int64_t)cb + r] = cz[cb + r].v;
            if (vals) {
                for (int l = 0; l < W; ++l) {
                    const int64_t col = l < nloc ? (i - 1) * (int64_t)blk + l : nvar - nv + (l - nloc);
                    scatter(col, l, row0, cb, cz.data());
                    if (last && np > 0) scatter(col, l, N * (int64_t)cb, np, cz.data() + cb);
                }
            }
        }
#pragma omp single
        if (nb > 0) {                                    // boundary rows: phi(X_1, X_N+1, V)   (DOCP_functions.jl:103-111)
            const int Wb = 2 * n + nv;
            std::vector<DN> x0(n), xf(n), v(nv > 0 ? nv : 1), out(nb);
            for (int k = 0; k < n; ++k) { x0[k] = DN::variable(xu[k], k); xf[k] = DN::variable(xu[N * (int64_t)blk + k], n + k); }
            for (int k = 0; k < nv; ++k) v[k] = DN::variable(xu[nvar - nv + k], 2 * n + k);
            P::template boundary<DN>(out.data(), x0.data(), xf.data(), v.data());
            const int64_t row0 = ncon - nb;
            if (c) for (int r = 0; r < nb; ++r) c[row0 + r] = out[r].v;
            if (vals)
                for (int l = 0; l < Wb; ++l) {
                    const int64_t col = l < n ? l : (l < 2 * n ? N * (int64_t)blk + (l - n) : nvar - nv + (l - 2 * n));
                    scatter(col, l, row0, nb, out.data());
                }
        }
    }
}

template <class P> static bool cons_jac_block(Docp& p, const double* xu, double* c, double* vals, int nthreads) {
    ensure_pattern(p);
    const Disc& d = p.disc;
    if (d.euler && !d.euler_explicit) return false;
    if (nthreads < 1) nthreads = 1;
    if (vals) std::memset(vals, 0, sizeof(double) * p.rowval.size());
    const int W = d.step_variables_block + p.dims.NLP_x + (d.final_control ? p.dims.NLP_u : 0) + p.dims.NLP_v;
    const int Wb = 2 * p.dims.NLP_x + p.dims.NLP_v;
    const int need = W > Wb ? W : Wb;
    if (need <= 16) cons_jac_block_cap<P, 16>(p, xu, c, vals, nthreads);
    else if (need <= 32) cons_jac_block_cap<P, 32>(p, xu, c, vals, nthreads);
    else if (need <= 64) cons_jac_block_cap<P, 64>(p, xu, c, vals, nthreads);
    else if (need <= 128) cons_jac_block_cap<P, 128>(p, xu, c, vals, nthreads);
    else return false;
    return true;
}

// one full column of the dense Jacobian (independent of any pattern)
template <class P> static void jacobian_column(const Docp& p, const double* xu, int64_t col, double* out) {
    const int64_t nvar = p.dim_NLP_variables, ncon = p.dim_NLP_constraints;
    std::vector<D1> z(nvar), cz(ncon);
    for (int64_t j = 0; j < nvar; ++j) z[j] = D1(xu[j], j == col ? 1.0 : 0.0);
    constraints<P, D1>(p, z.data(), cz.data());
    for (int64_t r = 0; r < ncon; ++r) out[r] = cz[r].d;
}

template <class P> static void objective_gradient(const Docp& p, const double* xu, double* g) {
    const int64_t nvar = p.dim_NLP_variables;
    std::vector<D1> z(nvar);
    for (int64_t j = 0; j < nvar; ++j) z[j] = D1(xu[j], 0.0);
    for (int64_t j = 0; j < nvar; ++j) {
        z[j].d = 1.0;
        g[j] = objective<P, D1>(p, z.data()).d;
        z[j].d = 0.0;
    }
}

// Lower triangle of DOCP_Hessian_pattern in CSC order: the rows/cols ADNLPModels' sparse Hessian backend reports through
// hess_structure! (NLPModels convention row >= col) and the order hess_coord! fills.
static void ensure_hess_pattern(Docp& p) {
    if (p.have_hess_pattern) return;
    if (p.pattern_mode == 2) {                         // traced lower triangle (see traced_patterns)
        IJ jac;
        std::vector<std::pair<int64_t, int64_t>> hl;   // (col, row), row >= col
        traced_patterns(p, jac, hl);
        std::sort(hl.begin(), hl.end());
        hl.erase(std::unique(hl.begin(), hl.end()), hl.end());
        p.hcolptr.assign(p.dim_NLP_variables + 1, 0);
        p.hrowval.clear();
        size_t k = 0;
        for (int64_t j = 0; j < p.dim_NLP_variables; ++j) {
            for (; k < hl.size() && hl[k].first == j; ++k) p.hrowval.push_back(hl[k].second);
            p.hcolptr[j + 1] = (int64_t)p.hrowval.size();
        }
        p.have_hess_pattern = true;
        return;
    }
    IJ ij;
    hessian_pattern_ij(p, ij);
    std::vector<int64_t> colptr, rowval;
    to_csc(ij, p.dim_NLP_variables, colptr, rowval);
    p.hcolptr.assign(p.dim_NLP_variables + 1, 0);
    p.hrowval.clear();
    for (int64_t j = 0; j < p.dim_NLP_variables; ++j) {
        for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k)
            if (rowval[k] >= j) p.hrowval.push_back(rowval[k]);
        p.hcolptr[j + 1] = (int64_t)p.hrowval.size();
    }
    p.have_hess_pattern = true;
}

// hess_coord!(nlp, x, y, vals; obj_weight): values of  obj_weight * d2 f + sum_i y_i d2 c_i  on the pattern above.
// Entries of the exact Hessian that the pattern does not hold are counted (dropped[0]: structurally present in the
// second-order sweep, dropped[1]: of those, numerically nonzero) -- ADNLPModels would silently lose them too.
template <class P> static void lagrangian_hessian(Docp& p, const double* xu, const double* y, double sigma, double* vals,
                                                  int64_t* dropped) {
    ensure_hess_pattern(p);
    const int64_t nvar = p.dim_NLP_variables, ncon = p.dim_NLP_constraints;
    std::vector<S2> z(nvar), cz(ncon);
    for (int64_t j = 0; j < nvar; ++j) z[j] = S2::variable(xu[j], j);
    S2::HVec acc;
    {
        S2 f = objective<P, S2>(p, z.data());
        for (auto& e : f.h) acc.emplace_back(e.first, sigma * e.second);
    }
    constraints<P, S2>(p, z.data(), cz.data());
    for (int64_t i = 0; i < ncon; ++i)
        for (auto& e : cz[i].h) acc.emplace_back(e.first, y[i] * e.second);
    s2detail::compress(acc);
    const int64_t nnz = (int64_t)p.hrowval.size();
    for (int64_t k = 0; k < nnz; ++k) vals[k] = 0.0;
    dropped[0] = dropped[1] = 0;
    for (auto& e : acc) {
        const int64_t row = (int64_t)(e.first >> 32), col = (int64_t)(e.first & 0xffffffffu);
        const int64_t* b = p.hrowval.data() + p.hcolptr[col];
        const int64_t* en = p.hrowval.data() + p.hcolptr[col + 1];
        const int64_t* it = std::lower_bound(b, en, row);
        if (it != en && *it == row) vals[it - p.hrowval.data()] = e.second;
        else { dropped[0]++; if (e.second != 0.0) dropped[1]++; }
    }
}

// ---------------------------------------------------------------------------------------------
// BLOCK mode of the Hessian of the Lagrangian: the Lagrangian is a sum over time steps (+ the boundary / Mayer point), so
// the sparse second-order sweep of `lagrangian_hessian` can run one step at a time on the one-step sub-problem of
// cons_jac_block (variables seeded with their GLOBAL indices) and in parallel over the steps.  Same number type, same
// `constraints` / `objective` templates; contributions of neighbouring steps to a shared entry are added atomically (their
// order is not fixed: a checker with a tolerance, not bit-reproducible).  Not how the reference obtains it; it exists so that
// EVERY Hessian entry can be checked at the full BASELINE sizes.  Not available for implicit Euler.
// ---------------------------------------------------------------------------------------------
template <class P> static void make_docp(Docp& p, int scheme, int64_t grid_size, const double* time_grid, int64_t time_grid_len);

template <class P> static bool hessian_block(Docp& p, const double* xu, const double* y, double sigma, double* vals, int64_t* dropped, int nthreads) {
    ensure_hess_pattern(p);
    const Disc& d = p.disc;
    if (d.euler && !d.euler_explicit) return false;
    const int n = p.dims.NLP_x, nv = p.dims.NLP_v, np = p.dims.path_cons, nb = p.dims.boundary_cons;
    const int64_t N = p.steps, nvar = p.dim_NLP_variables, ncon = p.dim_NLP_constraints;
    const int blk = d.step_variables_block, cb = d.state_stage_eqs_block + d.step_pathcons_block;
    if (nthreads < 1) nthreads = 1;
    const int64_t nnz = (int64_t)p.hrowval.size();
    for (int64_t k = 0; k < nnz; ++k) vals[k] = 0.0;
    const Docp& cp = p;
    int64_t drop0 = 0, drop1 = 0;
    auto scatter = [&](const S2::HVec& acc, int64_t& d0, int64_t& d1) {
        for (auto& e : acc) {
            const int64_t row = (int64_t)(e.first >> 32), col = (int64_t)(e.first & 0xffffffffu);
            const int64_t* b = cp.hrowval.data() + cp.hcolptr[col];
            const int64_t* en = cp.hrowval.data() + cp.hcolptr[col + 1];
            const int64_t* it = std::lower_bound(b, en, row);
            if (it != en && *it == row) {
                double& dst = vals[it - cp.hrowval.data()];
#pragma omp atomic
                dst += e.second;
            } else { d0++; if (e.second != 0.0) d1++; }
        }
    };
#pragma omp parallel num_threads(nthreads) reduction(+ : drop0, drop1)
    {
        Docp p1;
        p1.problem = cp.problem;
        p1.control_steps = cp.control_steps;
        make_docp<P>(p1, d.scheme, 1, nullptr, 0);
        p1.flags.mayer = false;                              // the Mayer term belongs to the boundary point below
        const int W = (int)p1.dim_NLP_variables, nloc = W - nv;
        std::vector<S2> z(W), cz(p1.dim_NLP_constraints);
#pragma omp for schedule(dynamic, 16)
        for (int64_t i = 1; i <= N; ++i) {
            p1.normalized_grid[0] = cp.normalized_grid[i - 1]; p1.normalized_grid[1] = cp.normalized_grid[i];
            p1.fixed_grid[0] = cp.fixed_grid[i - 1]; p1.fixed_grid[1] = cp.fixed_grid[i];
            const double* xs = xu + (i - 1) * (int64_t)blk;
            for (int l = 0; l < nloc; ++l) z[l] = S2::variable(xs[l], (i - 1) * (int64_t)blk + l);
            for (int k = 0; k < nv; ++k) z[nloc + k] = S2::variable(xu[nvar - nv + k], nvar - nv + k);
            S2::HVec acc;
            if (cp.flags.lagrange) {
                const S2 f = objective<P, S2>(p1, z.data());
                for (auto& e : f.h) acc.emplace_back(e.first, sigma * e.second);
            }
            constraints<P, S2>(p1, z.data(), cz.data());
            const int64_t row0 = (i - 1) * (int64_t)cb;
            for (int r = 0; r < cb; ++r)
                for (auto& e : cz[r].h) acc.emplace_back(e.first, y[row0 + r] * e.second);
            if (i == N)
                for (int r = 0; r < np; ++r)
                    for (auto& e : cz[cb + r].h) acc.emplace_back(e.first, y[N * (int64_t)cb + r] * e.second);
            s2detail::compress(acc);
            scatter(acc, drop0, drop1);
        }
#pragma omp single
        if (nb > 0 || cp.flags.mayer) {                      // boundary rows + Mayer cost: phi(X_1, X_N+1, V), g(X_1, X_N+1, V)
            std::vector<S2> x0(n), xf(n), v(nv > 0 ? nv : 1), out(nb > 0 ? nb : 1);
            for (int k = 0; k < n; ++k) { x0[k] = S2::variable(xu[k], k); xf[k] = S2::variable(xu[N * (int64_t)blk + k], N * (int64_t)blk + k); }
            for (int k = 0; k < nv; ++k) v[k] = S2::variable(xu[nvar - nv + k], nvar - nv + k);
            S2::HVec acc;
            if (nb > 0) {
                P::template boundary<S2>(out.data(), x0.data(), xf.data(), v.data());
                for (int r = 0; r < nb; ++r)
                    for (auto& e : out[r].h) acc.emplace_back(e.first, y[ncon - nb + r] * e.second);
            }
            if (cp.flags.mayer) {
                const S2 g = P::template mayer<S2>(x0.data(), xf.data(), v.data());
                for (auto& e : g.h) acc.emplace_back(e.first, sigma * e.second);
            }
            s2detail::compress(acc);
            scatter(acc, drop0, drop1);
        }
    }
    dropped[0] = drop0; dropped[1] = drop1;
    return true;
}

// ---------------------------------------------------------------------------------------------
// problem registry
// ---------------------------------------------------------------------------------------------
template <class P> struct Tag { using type = P; };
template <class F> static void dispatch(int pid, F&& f) {
    switch (pid) {
        case 0: f(Tag<Goddard>{}); break;
        case 1: f(Tag<GoddardAll>{}); break;
        case 2: f(Tag<DoubleIntegratorPath>{}); break;
        case 3: f(Tag<Quadrotor8>{}); break;
        case 4: f(Tag<Quadrotor12>{}); break;
        case 5: f(Tag<StagewiseScalar>{}); break;
        case 6: f(Tag<EstimateInitialCondition>{}); break;
        case 7: f(Tag<EstimateRotationRate>{}); break;
        case 8: f(Tag<LeastSquaresConstraint>{}); break;
        case 9: f(Tag<DoubleIntegratorFreeT0Tf>{}); break;
        case 10: f(Tag<GoddardAllF0F1>{}); break;
        case 11: f(Tag<AlgalBacterial>{}); break;
        default: throw std::runtime_error("unknown problem id");
    }
}

template <class P> static void traced_patterns_for(const Docp& p, IJ& jac, std::vector<std::pair<int64_t, int64_t>>& hl) {
    const int64_t nvar = p.dim_NLP_variables, ncon = p.dim_NLP_constraints;
    std::vector<S2> z(nvar), cz(ncon);
    for (int64_t j = 0; j < nvar; ++j) z[j] = S2::variable(0.3 + 0.001 * (double)(j % 97), j);      // any point: only the structure is read
    constraints<P, S2>(p, z.data(), cz.data());
    for (int64_t i = 0; i < ncon; ++i) {
        for (auto& e : cz[i].g) jac.single(i + 1, e.first + 1);
        for (auto& e : cz[i].h) hl.emplace_back((int64_t)(e.first & 0xffffffffu), (int64_t)(e.first >> 32));
    }
    const S2 f = objective<P, S2>(p, z.data());
    for (auto& e : f.h) hl.emplace_back((int64_t)(e.first & 0xffffffffu), (int64_t)(e.first >> 32));
}
static void traced_patterns(Docp& p, IJ& jac, std::vector<std::pair<int64_t, int64_t>>& hl) {
    dispatch(p.problem, [&](auto tag) { traced_patterns_for<typename decltype(tag)::type>(p, jac, hl); });
}

template <class P> static void make_docp(Docp& p, int scheme, int64_t grid_size, const double* time_grid, int64_t time_grid_len) {
    if (p.control_steps < 1) throw std::runtime_error("control_steps must be >= 1");
    p.dims = Dims{P::n, P::m, P::nv, P::p, P::bc};
    p.flags = Flags{P::freet0, P::freetf, P::has_lagrange, P::has_mayer, P::maximize};
    // DOCPtime  (DOCP_data.jl:176-214)
    if (time_grid == nullptr) {
        p.steps = (int)grid_size;
        p.normalized_grid.resize(grid_size + 1);
        // collect(LinRange(0, 1, N+1)): element i = (1 - t)*0 + t*1 with t = (i-1)/N
        for (int64_t i = 0; i <= grid_size; ++i) {
            double t = (double)i / (double)grid_size;
            p.normalized_grid[i] = (1 - t) * 0.0 + t * 1.0;
        }
    } else {
        for (int64_t i = 1; i < time_grid_len; ++i)
            if (!(time_grid[i - 1] < time_grid[i])) throw std::invalid_argument("given time grid is not strictly increasing. Aborting...");
        p.steps = (int)(time_grid_len - 1);
        p.normalized_grid.assign(time_grid, time_grid + time_grid_len);
        if (time_grid[0] != 0 || time_grid[time_grid_len - 1] != 1) {
            double t0 = time_grid[0], tf = time_grid[time_grid_len - 1];
            for (auto& t : p.normalized_grid) t = (t - t0) / (tf - t0);
        }
    }
    p.fixed_grid.assign(p.steps + 1, 0.0);
    if (!(P::freet0 || P::freetf)) {
        double t0 = P::template t0<double>(nullptr), tf = P::template tf<double>(nullptr);
        for (int64_t i = 0; i <= p.steps; ++i) p.fixed_grid[i] = t0 + (p.normalized_grid[i] * (tf - t0));
    }
    build_scheme(p, scheme);
    variables_bounds<P>(p);
    constraints_bounds<P>(p);
}

}  // namespace orc

// =================================================================================================
// C interface for the Python test harness (ctypes).  Status: 0 ok, nonzero = error (orc_last_error).
// =================================================================================================
using orc::Docp;
static std::string g_err;

extern "C" {

int orc_create_cs(int problem, int scheme, int64_t grid_size, const double* time_grid, int64_t time_grid_len, int control_steps, void** out);
int orc_create(int problem, int scheme, int64_t grid_size, const double* time_grid, int64_t time_grid_len, void** out) {
    return orc_create_cs(problem, scheme, grid_size, time_grid, time_grid_len, 1, out);
}
// DOCP(ocp, grid_size, control_steps, scheme, time_grid), src/DOCP_data.jl:293
int orc_create_cs(int problem, int scheme, int64_t grid_size, const double* time_grid, int64_t time_grid_len, int control_steps, void** out) {
    try {
        auto p = std::make_unique<Docp>();
        p->problem = problem;
        p->control_steps = control_steps;
        orc::dispatch(problem, [&](auto tag) { orc::make_docp<typename decltype(tag)::type>(*p, scheme, grid_size, time_grid, time_grid_len); });
        *out = p.release();
        return 0;
    } catch (const std::invalid_argument& e) { g_err = e.what(); return 2; }
    catch (const std::exception& e) { g_err = e.what(); return 1; }
}
void orc_destroy(void* h) { delete (Docp*)h; }
const char* orc_last_error() { return g_err.c_str(); }

// out[0..11] = n, m, nv, path, boundary, steps, nvar, ncon, step_variables_block, state_stage_eqs_block, stage, final_control
void orc_dims(void* h, int64_t* out) {
    Docp& p = *(Docp*)h;
    out[0] = p.dims.NLP_x; out[1] = p.dims.NLP_u; out[2] = p.dims.NLP_v; out[3] = p.dims.path_cons; out[4] = p.dims.boundary_cons;
    out[5] = p.steps; out[6] = p.dim_NLP_variables; out[7] = p.dim_NLP_constraints;
    out[8] = p.disc.step_variables_block; out[9] = p.disc.state_stage_eqs_block; out[10] = p.disc.stage; out[11] = p.disc.final_control;
}
// flags out[0..4] = freet0, freetf, lagrange, mayer, max
void orc_flags(void* h, int32_t* out) {
    Docp& p = *(Docp*)h;
    out[0] = p.flags.freet0; out[1] = p.flags.freetf; out[2] = p.flags.lagrange; out[3] = p.flags.mayer; out[4] = p.flags.max;
}
void orc_butcher(void* h, double* a9, double* b3, double* c3) {
    Docp& p = *(Docp*)h;
    for (int i = 0; i < 3; ++i) { b3[i] = p.disc.b[i]; c3[i] = p.disc.c[i]; for (int j = 0; j < 3; ++j) a9[3 * i + j] = p.disc.a[i][j]; }
}
void orc_grids(void* h, double* normalized, double* fixed) {
    Docp& p = *(Docp*)h;
    std::memcpy(normalized, p.normalized_grid.data(), sizeof(double) * (p.steps + 1));
    std::memcpy(fixed, p.fixed_grid.data(), sizeof(double) * (p.steps + 1));
}
void orc_bounds(void* h, double* lvar, double* uvar, double* lcon, double* ucon) {
    Docp& p = *(Docp*)h;
    std::memcpy(lvar, p.var_l.data(), sizeof(double) * p.dim_NLP_variables);
    std::memcpy(uvar, p.var_u.data(), sizeof(double) * p.dim_NLP_variables);
    std::memcpy(lcon, p.con_l.data(), sizeof(double) * p.dim_NLP_constraints);
    std::memcpy(ucon, p.con_u.data(), sizeof(double) * p.dim_NLP_constraints);
}
void orc_initial_guess(void* h, int use_problem_init, double* x0) {
    Docp& p = *(Docp*)h;
    orc::dispatch(p.problem, [&](auto tag) { orc::initial_guess<typename decltype(tag)::type>(p, use_problem_init != 0, x0); });
}
void orc_initial_guess_sampled(void* h, int64_t K, const double* T, const double* Xs, const double* Us, const double* v, double* x0) {
    Docp& p = *(Docp*)h;
    orc::dispatch(p.problem, [&](auto tag) { orc::initial_guess_sampled<typename decltype(tag)::type>(p, K, T, Xs, Us, v, x0); });
}
void orc_constraints(void* h, const double* xu, double* c) {
    Docp& p = *(Docp*)h;
    orc::dispatch(p.problem, [&](auto tag) { orc::constraints<typename decltype(tag)::type, double>(p, xu, c); });
}
double orc_objective(void* h, const double* xu) {
    Docp& p = *(Docp*)h;
    double f = 0;
    orc::dispatch(p.problem, [&](auto tag) { f = orc::objective<typename decltype(tag)::type, double>(p, xu); });
    return f;
}
void orc_gradient(void* h, const double* xu, double* g) {
    Docp& p = *(Docp*)h;
    orc::dispatch(p.problem, [&](auto tag) { orc::objective_gradient<typename decltype(tag)::type>(p, xu, g); });
}
void orc_set_pattern_mode(void* h, int mode) {
    Docp& p = *(Docp*)h;
    if (p.pattern_mode != mode) { p.pattern_mode = mode; p.have_pattern = false; p.have_hess_pattern = false; }
}
int64_t orc_jac_nnz(void* h) {
    Docp& p = *(Docp*)h;
    orc::ensure_pattern(p);
    return (int64_t)p.rowval.size();
}
int orc_jac_ncolors(void* h) {
    Docp& p = *(Docp*)h;
    orc::ensure_pattern(p);
    return p.ncolors;
}
// 0-based CSC
void orc_jac_pattern(void* h, int64_t* colptr, int64_t* rowval) {
    Docp& p = *(Docp*)h;
    orc::ensure_pattern(p);
    std::memcpy(colptr, p.colptr.data(), sizeof(int64_t) * p.colptr.size());
    std::memcpy(rowval, p.rowval.data(), sizeof(int64_t) * p.rowval.size());
}
void orc_jac_coord(void* h, const double* xu, double* vals) {
    Docp& p = *(Docp*)h;
    orc::dispatch(p.problem, [&](auto tag) { orc::jacobian_colored<typename decltype(tag)::type>(p, xu, vals); });
}
void orc_jac_coord_mt(void* h, const double* xu, double* vals, int nthreads) {
    Docp& p = *(Docp*)h;
    orc::dispatch(p.problem, [&](auto tag) { orc::jacobian_colored_mt<typename decltype(tag)::type>(p, xu, vals, nthreads); });
}
// block mode: fused c + Jacobian values (either output may be NULL); returns 0 when the scheme has no block mode
int orc_cons_jac_block(void* h, const double* xu, double* c, double* vals, int nthreads) {
    Docp& p = *(Docp*)h;
    bool ok = false;
    orc::dispatch(p.problem, [&](auto tag) { ok = orc::cons_jac_block<typename decltype(tag)::type>(p, xu, c, vals, nthreads); });
    return ok ? 1 : 0;
}
// block mode of hess_coord (returns 0 when the scheme has none)
int orc_hess_coord_block(void* h, const double* xu, const double* y, double obj_weight, double* vals, int64_t* dropped2, int nthreads) {
    Docp& p = *(Docp*)h;
    bool ok = false;
    orc::dispatch(p.problem, [&](auto tag) { ok = orc::hessian_block<typename decltype(tag)::type>(p, xu, y, obj_weight, vals, dropped2, nthreads); });
    return ok ? 1 : 0;
}
void orc_jac_column(void* h, const double* xu, int64_t col, double* out) {
    Docp& p = *(Docp*)h;
    orc::dispatch(p.problem, [&](auto tag) { orc::jacobian_column<typename decltype(tag)::type>(p, xu, col, out); });
}
// Hessian pattern: number of entries of the full symmetric pattern and of its lower triangle (nnzh as ADNLPModels reports it)
void orc_hess_nnz(void* h, int64_t* full, int64_t* lower) {
    Docp& p = *(Docp*)h;
    orc::IJ ij;
    orc::hessian_pattern_ij(p, ij);
    std::vector<int64_t> colptr, rowval;
    orc::to_csc(ij, p.dim_NLP_variables, colptr, rowval);
    *full = (int64_t)rowval.size();
    int64_t lo = 0;
    for (int64_t j = 0; j < p.dim_NLP_variables; ++j)
        for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k)
            if (rowval[k] >= j) ++lo;
    *lower = lo;
}

int64_t orc_hess_lower_nnz(void* h) {
    Docp& p = *(Docp*)h;
    orc::ensure_hess_pattern(p);
    return (int64_t)p.hrowval.size();
}
// 0-based CSC of the lower triangle
void orc_hess_pattern(void* h, int64_t* colptr, int64_t* rowval) {
    Docp& p = *(Docp*)h;
    orc::ensure_hess_pattern(p);
    std::memcpy(colptr, p.hcolptr.data(), sizeof(int64_t) * p.hcolptr.size());
    std::memcpy(rowval, p.hrowval.data(), sizeof(int64_t) * p.hrowval.size());
}
void orc_hess_coord(void* h, const double* xu, const double* y, double obj_weight, double* vals, int64_t* dropped2) {
    Docp& p = *(Docp*)h;
    orc::dispatch(p.problem, [&](auto tag) { orc::lagrangian_hessian<typename decltype(tag)::type>(p, xu, y, obj_weight, vals, dropped2); });
}

}  // extern "C"
