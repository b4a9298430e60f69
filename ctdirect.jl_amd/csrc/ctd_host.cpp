// ctd_host.cpp -- host model: DOCP sizes, time grid, bounds, initial guess, Jacobian sparsity pattern (CSC) and the
// emit tables of the kernels.  See ctd_host.hpp.
#include "ctd_host.hpp"
#include "ctd_kernel_body.hpp"
#include "ctd_hess_body.hpp"
#include "ctd_jit.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>

namespace ctd {

// status codes (values as in include/ctdirect_hip.h)
enum { ST_OK = 0, ST_EINVAL = 1, ST_EGRID = 2, ST_ESCHEME = 3, ST_EPATTERN = 4, ST_EPROBLEM = 5 };

// ------------------------------------------------------------------------------------------------------
// scheme structs: sizes and Butcher tables
// ------------------------------------------------------------------------------------------------------
static void set_butcher(Layout& L, int s) {
    // Float64 expressions exactly as the reference's constructors write them:
    // irk.jl:41-43 (s=1), irk.jl:77-79 / irk_stagewise.jl:61-64 (s=2), irk.jl:111-119 / irk_stagewise.jl:103-109 (s=3)
    for (double& x : L.a) x = 0.0;
    for (int i = 0; i < 3; ++i) L.b[i] = L.c[i] = 0.0;
    const double r3 = std::sqrt(3.0), r15 = std::sqrt(15.0);
    if (s == 1) {
        L.a[0] = 0.5; L.b[0] = 1.0; L.c[0] = 0.5;
    } else if (s == 2) {
        L.a[0] = 0.25;            L.a[1] = 0.25 - r3 / 6;
        L.a[3] = 0.25 + r3 / 6;   L.a[4] = 0.25;
        L.b[0] = 0.5; L.b[1] = 0.5;
        L.c[0] = 0.5 - r3 / 6; L.c[1] = 0.5 + r3 / 6;
    } else {
        L.a[0] = 5.0 / 36.0;            L.a[1] = 2.0 / 9 - r15 / 15;  L.a[2] = 5.0 / 36 - r15 / 30;
        L.a[3] = 5.0 / 36.0 + r15 / 24; L.a[4] = 2.0 / 9.0;           L.a[5] = 5.0 / 36.0 - r15 / 24;
        L.a[6] = 5.0 / 36 + r15 / 30;   L.a[7] = 2.0 / 9 + r15 / 15;  L.a[8] = 5.0 / 36.0;
        L.b[0] = 5.0 / 18.0; L.b[1] = 4.0 / 9.0; L.b[2] = 5.0 / 18.0;
        L.c[0] = 0.5 - 0.1 * r15; L.c[1] = 0.5; L.c[2] = 0.5 + 0.1 * r15;
    }
}

static int build_layout(Model& mo, int scheme, int64_t N, int control_steps, std::string& err) {
    Layout& L = mo.L;
    const ProblemInfo& pi = mo.info;
    std::memset(&L, 0, sizeof(L));
    L.scheme = scheme;
    L.cs = control_steps < 1 ? 1 : control_steps;
    // control_steps > 1: the reference sizes every scheme's block with it (trapeze.jl:20, euler.jl:22, irk.jl:141) but only the
    // midpoint scheme integrates over the control sub-steps (midpoint.jl:47-72,99-116,137-155); elsewhere the extra controls would
    // be variables nothing reads.  Offered where it means something.
    if (L.cs > 1 && scheme != 1) { err = "control_steps > 1 is only available with the :midpoint scheme (the only one whose step integrates over the control sub-steps, src/ode/midpoint.jl:137-155)"; return ST_ESCHEME; }
    L.n = pi.n; L.m = pi.m; L.nv = pi.nv; L.p = pi.npath; L.bc = pi.nbc;
    L.N = N;
    L.it0 = pi.it0; L.itf = pi.itf; L.t0 = pi.t0; L.tf = pi.tf;
    L.free_time = (pi.it0 >= 0 || pi.itf >= 0) ? 1 : 0;
    switch (scheme) {
        case 0:   // Trapeze: trapeze.jl:14-42
            L.sc = SC_TRAPEZE; L.s = 0; L.cu = L.m; L.final_control = 1;
            L.blk = L.n + L.m; L.eqs = L.n;
            L.nvar = N * L.blk + L.n + L.nv + L.m;
            break;
        case 1:   // Midpoint: midpoint.jl:17-39 (block = state + control_steps controls, :20)
            L.sc = SC_MIDPOINT; L.s = 0; L.cu = L.m * L.cs;
            L.blk = L.n + L.cu; L.eqs = L.n;
            L.nvar = N * L.blk + L.n + L.nv;
            break;
        case 7: case 8:           // Euler explicit / implicit: euler.jl:19-49 (midpoint's layout, evaluation at t_i or t_{i+1})
            L.sc = SC_MIDPOINT; L.s = 0; L.cu = L.m; L.euler = scheme - 6;
            L.blk = L.n + L.m; L.eqs = L.n;
            L.nvar = N * L.blk + L.n + L.nv;
            break;
        case 2: case 3: case 4:   // Gauss_Legendre_{1,2,3}, constant control: irk.jl:138-160
            L.sc = SC_IRK; L.s = scheme - 1; L.cu = L.m;
            set_butcher(L, L.s);
            L.blk = L.n + L.m + L.n * L.s; L.eqs = L.n * (1 + L.s);
            L.nvar = N * L.blk + L.n + L.nv;
            break;
        case 5: case 6:           // Gauss_Legendre_{2,3}_Stagewise: irk_stagewise.jl:136-163
            L.sc = SC_IRK; L.s = scheme - 3; L.stagewise = 1; L.cu = L.m * L.s;
            set_butcher(L, L.s);
            L.blk = L.n + L.cu + L.s * L.n; L.eqs = L.n * (1 + L.s);
            L.nvar = N * L.blk + L.n + L.nv;
            break;
        default:
            err = "Unknown discretization method (valid: trapeze, midpoint, euler, euler_implicit, gauss_legendre_1, gauss_legendre_2/3[_constant_control])";
            return ST_ESCHEME;
    }
    L.cb = L.eqs + L.p;
    L.ncon = N * L.cb + L.p + L.bc;
    L.v_off = L.nvar - L.nv;
    mo.R = make_rec_layout(L.n, L.m, L.nv, L.p, L.bc, L.s, L.cb, scheme == 1 ? L.cu : L.m, mo.n_f, mo.n_g);
    return ST_OK;
}

// ------------------------------------------------------------------------------------------------------
// DOCPtime: src/DOCP_data.jl:176-214
// ------------------------------------------------------------------------------------------------------
static int build_time(Model& mo, const HostDesc& d, int64_t& N, std::string& err) {
    if (d.time_grid == nullptr) {
        if (d.grid_size < 1) { err = "grid_size must be >= 1"; return ST_EINVAL; }
        N = d.grid_size;
        mo.uniform = true;
        mo.tau.resize(N + 1);
        for (int64_t i = 0; i <= N; ++i) mo.tau[i] = (double)i / (double)N;      // collect(LinRange(0, 1, N+1))
    } else {
        if (d.time_grid_len < 2) { err = "time grid needs at least two points"; return ST_EINVAL; }
        for (int64_t i = 1; i < d.time_grid_len; ++i)
            if (!(d.time_grid[i - 1] < d.time_grid[i])) {
                err = "given time grid is not strictly increasing. Aborting...";   // DOCP_data.jl:187
                return ST_EGRID;
            }
        N = d.time_grid_len - 1;
        mo.uniform = false;
        mo.tau.assign(d.time_grid, d.time_grid + d.time_grid_len);
        if (d.time_grid[0] != 0 || d.time_grid[N] != 1) {
            const double t0 = d.time_grid[0], tf = d.time_grid[N];
            for (double& t : mo.tau) t = (t - t0) / (tf - t0);
        }
    }
    return ST_OK;
}

// ------------------------------------------------------------------------------------------------------
// bounds: __variables_bounds! (DOCP_variables.jl:21-63, irk_stagewise.jl:250-300), __constraints_bounds!
// (DOCP_functions.jl:163-191)
// ------------------------------------------------------------------------------------------------------
static void bounds_block(int dim, const std::vector<BoxItem>& box, std::vector<double>& lb, std::vector<double>& ub) {
    lb.assign(dim, -kInf);
    ub.assign(dim, kInf);
    for (const BoxItem& e : box) { lb[e.index] = e.lb; ub[e.index] = e.ub; }
}

// written straight into the caller's arrays (any of them may be null): nothing of size nvar / ncon is kept on the handle
void Model::fill_bounds(double* var_l, double* var_u, double* con_l, double* con_u) const {
    const ProblemInfo& pi = info;
    std::vector<double> xl, xu, ul, uu, vl, vu;
    bounds_block(L.n, pi.state_box, xl, xu);
    bounds_block(L.m, pi.control_box, ul, uu);
    if (L.nv > 0) bounds_block(L.nv, pi.variable_box, vl, vu);
    auto fill_var = [&](double* dst, const std::vector<double>& xb, const std::vector<double>& ub, const std::vector<double>& vb, double dflt) {
        if (!dst) return;
        for (int64_t k = 0; k < L.nvar; ++k) dst[k] = dflt;
        for (int64_t i = 0; i <= L.N; ++i)
            for (int k = 0; k < L.n; ++k) dst[i * L.blk + k] = xb[k];
        if (L.m > 0) {
            if (L.stagewise) {
                for (int64_t i = 0; i < L.N; ++i)
                    for (int j = 0; j < L.s; ++j)
                        for (int k = 0; k < L.m; ++k) dst[i * L.blk + L.n + j * L.m + k] = ub[k];
            } else {
                const int64_t last = L.final_control ? L.N : L.N - 1;     // set_control_at_time_step!, common.jl:209-223
                for (int64_t i = 0; i <= last; ++i)
                    for (int j = 0; j < L.cs; ++j)                        // DOCP_variables.jl:44-47: every control of the step
                        for (int k = 0; k < L.m; ++k) dst[i * L.blk + L.n + j * L.m + k] = ub[k];
            }
        }
        for (int k = 0; k < L.nv; ++k) dst[L.v_off + k] = vb[k];
    };
    fill_var(var_l, xl, ul, vl, -kInf);
    fill_var(var_u, xu, uu, vu, kInf);
    auto fill_con = [&](double* dst, const std::vector<double>& pb, const std::vector<double>& bb) {
        if (!dst) return;
        for (int64_t k = 0; k < L.ncon; ++k) dst[k] = 0.0;
        int64_t off = 0;
        for (int64_t i = 0; i <= L.N; ++i) {
            if (i < L.N) off += L.eqs;
            for (int k = 0; k < L.p; ++k) dst[off + k] = pb[k];
            off += L.p;
        }
        for (int k = 0; k < L.bc; ++k) dst[off + k] = bb[k];
    };
    fill_con(con_l, pi.path_lb, pi.bc_lb);
    fill_con(con_u, pi.path_ub, pi.bc_ub);
}

// __initial_guess: DOCP_variables.jl:122-145, irk_stagewise.jl:302-335
// trajectory sampled at increasing times, evaluated by linear interpolation (end values held outside the span)
static void sample_at(const InitSamples& sm, const double* data, int dim, double t, double* out) {
    const int64_t K = sm.n;
    if (K == 1 || t <= sm.t[0]) { for (int k = 0; k < dim; ++k) out[k] = data[k]; return; }
    if (t >= sm.t[K - 1]) { for (int k = 0; k < dim; ++k) out[k] = data[(K - 1) * dim + k]; return; }
    const int64_t hi = std::upper_bound(sm.t, sm.t + K, t) - sm.t, lo = hi - 1;      // t[lo] <= t < t[hi]
    const double w = (t - sm.t[lo]) / (sm.t[hi] - sm.t[lo]);
    for (int k = 0; k < dim; ++k) out[k] = data[lo * dim + k] + w * (data[hi * dim + k] - data[lo * dim + k]);
}

void model_initial_guess(const Model& mo, double* X, bool use_default, const double* state, const double* control,
                         const double* variable, const InitSamples& sm) {
    const Layout& L = mo.L;
    const ProblemInfo& pi = mo.info;
    for (int64_t k = 0; k < L.nvar; ++k) X[k] = 0.1;
    std::vector<double> tmp(std::max(std::max(L.n, L.m), std::max(L.nv, 1)));
    // variable first (needed for the time grid when times are free)
    if (L.nv > 0) {
        if (variable) for (int k = 0; k < L.nv; ++k) X[L.v_off + k] = variable[k];
        else if (use_default && pi.init_variable(tmp.data())) for (int k = 0; k < L.nv; ++k) X[L.v_off + k] = tmp[k];
    }
    const double t0 = L.it0 >= 0 ? X[L.v_off + L.it0] : L.t0;
    const double tf = L.itf >= 0 ? X[L.v_off + L.itf] : L.tf;
    auto grid = [&](int64_t i) { return t0 + mo.tau[i] * (tf - t0); };
    auto get_state = [&](double t, double* out) -> bool {
        if (sm.n > 0 && sm.state) { sample_at(sm, sm.state, L.n, t, out); return true; }
        if (state) { for (int k = 0; k < L.n; ++k) out[k] = state[k]; return true; }
        return use_default && pi.init_state(t, out);
    };
    auto get_control = [&](double t, double* out) -> bool {
        if (sm.n > 0 && sm.control) { sample_at(sm, sm.control, L.m, t, out); return true; }
        if (control) { for (int k = 0; k < L.m; ++k) out[k] = control[k]; return true; }
        return use_default && pi.init_control(t, out);
    };
    for (int64_t i = 0; i <= L.N; ++i) {
        const double ti = grid(i);
        if (get_state(ti, tmp.data())) for (int k = 0; k < L.n; ++k) X[i * L.blk + k] = tmp[k];
        if (L.m > 0 && !L.stagewise && (i < L.N || L.final_control))
            if (get_control(ti, tmp.data()))
                for (int j = 0; j < L.cs; ++j)                        // DOCP_variables.jl:138-140: init.control(t_i) for every control of the step
                    for (int k = 0; k < L.m; ++k) X[i * L.blk + L.n + j * L.m + k] = tmp[k];
    }
    if (L.m > 0 && L.stagewise) {
        for (int64_t i = 0; i < L.N; ++i) {
            const double ti = grid(i), hi = grid(i + 1) - ti;
            for (int j = 0; j < L.s; ++j) {
                const double tij = ti + L.c[j] * hi;
                if (get_control(tij, tmp.data()))
                    for (int k = 0; k < L.m; ++k) X[i * L.blk + L.n + j * L.m + k] = tmp[k];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// DOCP_Jacobian_pattern: the add_nonzero_block! calls of each scheme (0-based, half-open)
// ------------------------------------------------------------------------------------------------------
static inline void push_block(std::vector<Block>& out, int64_t r0, int64_t r1, int64_t c0, int64_t c1) {
    if (r1 > r0 && c1 > c0) out.push_back(Block{r0, r1, c0, c1});
}

void Model::step_blocks(int64_t i, std::vector<Block>& out) const {
    const int64_t n = L.n, m = L.m, cb = L.cb, blk = L.blk;
    const int64_t co = i * cb, vo = i * blk, v0 = L.v_off, v1 = L.nvar;
    if (L.sc == SC_TRAPEZE) {                                     // trapeze.jl:168-204
        const int64_t dyn0 = co, dyn1 = co + n, path0 = co + n, path1 = co + cb;
        push_block(out, dyn0, dyn1, vo, vo + n);                  // :191  x_i
        push_block(out, dyn0, dyn1, vo + n, vo + n + m + n);      // :192  u_i, x_i+1
        push_block(out, dyn0, dyn1, vo + 2 * n + m, vo + 2 * n + 2 * m);   // :193  u_i+1
        push_block(out, path0, path1, vo, vo + n);                // :197
        push_block(out, path0, path1, vo + n, vo + n + m);        // :198
        push_block(out, path0, path1, v0, v1);                    // :203  (path rows only: hazard H1)
        if (pattern_mode >= 1) push_block(out, dyn0, dyn1, v0, v1);   // STRUCTURAL: the block the comment at :202 intends
    } else if (L.sc == SC_MIDPOINT) {                             // midpoint.jl:175-204; euler.jl:208-232 lists the same blocks
        push_block(out, co, co + n, vo, vo + blk + n);            // :192-194  x_i, u_i, x_i+1
        push_block(out, co + n, co + cb, vo, vo + n + m);         // :197-199
        push_block(out, co, co + cb, v0, v1);                     // :202
        // STRUCTURAL / OPTIMIZED: implicit Euler evaluates the path constraints of node i >= 1 with u(t_i) = U_{i-1}
        // (euler.jl:59-72), an entry euler.jl:231 does not list
        if (L.euler == 2 && pattern_mode >= 1 && i >= 1) push_block(out, co + n, co + cb, vo - blk + n, vo - blk + n + m);
    } else {                                                      // irk.jl:330-380 / irk_stagewise.jl:483-524
        const int64_t s = L.s;
        const int64_t dyn0 = co, dyn1 = co + n, st0 = co + n, st1 = co + (s + 1) * n, path0 = st1, path1 = co + cb;
        const int64_t xi0 = vo, xi1 = vo + n, ui1 = vo + n + L.cu, ki0 = ui1, ki1 = vo + blk, xip1_1 = vo + blk + n;
        push_block(out, dyn0, dyn1, xi0, xi1);
        push_block(out, dyn0, dyn1, ki0, xip1_1);
        push_block(out, dyn0, dyn1, v0, v1);
        push_block(out, st0, st1, xi0, ki1);
        push_block(out, st0, st1, v0, v1);
        push_block(out, path0, path1, xi0, ui1);
        push_block(out, path0, path1, v0, v1);
    }
}

static void build_tail_blocks(Model& mo) {
    const Layout& L = mo.L;
    const int64_t N = L.N, n = L.n, m = L.m;
    std::vector<Block>& out = mo.tail;
    out.clear();
    // 2. final path constraints (xf, uf, v): trapeze.jl:206-217, midpoint.jl:206-216, irk.jl:383-393, irk_stagewise.jl:526-537
    const int64_t fp0 = N * L.cb, fp1 = fp0 + L.p;
    const int64_t xf0 = N * L.blk, xf1 = xf0 + n;
    push_block(out, fp0, fp1, xf0, xf1);
    if (L.sc == SC_TRAPEZE) push_block(out, fp0, fp1, xf1, xf1 + m);
    else push_block(out, fp0, fp1, (N - 1) * L.blk + n, (N - 1) * L.blk + n + (L.stagewise ? L.cu : m));    // u(tf) = U_N convention (its first control: common.jl:140-155)
    push_block(out, fp0, fp1, L.v_off, L.nvar);
    // 3. boundary constraints (x0, xf, v)
    const int64_t b0 = fp1, b1 = L.ncon;
    push_block(out, b0, b1, 0, n);
    push_block(out, b0, b1, xf0, xf1);
    push_block(out, b0, b1, L.v_off, L.nvar);
    // leftover "lagrange state" entry of the stagewise scheme: irk_stagewise.jl:550-552 (hazard H2)
    // (euler.jl:257-259 has the same leftover)
    if ((L.stagewise || L.euler) && mo.info.lagrange && n > 0) push_block(out, L.ncon - 1, L.ncon, n - 1, n);
}

// rows of column j among the candidate blocks, sorted, each once
void Model::rows_from_blocks(int64_t j, const std::vector<Block>& cand, std::vector<int64_t>& rows) const {
    rows.clear();
    std::vector<std::pair<int64_t, int64_t>> iv;
    for (const Block& b : cand)
        if (j >= b.c0 && j < b.c1) iv.emplace_back(b.r0, b.r1);
    std::sort(iv.begin(), iv.end());
    int64_t next = -1;
    for (auto& p : iv) {
        int64_t r = std::max(p.first, next);
        for (; r < p.second; ++r)
            if (pattern_mode != 2 || opt_dep(r, j)) rows.push_back(r);      // OPTIMIZED: the traced subset of the structural blocks
        next = std::max(next, p.second);
    }
}

void Model::gen_column(int64_t j, std::vector<int64_t>& rows) const {
    std::vector<Block> cand;
    if (j >= L.v_off) {
        for (int64_t i = 0; i < L.N; ++i) step_blocks(i, cand);
    } else {
        const int64_t sj = j / L.blk;
        if (sj - 1 >= 0 && sj - 1 < L.N) step_blocks(sj - 1, cand);
        if (sj < L.N) step_blocks(sj, cand);
        if (L.euler == 2 && pattern_mode >= 1 && sj + 1 < L.N) step_blocks(sj + 1, cand);      // path rows of the next node read U_sj
    }
    for (const Block& b : tail) cand.push_back(b);
    rows_from_blocks(j, cand, rows);
}

// the piece of V column k (an optimisation variable: a row in every step) that the steps [i0, i1) contribute, and / or its tail rows
// (final path + boundary): the table builder looks at a few steps and the tail instead of generating all N of them
void Model::gen_vcolumn_piece(int k, int64_t i0, int64_t i1, bool with_tail, std::vector<int64_t>& rows) const {
    std::vector<Block> cand;
    for (int64_t i = i0; i < i1; ++i) step_blocks(i, cand);
    if (with_tail) for (const Block& b : tail) cand.push_back(b);
    rows_from_blocks(L.v_off + k, cand, rows);
}

// CTD_PATTERN_OPTIMIZED: does constraint `row` depend on variable `col` at the operator level?  The rules restate what a
// global tracer sees when it runs the reference's __constraints! (src/DOCP_functions.jl:80-115 with the scheme's step
// function): own-state terms, the step length h (free times), and the arguments the OCP function is called with --
// x_m = (X_i + X_{i+1}) / 2 (midpoint.jl:53-66), x_ij = X_i + h sum_l a_jl K^l (irk_stagewise.jl:424-446), the b-averaged
// stage control of the path constraints (irk_stagewise.jl:197-205), ...  SURVEY.md Appendix A.6 derives Goddard / midpoint:
// 6 + 8 + 4 entries per step + 4 boundary singletons = 4504 at N = 250 (test/ci/test_modeler_solver.jl:32).
bool Model::opt_dep(int64_t row, int64_t col) const {
    const int64_t N = L.N;
    const int n = L.n, m = L.m, nv = L.nv, vd = n + m;
    uint32_t TV = 0;
    if (L.it0 >= 0) TV |= 1u << (vd + L.it0);
    if (L.itf >= 0) TV |= 1u << (vd + L.itf);
    auto xbit = [&](uint32_t mask, int c) { return (mask >> c & 1u) != 0; };
    auto ubit = [&](uint32_t mask, int c) { return (mask >> (n + c) & 1u) != 0; };
    auto vbit = [&](uint32_t mask, int k) { return (mask >> (vd + k) & 1u) != 0; };
    const uint32_t xmask = (n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u));
    const bool isv = col >= L.v_off;
    const int kv = isv ? (int)(col - L.v_off) : 0;
    if (row >= N * L.cb + L.p) {                       // boundary row: phi_b(X_1, X_{N+1}, v)
        const uint32_t mk = dep_b[row - N * L.cb - L.p];
        if (isv) return (mk >> (2 * n + kv) & 1u) != 0;
        if (col < n) return (mk >> col & 1u) != 0;
        const int64_t q = col - N * L.blk;
        return q >= 0 && q < n && (mk >> (n + q) & 1u) != 0;
    }
    if (row >= N * L.cb) {                             // final-time path row: g_q(t_f, X_{N+1}, u_f, v)
        const uint32_t mk = dep_g[row - N * L.cb];
        if (isv) return vbit(mk, kv);
        const int64_t q = col - N * L.blk;
        if (q >= 0 && q < n) return xbit(mk, (int)q);
        if (L.sc == SC_TRAPEZE) return q >= n && q < n + m && ubit(mk, (int)(q - n));
        const int64_t o = col - ((N - 1) * L.blk + n);                       // u(t_f) := controls of step N
        return o >= 0 && o < (L.stagewise ? L.cu : m) && ubit(mk, (int)(L.stagewise ? o % m : o));
    }
    const int64_t i = row / L.cb;
    const int lr = (int)(row % L.cb);
    const int64_t q = isv ? -1 : col - i * L.blk;      // local column: own block [0, blk), X_{i+1} [blk, blk+n), U_{i+1} (trapeze)
    if (L.euler == 2 && lr >= L.eqs && i >= 1 && !isv) {      // implicit Euler: u(t_i) = U_{i-1} (euler.jl:59-72)
        const uint32_t mk = dep_g[lr - L.eqs];
        if (q >= 0 && q < n) return xbit(mk, (int)q);
        const int64_t o = q + L.blk - n;
        return o >= 0 && o < m && ubit(mk, (int)o);
    }
    if (!isv && (q < 0 || q >= L.blk + n + m)) return false;
    if (lr >= L.eqs) {                                 // path row at node i: g_q(t_i, X_i, u_i, v)
        const uint32_t mk = dep_g[lr - L.eqs];
        if (isv) return vbit(mk, kv);
        if (q < n) return xbit(mk, (int)q);
        if (q < n + (L.stagewise ? L.cu : m)) return ubit(mk, (int)(L.stagewise ? (q - n) % m : q - n));     // stagewise: every stage control (average)
        return false;
    }
    if (L.sc == SC_IRK) {
        const int s = L.s;
        if (lr < n) {                                  // state row: X_{i+1} - (X_i + h sum_j b_j K^j)
            if (isv) return (TV >> (vd + kv) & 1u) != 0;
            if (q < n) return q == lr;
            if (q < n + L.cu) return false;
            if (q < L.blk) return (q - n - L.cu) % n == lr;
            return q - L.blk == lr && q < L.blk + n;
        }
        const int j = (lr - n) / n, r = (lr - n) % n;  // stage row: K^j - f(t_ij, x_ij, u_ij, v)
        const uint32_t mk = dep_f[r];
        const bool through_x = (mk & xmask) != 0;      // x_ij carries X_i, every K^l and (free times) h
        if (isv) return vbit(mk, kv) || (through_x && (TV >> (vd + kv) & 1u));
        if (q < n) return xbit(mk, (int)q);
        if (q < n + L.cu) {
            if (L.stagewise) return (q - n) / m == j && ubit(mk, (int)((q - n) % m));
            return ubit(mk, (int)(q - n));
        }
        if (q < L.blk) {
            const int l = (int)(q - n - L.cu) / n, c = (int)(q - n - L.cu) % n;
            return (l == j && c == r) || xbit(mk, c);
        }
        (void)s;
        return false;
    }
    // trapeze / midpoint / Euler: X_{i+1} - (X_i + h * (...f...))
    const int r = lr;
    const uint32_t mk = dep_f[r];
    if (isv) return vbit(mk, kv) || (TV >> (vd + kv) & 1u);
    const bool at_i = (L.sc == SC_TRAPEZE) || L.euler == 0 || L.euler == 1;       // f reads X_i (U_i always)
    const bool at_ip1 = (L.sc == SC_TRAPEZE) || L.euler == 0 || L.euler == 2;     // f reads X_{i+1}
    if (q < n) return q == r || (at_i && xbit(mk, (int)q));
    if (q < L.blk) return ubit(mk, (int)((q - n) % (m > 0 ? m : 1)));                 // (every control of the step: control_steps)
    if (q < L.blk + n) return q - L.blk == r || (at_ip1 && xbit(mk, (int)(q - L.blk)));
    return L.sc == SC_TRAPEZE && ubit(mk, (int)(q - L.blk - n));                   // U_{i+1}
}

// ------------------------------------------------------------------------------------------------------
// classification of a Jacobian position -> where its value comes from
// ------------------------------------------------------------------------------------------------------
namespace {
struct Loc { int ci, di, beta; bool data_next; };
const Loc kZero{C_ZERO, 0, 0, false};
inline bool const_coef(int ci) { return ci == C_ZERO || ci == C_ONE || ci == C_NEG1 || (ci >= C_B && ci < C_B + 3); }
}  // namespace

// d(row lr of a step) / d(local variable q); q in [0,blk): own block, [blk, blk+n): X_{i+1}, then U_{i+1} (trapeze)
static Loc local_entry(const Model& mo, int lr, int q, int64_t step) {
    const Layout& L = mo.L;
    const RecLayout& R = mo.R;
    const int n = L.n, m = L.m;
    // record offset of d f_r / d x_c (d f_r / d u_c; c over all control blocks of the step) of eval block j, or -1: structurally
    // zero -- sparse eval blocks hold no slot for it and the entry is the constant part of its code
    auto F = [&](int j, int r, int c) {
        if (mo.n_f < 0) return R.oEval + j * R.eval_sz + R.oF + r * R.ldx + c;
        const int s = mo.map_f[r * n + c];
        return s < 0 ? -1 : R.oEval + j * R.eval_sz + R.oF + s;
    };
    auto G = [&](int j, int r, int c) {
        if (mo.n_g < 0) return R.oEval + j * R.eval_sz + R.oG + r * R.ldg + c;
        const int s = mo.map_g[r * m + c % m];
        return s < 0 ? -1 : R.oEval + j * R.eval_sz + R.oG + (c / m) * mo.n_g + s;
    };
    // coefficient * (a partial of the dynamics) + beta
    auto D = [&](int ci, int di, int beta, bool next) { return di < 0 ? Loc{C_ZERO, 0, beta, false} : Loc{ci, di, beta, next}; };
    const bool is_path = lr >= L.eqs;
    if (L.sc == SC_IRK) {
        // column kind
        int kind, l = 0, c = 0;     // 0 x_i, 1 control, 2 K, 3 x_{i+1}, 4 none
        if (q < n) { kind = 0; c = q; }
        else if (q < n + L.cu) { kind = 1; if (L.stagewise) { l = (q - n) / m; c = (q - n) % m; } else c = q - n; }
        else if (q < L.blk) { kind = 2; l = (q - n - L.cu) / n; c = (q - n - L.cu) % n; }
        else if (q < L.blk + n) { kind = 3; c = q - L.blk; }
        else kind = 4;
        if (is_path) {
            const int pq = lr - L.eqs;
            if (kind == 0) return Loc{C_ONE, R.oPx + pq * R.ldx + c, 0, false};
            if (kind == 1) return Loc{L.stagewise ? C_B + l : C_ONE, R.oPu + pq * R.ldu + c, 0, false};
            return kZero;
        }
        if (lr < n) {   // state equation row
            const int r = lr;
            if (kind == 0) return r == c ? Loc{C_NEG1, 0, 0, false} : kZero;
            if (kind == 2) return r == c ? Loc{C_HB + l, 0, 0, false} : kZero;
            if (kind == 3) return r == c ? Loc{C_ONE, 0, 0, false} : kZero;
            return kZero;
        }
        const int j = (lr - n) / n, r = (lr - n) % n;   // stage equation row
        if (kind == 0) return D(C_NEG1, F(j, r, c), 0, false);
        if (kind == 1) return (!L.stagewise || l == j) ? D(C_NEG1, G(j, r, c), 0, false) : kZero;
        if (kind == 2) return D(C_HA + 3 * j + l, F(j, r, c), (j == l && r == c) ? 1 : 0, false);
        return kZero;
    }
    // trapeze / midpoint
    int kind, c = 0;   // 0 x_i, 1 u_i, 3 x_{i+1}, 5 u_{i+1}, 4 none
    if (q < n) { kind = 0; c = q; }
    else if (q < L.blk) { kind = 1; c = q - n; }
    else if (q < L.blk + n) { kind = 3; c = q - L.blk; }
    else if (q < L.blk + n + m) { kind = 5; c = q - L.blk - n; }
    else kind = 4;
    if (is_path) {
        const int pq = lr - L.eqs;
        if (kind == 0) return Loc{C_ONE, R.oPx + pq * R.ldx + c, 0, false};
        // implicit Euler evaluates the path constraints of node i >= 1 with U_{i-1} (euler.jl:59-72): the pattern's
        // (path_i, U_i) entries are structural zeros there
        if (kind == 1) return ((L.euler == 2 && step >= 1) || c >= m) ? kZero : Loc{C_ONE, R.oPu + pq * R.ldu + c, 0, false};
        return kZero;
    }
    const int r = lr;
    if (L.sc == SC_MIDPOINT && L.euler == 1) {       // x_{i+1} - (x_i + h f(t_i, x_i, u_i)): euler.jl:141-159
        if (kind == 0) return D(C_NH, F(0, r, c), r == c ? 2 : 0, false);
        if (kind == 1) return D(C_NH, G(0, r, c), 0, false);
        if (kind == 3) return Loc{C_ZERO, 0, r == c ? 1 : 0, false};
        return kZero;
    }
    if (L.sc == SC_MIDPOINT && L.euler == 2) {       // x_{i+1} - (x_i + h f(t_{i+1}, x_{i+1}, u_i))
        if (kind == 0) return Loc{C_ZERO, 0, r == c ? 2 : 0, false};
        if (kind == 1) return D(C_NH, G(0, r, c), 0, false);
        if (kind == 3) return D(C_NH, F(0, r, c), r == c ? 1 : 0, false);
        return kZero;
    }
    if (L.sc == SC_MIDPOINT) {
        if (kind == 0) return D(C_NHH, F(0, r, c), r == c ? 2 : 0, false);
        if (kind == 1) return D(C_NH, G(0, r, c), 0, false);
        if (kind == 3) return D(C_NHH, F(0, r, c), r == c ? 1 : 0, false);
        return kZero;
    }
    if (kind == 0) return D(C_NHH, F(0, r, c), r == c ? 2 : 0, false);
    if (kind == 1) return D(C_NHH, G(0, r, c), 0, false);
    if (kind == 3) return D(C_NHH, F(0, r, c), r == c ? 1 : 0, true);
    if (kind == 5) return D(C_NHH, G(0, r, c), 0, true);
    return kZero;
}

// d(row lr of a step) / d v_k
static Loc local_entry_v(const Model& mo, int lr, int k) {
    const Layout& L = mo.L;
    const RecLayout& R = mo.R;
    const int n = L.n, nv = L.nv;
    if (lr >= L.eqs) return Loc{C_ONE, R.oPv + (lr - L.eqs) * nv + k, 0, false};
    if (lr < n) return Loc{C_ONE, R.oSv + lr * nv + k, 0, false};
    const int j = (lr - n) / n, r = (lr - n) % n;
    return Loc{C_NEG1, R.oEval + j * R.eval_sz + R.oW + r * nv + k, 0, false};
}

Model::Entry Model::classify(int64_t row, int64_t col) const {
    const RecLayout& Rr = R;
    const int64_t N = L.N;
    const int n = L.n, m = L.m, nv = L.nv;
    Entry e{0, 0, 0, C_ZERO, 0, 0, true};
    auto set = [&](const Loc& lc, int64_t step) {
        e.cstep = step;
        e.dstep = lc.data_next ? step + 1 : step;
        e.ci = lc.ci; e.di = lc.di; e.beta = lc.beta;
        e.cconst = const_coef(lc.ci);
    };
    if (row < N * L.cb) {
        const int64_t i = row / L.cb;
        const int lr = (int)(row % L.cb);
        e.kind = 0;
        e.cstep = e.dstep = i;
        if (col >= L.v_off) { set(local_entry_v(*this, lr, (int)(col - L.v_off)), i); return e; }
        const int64_t q = col - i * L.blk;
        if (L.euler == 2 && lr >= L.eqs && i >= 1 && q >= n - L.blk && q < n + m - L.blk) {
            // implicit Euler: d path_i / d U_{i-1} (the control the path constraints of node i are evaluated with)
            set(Loc{C_ONE, Rr.oPu + (lr - L.eqs) * Rr.ldu + (int)(q + L.blk - n), 0, false}, i);
            return e;
        }
        if (q < 0 || q >= L.blk + n + m || col >= L.v_off) return e;   // not a local variable of this step: zero
        set(local_entry(*this, lr, (int)q, i), i);
        return e;
    }
    if (row < N * L.cb + L.p) {   // final path row
        const int pq = (int)(row - N * L.cb);
        e.kind = 1;
        if (col >= L.v_off) { e.ci = C_ONE; e.di = Rr.oPv + pq * nv + (int)(col - L.v_off); return e; }
        const int64_t xf0 = N * L.blk;
        if (col >= xf0 && col < xf0 + n) { e.ci = C_ONE; e.di = Rr.oPx + pq * Rr.ldx + (int)(col - xf0); return e; }
        if (L.sc == SC_TRAPEZE) {
            if (col >= xf0 + n && col < xf0 + n + m) { e.ci = C_ONE; e.di = Rr.oPu + pq * Rr.ldu + (int)(col - xf0 - n); }
            return e;
        }
        const int64_t u0 = (N - 1) * L.blk + n;
        if (col >= u0 && col < u0 + (L.stagewise ? L.cu : m)) {
            const int o = (int)(col - u0);
            if (L.stagewise) { e.ci = C_B + o / m; e.di = Rr.oPu + pq * Rr.ldu + o % m; }
            else { e.ci = C_ONE; e.di = Rr.oPu + pq * Rr.ldu + o; }
        }
        return e;
    }
    // boundary row
    const int r = (int)(row - N * L.cb - L.p);
    e.kind = 2;
    const int64_t xf0 = N * L.blk;
    if (col >= L.v_off) { e.ci = C_ONE; e.di = Rr.oBv + r * nv + (int)(col - L.v_off); return e; }
    if (col < n) { e.ci = C_ONE; e.di = Rr.oB0 + r * Rr.ldx + (int)col; return e; }
    if (col >= xf0 && col < xf0 + n) { e.ci = C_ONE; e.di = Rr.oBf + r * Rr.ldx + (int)(col - xf0); return e; }
    return e;
}

int64_t Model::column_start(int64_t j) const {
    const int64_t h = reg_first * L.blk, t = reg_last * L.blk;
    if (j < h) return cp_head[j];
    if (j < t) {
        const int64_t i = j / L.blk;
        return seg_base + (i - reg_first) * (int64_t)Lseg + cp_tmpl[j - i * L.blk];
    }
    return cp_tail[j - t];
}

// Columns of row r, sorted: the same union of add_nonzero_block! rectangles, read along a row (CSR order of the value array).
// A step row only meets the blocks of its own step (step_blocks(i) lists the U_{i-1} columns of implicit Euler's path rows too),
// the p + bc tail rows only the tail blocks.
void Model::gen_row(int64_t r, std::vector<int64_t>& cols) const {
    cols.clear();
    std::vector<Block> cand;
    if (r < L.N * (int64_t)L.cb) step_blocks(r / L.cb, cand);
    else cand = tail;
    std::vector<std::pair<int64_t, int64_t>> iv;
    for (const Block& b : cand)
        if (r >= b.r0 && r < b.r1) iv.emplace_back(b.c0, b.c1);
    std::sort(iv.begin(), iv.end());
    int64_t next = -1;
    for (auto& p : iv) {
        int64_t c = std::max(p.first, next);
        for (; c < p.second; ++c)
            if (pattern_mode != 2 || opt_dep(r, c)) cols.push_back(c);
        next = std::max(next, p.second);
    }
}

int64_t Model::row_start(int64_t r) const {
    const int64_t h = reg_first * L.cb, t = reg_last * L.cb;
    if (r < h) return cp_head[r];
    if (r < t) {
        const int64_t i = r / L.cb;
        return seg_base + (i - reg_first) * (int64_t)Lseg + cp_tmpl[r - i * L.cb];
    }
    return cp_tail[r - t];
}

// CSC: the contiguous range of the shard's step columns (its slices of the V columns and, for the neighbours of a one-point
// scheme, nothing else).  CSR: the shard's ONE range -- its step rows; the last shard's range runs to the end of the array (the
// p + bc tail rows follow the step rows)
int64_t Model::shard_vals_begin(int64_t step_begin) const {
    return order == 1 ? row_start(step_begin * L.cb) : column_start(step_begin * L.blk);
}
int64_t Model::shard_vals_end(int64_t step_end) const {
    if (order == 1) return step_end == L.N ? nnzj : row_start(step_end * L.cb);
    return column_start(step_end * L.blk);
}

// relative template codes of the step-periodic segment of step i
static bool segment_codes(const Model& mo, int64_t i, std::vector<uint32_t>& codes, std::vector<int64_t>& cp,
                          std::vector<int64_t>& relrows, int& need_prev) {
    const Layout& L = mo.L;
    codes.clear(); relrows.clear();
    cp.assign(L.blk + 1, 0);
    std::vector<int64_t> rows;
    for (int lc = 0; lc < L.blk; ++lc) {
        const int64_t col = i * L.blk + lc;
        mo.gen_column(col, rows);
        for (int64_t row : rows) {
            Model::Entry e = mo.classify(row, col);
            if (e.kind != 0) return false;
            int64_t crel = i - e.cstep, drel = i - e.dstep;
            if (e.cconst) crel = 0;
            if (e.di == 0) drel = 0;
            if (crel < 0 || crel > 1 || drel < -1 || drel > 1) return false;
            if (crel == 1 || drel == 1) need_prev = 1;
            if (drel == -1) need_prev |= 2;            // data of the NEXT step's record (record code kRecNext)
            codes.push_back(pack_code(e.di, e.ci, e.beta, drel == -1 ? kRecNext : (int)drel, (int)crel));
            relrows.push_back(row - i * L.cb);
        }
        cp[lc + 1] = (int64_t)codes.size();
    }
    return true;
}

static int build_tables(Model& mo, std::string& err) {
    const Layout& L = mo.L;
    const int64_t N = L.N;
    if (L.nv > kMaxNV) { err = "more than 4 optimisation variables are not supported by the emit tables"; return ST_EPATTERN; }
    if (mo.R.stride >= 65536 || mo.R.bsize >= 65536) { err = "per-step record too large for 16-bit data indices"; return ST_EPATTERN; }
    // halo of a tile: trapeze reads the next node and the previous step's coefficients; midpoint the previous step
    mo.HL = (L.sc == SC_IRK) ? 0 : 1;
    mo.HH = (L.sc == SC_TRAPEZE) ? 1 : 0;

    // ---- regular range --------------------------------------------------------------------------------
    std::vector<uint32_t> c1, c2;
    std::vector<int64_t> cp1, cp2, rr1, rr2;
    int need_prev = 0;
    mo.reg_first = mo.reg_last = N;    // all-edge mode by default (small N)
    if (N >= 5) {
        bool ok = segment_codes(mo, 1, c1, cp1, rr1, need_prev);
        if (ok) {
            for (int64_t chk : {(int64_t)2, N - 2}) {
                int np2 = 0;
                if (!segment_codes(mo, chk, c2, cp2, rr2, np2) || c2 != c1 || cp2 != cp1 || rr2 != rr1) { ok = false; break; }
            }
        }
        if (!ok) { err = "Jacobian pattern is not step-periodic"; return ST_EPATTERN; }
        if ((need_prev & 1) && mo.HL == 0) { err = "internal: template needs the previous step but the tile has no halo"; return ST_EPATTERN; }
        if (need_prev & 2) mo.HH = 1;                  // implicit Euler: the path rows of node i+1 sit in the columns of U_i
        mo.reg_first = 1;
        mo.reg_last = N - 1;
        int np2 = 0;
        if (segment_codes(mo, N - 1, c2, cp2, rr2, np2) && c2 == c1 && cp2 == cp1 && rr2 == rr1) mo.reg_last = N;
        mo.tmpl = c1;
        mo.cp_tmpl = cp1;
        mo.Lseg = (int)c1.size();
    } else {
        mo.tmpl.clear(); mo.cp_tmpl.assign(L.blk + 1, 0); mo.Lseg = 0;
    }

    // ---- column starts ----------------------------------------------------------------------------------
    std::vector<int64_t> rows;
    const int64_t head_cols = mo.reg_first * L.blk;
    mo.cp_head.assign(head_cols + 1, 0);
    int64_t nz = 0;
    for (int64_t j = 0; j < head_cols; ++j) { mo.cp_head[j] = nz; mo.gen_column(j, rows); nz += (int64_t)rows.size(); }
    mo.cp_head[head_cols] = nz;
    mo.seg_base = nz;
    nz += (mo.reg_last - mo.reg_first) * (int64_t)mo.Lseg;
    const int64_t tail0 = mo.reg_last * L.blk;
    const int64_t tail_cols = L.nvar - tail0;
    mo.cp_tail.assign(tail_cols + 1, 0);
    // V columns (one row set per step + tail rows): the local rows of a step come from step 0, the periodicity the tiles rely on is
    // checked on sample steps (the step blocks are the same function of the step offset for every step), the tail rows apart --
    // nothing of size N is generated here (N = 10^6 .. 10^7 steps: seconds and gigabytes otherwise)
    std::vector<std::vector<int64_t>> vtail(L.nv);
    std::vector<int> lrows;
    mo.vr = 0;
    if (L.nv > 0) {
        mo.gen_vcolumn_piece(0, 0, 1, false, rows);
        for (int64_t r : rows) lrows.push_back((int)r);
        mo.vr = (int)lrows.size();
        const int64_t samples[] = {0, 1, 2, N / 2, N - 2, N - 1};
        for (int k = 0; k < L.nv; ++k) {
            for (int64_t i : samples) {
                if (i < 0 || i >= N) continue;
                mo.gen_vcolumn_piece(k, i, i + 1, false, rows);
                bool same = (int)rows.size() == mo.vr;
                for (int e = 0; same && e < mo.vr; ++e) same = rows[e] == i * L.cb + lrows[e];
                if (!same) { err = "V column is not step-periodic"; return ST_EPATTERN; }
            }
            mo.gen_vcolumn_piece(k, 0, 0, true, vtail[k]);
        }
    }
    for (int64_t jj = 0; jj < tail_cols; ++jj) {
        const int64_t j = tail0 + jj;
        mo.cp_tail[jj] = nz;
        if (j >= L.v_off) { nz += N * (int64_t)mo.vr + (int64_t)vtail[j - L.v_off].size(); continue; }
        mo.gen_column(j, rows);
        nz += (int64_t)rows.size();
    }
    mo.cp_tail[tail_cols] = nz;
    mo.nnzj = nz;

    // ---- V columns: per-step periodic part ----------------------------------------------------------------
    mo.vtmpl.clear();
    for (int k = 0; k < L.nv; ++k) {
        mo.vcol_base[k] = mo.column_start(L.v_off + k);
        for (int e = 0; e < mo.vr; ++e) {
            Loc lc = local_entry_v(mo, lrows[e], k);
            mo.vtmpl.push_back(pack_code(lc.di, lc.ci, lc.beta, 0, 0));
        }
    }

    // ---- edge entries -----------------------------------------------------------------------------------------
    struct Raw { int64_t idx; Model::Entry e; };
    std::vector<Raw> first, last;
    std::set<int64_t> need;
    need.insert(0);
    need.insert(N - 1);
    auto scan_col = [&](int64_t j, std::vector<Raw>& dst) {
        const bool vcol = j >= L.v_off;                          // V-column step rows are emitted by the tiles: only its tail rows here
        if (!vcol) mo.gen_column(j, rows);
        const std::vector<int64_t>& rws = vcol ? vtail[j - L.v_off] : rows;
        const int64_t base = mo.column_start(j) + (vcol ? N * (int64_t)mo.vr : 0);
        for (size_t t = 0; t < rws.size(); ++t) {
            const int64_t row = rws[t];
            Model::Entry e = mo.classify(row, j);
            if (e.kind == 0) {
                if (!e.cconst) need.insert(e.cstep);
                if (e.di != 0) need.insert(e.dstep);
            }
            dst.push_back(Raw{base + (int64_t)t, e});
        }
    };
    // head entries in column order, i.e. grouped by step: head_ptr[s] = entries before the columns of step s (a shard emits
    // the groups of its own steps -- for N < 5 every step column is irregular and sits here)
    mo.head_ptr.assign(mo.reg_first + 1, 0);
    for (int64_t j = 0; j < head_cols; ++j) {
        if (j % L.blk == 0) mo.head_ptr[j / L.blk] = (int)first.size();
        scan_col(j, first);
    }
    mo.head_ptr[mo.reg_first] = (int)first.size();
    for (int64_t j = tail0; j < L.nvar; ++j) scan_col(j, last);
    if ((int)need.size() > kMaxEdgeSlots) { err = "internal: too many edge records"; return ST_EPATTERN; }
    std::map<int64_t, int> slot_of;
    mo.n_edge_slots = 0;
    for (int64_t s : need) { slot_of[s] = mo.n_edge_slots; mo.edge_steps[mo.n_edge_slots++] = s; }
    mo.edge_fp = mo.n_edge_slots;
    mo.edge_b = mo.n_edge_slots + 1;
    mo.edge_slot_first = slot_of[0];
    mo.edge_slot_last = slot_of[N - 1];
    mo.edge_idx.clear(); mo.edge_code.clear();
    auto emit = [&](const std::vector<Raw>& src) {
        for (const Raw& r : src) {
            const Model::Entry& e = r.e;
            int crec, drec;
            if (e.kind == 0) {
                drec = (e.di != 0) ? slot_of[e.dstep] : slot_of[0];
                crec = (!e.cconst) ? slot_of[e.cstep] : drec;
            } else {
                crec = drec = (e.kind == 1) ? mo.edge_fp : mo.edge_b;
            }
            mo.edge_idx.push_back(r.idx);
            mo.edge_code.push_back(pack_code(e.di, e.ci, e.beta, drec, crec));
        }
    };
    emit(first);
    mo.edge_split = (int)mo.edge_idx.size();
    // tail of c: final path values and boundary values (computed by every shard)
    for (int q = 0; q < L.p; ++q) {
        mo.edge_idx.push_back(kEdgeCBit | (N * L.cb + q));
        mo.edge_code.push_back(pack_code(mo.R.oR + q, C_ONE, 0, mo.edge_fp, mo.edge_fp));
    }
    for (int r = 0; r < L.bc; ++r) {
        mo.edge_idx.push_back(kEdgeCBit | (N * L.cb + L.p + r));
        mo.edge_code.push_back(pack_code(mo.R.oBval + r, C_ONE, 0, mo.edge_b, mo.edge_b));
    }
    mo.edge_split2 = (int)mo.edge_idx.size();
    emit(last);

    // ---- early emission (Gauss-Legendre schemes whose OCP functions are differentiated in one pass) -----------------------
    // a position is early when its value only reads the step's own record at fields the lead role fills: the constant 1.0, the
    // state rows R[0, n), their d/dv Sv, and any coefficient
    mo.pos_order.clear();
    mo.n_late = mo.Lseg; mo.n_early = mo.c_early = mo.vr_early = 0;
    if (L.sc == SC_IRK && mo.fused && mo.Lseg > 0 && mo.Lseg < 65536) {
        std::vector<uint16_t> late, early;
        for (int k = 0; k < mo.Lseg; ++k) {
            const uint32_t c = mo.tmpl[k];
            const int di = code_di(c);
            const bool lead_field = di == 0 || (di >= mo.R.oR && di < mo.R.oR + L.n) || (di >= mo.R.oSv && di < mo.R.oSv + L.n * L.nv);
            (lead_field && code_drec_raw(c) == 0 && code_crec(c) == 0 ? early : late).push_back((uint16_t)k);
        }
        int vre = 0;
        if (L.nv > 0 && mo.vr > 0)
            for (int r : lrows) { if (r < L.n) ++vre; else break; }       // (rows of a V column are sorted: the state rows lead)
        if (!early.empty()) {
            mo.n_late = (int)late.size(); mo.n_early = (int)early.size();
            mo.c_early = L.n; mo.vr_early = vre;
            mo.pos_order = late;
            mo.pos_order.insert(mo.pos_order.end(), early.begin(), early.end());
        }
    }

    return ST_OK;
}

// ------------------------------------------------------------------------------------------------------
// CSR value order (ctd_desc.value_order = CTD_ORDER_CSR).  The reference assembles (Is, Js) and lets SparseArrays.sparse sort them
// by column (midpoint.jl:229-232, irk_stagewise.jl:555-558); a GPU KKT consumer (rocSPARSE / hipSOLVER) takes rows.  Read by rows
// the pattern is simpler than by columns: row r of step i only meets columns of its own block, of X_{i+1} (U_{i+1} for trapeze,
// U_{i-1} for implicit Euler's path rows) and V, so step i owns ONE contiguous range of Lseg values -- V entries inline -- for
// EVERY step (no irregular first / last step columns, no V streams), and only the p + bc tail rows are explicit edge entries.
// The emit phase is unchanged: the same 32-bit codes, one template of Lseg codes per step, kp.vr = 0.
// ------------------------------------------------------------------------------------------------------
static bool row_segment_codes(const Model& mo, int64_t i, std::vector<uint32_t>& codes, std::vector<int64_t>& rp,
                              std::vector<int64_t>& relcols, int& need) {
    const Layout& L = mo.L;
    codes.clear(); relcols.clear();
    rp.assign(L.cb + 1, 0);
    std::vector<int64_t> cols;
    for (int lr = 0; lr < L.cb; ++lr) {
        const int64_t row = i * L.cb + lr;
        mo.gen_row(row, cols);
        for (int64_t col : cols) {
            Model::Entry e = mo.classify(row, col);
            if (e.kind != 0) return false;
            int64_t crel = i - e.cstep, drel = i - e.dstep;
            if (e.cconst) crel = 0;
            if (e.di == 0) drel = 0;
            if (crel != 0 || drel < -1 || drel > 0) return false;      // a row reads its own step's record (trapeze: + the next node's)
            if (drel == -1) need |= 2;
            codes.push_back(pack_code(e.di, e.ci, e.beta, drel == -1 ? kRecNext : 0, 0));
            relcols.push_back(col >= L.v_off ? ((int64_t)1 << 40) + (col - L.v_off) : col - i * L.blk);
        }
        rp[lr + 1] = (int64_t)codes.size();
    }
    return true;
}

static int build_tables_csr(Model& mo, std::string& err) {
    const Layout& L = mo.L;
    const int64_t N = L.N;
    if (L.nv > kMaxNV) { err = "more than 4 optimisation variables are not supported by the emit tables"; return ST_EPATTERN; }
    if (mo.R.stride >= 65536 || mo.R.bsize >= 65536) { err = "per-step record too large for 16-bit data indices"; return ST_EPATTERN; }
    // halo records of a tile: rows never read the previous step's record; implicit Euler's path rows are EVALUATED with U_{i-1}, which
    // the staged tiles find in the previous slot (path_control); trapeze reads the next node's record
    mo.HL = (L.euler == 2) ? 1 : 0;
    mo.HH = (L.sc == SC_TRAPEZE) ? 1 : 0;
    std::vector<uint32_t> c1, c2;
    std::vector<int64_t> rp1, rp2, rc1, rc2;
    int need = 0;
    mo.vr = 0; mo.vtmpl.clear();
    for (int k = 0; k < kMaxNV; ++k) mo.vcol_base[k] = 0;
    // By rows every step has the same segment -- also the first and the last one, for any N >= 1 (no all-edge mode for tiny grids) --
    // except implicit Euler's step 0 on the structural / optimized patterns (its path rows have no U_{-1} columns): the template is
    // step 1's (step 0's when N = 1), steps that differ from it become explicit edge entries
    {
        const int64_t ref = N >= 2 ? 1 : 0;
        if (!row_segment_codes(mo, ref, c1, rp1, rc1, need)) { err = "Jacobian pattern is not step-local by rows"; return ST_EPATTERN; }
        auto same = [&](int64_t i) { int n2 = 0; return row_segment_codes(mo, i, c2, rp2, rc2, n2) && c2 == c1 && rp2 == rp1 && rc2 == rc1; };
        mo.reg_first = same(0) ? 0 : 1;
        mo.reg_last = same(N - 1) ? N : N - 1;
        if (mo.reg_last < mo.reg_first) mo.reg_last = mo.reg_first;
        for (int64_t chk : {(int64_t)2, N / 2, N - 2})
            if (chk > mo.reg_first && chk < mo.reg_last - 1 && !same(chk)) { err = "Jacobian pattern is not step-periodic by rows"; return ST_EPATTERN; }
        if ((need & 2) && mo.HH == 0) { err = "internal: row template needs the next step's record but the tile has no halo"; return ST_EPATTERN; }
        mo.tmpl = c1;
        mo.cp_tmpl = rp1;
        mo.Lseg = (int)c1.size();
    }
    // ---- row starts ------------------------------------------------------------------------------------------
    std::vector<int64_t> cols;
    const int64_t head_rows = mo.reg_first * L.cb;
    mo.cp_head.assign(head_rows + 1, 0);
    int64_t nz = 0;
    for (int64_t r = 0; r < head_rows; ++r) { mo.cp_head[r] = nz; mo.gen_row(r, cols); nz += (int64_t)cols.size(); }
    mo.cp_head[head_rows] = nz;
    mo.seg_base = nz;
    nz += (mo.reg_last - mo.reg_first) * (int64_t)mo.Lseg;
    const int64_t tail0 = mo.reg_last * L.cb, tail_rows = L.ncon - tail0;
    mo.cp_tail.assign(tail_rows + 1, 0);
    for (int64_t rr = 0; rr < tail_rows; ++rr) { mo.cp_tail[rr] = nz; mo.gen_row(tail0 + rr, cols); nz += (int64_t)cols.size(); }
    mo.cp_tail[tail_rows] = nz;
    mo.nnzj = nz;
    // ---- edge entries: the rows of irregular leading / trailing steps (all steps when N < 5) and the p + bc tail rows ----------
    struct Raw { int64_t idx; Model::Entry e; };
    std::vector<Raw> first, last;
    std::set<int64_t> needrec;
    needrec.insert(0);
    needrec.insert(N - 1);
    auto scan_row = [&](int64_t r, std::vector<Raw>& dst) {
        mo.gen_row(r, cols);
        const int64_t base = mo.row_start(r);
        for (size_t t = 0; t < cols.size(); ++t) {
            Model::Entry e = mo.classify(r, cols[t]);
            if (e.kind == 0) {
                if (!e.cconst) needrec.insert(e.cstep);
                if (e.di != 0) needrec.insert(e.dstep);
            }
            dst.push_back(Raw{base + (int64_t)t, e});
        }
    };
    mo.head_ptr.assign(mo.reg_first + 1, 0);       // head entries grouped by step: a shard emits the groups of its own steps
    for (int64_t r = 0; r < head_rows; ++r) {
        if (r % L.cb == 0) mo.head_ptr[r / L.cb] = (int)first.size();
        scan_row(r, first);
    }
    mo.head_ptr[mo.reg_first] = (int)first.size();
    for (int64_t r = tail0; r < L.ncon; ++r) scan_row(r, last);
    if ((int)needrec.size() > kMaxEdgeSlots) { err = "internal: too many edge records"; return ST_EPATTERN; }
    std::map<int64_t, int> slot_of;
    mo.n_edge_slots = 0;
    for (int64_t s : needrec) { slot_of[s] = mo.n_edge_slots; mo.edge_steps[mo.n_edge_slots++] = s; }
    mo.edge_fp = mo.n_edge_slots;
    mo.edge_b = mo.n_edge_slots + 1;
    mo.edge_slot_first = slot_of[0];
    mo.edge_slot_last = slot_of[N - 1];
    mo.edge_idx.clear(); mo.edge_code.clear();
    auto emit = [&](const std::vector<Raw>& src) {
        for (const Raw& r : src) {
            const Model::Entry& e = r.e;
            int crec, drec;
            if (e.kind == 0) {
                drec = (e.di != 0) ? slot_of[e.dstep] : slot_of[0];
                crec = (!e.cconst) ? slot_of[e.cstep] : drec;
            } else {
                crec = drec = (e.kind == 1) ? mo.edge_fp : mo.edge_b;
            }
            mo.edge_idx.push_back(r.idx);
            mo.edge_code.push_back(pack_code(e.di, e.ci, e.beta, drec, crec));
        }
    };
    emit(first);
    mo.edge_split = (int)mo.edge_idx.size();
    for (int q = 0; q < L.p; ++q) {                // tail of c: final path values and boundary values (computed by every shard)
        mo.edge_idx.push_back(kEdgeCBit | (N * L.cb + q));
        mo.edge_code.push_back(pack_code(mo.R.oR + q, C_ONE, 0, mo.edge_fp, mo.edge_fp));
    }
    for (int r = 0; r < L.bc; ++r) {
        mo.edge_idx.push_back(kEdgeCBit | (N * L.cb + L.p + r));
        mo.edge_code.push_back(pack_code(mo.R.oBval + r, C_ONE, 0, mo.edge_b, mo.edge_b));
    }
    mo.edge_split2 = (int)mo.edge_idx.size();
    emit(last);
    mo.pos_order.clear();                          // (early emission is a CSC-order experiment)
    mo.n_late = mo.Lseg; mo.n_early = mo.c_early = mo.vr_early = 0;
    return ST_OK;
}

// structural nonzeros the selected pattern leaves out (hazard H1; implicit Euler's (path_i, U_{i-1}) entries)
static void count_dropped(Model& mo) {
    const Layout& L = mo.L;
    const int64_t N = L.N;
    mo.dropped = 0;
    if (mo.pattern_mode == 0 && L.sc == SC_TRAPEZE && L.nv > 0 && (L.free_time || mo.dyn_v))
        mo.dropped = N * (int64_t)L.n * L.nv;
    if (L.euler == 2 && L.p > 0 && L.m > 0 && mo.pattern_mode == 0) {
        int64_t pairs = 0;                                  // (path row, control) pairs the path functions really couple
        for (int q = 0; q < L.p; ++q)
            for (int c = 0; c < L.m; ++c) pairs += (mo.dep_g.size() > (size_t)q && (mo.dep_g[q] >> (L.n + c) & 1u)) ? 1 : 0;
        mo.dropped += (N - 1) * pairs;
    }
}

int default_tile(const Model& mo, int64_t nsteps) {
    // steps per 256-lane workgroup, from the MI355X tuning sweeps (profiles/): the largest power of two whose tile fits
    // ~60 KiB of LDS (two workgroups per CU stay resident), at most 32 for the Gauss-Legendre schemes and 64 for
    // trapeze / midpoint (small records).  Small grids: when that would give fewer than ~480 workgroups the tile shrinks
    // (not necessarily to a power of two) so that every one of the 256 CUs holds about two of them -- 10 000 steps run
    // best with 21-step tiles (478 workgroups, 7.9 us vs 8.3 us at 32 steps)
    const Layout& L = mo.L;
    const int64_t per_step = (int64_t)(L.blk + mo.R.stride + 1) * 8;
    const int64_t fit = (60 * 1024) / per_step - mo.HL - mo.HH;
    const int64_t cap = (L.sc == SC_IRK) ? 32 : 64;
    int64_t T = 1;
    while (T * 2 <= fit && T * 2 <= cap) T *= 2;
    // wide OCPs (several direction chunks per evaluation point: the quadrotors).  Their evaluation is bound by FP64 ISSUE, not by
    // latency: the generated dynamics code of a point is ~500 instructions per part, a wave instruction costs 4 cycles whatever
    // the number of active lanes, and a 7-step tile of Gauss-Legendre 3 runs it with 21 of 64 lanes.  With the sparse eval blocks
    // (a 12-state quadrotor step record on Gauss-Legendre 3: 307 doubles instead of 857) a tile holds as many steps as the evaluating wave has lanes
    // for: stage points + path points <= 64.  What limits the tile from above is the OUTPUT per step: the emit phase runs at
    // the chip's write rate only while other workgroups' evaluations overlap it, so tiles that write ~128 KiB measured best
    // (profiles/r03_tile_sweeps.log: 12-state quadrotor GL3 N = 20 000, manual pattern (23 KiB per step) 5 - 8 steps 109 - 112 us,
    // 16 steps 125; optimized pattern (3.7 KiB per step) 7 steps 36.3, 16 steps 25.8; 8-state quadrotor 11 KiB per step: 10 - 12
    // steps; one-point schemes flat from 16 steps on)
    const bool wide = mo.nch_dyn > 1;
    const bool multi_u = wide && L.sc == SC_MIDPOINT && L.cs > 1;
    if (wide) {
        const int pts = L.sc == SC_IRK ? L.s : 1;
        // (several controls per step: the evaluating lane runs the dynamics code once per control, a fixed cost per tile whatever
        // its size, so the FEWEST rounds of resident workgroups win: up to 48 steps on two resident workgroups per CU, then the steps
        // spread evenly over that many rounds of 512 tiles -- profiles/r03_control_steps.md: 12-state quadrotor, 2 controls, N = 20 000: 24 steps = 4 tiles
        // per CU 26.0 us, 32 steps = 3 per CU 28.2, 40 steps = 2 per CU 19.3, 48 steps 19.8; 8-state: 17.1 / 12.2 / 11.4 / 11.9)
        const int64_t lane_cap = L.sc == SC_IRK ? 64 / (pts + 1) : (multi_u ? 48 : 24);
        const double out_bytes = 8.0 * (double)(mo.Lseg + L.cb + (int64_t)L.nv * mo.vr);
        T = std::max<int64_t>(4, std::min<int64_t>(lane_cap, (int64_t)std::lround(131072.0 / std::max(out_bytes, 1.0))));
        if (multi_u) T = lane_cap;
        // ... and the largest such tile that leaves three workgroups per CU (exact LDS accounting: 1280-byte granules)
        for (; T > 4; --T) {
            KParams kp;
            mo.fill_kparams(kp, 0, std::min<int64_t>(L.N, T * 4), (int)T);
            if (wgs_per_cu(lds_doubles(kp) * 8) >= (multi_u ? 2 : 3)) break;
        }
        if (multi_u) {
            const int64_t ns = nsteps > 0 ? nsteps : L.N, rounds = (ns + T * 512 - 1) / (T * 512);     // (rounds of two workgroups per CU)
            T = std::max<int64_t>(4, std::min<int64_t>(T, (ns + rounds * 512 - 1) / (rounds * 512)));
        }
    }
    if (nsteps <= 0) nsteps = L.N;               // steps this handle evaluates (a shard of the grid, or all of it)
    if ((nsteps + T - 1) / T < 480) T = std::max<int64_t>(4, std::min<int64_t>(T, (nsteps + 479) / 480));
    // light steps on long grids (double integrator, midpoint, 100 000 steps: 152 bytes of output per step): ONE round of ~512
    // workgroups (two per CU) when such a tile still fits 80 KiB of LDS -- 196-step tiles 6.1 us vs 64-step tiles 7.6 us
    // (profiles/r02_tile_sweeps.log); heavier steps keep the smaller tile
    // (OCPs with path constraints of their own direction chunks run the staged driver: the same holds up to ~40 KiB --
    // goddard_all, trapeze, N = 20 000: 40 steps 6.1 us, the 8 steps of round 1's rule 12.0 us)
    if (!wide) {
        const int64_t T1 = (nsteps + 511) / 512;
        if (T1 > T && (T1 + mo.HL + mo.HH + 1) * per_step <= (mo.fused ? 80 : 40) * 1024) T = T1;
    }
    return (int)T;
}

void Model::fill_kparams(KParams& kp, int64_t step_begin, int64_t step_end, int tile) const {
    std::memset(&kp, 0, sizeof(kp));
    kp.L = L;
    kp.R = R;
    kp.T = tile;
    kp.HL = HL; kp.HH = HH;
    kp.step_begin = step_begin; kp.step_end = step_end;
    kp.ntiles = (int)((step_end - step_begin + tile - 1) / tile);
    kp.xcd_remap = 0;
    kp.Lseg = Lseg; kp.vr = vr;
    kp.div_cb = make_fastdiv((uint32_t)L.cb);
    kp.div_Lseg = make_fastdiv((uint32_t)(Lseg > 0 ? Lseg : 1));
    kp.div_vr = make_fastdiv((uint32_t)(vr > 0 ? vr : 1));
    kp.div_blk = make_fastdiv((uint32_t)(L.blk > 0 ? L.blk : 1));
    kp.seg_base = seg_base; kp.reg_first = reg_first; kp.reg_last = reg_last;
    for (int k = 0; k < kMaxNV; ++k) kp.vcol_base[k] = vcol_base[k];
    const bool owns_last = step_end == L.N;
    kp.edge_begin = head_ptr[std::min<int64_t>(step_begin, reg_first)];
    kp.edge_end = head_ptr[std::min<int64_t>(step_end, reg_first)];
    kp.edge2_begin = edge_split;                       // tail rows of c (every shard), then the trailing columns (last shard)
    kp.edge2_end = owns_last ? (int)edge_idx.size() : edge_split2;
    {   // edge workgroups: one, or -- long explicit lists (wide OCPs: 6432 entries for the 12-state quadrotor on Gauss-Legendre 3) --
        // one per ~512 entries up to 16, each evaluating the edge records and emitting its share (edge_share_begin, ctd_kernel_body.hpp)
        const int nedge = (kp.edge_end - kp.edge_begin) + (kp.edge2_end - kp.edge2_begin);
        int E = nedge > 0 ? 1 : 0;
        if (nedge > 1024) E = std::min(16, (nedge + 511) / 512);
        if (const char* e = std::getenv("CTD_EDGE_BLOCKS")) { const int v = std::atoi(e); if (v >= 1 && nedge > 0) E = std::min(v, 64); }
        kp.has_edge = E;
    }
    kp.n_edge_slots = n_edge_slots;
    kp.edge_fp = edge_fp; kp.edge_b = edge_b;
    kp.edge_slot_first = edge_slot_first; kp.edge_slot_last = edge_slot_last;
    for (int k = 0; k < kMaxEdgeSlots; ++k) kp.edge_steps[k] = edge_steps[k];
    // emit templates in LDS: short periods only, and never at the price of a resident workgroup (12-state quadrotor, optimized
    // pattern: 2 KiB of codes pushed the 7-step tile from three workgroups per CU to two -- the LDS granule is 1280 bytes)
    kp.n_late = Lseg; kp.n_early = kp.c_early = kp.vr_early = 0;      // (the engine switches early emission on: device table + launch geometry)
    kp.div_late = make_fastdiv((uint32_t)(Lseg > 0 ? Lseg : 1));
    kp.stage_codes = (Lseg + L.nv * vr <= kMaxStagedCodes) ? 1 : 0;
    if (kp.stage_codes) {
        const int with = wgs_per_cu(lds_doubles(kp) * 8);
        kp.stage_codes = 0;
        const int without = wgs_per_cu(lds_doubles(kp) * 8);
        kp.stage_codes = (without > with && with < 4) ? 0 : 1;
    }
}

int build_model(const HostDesc& d, Model& mo, std::string& err) {
    mo.problem = d.problem;
    mo.pattern_mode = d.pattern_mode;
    if (d.pattern_mode < 0 || d.pattern_mode > 2) { err = "unknown pattern mode"; return ST_EPATTERN; }
    bool found = for_problem(d.problem, [&](auto tag) {
        using P = typename decltype(tag)::type;
        mo.info = P::info();
        mo.dyn_t = P::DYN_T;
        mo.dyn_v = P::DYN_V;
        mo.nch_dyn = Dirs<P>::NCH_DYN;
        mo.nch_path = (P::NPATH > 0) ? Dirs<P>::NCH_PATH : 0;
        mo.fused = Dirs<P>::FUSED;
        mo.H.hk = HessK<P>::value;
        mo.n_f = mo.n_g = -1;
        if (DynNZ<P>::sparse) {
            mo.n_f = DynNZ<P>::nF; mo.n_g = DynNZ<P>::nG;
            mo.map_f.assign(P::NX * P::NX, -1); mo.map_g.assign(P::NX * P::NU, -1);
            for (int r = 0; r < P::NX; ++r) {
                for (int c = 0; c < P::NX; ++c) mo.map_f[r * P::NX + c] = DynNZ<P>::fx(r, c);
                for (int c = 0; c < P::NU; ++c) mo.map_g[r * P::NU + c] = DynNZ<P>::gu(r, c);
            }
        }
    });
    if (!found) {
        const RtOcp* ro = runtime_ocp(d.problem);          // registered at run time (ctd_register_ocp)
        if (!ro) { err = "problem id not in the compiled registry"; return ST_EPROBLEM; }
        mo.info = ro->info;
        mo.dyn_t = ro->dyn_t;
        mo.dyn_v = ro->dyn_v;
        const int dyn = ro->info.n + ro->info.m + (ro->dyn_t ? 1 : 0) + (ro->dyn_v ? ro->info.nv : 0);
        const int pth = ro->info.n + ro->info.m + (ro->path_t ? 1 : 0) + (ro->path_v ? ro->info.nv : 0);
        mo.nch_dyn = (dyn + ro->dc - 1) / ro->dc;
        mo.nch_path = ro->info.npath > 0 ? (pth + ro->dc - 1) / ro->dc : 0;
        mo.fused = mo.nch_dyn == 1 && (ro->info.npath == 0 || mo.nch_path == 1);      // Dirs<P>::FUSED
        mo.H.hk = ro->hk;
        mo.n_f = mo.n_g = -1;
        if (ro->dyn_nz.sparse) { mo.n_f = ro->dyn_nz.n_f; mo.n_g = ro->dyn_nz.n_g; mo.map_f = ro->dyn_nz.map_f; mo.map_g = ro->dyn_nz.map_g; }
    }
    if (d.scheme < 0 || d.scheme > 8) { err = "Unknown discretization method"; return ST_ESCHEME; }
    int64_t N = 0;
    int st = build_time(mo, d, N, err);
    if (st) return st;
    st = build_layout(mo, d.scheme, N, d.control_steps, err);
    if (st) return st;
    build_tail_blocks(mo);
    compute_dep_masks(mo);          // operator-level dependence masks: the OPTIMIZED pattern, and the count of dropped nonzeros
    if (d.value_order != 0 && d.value_order != 1) { err = "unknown value order (CTD_ORDER_CSC = 0, CTD_ORDER_CSR = 1)"; return ST_EINVAL; }
    mo.order = d.value_order;
    st = mo.order == 1 ? build_tables_csr(mo, err) : build_tables(mo, err);
    if (st) return st;
    count_dropped(mo);
    return build_hess_model(mo, err);
}

}  // namespace ctd
