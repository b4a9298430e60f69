// ctd_hess_host.cpp -- host side of the Hessian of the Lagrangian: the lower triangle of DOCP_Hessian_pattern in CSC
// order (without materialising it for the step-periodic middle) and the term tables the Hessian kernel consumes.
// See ctd_hess.hpp for the decomposition.
#include <cstdlib>
#include "ctd_host.hpp"
#include "ctd_jit.hpp"
#include "ctd_hess_body.hpp"

#include <algorithm>
#include <map>
#include <set>

namespace ctd {

enum { ST_OK = 0, ST_EPATTERN = 4 };

// ------------------------------------------------------------------------------------------------------
// DOCP_Hessian_pattern: the add_nonzero_block! calls of each scheme (0-based, half-open; `sym` blocks pushed both ways)
// ------------------------------------------------------------------------------------------------------
static inline void hpush(std::vector<Block>& out, int64_t r0, int64_t r1, int64_t c0, int64_t c1, bool sym = false) {
    if (r1 > r0 && c1 > c0) {
        out.push_back(Block{r0, r1, c0, c1});
        if (sym) out.push_back(Block{c0, c1, r0, r1});
    }
}

void Model::hess_step_blocks(int64_t i, std::vector<Block>& out) const {
    const int64_t blk = L.blk, vo = i * blk, v0 = L.v_off, v1 = L.nvar;
    if (L.euler) {                                        // euler.jl:297-326
        const int64_t x0 = vo, x1 = vo + L.n, u0 = x1, u1 = x1 + L.m, y0 = vo + blk, y1 = y0 + L.n;
        hpush(out, u0, u1, u0, u1);
        hpush(out, u0, u1, v0, v1, true);
        if (L.euler == 2) {                               // implicit: x_{i+1} carries the dynamics' second-order terms
            hpush(out, y0, y1, y0, y1);
            hpush(out, y0, y1, u0, u1, true);
            hpush(out, y0, y1, v0, v1, true);
        }
        hpush(out, x0, x1, x0, x1);                       // explicit: dynamics + path; implicit: path part (:321-325)
        hpush(out, x0, x1, u0, u1, true);
        hpush(out, x0, x1, v0, v1, true);
        return;
    }
    int64_t hi;
    if (L.sc == SC_TRAPEZE) hi = vo + 2 * blk;            // trapeze.jl:262-281  [X_i U_i X_i+1 U_i+1]
    else if (L.sc == SC_MIDPOINT) hi = vo + blk + L.n;    // midpoint.jl:262-281 [X_i U_i X_i+1]
    else hi = vo + blk;                                   // irk.jl:445-462 / irk_stagewise.jl:587-604 [X_i U_i K_i]
    hpush(out, vo, hi, vo, hi);
    hpush(out, vo, hi, v0, v1, true);
}

static void build_hess_tail(Model& mo) {
    const Layout& L = mo.L;
    const int64_t N = L.N, n = L.n;
    std::vector<Block>& out = mo.H.tail;
    out.clear();
    hpush(out, L.v_off, L.nvar, L.v_off, L.nvar);                       // variable x variable (first call of every scheme)
    const int64_t xf0 = N * L.blk, xf1 = xf0 + n;
    if (L.sc == SC_TRAPEZE) {                                           // trapeze.jl:284-293
        if (mo.info.mayer || L.bc > 0) hpush(out, 0, n, xf0, xf1, true);
        return;
    }
    if (L.sc == SC_MIDPOINT) {                                          // midpoint.jl:284-290, euler.jl:337-343
        hpush(out, 0, n, xf0, xf1, true);
        // explicit Euler's step blocks never reach the final state (euler.jl:297-326), so the reference's pattern has no
        // (x_f, x_f) / (x_f, v) block although the boundary / Mayer / final-time path terms live there; the OPTIMIZED pattern
        // (what a tracer finds) has them
        if (mo.pattern_mode == 2 && L.euler == 1) {
            hpush(out, xf0, xf1, xf0, xf1);
            hpush(out, xf0, xf1, L.v_off, L.nvar, true);
        }
        return;
    }
    // irk.jl:465-486 / irk_stagewise.jl:607-628 (u(tf) = U_N convention)
    const int64_t uf0 = (N - 1) * L.blk + n, uf1 = uf0 + L.cu;
    hpush(out, xf0, xf1, xf0, xf1);
    hpush(out, uf0, uf1, uf0, uf1);
    hpush(out, xf0, xf1, uf0, uf1, true);
    // the reference's (xf_start:uf_end) x v block is an empty range (xf_start > uf_end): nothing pushed.  STRUCTURAL
    // mode adds the block the surrounding comment intends
    if (mo.pattern_mode >= 1) hpush(out, xf0, xf1, L.v_off, L.nvar, true);
    hpush(out, uf0, uf1, L.v_off, L.nvar, true);
    hpush(out, 0, n, xf0, xf1, true);
}

// CTD_PATTERN_OPTIMIZED keeps the entries of the (structural) blocks that receive at least one structurally nonzero term --
// the pairs the dependency-mask probe finds at the evaluation points, mapped through the scheme's chain rule: what a global
// second-order tracer over the Lagrangian reports (nnzh 5259 for Goddard, midpoint, N = 250, test/ci/test_modeler_solver.jl:32)
static bool hess_opt_keep(const Model& mo, int64_t row, int64_t col);

void Model::hess_gen_column(int64_t j, std::vector<int64_t>& rows) const {
    rows.clear();
    std::vector<Block> cand;
    if (j < L.v_off) {
        const int64_t sj = j / L.blk;
        for (int64_t s = sj - 1; s <= sj; ++s)
            if (s >= 0 && s < L.N) hess_step_blocks(s, cand);
    }
    // (V columns: only rows >= j matter, and those lie in the V x V tail block)
    for (const Block& b : H.tail) cand.push_back(b);
    std::vector<std::pair<int64_t, int64_t>> iv;
    for (const Block& b : cand)
        if (j >= b.c0 && j < b.c1 && b.r1 > j) iv.emplace_back(std::max(b.r0, j), b.r1);
    std::sort(iv.begin(), iv.end());
    int64_t next = -1;
    for (auto& p : iv) {
        int64_t r = std::max(p.first, next);
        for (; r < p.second; ++r)
            if (pattern_mode != 2 || hess_opt_keep(*this, r, j)) rows.push_back(r);
        next = std::max(next, p.second);
    }
}

int64_t Model::hess_column_start(int64_t j) const {
    const int64_t h = H.reg_first * L.blk, t = H.reg_last * L.blk;
    if (j < h) return H.cp_head[j];
    if (j < t) {
        const int64_t i = j / L.blk;
        return H.seg_base + (i - H.reg_first) * (int64_t)H.Lseg + H.cp_tmpl[j - i * L.blk];
    }
    return H.cp_tail[j - t];
}

// ------------------------------------------------------------------------------------------------------
// structural sparsity of the evaluation points' Hessians: dependency masks pushed through the OCP functors
// ------------------------------------------------------------------------------------------------------
// value -> (set of directions it depends on, set of direction pairs with a structurally nonzero second derivative)
struct SP {
    uint32_t m1;
    uint32_t m2[32];
    SP() : m1(0) { for (uint32_t& r : m2) r = 0; }
    SP(double) : m1(0) { for (uint32_t& r : m2) r = 0; }
    static SP dir(uint32_t mask) { SP r; r.m1 = mask; return r; }
};
static inline SP sp_lin(const SP& a, const SP& b) {
    SP r; r.m1 = a.m1 | b.m1;
    for (int i = 0; i < 32; ++i) r.m2[i] = a.m2[i] | b.m2[i];
    return r;
}
static inline void sp_outer(SP& r, uint32_t a, uint32_t b) {
    for (int i = 0; i < 32; ++i) {
        if (a >> i & 1u) r.m2[i] |= b;
        if (b >> i & 1u) r.m2[i] |= a;
    }
}
static inline SP sp_nonlin(const SP& a) { SP r = a; sp_outer(r, a.m1, a.m1); return r; }
inline SP operator+(const SP& a, const SP& b) { return sp_lin(a, b); }
inline SP operator-(const SP& a, const SP& b) { return sp_lin(a, b); }
inline SP operator-(const SP& a) { return a; }
inline SP operator*(const SP& a, const SP& b) { SP r = sp_lin(a, b); sp_outer(r, a.m1, b.m1); return r; }
inline SP operator/(const SP& a, const SP& b) { return a * sp_nonlin(b); }
inline SP operator+(const SP& a, double) { return a; }
inline SP operator+(double, const SP& a) { return a; }
inline SP operator-(const SP& a, double) { return a; }
inline SP operator-(double, const SP& a) { return a; }
inline SP operator*(const SP& a, double) { return a; }
inline SP operator*(double, const SP& a) { return a; }
inline SP operator/(const SP& a, double) { return a; }
inline SP operator/(double, const SP& a) { return sp_nonlin(a); }
inline SP d_exp(const SP& a) { return sp_nonlin(a); }
inline SP d_sin(const SP& a) { return sp_nonlin(a); }
inline SP d_cos(const SP& a) { return sp_nonlin(a); }
inline SP d_sqr(const SP& a) { return sp_nonlin(a); }

// the OCP functions on dependency masks: compiled registry entry ...
template <class P> struct RegistryFns {
    void dynamics(SP* f, const SP& t, const SP* x, const SP* u, const SP* v) const { P::template dynamics<SP>(f, t, x, u, v); }
    SP lagrange(const SP& t, const SP* x, const SP* u, const SP* v) const { return P::template lagrange<SP>(t, x, u, v); }
    SP mayer(const SP* x0, const SP* xf, const SP* v) const { return P::template mayer<SP>(x0, xf, v); }
    void path(SP* r, const SP& t, const SP* x, const SP* u, const SP* v) const { P::template path<SP>(r, t, x, u, v); }
    void boundary(SP* r, const SP* x0, const SP* xf, const SP* v) const { P::template boundary<SP>(r, x0, xf, v); }
};
// ... or the postfix programs of a run-time OCP (ctd_jit.hpp)
struct RuntimeFns {
    const RtOcp& ro;
    static SP run(const RtProgram& prog, const SP* t, const SP* x, const SP* u, const SP* v, const SP* x0, const SP* xf) {
        std::vector<SP> st;
        for (const RtOp& o : prog) {
            switch (o.kind) {
                case RT_CONST: st.emplace_back(); break;
                case RT_T: st.push_back(*t); break;
                case RT_X: st.push_back(x[o.k]); break;
                case RT_U: st.push_back(u[o.k]); break;
                case RT_V: st.push_back(v[o.k]); break;
                case RT_X0: st.push_back(x0[o.k]); break;
                case RT_XF: st.push_back(xf[o.k]); break;
                case RT_NEG: break;
                case RT_NONLIN: st.back() = sp_nonlin(st.back()); break;
                case RT_ZERO: st.back() = SP(); break;
                case RT_POW: if (o.k == 0) st.back() = SP(); else if (o.k >= 2) st.back() = sp_nonlin(st.back()); break;
                default: {
                    const SP b = st.back(); st.pop_back();
                    const SP a = st.back(); st.pop_back();
                    st.push_back(o.kind == RT_MUL ? a * b : (o.kind == RT_DIV ? a / b : sp_lin(a, b)));      // (RT_ADD, RT_SUB, RT_MAX)
                }
            }
        }
        return st.empty() ? SP() : st.back();
    }
    void dynamics(SP* f, const SP& t, const SP* x, const SP* u, const SP* v) const {
        for (size_t r = 0; r < ro.p_dynamics.size(); ++r) f[r] = run(ro.p_dynamics[r], &t, x, u, v, nullptr, nullptr);
    }
    SP lagrange(const SP& t, const SP* x, const SP* u, const SP* v) const { return run(ro.p_lagrange, &t, x, u, v, nullptr, nullptr); }
    SP mayer(const SP* x0, const SP* xf, const SP* v) const { return run(ro.p_mayer, nullptr, nullptr, nullptr, v, x0, xf); }
    void path(SP* r, const SP& t, const SP* x, const SP* u, const SP* v) const {
        for (size_t q = 0; q < ro.p_path.size(); ++q) r[q] = run(ro.p_path[q], &t, x, u, v, nullptr, nullptr);
    }
    void boundary(SP* r, const SP* x0, const SP* xf, const SP* v) const {
        for (size_t q = 0; q < ro.p_boundary.size(); ++q) r[q] = run(ro.p_boundary[q], nullptr, nullptr, nullptr, v, x0, xf);
    }
};

// First-order masks of every output of the OCP functions, the way a global operator-overloading tracer sees them
// (SparseConnectivityTracer behind ADNLPModels' default backend, src/collocation.jl:131-134): a dependence survives
// `u * 0`, and time enters as the free-time variables.
template <class F>
static void probe_first_order(Model& mo, const F& fn) {
    const Layout& L = mo.L;
    const int n = L.n, m = L.m, nv = L.nv, np = L.p, nb = L.bc, vd = n + m;
    uint32_t TV = 0;
    if (L.it0 >= 0) TV |= 1u << (vd + L.it0);
    if (L.itf >= 0) TV |= 1u << (vd + L.itf);
    const SP t = SP::dir(TV);
    std::vector<SP> x(n > 0 ? n : 1), u(m > 0 ? m : 1), v(nv > 0 ? nv : 1);
    for (int r = 0; r < n; ++r) x[r] = SP::dir(1u << r);
    for (int b = 0; b < m; ++b) u[b] = SP::dir(1u << (n + b));
    for (int k = 0; k < nv; ++k) v[k] = SP::dir(1u << (vd + k));
    std::vector<SP> f(n > 0 ? n : 1);
    fn.dynamics(f.data(), t, x.data(), u.data(), v.data());
    mo.dep_f.assign(n, 0);
    for (int r = 0; r < n; ++r) mo.dep_f[r] = f[r].m1;
    mo.dep_g.assign(np, 0);
    if (np > 0) {
        std::vector<SP> g(np);
        fn.path(g.data(), t, x.data(), u.data(), v.data());
        for (int q = 0; q < np; ++q) mo.dep_g[q] = g[q].m1;
    }
    mo.dep_b.assign(nb, 0);
    if (nb > 0) {
        std::vector<SP> x0(n > 0 ? n : 1), xf(n > 0 ? n : 1), vb(nv > 0 ? nv : 1), r_(nb);
        for (int r = 0; r < n; ++r) { x0[r] = SP::dir(1u << r); xf[r] = SP::dir(1u << (n + r)); }
        for (int k = 0; k < nv; ++k) vb[k] = SP::dir(1u << (2 * n + k));
        fn.boundary(r_.data(), x0.data(), xf.data(), vb.data());
        for (int b = 0; b < nb; ++b) mo.dep_b[b] = r_[b].m1;
    }
}

void compute_dep_masks(Model& mo) {
    if (!for_problem(mo.problem, [&](auto tag) { probe_first_order(mo, RegistryFns<typename decltype(tag)::type>{}); })) {
        const RtOcp* ro = runtime_ocp(mo.problem);
        if (ro) probe_first_order(mo, RuntimeFns{*ro});
    }
}

template <class F>
static void probe_structure(Model& mo, const F& fn) {
    const Layout& L = mo.L;
    const int n = L.n, m = L.m, nv = L.nv, np = L.p, nb = L.bc;
    HessModel& H = mo.H;
    const int md = H.R.md, mdb = H.R.mdb, vd = n + m;
    const bool free_time = L.free_time != 0;
    uint32_t TV = 0;                                   // directions that move the time grid
    if (L.it0 >= 0) TV |= 1u << (vd + L.it0);
    if (L.itf >= 0) TV |= 1u << (vd + L.itf);
    auto fill = [&](const SP& phi, int dim, std::vector<uint8_t>& out) {
        out.assign((size_t)dim * dim, 0);
        for (int p = 0; p < dim; ++p)
            for (int q = 0; q < dim; ++q)
                if ((phi.m2[p] >> q & 1u) || (phi.m2[q] >> p & 1u)) out[p * dim + q] = 1;
    };
    const SP hh = SP::dir(TV), t = SP::dir(TV);
    std::vector<SP> x(n > 0 ? n : 1), u(m > 0 ? m : 1), v(nv > 0 ? nv : 1);
    for (int r = 0; r < n; ++r) x[r] = SP::dir((1u << r) | ((L.sc == SC_IRK && free_time) ? TV : 0u));
    for (int b = 0; b < m; ++b) u[b] = SP::dir(1u << (n + b));
    for (int k = 0; k < nv; ++k) v[k] = SP::dir(1u << (vd + k));
    // stage-type point
    {
        std::vector<SP> f(n > 0 ? n : 1);
        fn.dynamics(f.data(), t, x.data(), u.data(), v.data());
        SP phi;
        for (int r = 0; r < n; ++r) phi = phi + f[r] * (L.sc == SC_IRK ? SP() : hh);
        if (mo.info.lagrange) phi = phi + hh * fn.lagrange(t, x.data(), u.data(), v.data());
        if (L.sc == SC_TRAPEZE && np > 0) {
            std::vector<SP> g(np);
            fn.path(g.data(), t, x.data(), u.data(), v.data());
            for (int r = 0; r < np; ++r) phi = phi + g[r];
        }
        fill(phi, md, H.need_stage);
        H.need_rk.assign((size_t)(nv > 0 ? nv : 1) * (n > 0 ? n : 1), 0);
        if (L.sc == SC_IRK && free_time)
            for (int k = 0; k < nv; ++k)
                for (int a = 0; a < n; ++a) {
                    const bool time_var = (k == L.it0 || k == L.itf);
                    if (H.need_stage[a * md + vd + k] || (time_var && (phi.m1 >> a & 1u))) H.need_rk[k * n + a] = 1;
                }
    }
    // path point (x = X_i exactly: no dependence on the time grid through the state)
    H.need_path.assign((size_t)md * md, 0);
    if (np > 0 && L.sc != SC_TRAPEZE) {
        std::vector<SP> xp(n > 0 ? n : 1), g(np);
        for (int r = 0; r < n; ++r) xp[r] = SP::dir(1u << r);
        fn.path(g.data(), t, xp.data(), u.data(), v.data());
        SP phi;
        for (int r = 0; r < np; ++r) phi = phi + g[r];
        fill(phi, md, H.need_path);
    }
    // boundary + Mayer point
    H.need_bnd.assign((size_t)mdb * mdb, 0);
    {
        std::vector<SP> x0(n > 0 ? n : 1), xf(n > 0 ? n : 1), vb(nv > 0 ? nv : 1);
        for (int r = 0; r < n; ++r) { x0[r] = SP::dir(1u << r); xf[r] = SP::dir(1u << (n + r)); }
        for (int k = 0; k < nv; ++k) vb[k] = SP::dir(1u << (2 * n + k));
        SP phi;
        if (nb > 0) {
            std::vector<SP> r_(nb);
            fn.boundary(r_.data(), x0.data(), xf.data(), vb.data());
            for (int r = 0; r < nb; ++r) phi = phi + r_[r];
        }
        if (mo.info.mayer) phi = phi + fn.mayer(x0.data(), xf.data(), vb.data());
        fill(phi, mdb, H.need_bnd);
    }
}

// ------------------------------------------------------------------------------------------------------
// which evaluation points couple two NLP variables, and through which directions / coefficients
// ------------------------------------------------------------------------------------------------------
namespace {
enum { VK_X = 0, VK_U = 1, VK_K = 2, VK_V = 3 };
struct Var { int kind; int64_t s; int l; int c; };
enum { PT_STAGE = 0, PT_PATH = 1, PT_LIN = 2, PT_FPATH = 3, PT_BND = 4 };
struct Term { int pt; int64_t step; int di, c1, c2; };
struct Dir { bool ok; int d, coef; };

Var decode_var(const Layout& L, int64_t idx) {
    if (idx >= L.v_off) return Var{VK_V, -1, 0, (int)(idx - L.v_off)};
    const int64_t s = idx / L.blk;
    const int q = (int)(idx - s * L.blk);
    if (q < L.n) return Var{VK_X, s, 0, q};
    if (q < L.n + L.cu) {
        if (L.stagewise || L.cs > 1) return Var{VK_U, s, (q - L.n) / L.m, (q - L.n) % L.m};      // (l: stage / control of the step)
        return Var{VK_U, s, -1, q - L.n};
    }
    return Var{VK_K, s, (q - L.n - L.cu) / L.n, (q - L.n - L.cu) % L.n};
}

// stage point (s, j): Gauss-Legendre stage, the midpoint, or the trapeze node s
Dir map_stage(const Layout& L, int64_t s, int j, const Var& v) {
    const int n = L.n, m = L.m;
    if (v.kind == VK_V) return Dir{true, n + m + v.c, HC_ONE};
    if (L.sc == SC_IRK) {
        if (v.s != s) return Dir{false, 0, 0};
        if (v.kind == VK_X) return Dir{true, v.c, HC_ONE};
        if (v.kind == VK_K) return Dir{true, v.c, HC_HA + 3 * j + v.l};
        if (L.stagewise && v.l != j) return Dir{false, 0, 0};
        return Dir{true, n + v.c, HC_ONE};
    }
    if (L.sc == SC_MIDPOINT) {
        if (L.euler == 0 && v.kind == VK_X && (v.s == s || v.s == s + 1)) return Dir{true, v.c, HC_HALF};
        if (L.euler == 1 && v.kind == VK_X && v.s == s) return Dir{true, v.c, HC_ONE};          // f(t_i, X_i, U_i)
        if (L.euler == 2 && v.kind == VK_X && v.s == s + 1) return Dir{true, v.c, HC_ONE};      // f(t_{i+1}, X_{i+1}, U_i)
        if (v.kind == VK_U && v.s == s && (L.cs == 1 || v.l == j)) return Dir{true, n + v.c, HC_ONE};     // cs > 1: point j reads U_s^j
        return Dir{false, 0, 0};
    }
    if (v.s != s) return Dir{false, 0, 0};
    return v.kind == VK_X ? Dir{true, v.c, HC_ONE} : Dir{true, n + v.c, HC_ONE};
}
// path point of step s (x = X_s, u = get_OCP_control_at_time_step; stagewise: sum_l b_l U_s^l, irk_stagewise.jl:197-205);
// xs is the step whose state the point reads (s, or N for the final-time point which keeps the controls of step N-1)
Dir map_path(const Layout& L, int64_t xs, int64_t us, const Var& v) {
    const int n = L.n, m = L.m;
    if (v.kind == VK_V) return Dir{true, n + m + v.c, HC_ONE};
    if (v.kind == VK_X && v.s == xs) return Dir{true, v.c, HC_ONE};
    if (v.kind == VK_U && v.s == us && L.cs > 1) return v.l == 0 ? Dir{true, n + v.c, HC_ONE} : Dir{false, 0, 0};   // first control of the step
    if (v.kind == VK_U && v.s == us) return Dir{true, n + v.c, L.stagewise ? HC_B + v.l : HC_ONE};
    return Dir{false, 0, 0};
}
Dir map_bnd(const Layout& L, const Var& v) {
    if (v.kind == VK_V) return Dir{true, 2 * L.n + v.c, HC_ONE};
    if (v.kind == VK_X && v.s == 0) return Dir{true, v.c, HC_ONE};
    if (v.kind == VK_X && v.s == L.N) return Dir{true, L.n + v.c, HC_ONE};
    return Dir{false, 0, 0};
}
inline int sym_index(int md, int a, int b) { return a <= b ? a * md + b : b * md + a; }
}  // namespace

// every term of d2 L / d(row) d(col); both variables V is handled by the V x V reduction, not here
static void collect_terms(const Model& mo, int64_t row, int64_t col, std::vector<Term>& out) {
    const Layout& L = mo.L;
    const HessRecLayout& R = mo.H.R;
    const int S = R.S;
    const Var vr = decode_var(L, row), vc = decode_var(L, col);
    out.clear();
    if (vr.kind == VK_V && vc.kind == VK_V) return;
    std::set<int64_t> steps;
    for (const Var* v : {&vr, &vc}) {
        if (v->kind == VK_V) continue;
        steps.insert(v->s);
        if (L.sc == SC_MIDPOINT && v->kind == VK_X) steps.insert(v->s - 1);
        if (L.euler == 2 && v->kind == VK_U) steps.insert(v->s + 1);      // path point of node s+1 uses U_s (euler.jl:59-72)
    }
    const int64_t last_pt = (L.sc == SC_TRAPEZE) ? L.N : L.N - 1;
    const bool has_path_pt = L.p > 0 && L.sc != SC_TRAPEZE;
    const bool rk = L.sc == SC_IRK && L.free_time;
    for (int64_t s : steps) {
        if (s < 0 || s > last_pt) continue;
        for (int j = 0; j < S; ++j) {
            const Dir a = map_stage(L, s, j, vr), b = map_stage(L, s, j, vc);
            if (!a.ok || !b.ok) continue;
            // (summed points: pairs without a control direction are read from point 0, which holds the sum over the points)
            auto is_u = [&](int d) { return d >= L.n && d < L.n + L.m; };
            if (j > 0 && hess_sums_stages(L.sc, L.cs) && !is_u(a.d) && !is_u(b.d)) continue;
            const int base = R.oStage + j * R.stage_sz;
            const Var* kv = (vr.kind == VK_K) ? &vr : (vc.kind == VK_K ? &vc : nullptr);
            const Var* vv = (vr.kind == VK_V) ? &vr : (vc.kind == VK_V ? &vc : nullptr);
            if (rk && kv && vv) {  // d2/dK^l_a dV_k: a_jl (h HD[x_a][V_k] + dh/dv_k dPhi/dx_a)
                if (mo.H.need_rk[vv->c * L.n + kv->c])
                    out.push_back(Term{PT_STAGE, s, base + R.oRK + vv->c * L.n + kv->c, HC_A + 3 * j + kv->l, HC_ONE});
            } else if (mo.H.need_stage[sym_index(R.md, a.d, b.d)]) {
                out.push_back(Term{PT_STAGE, s, base + hess_tri(R.md, a.d, b.d), a.coef, b.coef});
            }
        }
        if (rk) {   // state-equation row -h sum_l b_l y'K^l: d2/dK^l_a dV_k = -b_l dh/dv_k y_a for the time variables
            const Var* kv = (vr.kind == VK_K) ? &vr : (vc.kind == VK_K ? &vc : nullptr);
            const Var* vv = (vr.kind == VK_V) ? &vr : (vc.kind == VK_V ? &vc : nullptr);
            if (kv && vv && kv->s == s && (vv->c == L.it0 || vv->c == L.itf))
                out.push_back(Term{PT_LIN, s, R.oYX + kv->c, HC_NBH + 3 * vv->c + kv->l, HC_ONE});
        }
        if (has_path_pt && s < L.N) {
            const int64_t us = (L.euler == 2 && s >= 1) ? s - 1 : s;
            const Dir a = map_path(L, s, us, vr), b = map_path(L, s, us, vc);
            if (a.ok && b.ok && mo.H.need_path[sym_index(R.md, a.d, b.d)])
                out.push_back(Term{PT_PATH, s, R.oHP + hess_tri(R.md, a.d, b.d), a.coef, b.coef});
        }
    }
    if (has_path_pt) {
        const Dir a = map_path(L, L.N, L.N - 1, vr), b = map_path(L, L.N, L.N - 1, vc);
        if (a.ok && b.ok && mo.H.need_path[sym_index(R.md, a.d, b.d)])
            out.push_back(Term{PT_FPATH, 0, R.oHP + hess_tri(R.md, a.d, b.d), a.coef, b.coef});
    }
    if (L.bc > 0 || mo.info.mayer) {
        const Dir a = map_bnd(L, vr), b = map_bnd(L, vc);
        if (a.ok && b.ok && mo.H.need_bnd[sym_index(R.mdb, a.d, b.d)])
            out.push_back(Term{PT_BND, 0, hess_tri(R.mdb, a.d, b.d), HC_ONE, HC_ONE});
    }
}

static bool hess_opt_keep(const Model& mo, int64_t row, int64_t col) {
    const Layout& L = mo.L;
    const HessRecLayout& R = mo.H.R;
    if (row >= L.v_off && col >= L.v_off) {           // V x V: summed over every evaluation point
        const int a = (int)(col - L.v_off), b = (int)(row - L.v_off), vd = L.n + L.m;
        if (!mo.H.need_stage.empty() && mo.H.need_stage[sym_index(R.md, vd + a, vd + b)]) return true;
        if (L.p > 0 && L.sc != SC_TRAPEZE && !mo.H.need_path.empty() && mo.H.need_path[sym_index(R.md, vd + a, vd + b)]) return true;
        if ((L.bc > 0 || mo.info.mayer) && !mo.H.need_bnd.empty() && mo.H.need_bnd[sym_index(R.mdb, 2 * L.n + a, 2 * L.n + b)]) return true;
        return false;
    }
    std::vector<Term> tt;
    collect_terms(mo, row, col, tt);
    return !tt.empty();
}

// id of the coefficient product C[c1] * C[c2] (registered on first use; -1 when the table is full)
static int pair_id(HessModel& H, int c1, int c2) {
    if (c1 > c2) std::swap(c1, c2);
    const uint16_t code = (uint16_t)(c1 | (c2 << 8));
    for (size_t i = 0; i < H.pairs.size(); ++i)
        if (H.pairs[i] == code) return (int)i;
    if ((int)H.pairs.size() >= kMaxPairs) return -1;
    H.pairs.push_back(code);
    return (int)H.pairs.size() - 1;
}

// template of the step-periodic segment of step i: per entry the relative row and its relative terms
struct SegTmpl {
    std::vector<int64_t> cp, relrow;
    std::vector<uint32_t> tptr, terms;
    bool operator==(const SegTmpl& o) const { return cp == o.cp && relrow == o.relrow && tptr == o.tptr && terms == o.terms; }
};

static bool segment_template(Model& mo, int64_t i, SegTmpl& t) {
    const Layout& L = mo.L;
    t.cp.assign(L.blk + 1, 0);
    t.relrow.clear(); t.tptr.assign(1, 0); t.terms.clear();
    std::vector<int64_t> rows;
    std::vector<Term> tt;
    for (int lc = 0; lc < L.blk; ++lc) {
        const int64_t col = i * L.blk + lc;
        mo.hess_gen_column(col, rows);
        for (int64_t row : rows) {
            collect_terms(mo, row, col, tt);
            for (const Term& x : tt) {
                if (x.pt == PT_FPATH || x.pt == PT_BND) return false;
                const int64_t rel = i - x.step;            // 0 own step, 1 previous, -1 next (slot code 2)
                if (rel < -mo.H.HH || rel > mo.H.HL) return false;
                const int pid = pair_id(mo.H, x.c1, x.c2);
                if (pid < 0) return false;
                t.terms.push_back(pack_term(x.di, pid, rel < 0 ? 2 : (int)rel));
            }
            if (tt.size() > (size_t)kMaxTerms) return false;
            t.tptr.push_back((uint32_t)t.terms.size());
            t.relrow.push_back(row >= L.v_off ? ((int64_t)1 << 40) + (row - L.v_off) : row - i * L.blk);
        }
        t.cp[lc + 1] = (int64_t)t.relrow.size();
    }
    return true;
}

int build_hess_model(Model& mo, std::string& err) {
    const Layout& L = mo.L;
    HessModel& H = mo.H;
    const int64_t N = L.N;
    // (midpoint with control_steps > 1: one stage-type point per control of the step, midpoint.jl:61-69,108-112)
    H.R = make_hess_layout(L.n, L.m, L.nv, L.p, L.sc, (L.sc == SC_MIDPOINT && L.cs > 1) ? L.cs : L.s, L.free_time != 0);
    if (H.R.stride >= 65536) { err = "per-step Hessian record too large for 16-bit data indices"; return ST_EPATTERN; }
    H.HL = (L.sc == SC_MIDPOINT) ? 1 : 0;
    H.HH = (L.euler == 2 && L.p > 0) ? 1 : 0;       // implicit Euler: the path point of node i+1 couples X_{i+1} with U_i
    H.pairs.clear();
    pair_id(H, HC_ONE, HC_ONE);           // pair 0
    if (H.R.md > 31 || H.R.mdb > 31 || H.hk > 4) { err = "more than 31 Hessian directions per evaluation point are not supported"; return ST_EPATTERN; }
    if (!for_problem(mo.problem, [&](auto tag) { probe_structure(mo, RegistryFns<typename decltype(tag)::type>{}); })) {
        const RtOcp* ro = runtime_ocp(mo.problem);       // run-time OCP: the probe walks its postfix programs
        if (!ro) { err = "problem id not in the registry"; return ST_EPATTERN; }
        probe_structure(mo, RuntimeFns{*ro});
    }
    build_hess_tail(mo);

    // ---- regular range ---------------------------------------------------------------------------------------------
    SegTmpl t1, t2;
    H.reg_first = H.reg_last = N;
    H.Lseg = 0;
    H.tptr.assign(1, 0); H.terms.clear(); H.cp_tmpl.assign(L.blk + 1, 0);
    if (N >= 5) {
        bool ok = segment_template(mo, 1, t1);
        if (ok)
            for (int64_t chk : {(int64_t)2, N - 2})
                if (!segment_template(mo, chk, t2) || !(t2 == t1)) { ok = false; break; }
        if (!ok) { err = "Hessian pattern is not step-periodic"; return ST_EPATTERN; }
        H.reg_first = 1;
        H.reg_last = N - 1;
        if (segment_template(mo, N - 1, t2) && t2 == t1) H.reg_last = N;
        H.Lseg = (int)t1.relrow.size();
        H.tptr = t1.tptr; H.terms = t1.terms; H.cp_tmpl = t1.cp; H.relrow = t1.relrow;
    }

    // ---- column starts ----------------------------------------------------------------------------------------------
    std::vector<int64_t> rows;
    const int64_t head_cols = H.reg_first * L.blk;
    H.cp_head.assign(head_cols + 1, 0);
    int64_t nz = 0;
    for (int64_t j = 0; j < head_cols; ++j) { H.cp_head[j] = nz; mo.hess_gen_column(j, rows); nz += (int64_t)rows.size(); }
    H.cp_head[head_cols] = nz;
    H.seg_base = nz;
    nz += (H.reg_last - H.reg_first) * (int64_t)H.Lseg;
    const int64_t tail0 = H.reg_last * L.blk, tail_cols = L.nvar - tail0;
    H.cp_tail.assign(tail_cols + 1, 0);
    for (int64_t jj = 0; jj < tail_cols; ++jj) { H.cp_tail[jj] = nz; mo.hess_gen_column(tail0 + jj, rows); nz += (int64_t)rows.size(); }
    H.cp_tail[tail_cols] = nz;
    H.nnzh = nz;

    // ---- V x V entries: every step contributes the same terms ---------------------------------------------------------------
    H.nvv = L.nv * (L.nv + 1) / 2;
    H.vptr.assign(1, 0); H.vterms.clear();
    {
        int e = 0;
        const int md = H.R.md, vd = L.n + L.m;
        for (int kc = 0; kc < L.nv; ++kc) {
            mo.hess_gen_column(L.v_off + kc, rows);
            if (mo.pattern_mode != 2 && (int)rows.size() != L.nv - kc) { err = "internal: V x V block is not dense lower-triangular"; return ST_EPATTERN; }
            for (int kr = kc; kr < L.nv; ++kr, ++e) {
                // (optimized pattern: an entry no evaluation point feeds is not in the pattern: position -1, no terms)
                const auto it = std::find(rows.begin(), rows.end(), L.v_off + kr);
                H.vv_idx[e] = it == rows.end() ? -1 : mo.hess_column_start(L.v_off + kc) + (int64_t)(it - rows.begin());
                if (it == rows.end()) { H.vptr.push_back((uint32_t)H.vterms.size()); continue; }
                for (int j = 0; j < (hess_sums_stages(L.sc, L.cs) ? 1 : H.R.S); ++j)
                    if (H.need_stage[sym_index(md, vd + kc, vd + kr)])
                        H.vterms.push_back(pack_term(H.R.oStage + j * H.R.stage_sz + hess_tri(md, vd + kc, vd + kr), 0, 0));
                if (L.p > 0 && L.sc != SC_TRAPEZE && H.need_path[sym_index(md, vd + kc, vd + kr)])
                    H.vterms.push_back(pack_term(H.R.oHP + hess_tri(md, vd + kc, vd + kr), 0, 0));
                H.vptr.push_back((uint32_t)H.vterms.size());
            }
        }
    }

    // ---- edge entries ---------------------------------------------------------------------------------------------------------
    struct Raw { int64_t idx; std::vector<Term> tt; };
    std::vector<Raw> raws;
    std::set<int64_t> need;
    need.insert(0);          // the boundary point reads X_0 from the record inputs of step 0,
    need.insert(N - 1);      // the final-time points X_N and the controls of step N-1 from those of step N-1
    std::vector<Term> tt;
    auto scan_col = [&](int64_t j) {
        if (j >= L.v_off) return;                  // V columns hold V x V rows only
        mo.hess_gen_column(j, rows);
        const int64_t base = mo.hess_column_start(j);
        for (size_t t = 0; t < rows.size(); ++t) {
            collect_terms(mo, rows[t], j, tt);
            for (const Term& x : tt)
                if (x.pt != PT_FPATH && x.pt != PT_BND) need.insert(x.step);
            raws.push_back(Raw{base + (int64_t)t, tt});
        }
    };
    H.head_ptr.assign(H.reg_first + 1, 0);     // head entries grouped by step (see Model::head_ptr)
    for (int64_t j = 0; j < head_cols; ++j) {
        if (j % L.blk == 0) H.head_ptr[j / L.blk] = (int)raws.size();
        scan_col(j);
    }
    H.head_ptr[H.reg_first] = (int)raws.size();
    H.edge_split = (int)raws.size();           // entries before: owners of the leading steps; after: owner of step N-1 (shards)
    for (int64_t j = tail0; j < L.nvar; ++j) scan_col(j);
    if (L.sc == SC_TRAPEZE) need.insert(N);        // node N: its V x V share is summed by the edge block
    if ((int)need.size() > kMaxHessEdgeSlots) { err = "internal: too many edge records (Hessian)"; return ST_EPATTERN; }
    std::map<int64_t, int> slot_of;
    H.n_edge_slots = 0;
    for (int64_t s : need) { slot_of[s] = H.n_edge_slots; H.edge_steps[H.n_edge_slots++] = s; }
    H.edge_fp = H.n_edge_slots;
    H.edge_b = H.n_edge_slots + 1;
    H.edge_idx.clear(); H.eptr.assign(1, 0); H.eterms.clear();
    for (const Raw& r : raws) {
        H.edge_idx.push_back(r.idx);
        for (const Term& x : r.tt) {
            const int slot = x.pt == PT_FPATH ? H.edge_fp : (x.pt == PT_BND ? H.edge_b : slot_of[x.step]);
            const int pid = pair_id(H, x.c1, x.c2);
            if (pid < 0) { err = "internal: coefficient pair table full (Hessian)"; return ST_EPATTERN; }
            H.eterms.push_back(pack_term(x.di, pid, slot));
        }
        H.eptr.push_back((uint32_t)H.eterms.size());
    }
    // V x V share of the edge block: final-path point, boundary point, trapeze node N
    H.evptr.assign(1, (uint32_t)H.eterms.size());
    {
        const int md = H.R.md, vd = L.n + L.m, mdb = H.R.mdb;
        for (int kc = 0; kc < L.nv; ++kc)
            for (int kr = kc; kr < L.nv; ++kr) {
                if (L.p > 0 && L.sc != SC_TRAPEZE && H.need_path[sym_index(md, vd + kc, vd + kr)])
                    H.eterms.push_back(pack_term(H.R.oHP + hess_tri(md, vd + kc, vd + kr), 0, H.edge_fp));
                if ((L.bc > 0 || mo.info.mayer) && H.need_bnd[sym_index(mdb, 2 * L.n + kc, 2 * L.n + kr)])
                    H.eterms.push_back(pack_term(hess_tri(mdb, 2 * L.n + kc, 2 * L.n + kr), 0, H.edge_b));
                if (L.sc == SC_TRAPEZE && H.need_stage[sym_index(md, vd + kc, vd + kr)])
                    H.eterms.push_back(pack_term(H.R.oStage + hess_tri(md, vd + kc, vd + kr), 0, slot_of[N]));
                H.evptr.push_back((uint32_t)H.eterms.size());
            }
    }

    // ---- eval tasks: a lane differentiates along one outer direction p and up to hk inner directions q_i and produces
    // the pairs (p, q_i).  Every structurally nonzero pair {a, b} (and every pair an RK entry needs) is assigned to one
    // lane, with p = a or p = b (the Hessian is symmetric): greedy packing, busiest direction first -----------------------
    auto make_tasks = [&](int md, const std::vector<uint8_t>& need, bool with_rk, std::vector<uint32_t>& out) {
        out.clear();
        const int vd = L.n + L.m;
        std::vector<std::vector<int>> adj(md);                 // remaining pairs per direction (self pair: d in adj[d])
        auto has = [&](int a, int b) {
            if (need[a * md + b] || need[b * md + a]) return true;
            if (with_rk) {
                const int x = std::min(a, b), v = std::max(a, b);
                if (x < L.n && v >= vd && H.need_rk[(v - vd) * L.n + x]) return true;
            }
            return false;
        };
        for (int a = 0; a < md; ++a)
            for (int b = a; b < md; ++b)
                if (has(a, b)) { adj[a].push_back(b); if (b != a) adj[b].push_back(a); }
        auto drop = [&](int a, int b) {
            adj[a].erase(std::find(adj[a].begin(), adj[a].end(), b));
            if (a != b) adj[b].erase(std::find(adj[b].begin(), adj[b].end(), a));
        };
        for (;;) {
            int best = -1;
            for (int d = 0; d < md; ++d)
                if (!adj[d].empty() && (best < 0 || adj[d].size() > adj[best].size())) best = d;
            if (best < 0) break;
            // partners with the fewest other pairs first: they are the hardest to place elsewhere
            std::vector<int> part = adj[best];
            std::stable_sort(part.begin(), part.end(), [&](int x, int y) { return adj[x].size() < adj[y].size(); });
            uint32_t code = (uint32_t)best;
            int cnt = 0;
            for (int q : part) {
                if (cnt == H.hk) break;
                code |= (uint32_t)q << (5 + 5 * cnt);
                drop(best, q);
                ++cnt;
            }
            for (; cnt < 4; ++cnt) code |= 31u << (5 + 5 * cnt);
            out.push_back(code);
        }
    };
    make_tasks(H.R.md, H.need_stage, L.sc == SC_IRK && L.free_time, H.tasks);
    // run-time OCP with symbolically differentiated stage functions: one lane per stage point writes every second
    // derivative of the point (hess_eval_stage_sym)
    {
        const RtOcp* ro = runtime_ocp(mo.problem);
        H.sym_stage = (ro && ro->has_sym) || registry_has_sym(mo.problem);
        bool lag_t = ro ? ro->lag_t : true;
        for_problem(mo.problem, [&](auto tag) { lag_t = decltype(tag)::type::LAG_T; });
        if (L.cs > 1 && mo.info.lagrange && lag_t) H.sym_stage = false;      // hess_uses_sym (ctd_hess_body.hpp)
        if (H.sym_stage) H.tasks.assign(1, 0u);
    }
    make_tasks(H.R.md, H.need_path, false, H.ptasks);
    make_tasks(H.R.mdb, H.need_bnd, false, H.btasks);
    // ---- what the tiles read: terms as LDS offsets, coefficient pairs as constant * step-dependent factors ---------------
    {
        const int np = (int)H.pairs.size();
        H.tcode.clear();
        for (uint32_t c : H.terms) {
            const int sl = term_slot(c), sd = sl == 1 ? 1 : (sl == 2 ? -1 : 0);
            const int a = term_pair(c) - sd * np, b = term_di(c) - sd * H.R.stride;
            if (a < -32768 || a > 32767 || b < -32768 || b > 32767) { err = "per-step Hessian record too large for 16-bit LDS offsets"; return ST_EPATTERN; }
            H.tcode.push_back(pack_tile_term(a, b));
        }
        // Segments that are mostly structural zeros of the pattern (the manual patterns of the Gauss-Legendre schemes: 85 % of
        // the 12-state quadrotor's 1890 entries per step): the tiles zero-fill their part of vals and walk only the entries
        // that have terms -- 2 passes of a workgroup over the segment instead of 8.  CTD_HESS_COMPACT=0/1 overrides.
        int nzero = 0;
        for (int e = 0; e < H.Lseg; ++e) nzero += H.tptr[e + 1] == H.tptr[e];
        // Fewer zeros (goddard, 3 Gauss-Legendre stages: 30 of 135): the same walk -- 105 entries let every lane take every
        // second step instead of every step -- and explicit zero stores at the listed positions.
        // (37.7 vs 38.5 us at N = 80000; taken only where the shorter walk lets a lane skip steps)
        {
            const int ncz = H.Lseg - nzero, g0 = H.Lseg < 256 ? 256 / std::max(1, H.Lseg) : 1, g1 = ncz < 256 ? 256 / std::max(1, ncz) : 1;
            H.compact = nzero == 0 ? 0 : (2 * nzero >= H.Lseg ? 1 : ((g1 > g0 && 8 * nzero >= H.Lseg) ? 2 : 0));
        }
        if (const char* ev = std::getenv("CTD_HESS_COMPACT"); ev && *ev) H.compact = nzero > 0 ? std::atoi(ev) : 0;
        H.cpos.clear(); H.zpos.clear(); H.ctptr = H.tptr;
        if (H.compact) {
            H.ctptr.assign(1, 0u);
            for (int e = 0; e < H.Lseg; ++e) {
                if (H.tptr[e + 1] > H.tptr[e]) { H.cpos.push_back((uint32_t)e); H.ctptr.push_back(H.tptr[e + 1]); }
                else if (H.compact == 2) H.zpos.push_back((uint32_t)e);
            }
        }
        auto factor = [&](int ci, double& cst) -> int {       // C[ci] = cst * F(kind)   (HC_* in ctd_hess.hpp)
            auto abc = [&](int e) { return e < 9 ? L.a[e] : L.b[e - 9]; };
            if (ci == HC_ONE) { cst = 1.0; return HF_ONE; }
            if (ci == HC_HALF) { cst = 0.5; return HF_ONE; }
            if (ci < HC_A) { cst = abc(ci - HC_HA); return HF_H; }
            if (ci < HC_B) { cst = abc(ci - HC_A); return HF_ONE; }
            if (ci < HC_NBH) { cst = abc(9 + ci - HC_B); return HF_ONE; }
            const int k = (ci - HC_NBH) / 3, l = (ci - HC_NBH) - 3 * k;
            cst = -abc(9 + l);
            return HF_DH + k;
        };
        H.pair_kind.assign(np, 0);
        H.pair_c.assign(np, 0.0);
        for (int i = 0; i < np; ++i) {
            double c1 = 0.0, c2 = 0.0;
            const int k1 = factor(H.pairs[i] & 0xFF, c1), k2 = factor(H.pairs[i] >> 8, c2);
            H.pair_kind[i] = (uint16_t)(k1 | (k2 << 8));
            H.pair_c[i] = c1 * c2;
        }
    }
    return ST_OK;
}

bool build_hess_step_tables(const Model& mo, const short* pairs, int nout, int chunk, std::vector<int32_t>& src, std::vector<int32_t>& chunk_pos) {
    const Layout& L = mo.L;
    const HessModel& H = mo.H;
    if (!pairs || nout < 0 || H.Lseg <= 0 || (int)H.relrow.size() != H.Lseg || (int)H.cp_tmpl.size() != L.blk + 1) return false;
    if (H.HL != 0 || H.HH != 0) return false;
    src.assign(H.Lseg, -1);
    std::vector<int32_t> pos(nout);
    for (int c = 0; c < nout; ++c) {
        const int a = pairs[2 * c], b = pairs[2 * c + 1];
        if (b < 0 || b >= L.blk) return false;
        const int64_t want = a < L.blk ? (int64_t)a : ((int64_t)1 << 40) + (a - L.blk);
        int e = -1;
        for (int64_t q = H.cp_tmpl[b]; q < H.cp_tmpl[b + 1]; ++q)
            if (H.relrow[q] == want) { e = (int)q; break; }
        if (e < 0) return false;                                 // the pattern lacks an entry the step function produces
        if (c > 0 && e <= pos[c - 1]) return false;              // (outputs come in CSC order)
        pos[c] = e;
        src[e] = c;
    }
    chunk_pos.assign(1, 0);
    for (int c = chunk; c < nout; c += chunk) chunk_pos.push_back(pos[c]);
    if (nout > 0) chunk_pos.push_back(H.Lseg);
    return true;
}

int default_hess_tile(const Model& mo) {
    // The eval phase runs (steps per tile) x (lanes per step) second-order tasks on 256 lanes: tiles are sized so that the
    // tasks fill whole passes (MI355X sweeps, profiles/r01_hessian_kernel.md): one full pass when that gives a tile of >= 24
    // steps (trapeze / midpoint / Euler / light Gauss-Legendre problems), else two passes; clamped to ~60 KiB of LDS, and
    // halved while the grid would not give every CU at least one workgroup
    const Layout& L = mo.L;
    const HessModel& H = mo.H;
    const int tps = std::max(1, H.R.S * (int)H.tasks.size() + (int)H.ptasks.size());
    const int64_t per_step = (int64_t)(L.blk + L.cb + H.R.stride + (int64_t)H.pairs.size() + 2) * 8;
    const int64_t fit = std::max<int64_t>(1, (60 * 1024) / per_step - H.HL - H.HH - 1);
    if (H.sym_stage) {
        // symbolic stage functions (run-time OCPs): the eval phase is one short pass, so the tile is set by the other
        // phases -- MI355X sweeps (profiles/r01_hessian_symbolic.md): ~27 KiB of LDS per workgroup for light steps (five to six
        // workgroups per CU), ~48 KiB for steps of 2 KiB and more (the CSC period is then several wave passes per step)
        // (records hold packed triangles since round 2: sweeps in profiles/r02_hessian_tiles.log -- the 12-state quadrotor,
        // 6.3 KiB per step: 8 steps = 57 KiB 90 us, 6 steps 101 us, 10 steps 95 us, 12 steps = one workgroup per CU 133 us)
        // (optimized pattern, round 4: no compact segment to zero-fill, a fifth of the manual pattern's entries per step -- the emission
        // is short and the largest tile that still leaves two workgroups per CU wins: 12-state quadrotor GL3, N = 20 000: 8 steps 60.5 us,
        // 10 steps = 69 KiB 55.3, 12 steps = one workgroup per CU 90.5: profiles/r04_hess_tiles_optimized.log)
        const int64_t budget = per_step >= 5000 ? (mo.pattern_mode == 2 ? 70 * 1024 : 58 * 1024) : (per_step >= 2048 ? 48 * 1024 : 27 * 1024);
        int64_t Ts = std::max<int64_t>(1, std::min<int64_t>(128, budget / per_step - H.HL - H.HH - 1));
        if (Ts < 6) Ts = std::max<int64_t>(Ts, std::min<int64_t>(6, (78 * 1024) / per_step - H.HL - H.HH - 1));
        // One-point schemes (midpoint, Euler, trapeze; any number of controls per step) with heavy steps: the kernel keeps TWO
        // workgroups per CU resident (registers of the symbolic stage function), a round of 512 tiles, and its time follows
        // rounds x steps per tile -- what counts is how full the last round is (8-state quadrotor, midpoint, N = 20 000: 40 steps =
        // 502 tiles 14.4 us, 32 steps = 627 tiles 19.9, 24 steps = 836 tiles 18.9, 20 steps = 1002 tiles 14.5, 16 steps = 1252 tiles 17.6;
        // 12-state: 15 steps 29.2 us, 20 steps = 1003 tiles 22.1; with 2 controls per step 7 steps 87.5 us, 16 steps 30.2, 20 steps =
        // ONE resident workgroup 92.3: profiles/r03_control_steps.md).  The largest tile that lets two workgroups share the LDS
        // fixes the number of rounds; the steps are then spread evenly over those rounds.
        if (L.sc != SC_IRK && per_step >= 1400) {
            const int64_t tmax = std::max<int64_t>(4, std::min<int64_t>(64 / L.cs, (70 * 1024) / per_step - H.HL - H.HH - 1));
            const int64_t rounds = (L.N + tmax * 512 - 1) / (tmax * 512);
            Ts = std::max<int64_t>(4, std::min<int64_t>(tmax, (L.N + rounds * 512 - 1) / (rounds * 512)));
        }
        // small grids: at least ~1.5 workgroups per CU (Goddard, 2 stages, N = 10 000: 21 steps = 477 tiles 10.3 us, 11 steps 11.0 us)
        while (Ts > 4 && (L.N + Ts - 1) / Ts < 400) Ts = (Ts + 1) / 2;
        return (int)Ts;
    }
    int64_t T = 256 / tps;
    if (T < 24) T = 512 / tps;
    T = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(T, 128), fit));
    // heavy steps (>= 100 lanes each: the 12-state quadrotor): such a tile is only ~4 steps; three passes on the largest
    // tile that still lets two workgroups share a CU's 160 KiB of LDS (6 steps, 77 KiB) measured 9 % faster -- one step more
    // (88 KiB, one workgroup per CU) 50 % slower
    if (T < 6 && tps >= 100) {
        const int64_t fit3 = (78 * 1024) / per_step - H.HL - H.HH - 1;
        T = std::max<int64_t>(T, std::min<int64_t>(768 / tps, fit3));
    }
    while (T > 4 && (L.N + T - 1) / T < 256) T = (T + 1) / 2;
    return (int)T;
}

void Model::fill_hparams(HParams& hp, int tile, int64_t step_begin, int64_t step_end) const {
    hp = HParams{};
    hp.L = L;
    hp.R = H.R;
    hp.T = tile;
    hp.HL = H.HL;
    hp.HH = H.HH;
    if (step_end <= 0) { step_begin = 0; step_end = L.N; }
    hp.step_begin = step_begin; hp.step_end = step_end;
    hp.ntiles = (int)((step_end - step_begin + tile - 1) / tile);
    hp.xcd_remap = 0;
    // shards: the owners of the leading irregular steps emit their head entries, the owner of step N-1 the tail entries and the V x V share of the
    // final-path / boundary / last-node points; the other shards have no edge work
    hp.edge_begin = H.head_ptr[std::min<int64_t>(step_begin, H.reg_first)];
    hp.edge_end = H.head_ptr[std::min<int64_t>(step_end, H.reg_first)];
    hp.edge2_begin = H.edge_split;
    hp.edge2_end = step_end == L.N ? (int)H.edge_idx.size() : H.edge_split;
    hp.edge_vv = step_end == L.N ? 1 : 0;
    // one edge workgroup per 256 edge entries (at most 16): every one of them evaluates the edge records, each emits its share
    {
        const int ntot = (hp.edge_end - hp.edge_begin) + (hp.edge2_end - hp.edge2_begin);
        hp.n_edge_blocks = std::max(1, std::min(16, (ntot + 255) / 256));
        if (const char* ev = std::getenv("CTD_HESS_EDGE_BLOCKS"); ev && *ev) hp.n_edge_blocks = std::max(1, std::min(64, std::atoi(ev)));
    }
    hp.Lseg = H.Lseg;
    hp.nterms = (int)H.terms.size();
    hp.nvterms = (int)H.vterms.size();
    hp.seg_base = H.seg_base; hp.reg_first = H.reg_first; hp.reg_last = H.reg_last;
    hp.nvv = H.nvv;
    for (int e = 0; e < H.nvv; ++e) hp.vv_idx[e] = H.vv_idx[e];
    hp.n_edge = (int)H.edge_idx.size();
    hp.n_edge_slots = H.n_edge_slots;
    hp.edge_fp = H.edge_fp; hp.edge_b = H.edge_b;
    for (int k = 0; k < kMaxHessEdgeSlots; ++k) hp.edge_steps[k] = H.edge_steps[k];
    hp.npairs = (int)H.pairs.size();
    for (int i = 0; i < hp.npairs; ++i) hp.pairs[i] = H.pair_kind[i];
    hp.ntask = (int)H.tasks.size();
    hp.nptask = (int)H.ptasks.size();
    hp.nbtask = (int)H.btasks.size();
    hp.slot_tasks = H.R.S * hp.ntask + hp.nptask;
    hp.div_ntask = make_fastdiv((uint32_t)(hp.ntask > 0 ? hp.ntask : 1));
    hp.div_stage_tasks = make_fastdiv((uint32_t)(H.R.S * hp.ntask > 0 ? H.R.S * hp.ntask : 1));
    hp.div_nptask = make_fastdiv((uint32_t)(hp.nptask > 0 ? hp.nptask : 1));
    hp.nc = (int)H.ctptr.size() - 1;
    hp.compact = H.compact;
    hp.nz = (int)H.zpos.size();
    hp.div_nz = make_fastdiv((uint32_t)(hp.nz > 0 ? hp.nz : 1));
    hp.div_nc = make_fastdiv((uint32_t)(hp.nc > 0 ? hp.nc : 1));
    hp.div_npairs = make_fastdiv((uint32_t)(hp.npairs > 0 ? hp.npairs : 1));
}

}  // namespace ctd
