// TEST INFRASTRUCTURE ONLY -- serial CPU stepping of the engine's kernel phase functions.
//
// There is no GPU in the build container.  This file compiles the SAME phase templates the HIP kernel wraps
// (ctdirect.jl_amd/csrc/ctd_kernel_body.hpp) with g++ and runs them lane by lane, workgroup by workgroup, with the
// "LDS" on the heap (exact size, NaN-filled, AddressSanitizer-friendly).  It lets the CPU test-suite check the
// emit tables, indexing and chain-rule algebra against the oracle before a kernel ever runs on hardware.
// It is NOT part of the product: libctdirect_hip.so does not contain it, the C ABI cannot reach it, and the GPU
// parity tests (-m gpu) never use it.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <string>
#include <vector>

#include "../../ctdirect.jl_amd/csrc/ctd_host.cpp"
#include "../../ctdirect.jl_amd/csrc/ctd_jit.cpp"
#include "../../ctdirect.jl_amd/csrc/ctd_kernel_body.hpp"
#define ST_OK ST_OK_HESS
#define ST_EPATTERN ST_EPATTERN_HESS
#include "../../ctdirect.jl_amd/csrc/ctd_hess_host.cpp"
#undef ST_OK
#undef ST_EPATTERN
#include "../../ctdirect.jl_amd/csrc/ctd_hess_body.hpp"

using namespace ctd;

template <class P, int SC, int S>
static void run_blocks(const KParams& kp, const double* xu, int nthr) {
    const int nblocks = kp.ntiles + kp.has_edge;
    const int64_t nlds = lds_doubles(kp);
    // multi-tile workgroups (KParams::wg_stride, staged driver): workgroup w walks the blocks w, w + wg_stride, ... on ONE LDS image
    const int stride = DirectTile<P, SC>::value ? 0 : kp.wg_stride;
    const int nwg = stride > 0 ? std::min(nblocks, stride + kp.has_edge) : nblocks;
    for (int w = 0; w < nwg; ++w) {
        int b = w;
        std::vector<double> lds(nlds, std::numeric_limits<double>::quiet_NaN());
        if (DirectTile<P, SC>::value) {
            // direct driver (cons_jac_kernel): no staging of xu, codes prefetched per lane, one barrier before the emission
            BlockCtx cx = make_direct_ctx(kp, b, lds.data(), xu);
            std::vector<EmitPre> pre(nthr);
            for (int t = 0; t < nthr; ++t) pre[t] = emit_prefetch<P>(kp, cx, t, nthr);
            for (int t = 0; t < nthr; ++t) phase_eval<P, SC, S, RegEval<P, SC, S>::value, 1>(kp, cx, t, nthr, &pre[t]);
            if constexpr (SC == SC_IRK) {
                if (kp.n_early > 0 && !cx.is_edge) {          // early emission: the lead wave's stores (in lockstep on the GPU)
                    const int lb = early_leadbase<P, SC, S>(kp);
                    for (int l = 0; l < 64 && lb + l < nthr; ++l) early_emit<P, SC, S, 1>(kp, cx, l, pre[lb + l]);
                }
            }
            for (int t = 0; t < nthr; ++t) phase_emit<P, SC, S>(kp, cx, t, nthr, &pre[t]);
            continue;
        }
        BlockCtx cx = make_ctx(kp, b, lds.data());
        constexpr int NB = EmitN<P, SC, S>::value;          // (as cons_jac_body: codes of long periods prefetched per lane)
        std::vector<EmitPreT<NB>> pre(nthr);
        const bool use_pre = !cx.is_edge && !codes_staged(kp);
        if (use_pre)
            for (int t = 0; t < nthr; ++t) pre[t] = emit_prefetch<P, NB>(kp, cx, t, nthr);
        for (int t = 0; t < nthr; ++t) phase_load<P, SC, S>(kp, cx, xu, t, nthr);
        for (;;) {                                           // (the loop of cons_jac_body)
            for (int t = 0; t < nthr; ++t) phase_eval<P, SC, S>(kp, cx, t, nthr);
            for (int t = 0; t < nthr; ++t) phase_fin<P, SC, S>(kp, cx, t, nthr);
            for (int t = 0; t < nthr; ++t) phase_fin2<P, SC, S>(kp, cx, t, nthr);
            b += stride;
            const bool more = stride > 0 && !cx.is_edge && b < nblocks;
            BlockCtx nx = cx;
            std::vector<TileIn> tin(nthr);
            if (more) {
                nx = make_ctx(kp, b, lds.data());
                for (int t = 0; t < nthr; ++t) tin[t] = load_issue<P>(kp, nx, xu, t, nthr);
            }
            for (int t = 0; t < nthr; ++t) phase_emit<P, SC, S, NB>(kp, cx, t, nthr, use_pre ? &pre[t] : nullptr);
            if (!more) break;
            for (int t = 0; t < nthr; ++t) load_commit<P, SC, S>(kp, nx, xu, tin[t], t, nthr);
            cx = nx;
        }
    }
}

// serial stepping of hess_kernel + hess_finish_kernel (ctd_hess_kernels.hpp)
template <class P, int SC, int S>
static void run_hess_blocks(const HParams& hp, const double* xu, const double* y, int nthr) {
    const int64_t nlds = hess_lds_doubles(hp);
    for (int b = 0; b < hp.ntiles + hp.n_edge_blocks; ++b) {
        std::vector<double> lds(nlds, std::numeric_limits<double>::quiet_NaN());
        HBlockCtx cx = make_hctx(hp, b, lds.data());
        for (int t = 0; t < nthr; ++t) hess_phase_load<P>(hp, cx, xu, y, t, nthr);
        for (int t = 0; t < nthr; ++t) hess_phase_eval<P, SC, S>(hp, cx, t, nthr);
        for (int t = 0; t < nthr; ++t) hess_phase_stage_sum<P, SC, S>(hp, cx, t, nthr);
        for (int t = 0; t < nthr; ++t) hess_phase_emit<P, SC, S>(hp, cx, b, t, nthr);
        for (int t = 0; t < nthr; ++t) hess_phase_vvsum(hp, cx, b, t, nthr);
    }
    for (int e = 0; e < hp.nvv; ++e) {
        const int fthr = 256;                             // hess_finish_kernel always runs kHessBlock lanes: same tree
        std::vector<double> red(fthr);
        for (int t = 0; t < fthr; ++t) red[t] = hess_finish_partial(hp, e, t, fthr);
        for (int off = fthr >> 1; off > 0; off >>= 1)
            for (int t = 0; t < off; ++t) red[t] = red[t] + red[t + off];
        if (hp.vv_idx[e] >= 0) hp.vals[hp.vv_idx[e]] = red[0];
    }
}

static std::string g_err;
static int g_control_steps = 1;          // DOCP(..., control_steps, ...) of the models built below (tests set it around a call)
static int g_value_order = 0;            // ctd_desc.value_order of the models built below (0 CSC, 1 CSR)

extern "C" {

void emu_set_control_steps(int cs) { g_control_steps = cs < 1 ? 1 : cs; }
void emu_set_value_order(int o) { g_value_order = o; }

const char* emu_last_error() { return g_err.c_str(); }

// out[0..3] = nvar, ncon, nnzj, dropped
int emu_sizes(int problem, int scheme, int pattern_mode, int64_t N, const double* tg, int64_t tglen, int64_t* out) {
    Model mo;
    HostDesc d{problem, scheme, pattern_mode, N, tg, tglen, g_control_steps, g_value_order};
    int st = build_model(d, mo, g_err);
    if (st) return st;
    out[0] = mo.L.nvar; out[1] = mo.L.ncon; out[2] = mo.nnzj; out[3] = mo.dropped;
    return 0;
}

int emu_csc(int problem, int scheme, int pattern_mode, int64_t N, const double* tg, int64_t tglen, int64_t* colptr, int64_t* rowval) {
    Model mo;
    HostDesc d{problem, scheme, pattern_mode, N, tg, tglen, g_control_steps, g_value_order};
    int st = build_model(d, mo, g_err);
    if (st) return st;
    std::vector<int64_t> rows;
    int64_t nz = 0;
    for (int64_t j = 0; j < mo.L.nvar; ++j) {
        colptr[j] = nz;
        if (mo.column_start(j) != nz) { g_err = "column_start mismatch"; return 99; }
        mo.gen_column(j, rows);
        for (int64_t r : rows) rowval[nz++] = r;
    }
    colptr[mo.L.nvar] = nz;
    if (nz != mo.nnzj) { g_err = "nnz mismatch"; return 98; }
    return 0;
}

// rowptr / colind of the pattern by rows (+ consistency of Model::row_start with the generated rows for CSR-order models), and
// info[0..5] = reg_first, reg_last, Lseg, vr, HL, HH of the model
int emu_csr(int problem, int scheme, int pattern_mode, int64_t N, const double* tg, int64_t tglen, int64_t* rowptr, int64_t* colind, int64_t* info) {
    Model mo;
    HostDesc d{problem, scheme, pattern_mode, N, tg, tglen, g_control_steps, g_value_order};
    int st = build_model(d, mo, g_err);
    if (st) return st;
    std::vector<int64_t> cols;
    int64_t nz = 0;
    for (int64_t r = 0; r < mo.L.ncon; ++r) {
        rowptr[r] = nz;
        if (mo.order == 1 && mo.row_start(r) != nz) { g_err = "row_start mismatch"; return 99; }
        mo.gen_row(r, cols);
        for (int64_t c : cols) colind[nz++] = c;
    }
    rowptr[mo.L.ncon] = nz;
    if (nz != mo.nnzj) { g_err = "nnz mismatch"; return 98; }
    if (info) { info[0] = mo.reg_first; info[1] = mo.reg_last; info[2] = mo.Lseg; info[3] = mo.vr; info[4] = mo.HL; info[5] = mo.HH; }
    return 0;
}

// value range [begin, end) a shard [step_begin, step_end) of the grid owns (Model::shard_vals_begin / _end)
int emu_shard_range(int problem, int scheme, int pattern_mode, int64_t N, const double* tg, int64_t tglen, int64_t step_begin, int64_t step_end, int64_t* out2) {
    Model mo;
    HostDesc d{problem, scheme, pattern_mode, N, tg, tglen, g_control_steps, g_value_order};
    int st = build_model(d, mo, g_err);
    if (st) return st;
    out2[0] = mo.shard_vals_begin(step_begin);
    out2[1] = mo.shard_vals_end(step_end);
    return 0;
}

int emu_cons_jac(int problem, int scheme, int pattern_mode, int64_t N, const double* tg, int64_t tglen, int tile, int nthr,
                 int64_t step_begin, int64_t step_end, const double* x, double* c, double* vals) {
    Model mo;
    HostDesc d{problem, scheme, pattern_mode, N, tg, tglen, g_control_steps, g_value_order};
    int st = build_model(d, mo, g_err);
    if (st) return st;
    if (step_end <= 0) { step_begin = 0; step_end = mo.L.N; }
    if (tile <= 0) tile = default_tile(mo);
    KParams kp;
    mo.fill_kparams(kp, step_begin, step_end, tile);
    if (const char* e = std::getenv("CTD_XCD")) kp.xcd_remap = std::atoi(e);      // same ablation knob as the engine
    if (const char* e = std::getenv("CTD_EMU_WG_STRIDE")) kp.wg_stride = std::atoi(e);      // multi-tile workgroups
    kp.tau = mo.uniform ? nullptr : mo.tau.data();
    kp.tmpl = mo.tmpl.data();
    kp.vtmpl = mo.vtmpl.data();
    kp.edge_idx = mo.edge_idx.data();
    kp.edge_code = mo.edge_code.data();
    kp.c = c;
    kp.vals = vals;
    // early emission as ctd_create switches it on (CTD_EMU_EARLY=1; the same conditions on the launch geometry)
    if (std::getenv("CTD_EMU_EARLY") && std::atoi(std::getenv("CTD_EMU_EARLY")) && mo.n_early > 0 && mo.fused && tile <= 32) {
        int lgT = 0;
        while ((1 << lgT) < tile) ++lgT;
        const int leadbase = ((((mo.L.s * mo.nch_dyn + mo.nch_path) << lgT)) + 63) & ~63;
        if (leadbase + 64 <= nthr && mo.n_early + mo.c_early + mo.L.nv * mo.vr_early <= 64 && mo.n_late > 0 && mo.n_late <= nthr) {
            kp.pos = mo.pos_order.data();
            kp.n_late = mo.n_late; kp.n_early = mo.n_early; kp.c_early = mo.c_early; kp.vr_early = mo.vr_early;
            kp.div_late = make_fastdiv((uint32_t)mo.n_late);
        }
    }
    bool ok = for_problem(problem, [&](auto tag) {
        using P = typename decltype(tag)::type;
        switch (mo.L.sc) {
            case SC_TRAPEZE: run_blocks<P, SC_TRAPEZE, 1>(kp, x, nthr); break;
            case SC_MIDPOINT:
                if (mo.L.cs == 2) run_blocks<P, SC_MIDPOINT, 2>(kp, x, nthr);
                else if (mo.L.cs == 3) run_blocks<P, SC_MIDPOINT, 3>(kp, x, nthr);
                else run_blocks<P, SC_MIDPOINT, 1>(kp, x, nthr);
                break;
            default:
                if (mo.L.s == 1) run_blocks<P, SC_IRK, 1>(kp, x, nthr);
                else if (mo.L.s == 2) run_blocks<P, SC_IRK, 2>(kp, x, nthr);
                else run_blocks<P, SC_IRK, 3>(kp, x, nthr);
                break;
        }
    });
    return ok ? 0 : 5;
}

// Sharded iterate read in place (ctd_set_x_shards): G shards with the engine's balanced split; shard k evaluates from a buffer
// of its own that holds ONLY its own variables (+ the replicated v) and NaN everywhere else, the entries of other shards come
// through the XHalo table from the owners' buffers.  All shards write into the shared c / vals.
int emu_cons_jac_sharded(int problem, int scheme, int pattern_mode, int64_t N, const double* tg, int64_t tglen, int tile, int nthr,
                         int G, const double* x, double* c, double* vals) {
    Model mo;
    HostDesc d{problem, scheme, pattern_mode, N, tg, tglen, g_control_steps, g_value_order};
    int st = build_model(d, mo, g_err);
    if (st) return st;
    const Layout& L = mo.L;
    if (G < 1 || G > kMaxShards || G > L.N) { g_err = "bad shard count"; return 97; }
    if (tile <= 0) tile = default_tile(mo);
    std::vector<int64_t> sb(G + 1);
    const int64_t base = L.N / G, rem = L.N % G;
    for (int k = 0; k <= G; ++k) sb[k] = k * base + (k < rem ? k : rem);
    std::vector<std::vector<double>> xs(G, std::vector<double>(L.nvar, std::numeric_limits<double>::quiet_NaN()));
    for (int k = 0; k < G; ++k) {
        const int64_t lo = sb[k] * L.blk, hi = (k == G - 1) ? L.v_off : sb[k + 1] * L.blk;
        for (int64_t g = lo; g < hi; ++g) xs[k][g] = x[g];
        for (int64_t g = L.v_off; g < L.nvar; ++g) xs[k][g] = x[g];
    }
    for (int k = 0; k < G; ++k) {
        XHalo hl{};
        hl.G = G; hl.self = k;
        for (int j = 0; j < G; ++j) { hl.vbegin[j] = sb[j] * L.blk; hl.x[j] = j == k ? nullptr : xs[j].data(); }
        hl.vbegin[G] = L.v_off;
        KParams kp;
        mo.fill_kparams(kp, sb[k], sb[k + 1], tile);
        if (const char* e = std::getenv("CTD_EMU_WG_STRIDE")) kp.wg_stride = std::atoi(e);
        kp.tau = mo.uniform ? nullptr : mo.tau.data();
        kp.tmpl = mo.tmpl.data();
        kp.vtmpl = mo.vtmpl.data();
        kp.edge_idx = mo.edge_idx.data();
        kp.edge_code = mo.edge_code.data();
        kp.c = c;
        kp.vals = vals;
        kp.halo = &hl;
        kp.near = make_xnear(hl, L.blk, L.N, L.v_off);
        const double* xk = xs[k].data();
        bool ok = for_problem(problem, [&](auto tag) {
            using P = typename decltype(tag)::type;
            switch (mo.L.sc) {
                case SC_TRAPEZE: run_blocks<P, SC_TRAPEZE, 1>(kp, xk, nthr); break;
                case SC_MIDPOINT:
                    if (mo.L.cs == 2) run_blocks<P, SC_MIDPOINT, 2>(kp, xk, nthr);
                    else if (mo.L.cs == 3) run_blocks<P, SC_MIDPOINT, 3>(kp, xk, nthr);
                    else run_blocks<P, SC_MIDPOINT, 1>(kp, xk, nthr);
                    break;
                default:
                    if (mo.L.s == 1) run_blocks<P, SC_IRK, 1>(kp, xk, nthr);
                    else if (mo.L.s == 2) run_blocks<P, SC_IRK, 2>(kp, xk, nthr);
                    else run_blocks<P, SC_IRK, 3>(kp, xk, nthr);
                    break;
            }
        });
        if (!ok) return 5;
    }
    return 0;
}

// index map of ctd_stitch_c's unpack kernel (ctd_layout.hpp stitch_src): out[r] = position of global row r in the gathered blocks
void emu_stitch_src(int64_t N, int cb, int G, int64_t smax, int64_t ncon, int64_t* out) {
    for (int64_t r = 0; r < ncon; ++r) out[r] = stitch_src(r, N, cb, G, smax);
}

int64_t emu_hess_nnz(int problem, int scheme, int pattern_mode, int64_t N, const double* tg, int64_t tglen) {
    Model mo;
    HostDesc d{problem, scheme, pattern_mode, N, tg, tglen, g_control_steps, g_value_order};
    if (build_model(d, mo, g_err)) return -1;
    return mo.H.nnzh;
}

// lower triangle of DOCP_Hessian_pattern, 0-based CSC
int emu_hess_csc(int problem, int scheme, int pattern_mode, int64_t N, const double* tg, int64_t tglen, int64_t* colptr, int64_t* rowval) {
    Model mo;
    HostDesc d{problem, scheme, pattern_mode, N, tg, tglen, g_control_steps, g_value_order};
    int st = build_model(d, mo, g_err);
    if (st) return st;
    std::vector<int64_t> rows;
    int64_t nz = 0;
    for (int64_t j = 0; j < mo.L.nvar; ++j) {
        colptr[j] = nz;
        if (mo.hess_column_start(j) != nz) { g_err = "hess_column_start mismatch"; return 99; }
        mo.hess_gen_column(j, rows);
        for (int64_t r : rows) rowval[nz++] = r;
    }
    colptr[mo.L.nvar] = nz;
    if (nz != mo.H.nnzh) { g_err = "nnzh mismatch"; return 98; }
    return 0;
}

int emu_hess(int problem, int scheme, int pattern_mode, int64_t N, const double* tg, int64_t tglen, int tile, int nthr,
             const double* x, const double* y, double obj_weight, double* vals, int64_t step_begin, int64_t step_end) {
    Model mo;
    HostDesc d{problem, scheme, pattern_mode, N, tg, tglen, g_control_steps, g_value_order};
    int st = build_model(d, mo, g_err);
    if (st) return st;
    if (tile <= 0) tile = default_hess_tile(mo);
    HParams hp;
    mo.fill_hparams(hp, tile, step_begin, step_end);
    if (const char* e = std::getenv("CTD_XCD")) hp.xcd_remap = std::atoi(e);
    const HessModel& H = mo.H;
    hp.tau = mo.uniform ? nullptr : mo.tau.data();
    hp.tptr = H.ctptr.data(); hp.terms = H.tcode.data(); hp.pair_c = H.pair_c.data();
    hp.cpos = H.cpos.data(); hp.zpos = H.zpos.data();
    hp.vptr = H.vptr.data(); hp.vterms = H.vterms.data();
    hp.edge_idx = H.edge_idx.data(); hp.eptr = H.eptr.data(); hp.evptr = H.evptr.data(); hp.eterms = H.eterms.data();
    hp.tasks = H.tasks.data(); hp.ptasks = H.ptasks.data(); hp.btasks = H.btasks.data();
    hp.obj_weight = obj_weight;
    hp.vals = vals;
    std::vector<double> partials((size_t)(hp.ntiles + hp.n_edge_blocks) * (hp.nvv > 0 ? hp.nvv : 1), std::numeric_limits<double>::quiet_NaN());
    hp.partials = partials.data();
    if (mo.L.cs > 3 && mo.L.cs != 5) { g_err = "emulator: Hessian kernels with 1, 2, 3 and 5 controls per step"; return 5; }
    bool ok = for_problem(problem, [&](auto tag) {
        using P = typename decltype(tag)::type;
        switch (mo.L.sc) {
            case SC_TRAPEZE: run_hess_blocks<P, SC_TRAPEZE, 1>(hp, x, y, nthr); break;
            case SC_MIDPOINT:
                if (mo.L.cs == 2) run_hess_blocks<P, SC_MIDPOINT, 2>(hp, x, y, nthr);
                else if (mo.L.cs == 3) run_hess_blocks<P, SC_MIDPOINT, 3>(hp, x, y, nthr);
                else if (mo.L.cs == 5) run_hess_blocks<P, SC_MIDPOINT, 5>(hp, x, y, nthr);      // (summed points: hess_sums_stages)
                else run_hess_blocks<P, SC_MIDPOINT, 1>(hp, x, y, nthr);
                break;
            default:
                if (mo.L.s == 1) run_hess_blocks<P, SC_IRK, 1>(hp, x, y, nthr);
                else if (mo.L.s == 2) run_hess_blocks<P, SC_IRK, 2>(hp, x, y, nthr);
                else run_hess_blocks<P, SC_IRK, 3>(hp, x, y, nthr);
                break;
        }
    });
    return ok ? 0 : 5;
}

// Symbolic engine check (ctd_sym.hpp, test infrastructure): parses `expr` (kind-0 grammar) with t, x, u, v bound to the
// differentiation variables 0 .. n+m+nv (t first), differentiates it twice symbolically and compares every second
// derivative, evaluated at `point`, with central differences of the symbolic first derivatives.  Returns the largest
// difference relative to max(1, |value|) in *max_err and the number of DAG nodes in *nnodes.
int emu_sym_check(const char* expr, int n, int m, int nv, const double* point, double* max_err, int64_t* nnodes) {
    const int nz = 1 + n + m + nv;
    sym::Graph g;
    std::vector<int> X(n > 0 ? n : 1), U(m > 0 ? m : 1), V(nv > 0 ? nv : 1);
    for (int r = 0; r < n; ++r) X[r] = g.var(1 + r);
    for (int b = 0; b < m; ++b) U[b] = g.var(1 + n + b);
    for (int k = 0; k < nv; ++k) V[k] = g.var(1 + n + m + k);
    ExprCtx cx{n, m, nv, 0, {}, {}};
    const std::string text(expr);
    Parser ps(text, cx);
    ps.g = &g; ps.g_t = g.var(0); ps.g_x = X.data(); ps.g_u = U.data(); ps.g_v = V.data();
    Parser::Val val;
    if (!ps.expr(val)) { g_err = ps.err; return 1; }
    const std::vector<double> prm;
    std::vector<double> z(point, point + nz);
    double worst = 0.0;
    for (int i = 0; i < nz; ++i) {
        const int di = g.diff(val.node, i);
        for (int j = 0; j < nz; ++j) {
            const double s = g.eval(g.diff(di, j), prm, z);
            const double hstep = 1e-6 * std::max(1.0, std::fabs(z[j]));
            std::vector<double> zp = z, zm = z;
            zp[j] += hstep; zm[j] -= hstep;
            const double fd = (g.eval(di, prm, zp) - g.eval(di, prm, zm)) / (2.0 * hstep);
            worst = std::max(worst, std::fabs(s - fd) / std::max(1.0, std::fabs(s)));
        }
    }
    *max_err = worst;
    *nnodes = (int64_t)g.nodes.size();
    return 0;
}

}  // extern "C"

// ---- a run-time OCP in emulation (TEST INFRASTRUCTURE ONLY) --------------------------------------------------------------------
// Build with -DCTD_EMU_USER_OCP_HEADER='"file.hpp"' where file.hpp holds the functor text ctd_ocp_source returns for an OCP
// (namespace ctd { struct UserOCP ... }): the SAME phase templates then run with that functor, serially, with the LDS on the heap
// -- AddressSanitizer / UBSan can look at the tiles of a hiprtc-compiled kernel's logic (tests/test_jit_cpu.py).  The OCP is
// registered again inside this library (its own registry) from the same ctd_ocp_def, so host model and functor match.
#ifdef CTD_EMU_USER_OCP_HEADER
#include CTD_EMU_USER_OCP_HEADER
extern "C" {
int emu_user_register(const ctd_ocp_def* def, int* id) {
    const int st = register_runtime_ocp(def, id, g_err);
    return st;
}
int emu_user_cons_jac(int problem, int scheme, int pattern_mode, int64_t N, int tile, int nthr, const double* x, double* c, double* vals, int64_t* sizes) {
    Model mo;
    HostDesc d{problem, scheme, pattern_mode, N, nullptr, 0, g_control_steps, g_value_order};
    int st = build_model(d, mo, g_err);
    if (st) return st;
    if (sizes) { sizes[0] = mo.L.nvar; sizes[1] = mo.L.ncon; sizes[2] = mo.nnzj; }
    if (!x) return 0;
    if (tile <= 0) tile = default_tile(mo);
    KParams kp;
    mo.fill_kparams(kp, 0, mo.L.N, tile);
    kp.tau = mo.uniform ? nullptr : mo.tau.data();
    kp.tmpl = mo.tmpl.data();
    kp.vtmpl = mo.vtmpl.data();
    kp.edge_idx = mo.edge_idx.data();
    kp.edge_code = mo.edge_code.data();
    kp.c = c;
    kp.vals = vals;
    using P = UserOCP;
    switch (mo.L.sc) {
        case SC_TRAPEZE: run_blocks<P, SC_TRAPEZE, 1>(kp, x, nthr); break;
        case SC_MIDPOINT: run_blocks<P, SC_MIDPOINT, 1>(kp, x, nthr); break;
        default:
            if (mo.L.s == 1) run_blocks<P, SC_IRK, 1>(kp, x, nthr);
            else if (mo.L.s == 2) run_blocks<P, SC_IRK, 2>(kp, x, nthr);
            else run_blocks<P, SC_IRK, 3>(kp, x, nthr);
            break;
    }
    return 0;
}
}  // extern "C"
#endif

