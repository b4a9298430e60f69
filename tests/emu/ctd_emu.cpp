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
#include <limits>
#include <string>
#include <vector>

#include "../../ctdirect.jl_amd/csrc/ctd_host.cpp"
#include "../../ctdirect.jl_amd/csrc/ctd_kernel_body.hpp"

using namespace ctd;

template <class P, int SC, int S>
static void run_blocks(const KParams& kp, const double* xu, int nthr) {
    const int nblocks = kp.ntiles + (kp.has_edge ? 1 : 0);
    const int64_t nlds = lds_doubles(kp);
    for (int b = 0; b < nblocks; ++b) {
        std::vector<double> lds(nlds, std::numeric_limits<double>::quiet_NaN());
        BlockCtx cx = make_ctx(kp, b, lds.data());
        for (int t = 0; t < nthr; ++t) phase_load<P, SC, S>(kp, cx, xu, t, nthr);
        for (int t = 0; t < nthr; ++t) phase_eval<P, SC, S>(kp, cx, t, nthr);
        for (int t = 0; t < nthr; ++t) phase_fin<P, SC, S>(kp, cx, t, nthr);
        for (int t = 0; t < nthr; ++t) phase_fin2<P, SC, S>(kp, cx, t, nthr);
        for (int t = 0; t < nthr; ++t) phase_emit<P, SC, S>(kp, cx, t, nthr);
    }
}

static std::string g_err;

extern "C" {

const char* emu_last_error() { return g_err.c_str(); }

// out[0..3] = nvar, ncon, nnzj, dropped
int emu_sizes(int problem, int scheme, int pattern_mode, int64_t N, const double* tg, int64_t tglen, int64_t* out) {
    Model mo;
    HostDesc d{problem, scheme, pattern_mode, N, tg, tglen};
    int st = build_model(d, mo, g_err);
    if (st) return st;
    out[0] = mo.L.nvar; out[1] = mo.L.ncon; out[2] = mo.nnzj; out[3] = mo.dropped;
    return 0;
}

int emu_csc(int problem, int scheme, int pattern_mode, int64_t N, const double* tg, int64_t tglen, int64_t* colptr, int64_t* rowval) {
    Model mo;
    HostDesc d{problem, scheme, pattern_mode, N, tg, tglen};
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

int emu_cons_jac(int problem, int scheme, int pattern_mode, int64_t N, const double* tg, int64_t tglen, int tile, int nthr,
                 int64_t step_begin, int64_t step_end, const double* x, double* c, double* vals) {
    Model mo;
    HostDesc d{problem, scheme, pattern_mode, N, tg, tglen};
    int st = build_model(d, mo, g_err);
    if (st) return st;
    if (step_end <= 0) { step_begin = 0; step_end = mo.L.N; }
    if (tile <= 0) tile = default_tile(mo);
    KParams kp;
    mo.fill_kparams(kp, step_begin, step_end, tile);
    kp.tau = mo.uniform ? nullptr : mo.tau.data();
    kp.tmpl = mo.tmpl.data();
    kp.vtmpl = mo.vtmpl.data();
    kp.edge_idx = mo.edge_idx.data();
    kp.edge_code = mo.edge_code.data();
    kp.c = c;
    kp.vals = vals;
    bool ok = for_problem(problem, [&](auto tag) {
        using P = typename decltype(tag)::type;
        switch (mo.L.sc) {
            case SC_TRAPEZE: run_blocks<P, SC_TRAPEZE, 1>(kp, x, nthr); break;
            case SC_MIDPOINT: run_blocks<P, SC_MIDPOINT, 1>(kp, x, nthr); break;
            default:
                if (mo.L.s == 1) run_blocks<P, SC_IRK, 1>(kp, x, nthr);
                else if (mo.L.s == 2) run_blocks<P, SC_IRK, 2>(kp, x, nthr);
                else run_blocks<P, SC_IRK, 3>(kp, x, nthr);
                break;
        }
    });
    return ok ? 0 : 5;
}

}  // extern "C"
