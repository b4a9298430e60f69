// ctd_kernel_body.hpp -- the fused constraints + sparse-Jacobian evaluation, written as phase functions.
//
// One workgroup evaluates a TILE of consecutive time steps of the collocation grid (reference loop:
// `for i in 1:docp.time.steps`, src/DOCP_functions.jl:92-98), in four phases separated by workgroup barriers:
//
//   load   the tile's slice of the NLP vector xu (step-major, external layout) is copied once into LDS with
//          coalesced loads; every later read of X_i, U_i^j, K_i^j, X_{i+1} comes from LDS
//   eval   one lane per (step, eval point, direction chunk): the OCP dynamics (and path constraints) are evaluated
//          on forward duals in registers -> df/dx, df/du, df/dt, df/dv and the values land in the step's LDS record
//          (replaces setWorkArray + stepStateConstraints! + stepPathConstraints! AND the ncolors Dual passes of
//          ADNLPModels: trapeze.jl:50-71,118-142, midpoint.jl:47-72,124-140, irk.jl:236-308,
//          irk_stagewise.jl:394-460, DOCP_functions.jl:122-140)
//   fin    one lane per step: scheme chain rule pieces that need a whole step (residual rows, d/dv through the
//          free time grid of get_time_grid, DOCP_data.jl:437-458, per-step coefficients -h a_jl, -h b_j, ...)
//   emit   all lanes stream the outputs in their final external order with coalesced 8-byte stores:
//          c rows of the tile, the tile's contiguous range of CSC values (one 32-bit code per entry of the
//          step-periodic pattern: value = coef * record[di] + beta), and the tile's slice of every V column
//
// The first workgroup is the EDGE block: boundary constraints (DOCP_functions.jl:103-111), path constraints at
// the final time (:100) and the few CSC entries whose layout is not step-periodic (first and last step columns,
// final-state columns, tails of the V columns), driven by an explicit (index, code) list.
//
// The phase functions are plain templates over (OCP functor, scheme class); `tid`/`nthr` are the lane id and
// workgroup size.  ctd_kernels.hip wraps them in the __global__ kernel; tests/emu/ steps them serially on the CPU
// (test infrastructure only -- the C ABI never takes that path).
#pragma once
#include "ctd_layout.hpp"

namespace ctd {

struct BlockCtx {
    int is_edge;
    int nslots;        // records held by this block (step / node records)
    int in_stride;     // doubles between the inputs of consecutive slots
    int64_t a, b;      // steps [a, b) whose outputs this tile emits
    int64_t lo;        // step / node index of slot 0 (tile)
    double* in;        // staged slice of xu
    double* v;         // optimisation variables
    double* rec;       // records
};

CTD_HD int64_t slot_index(const KParams& kp, const BlockCtx& cx, int k) {
    return cx.is_edge ? kp.edge_steps[k] : cx.lo + k;
}

CTD_HD BlockCtx make_ctx(const KParams& kp, int block, double* lds) {
    BlockCtx cx;
    const Layout& L = kp.L;
    if (kp.has_edge && block == 0) {
        cx.is_edge = 1;
        cx.nslots = kp.n_edge_slots;
        cx.in_stride = L.blk + L.n + L.m;
        cx.a = cx.b = cx.lo = 0;
        cx.in = lds;
        cx.v = cx.in + (int64_t)cx.nslots * cx.in_stride;
        cx.rec = cx.v + kMaxNV;
    } else {
        const int tile = block - (kp.has_edge ? 1 : 0);
        cx.is_edge = 0;
        cx.a = kp.step_begin + (int64_t)tile * kp.T;
        cx.b = cx.a + kp.T < kp.step_end ? cx.a + kp.T : kp.step_end;
        cx.lo = cx.a - kp.HL;
        cx.nslots = (int)(cx.b - cx.a) + kp.HL + kp.HH;
        cx.in_stride = L.blk;
        cx.in = lds;
        cx.v = cx.in + (int64_t)(kp.T + kp.HL + kp.HH + 1) * L.blk + L.n + L.m;
        cx.rec = cx.v + kMaxNV;
    }
    return cx;
}

// LDS doubles a block needs (host uses this to size the launch)
inline int64_t lds_doubles(const KParams& kp) {
    const Layout& L = kp.L;
    int64_t tile = (int64_t)(kp.T + kp.HL + kp.HH + 1) * L.blk + L.n + L.m + kMaxNV + (int64_t)(kp.T + kp.HL + kp.HH) * kp.R.stride;
    int64_t edge = (int64_t)kp.n_edge_slots * (L.blk + L.n + L.m) + kMaxNV + (int64_t)(kp.n_edge_slots + 2) * kp.R.stride;
    return tile > edge ? tile : edge;
}

// normalized time of grid point i: collect(LinRange(0, 1, N+1))[i+1] = i / N (src/DOCP_data.jl:179-183), or the
// user grid normalised on the host (:191-199)
CTD_HD double tau_at(const KParams& kp, int64_t i) {
    return kp.tau ? kp.tau[i] : (double)i / (double)kp.L.N;
}

// get_time_grid (src/DOCP_data.jl:437-458): t_i = t0 + tau_i (tf - t0), t0/tf fixed or components of v
template <class P> CTD_HD void time_ends(const KParams& kp, const double* v, double& t0, double& tf) {
    t0 = (P::IT0 >= 0) ? v[P::IT0 >= 0 ? P::IT0 : 0] : kp.L.t0;
    tf = (P::ITF >= 0) ? v[P::ITF >= 0 ? P::ITF : 0] : kp.L.tf;
}
template <class P> CTD_HD double time_at(const KParams& kp, const double* v, int64_t i) {
    double t0, tf;
    time_ends<P>(kp, v, t0, tf);
    return t0 + tau_at(kp, i) * (tf - t0);
}
// d t_i / d v_k, following the dual arithmetic of the same expression
template <class P> CTD_HD double dtime_at(const KParams& kp, int64_t i, int k) {
    const double dt0 = (P::IT0 == k) ? 1.0 : 0.0;
    const double dtf = (P::ITF == k) ? 1.0 : 0.0;
    return dt0 + tau_at(kp, i) * (dtf - dt0);
}

// ------------------------------------------------------------------------------------------------------
// phase: load
// ------------------------------------------------------------------------------------------------------
template <class P, int SC>
CTD_HD void phase_load(const KParams& kp, const BlockCtx& cx, const double* __restrict__ xu, int tid, int nthr) {
    const Layout& L = kp.L;
    if (cx.is_edge) {
        const int per = cx.in_stride;
        for (int e = tid; e < cx.nslots * per; e += nthr) {
            const int k = e / per, o = e - k * per;
            const int64_t g = kp.edge_steps[k] * L.blk + o;
            cx.in[e] = (g < L.v_off) ? xu[g] : 0.0;
        }
    } else {
        const int64_t g0 = (cx.lo < 0 ? 0 : cx.lo) * (int64_t)L.blk;
        int64_t g1 = (cx.lo + cx.nslots) * (int64_t)L.blk + L.n + L.m;
        if (g1 > L.v_off) g1 = L.v_off;
        const int64_t shift = cx.lo * (int64_t)L.blk;
        for (int64_t g = g0 + tid; g < g1; g += nthr) cx.in[g - shift] = xu[g];
    }
    if (tid < kMaxNV) cx.v[tid] = (tid < P::NV) ? xu[L.v_off + tid] : 0.0;
}

// ------------------------------------------------------------------------------------------------------
// phase: eval (dual evaluation of the OCP functions)
// ------------------------------------------------------------------------------------------------------
template <class P> struct Dirs {
    static constexpr int DYN = P::NX + P::NU + (P::DYN_T ? 1 : 0) + (P::DYN_V ? P::NV : 0);
    static constexpr int PATH = P::NX + P::NU + (P::PATH_T ? 1 : 0) + (P::PATH_V ? P::NV : 0);
    static constexpr int BND = 2 * P::NX + P::NV;
    static constexpr int DC = P::DC;
    static constexpr int NCH_DYN = (DYN + DC - 1) / DC;
    static constexpr int NCH_PATH = (PATH + DC - 1) / DC;
    static constexpr int NCH_BND = (BND + DC - 1) / DC;
};

// the control seen by path constraints: U_i, or for stagewise schemes the b-weighted stage average
// (get_OCP_control_at_time_step, src/ode/common.jl:140-155 / irk_stagewise.jl:197-205)
template <class P> CTD_HD void node_control(const KParams& kp, const double* base, double* u) {
    const Layout& L = kp.L;
    if (L.stagewise) {
        for (int c = 0; c < P::NU; ++c) u[c] = L.b[0] * base[P::NX + c];
        for (int j = 1; j < L.s; ++j)
            for (int c = 0; c < P::NU; ++c) u[c] = u[c] + L.b[j] * base[P::NX + j * P::NU + c];
    } else {
        for (int c = 0; c < P::NU; ++c) u[c] = base[P::NX + c];
    }
}

// one dynamics evaluation on duals: slot k (step or node i), eval point j, direction chunk q
template <class P, int SC>
CTD_HD void eval_dynamics(const KParams& kp, const BlockCtx& cx, int k, int j, int q) {
    constexpr int n = P::NX, m = P::NU, nv = P::NV, DC = P::DC;
    using D = Dual<DC>;
    const Layout& L = kp.L;
    const RecLayout& R = kp.R;
    const int64_t i = slot_index(kp, cx, k);
    if (i < 0) return;
    if (SC == SC_TRAPEZE ? (i > L.N) : (i >= L.N)) return;
    const double* base = cx.in + (int64_t)k * cx.in_stride;
    double xv[n > 0 ? n : 1], uv[m > 0 ? m : 1];
    double t;
    if (SC == SC_TRAPEZE) {                       // f(t_i, X_i, U_i, v): trapeze.jl:60-69
        t = time_at<P>(kp, cx.v, i);
        for (int c = 0; c < n; ++c) xv[c] = base[c];
        for (int c = 0; c < m; ++c) uv[c] = base[n + c];
    } else if (SC == SC_MIDPOINT) {               // f(0.5(t_i+t_{i+1}), 0.5(X_i+X_{i+1}), U_i, v): midpoint.jl:53-66
        t = 0.5 * (time_at<P>(kp, cx.v, i) + time_at<P>(kp, cx.v, i + 1));
        for (int c = 0; c < n; ++c) xv[c] = 0.5 * (base[c] + base[L.blk + c]);
        for (int c = 0; c < m; ++c) uv[c] = base[n + c];
    } else {                                      // f(t_i + c_j h, X_i + h sum_l a_jl K^l, U_i^j | U_i, v): irk_stagewise.jl:424-446
        const double ti = time_at<P>(kp, cx.v, i);
        const double h = time_at<P>(kp, cx.v, i + 1) - ti;
        t = ti + L.c[j] * h;
        const double* K = base + n + L.cu;
        for (int c = 0; c < n; ++c) {
            double x = base[c];
            for (int l = 0; l < L.s; ++l) x = x + h * L.a[3 * j + l] * K[l * n + c];
            xv[c] = x;
        }
        const double* U = base + n + (L.stagewise ? j * m : 0);
        for (int c = 0; c < m; ++c) uv[c] = U[c];
    }
    // seed directions [x | u | t | v] of this chunk
    // (compare-and-select seeding keeps the dual arrays in registers: no runtime-indexed private arrays)
    D X[n > 0 ? n : 1], U[m > 0 ? m : 1], V[nv > 0 ? nv : 1], Tt, out[n > 0 ? n : 1];
    const int g0 = q * DC;
    constexpr int gT = n + m, gV = n + m + (P::DYN_T ? 1 : 0);
#pragma unroll
    for (int c = 0; c < n; ++c) {
        X[c].v = xv[c];
#pragma unroll
        for (int d = 0; d < DC; ++d) X[c].d[d] = (g0 + d == c) ? 1.0 : 0.0;
    }
#pragma unroll
    for (int c = 0; c < m; ++c) {
        U[c].v = uv[c];
#pragma unroll
        for (int d = 0; d < DC; ++d) U[c].d[d] = (g0 + d == n + c) ? 1.0 : 0.0;
    }
    Tt.v = t;
#pragma unroll
    for (int d = 0; d < DC; ++d) Tt.d[d] = (P::DYN_T && g0 + d == gT) ? 1.0 : 0.0;
#pragma unroll
    for (int c = 0; c < nv; ++c) {
        V[c].v = cx.v[c];
#pragma unroll
        for (int d = 0; d < DC; ++d) V[c].d[d] = (P::DYN_V && g0 + d == gV + c) ? 1.0 : 0.0;
    }
    P::template dynamics<D>(out, Tt, X, U, V);
    double* ev = cx.rec + (int64_t)k * R.stride + R.oEval + j * R.eval_sz;
#pragma unroll
    for (int d = 0; d < DC; ++d) {
        const int g = g0 + d;
        if (g < n) { for (int r = 0; r < n; ++r) ev[R.oF + r * n + g] = out[r].d[d]; }
        else if (g < n + m) { for (int r = 0; r < n; ++r) ev[R.oG + r * m + (g - n)] = out[r].d[d]; }
        else if (P::DYN_T && g == n + m) { for (int r = 0; r < n; ++r) ev[R.oft + r] = out[r].d[d]; }
        else if (P::DYN_V && g < Dirs<P>::DYN) {
            const int kk = g - n - m - (P::DYN_T ? 1 : 0);
            for (int r = 0; r < n; ++r) ev[R.oW + r * nv + kk] = out[r].d[d];
        }
    }
    if (q == 0) for (int r = 0; r < n; ++r) ev[R.of + r] = out[r].v;
}

// path constraints g(t, x, u, v) on duals into record `rec`: stepPathConstraints!, DOCP_functions.jl:122-140
template <class P>
CTD_HD void eval_path(const KParams& kp, double* rec, double t, const double* xv, const double* uv, const double* vv, int q,
                      int value_off) {
    constexpr int n = P::NX, m = P::NU, nv = P::NV, np = P::NPATH, DC = P::DC;
    using D = Dual<DC>;
    const RecLayout& R = kp.R;
    D X[n > 0 ? n : 1], U[m > 0 ? m : 1], V[nv > 0 ? nv : 1], Tt, out[np > 0 ? np : 1];
    const int g0 = q * DC;
    constexpr int gT = n + m, gV = n + m + (P::PATH_T ? 1 : 0);
#pragma unroll
    for (int c = 0; c < n; ++c) {
        X[c].v = xv[c];
#pragma unroll
        for (int d = 0; d < DC; ++d) X[c].d[d] = (g0 + d == c) ? 1.0 : 0.0;
    }
#pragma unroll
    for (int c = 0; c < m; ++c) {
        U[c].v = uv[c];
#pragma unroll
        for (int d = 0; d < DC; ++d) U[c].d[d] = (g0 + d == n + c) ? 1.0 : 0.0;
    }
    Tt.v = t;
#pragma unroll
    for (int d = 0; d < DC; ++d) Tt.d[d] = (P::PATH_T && g0 + d == gT) ? 1.0 : 0.0;
#pragma unroll
    for (int c = 0; c < nv; ++c) {
        V[c].v = vv[c];
#pragma unroll
        for (int d = 0; d < DC; ++d) V[c].d[d] = (P::PATH_V && g0 + d == gV + c) ? 1.0 : 0.0;
    }
    P::template path<D>(out, Tt, X, U, V);
#pragma unroll
    for (int d = 0; d < DC; ++d) {
        const int g = g0 + d;
        if (g < n) { for (int r = 0; r < np; ++r) rec[R.oPx + r * n + g] = out[r].d[d]; }
        else if (g < n + m) { for (int r = 0; r < np; ++r) rec[R.oPu + r * m + (g - n)] = out[r].d[d]; }
        else if (P::PATH_T && g == n + m) { for (int r = 0; r < np; ++r) rec[R.oPt + r] = out[r].d[d]; }
        else if (P::PATH_V && g < Dirs<P>::PATH) {
            const int kk = g - n - m - (P::PATH_T ? 1 : 0);
            for (int r = 0; r < np; ++r) rec[R.oPv + r * nv + kk] = out[r].d[d];
        }
    }
    if (q == 0) for (int r = 0; r < np; ++r) rec[value_off + r] = out[r].v;
}

template <class P, int SC>
CTD_HD void eval_step_path(const KParams& kp, const BlockCtx& cx, int k, int q) {
    constexpr int n = P::NX, m = P::NU;
    const Layout& L = kp.L;
    const int64_t i = slot_index(kp, cx, k);
    if (i < 0 || i >= L.N) return;
    const double* base = cx.in + (int64_t)k * cx.in_stride;
    double uv[m > 0 ? m : 1];
    node_control<P>(kp, base, uv);
    double xv[n > 0 ? n : 1];
    for (int c = 0; c < n; ++c) xv[c] = base[c];
    eval_path<P>(kp, cx.rec + (int64_t)k * kp.R.stride, time_at<P>(kp, cx.v, i), xv, uv, cx.v, q, kp.R.oR + L.eqs);
}

// path constraints at the final time (DOCP_functions.jl:100) with the convention u(tf) = U_N unless U_{N+1} exists
template <class P, int SC>
CTD_HD void eval_final_path(const KParams& kp, const BlockCtx& cx, int q) {
    constexpr int n = P::NX, m = P::NU;
    const Layout& L = kp.L;
    const double* base = cx.in + (int64_t)kp.edge_slot_last * cx.in_stride;
    double xv[n > 0 ? n : 1], uv[m > 0 ? m : 1];
    for (int c = 0; c < n; ++c) xv[c] = base[L.blk + c];
    if (SC == SC_TRAPEZE) { for (int c = 0; c < m; ++c) uv[c] = base[L.blk + n + c]; }
    else node_control<P>(kp, base, uv);
    eval_path<P>(kp, cx.rec + (int64_t)kp.edge_fp * kp.R.stride, time_at<P>(kp, cx.v, L.N), xv, uv, cx.v, q, kp.R.oR);
}

// boundary constraints phi(x0, xf, v) on duals: DOCP_functions.jl:103-111
template <class P>
CTD_HD void eval_boundary(const KParams& kp, const BlockCtx& cx, int q) {
    constexpr int n = P::NX, nv = P::NV, nb = P::NBC, DC = P::DC;
    using D = Dual<DC>;
    const Layout& L = kp.L;
    const RecLayout& R = kp.R;
    const double* b0 = cx.in + (int64_t)kp.edge_slot_first * cx.in_stride;
    const double* bf = cx.in + (int64_t)kp.edge_slot_last * cx.in_stride + L.blk;
    D X0[n > 0 ? n : 1], XF[n > 0 ? n : 1], V[nv > 0 ? nv : 1], out[nb > 0 ? nb : 1];
    const int g0 = q * DC;
#pragma unroll
    for (int c = 0; c < n; ++c) {
        X0[c].v = b0[c];
        XF[c].v = bf[c];
#pragma unroll
        for (int d = 0; d < DC; ++d) {
            X0[c].d[d] = (g0 + d == c) ? 1.0 : 0.0;
            XF[c].d[d] = (g0 + d == n + c) ? 1.0 : 0.0;
        }
    }
#pragma unroll
    for (int c = 0; c < nv; ++c) {
        V[c].v = cx.v[c];
#pragma unroll
        for (int d = 0; d < DC; ++d) V[c].d[d] = (g0 + d == 2 * n + c) ? 1.0 : 0.0;
    }
    P::template boundary<D>(out, X0, XF, V);
    double* rec = cx.rec + (int64_t)kp.edge_b * R.stride;
#pragma unroll
    for (int d = 0; d < DC; ++d) {
        const int g = g0 + d;
        if (g < n) { for (int r = 0; r < nb; ++r) rec[R.oB0 + r * n + g] = out[r].d[d]; }
        else if (g < 2 * n) { for (int r = 0; r < nb; ++r) rec[R.oBf + r * n + (g - n)] = out[r].d[d]; }
        else if (g < 2 * n + nv) { for (int r = 0; r < nb; ++r) rec[R.oBv + r * nv + (g - 2 * n)] = out[r].d[d]; }
    }
    if (q == 0) for (int r = 0; r < nb; ++r) rec[R.oBval + r] = out[r].v;
}

template <class P, int SC>
CTD_HD void phase_eval(const KParams& kp, const BlockCtx& cx, int tid, int nthr) {
    const RecLayout& R = kp.R;
    const int ns = cx.nslots;
    const int n_dyn = R.S * Dirs<P>::NCH_DYN * ns;
    const int n_path = (P::NPATH > 0) ? Dirs<P>::NCH_PATH * ns : 0;
    int n_fp = 0, n_b = 0;
    if (cx.is_edge) {
        n_fp = (P::NPATH > 0) ? Dirs<P>::NCH_PATH : 0;
        n_b = (P::NBC > 0) ? Dirs<P>::NCH_BND : 0;
    }
    const int total = n_dyn + n_path + n_fp + n_b;
    for (int task = tid; task < total; task += nthr) {
        if (task < n_dyn) {
            // slot fastest: neighbouring lanes run the same (eval point, chunk) on neighbouring steps
            const int k = task % ns, jq = task / ns;
            eval_dynamics<P, SC>(kp, cx, k, jq / Dirs<P>::NCH_DYN, jq % Dirs<P>::NCH_DYN);
        } else if (task < n_dyn + n_path) {
            const int t2 = task - n_dyn;
            eval_step_path<P, SC>(kp, cx, t2 % ns, t2 / ns);
        } else if (task < n_dyn + n_path + n_fp) {
            eval_final_path<P, SC>(kp, cx, task - n_dyn - n_path);
        } else {
            eval_boundary<P>(kp, cx, task - n_dyn - n_path - n_fp);
        }
    }
    // record header: [0] = 1.0 (every record of the block, used by constant entries of the pattern)
    const int nrec = cx.is_edge ? ns + 2 : ns;
    for (int k = tid; k < nrec; k += nthr) cx.rec[(int64_t)k * R.stride] = 1.0;
}

// ------------------------------------------------------------------------------------------------------
// phase: fin (one lane per record)
// ------------------------------------------------------------------------------------------------------
template <class P> CTD_HD void fill_const_coefs(const KParams& kp, double* C) {
    for (int e = 0; e < kNC; ++e) C[e] = 0.0;
    C[C_ZERO] = 0.0; C[C_ONE] = 1.0; C[C_NEG1] = -1.0;
    for (int j = 0; j < 3; ++j) C[C_B + j] = kp.L.b[j];
}

template <class P, int SC>
CTD_HD void finalize_step(const KParams& kp, const BlockCtx& cx, int k) {
    constexpr int n = P::NX, m = P::NU, nv = P::NV, np = P::NPATH;
    constexpr bool FREE = (P::IT0 >= 0) || (P::ITF >= 0);
    const Layout& L = kp.L;
    const RecLayout& R = kp.R;
    const int64_t i = slot_index(kp, cx, k);
    double* rec = cx.rec + (int64_t)k * R.stride;
    double* C = rec + R.oC;
    fill_const_coefs<P>(kp, C);
    if (i < 0 || i > L.N) return;
    const bool is_step = i < L.N;
    const double* base = cx.in + (int64_t)k * cx.in_stride;
    const double ti = time_at<P>(kp, cx.v, i);
    const double tip1 = is_step ? time_at<P>(kp, cx.v, i + 1) : ti;
    double dti[nv > 0 ? nv : 1], dti1[nv > 0 ? nv : 1], dh[nv > 0 ? nv : 1];
    for (int kk = 0; kk < nv; ++kk) {
        dti[kk] = FREE ? dtime_at<P>(kp, i, kk) : 0.0;
        dti1[kk] = (FREE && is_step) ? dtime_at<P>(kp, i + 1, kk) : dti[kk];
        dh[kk] = dti1[kk] - dti[kk];
    }
    // path rows: total d/dv = explicit + dg/dt * dt_i/dv
    if (np > 0 && is_step) {
        for (int r = 0; r < np; ++r)
            for (int kk = 0; kk < nv; ++kk) {
                double pv = P::PATH_V ? rec[R.oPv + r * nv + kk] : 0.0;
                if (P::PATH_T && FREE) pv = pv + rec[R.oPt + r] * dti[kk];
                rec[R.oPv + r * nv + kk] = pv;
            }
    }
    if (SC == SC_IRK) {
        if (!is_step) return;
        const double h = tip1 - ti;
        for (int j = 0; j < L.s; ++j) {
            for (int l = 0; l < L.s; ++l) C[C_HA + 3 * j + l] = -(h * L.a[3 * j + l]);
            C[C_HB + j] = -(h * L.b[j]);
        }
        const double* K = base + n + L.cu;
        double* Rr = rec + R.oR;
        for (int j = 0; j < L.s; ++j) {
            double* ev = rec + R.oEval + j * R.eval_sz;
            // stage rows: K_i^j - f(...)   (irk_stagewise.jl:448-451)
            for (int r = 0; r < n; ++r) Rr[n + j * n + r] = K[j * n + r] - ev[R.of + r];
            if (nv > 0) {
                for (int kk = 0; kk < nv; ++kk) {
                    // d x_ij / d v_kk = dh * sum_l a_jl K^l  (x_i itself does not depend on v)
                    double dx[n > 0 ? n : 1];
                    for (int c = 0; c < n; ++c) {
                        double acc = 0.0;
                        for (int l = 0; l < L.s; ++l) acc = acc + (dh[kk] * L.a[3 * j + l]) * K[l * n + c];
                        dx[c] = acc;
                    }
                    const double dtij = dti[kk] + L.c[j] * dh[kk];
                    for (int r = 0; r < n; ++r) {
                        double w = P::DYN_V ? ev[R.oW + r * nv + kk] : 0.0;
                        if (P::DYN_T && FREE) w = w + ev[R.oft + r] * dtij;
                        if (FREE) for (int c = 0; c < n; ++c) w = w + ev[R.oF + r * n + c] * dx[c];
                        ev[R.oW + r * nv + kk] = w;
                    }
                }
            }
        }
        // state rows: X_{i+1} - (X_i + h sum_j b_j K^j)   (irk_stagewise.jl:456-457)
        for (int r = 0; r < n; ++r) {
            double sumbk = L.b[0] * K[r];
            for (int j = 1; j < L.s; ++j) sumbk = sumbk + L.b[j] * K[j * n + r];
            Rr[r] = base[L.blk + r] - (base[r] + h * sumbk);
            for (int kk = 0; kk < nv; ++kk) rec[R.oSv + r * nv + kk] = -(dh[kk] * sumbk);
        }
    } else if (SC == SC_MIDPOINT) {
        if (!is_step) return;
        const double h = (tip1 - ti) / 1.0;
        C[C_NHH] = -(0.5 * h);
        C[C_NH] = -h;
        double* ev = rec + R.oEval;
        double* Rr = rec + R.oR;
        for (int r = 0; r < n; ++r) {
            Rr[r] = base[L.blk + r] - (base[r] + h * ev[R.of + r]);    // midpoint.jl:139
            for (int kk = 0; kk < nv; ++kk) {
                double w = P::DYN_V ? ev[R.oW + r * nv + kk] : 0.0;
                if (P::DYN_T && FREE) w = w + ev[R.oft + r] * (0.5 * (dti[kk] + dti1[kk]));
                ev[R.oW + r * nv + kk] = w;
                rec[R.oSv + r * nv + kk] = -(dh[kk] * ev[R.of + r] + h * w);
            }
        }
    } else {  // SC_TRAPEZE: node-level part; the step-level part needs the next node (finalize_trapeze_step)
        double* ev = rec + R.oEval;
        for (int r = 0; r < n; ++r)
            for (int kk = 0; kk < nv; ++kk) {
                double w = P::DYN_V ? ev[R.oW + r * nv + kk] : 0.0;
                if (P::DYN_T && FREE) w = w + ev[R.oft + r] * dti[kk];
                ev[R.oW + r * nv + kk] = w;
            }
        if (is_step) C[C_NHH] = -(0.5 * (tip1 - ti));
    }
    (void)m;
}

// trapeze: X_{i+1} - (X_i + h/2 (f_i + f_{i+1}))  (trapeze.jl:128-140); needs the record of node i+1
template <class P>
CTD_HD void finalize_trapeze_step(const KParams& kp, const BlockCtx& cx, int k) {
    constexpr int n = P::NX, nv = P::NV;
    constexpr bool FREE = (P::IT0 >= 0) || (P::ITF >= 0);
    const Layout& L = kp.L;
    const RecLayout& R = kp.R;
    const int64_t i = slot_index(kp, cx, k);
    if (i < 0 || i >= L.N || k + 1 >= cx.nslots) return;
    if (slot_index(kp, cx, k + 1) != i + 1) return;
    double* rec = cx.rec + (int64_t)k * R.stride;
    const double* nxt = cx.rec + (int64_t)(k + 1) * R.stride;
    const double* base = cx.in + (int64_t)k * cx.in_stride;
    const double ti = time_at<P>(kp, cx.v, i), tip1 = time_at<P>(kp, cx.v, i + 1);
    const double half_h = 0.5 * (tip1 - ti);
    const double* e0 = rec + R.oEval;
    const double* e1 = nxt + R.oEval;
    for (int r = 0; r < n; ++r) {
        const double fs = e0[R.of + r] + e1[R.of + r];
        rec[R.oR + r] = base[L.blk + r] - (base[r] + half_h * fs);
        for (int kk = 0; kk < nv; ++kk) {
            const double dhalf = FREE ? 0.5 * (dtime_at<P>(kp, i + 1, kk) - dtime_at<P>(kp, i, kk)) : 0.0;
            rec[R.oSv + r * nv + kk] = -(dhalf * fs + half_h * (e0[R.oW + r * nv + kk] + e1[R.oW + r * nv + kk]));
        }
    }
}

template <class P, int SC>
CTD_HD void phase_fin(const KParams& kp, const BlockCtx& cx, int tid, int nthr) {
    constexpr int nv = P::NV, np = P::NPATH;
    constexpr bool FREE = (P::IT0 >= 0) || (P::ITF >= 0);
    const RecLayout& R = kp.R;
    for (int k = tid; k < cx.nslots; k += nthr) finalize_step<P, SC>(kp, cx, k);
    if (cx.is_edge) {
        // final-path record and boundary record: coefficients, and total d/dv of the final path rows
        for (int e = tid; e < 2; e += nthr) {
            double* rec = cx.rec + (int64_t)(e == 0 ? kp.edge_fp : kp.edge_b) * R.stride;
            fill_const_coefs<P>(kp, rec + R.oC);
            if (e == 0 && np > 0) {
                for (int r = 0; r < np; ++r)
                    for (int kk = 0; kk < nv; ++kk) {
                        double pv = P::PATH_V ? rec[R.oPv + r * nv + kk] : 0.0;
                        if (P::PATH_T && FREE) pv = pv + rec[R.oPt + r] * dtime_at<P>(kp, kp.L.N, kk);
                        rec[R.oPv + r * nv + kk] = pv;
                    }
            }
        }
    }
}

template <class P, int SC>
CTD_HD void phase_fin2(const KParams& kp, const BlockCtx& cx, int tid, int nthr) {
    if (SC == SC_TRAPEZE && !cx.is_edge)
        for (int k = tid; k < cx.nslots; k += nthr) finalize_trapeze_step<P>(kp, cx, k);
}

// ------------------------------------------------------------------------------------------------------
// phase: emit
// ------------------------------------------------------------------------------------------------------
CTD_HD double eval_code(const KParams& kp, const double* rec_c, const double* rec_d, uint32_t code) {
    const double coef = rec_c[kp.R.oC + code_ci(code)];
    const double data = rec_d[code_di(code)];
    const int bt = code_beta(code);
    const double beta = bt == 0 ? 0.0 : (bt == 1 ? 1.0 : -1.0);
    return coef * data + beta;
}

template <class P, int SC>
CTD_HD void phase_emit(const KParams& kp, const BlockCtx& cx, int tid, int nthr) {
    const Layout& L = kp.L;
    const RecLayout& R = kp.R;
    if (cx.is_edge) {
        for (int e = kp.edge_begin + tid; e < kp.edge_end; e += nthr) {
            const uint32_t code = kp.edge_code[e];
            const int64_t idx = kp.edge_idx[e];
            const double val = eval_code(kp, cx.rec + (int64_t)code_crec(code) * R.stride,
                                         cx.rec + (int64_t)code_drec(code) * R.stride, code);
            if (idx & kEdgeCBit) { if (kp.c) kp.c[idx & ~kEdgeCBit] = val; }
            else if (kp.vals) kp.vals[idx] = val;
        }
        return;
    }
    const int nsteps = (int)(cx.b - cx.a);
    const int slot0 = (int)(cx.a - cx.lo);
    // (A) constraint rows of the tile: c[a*cb .. b*cb)
    if (kp.c) {
        const int total = nsteps * L.cb;
        double* out = kp.c + cx.a * (int64_t)L.cb;
        int s = tid / L.cb, r = tid - s * L.cb;
        const int ds = nthr / L.cb, dr = nthr - ds * L.cb;
        for (int e = tid; e < total; e += nthr) {
            out[e] = cx.rec[(int64_t)(slot0 + s) * R.stride + R.oR + r];
            s += ds; r += dr;
            if (r >= L.cb) { r -= L.cb; s += 1; }
        }
    }
    if (!kp.vals) return;
    // (B) step-periodic CSC segments of the regular steps of the tile
    {
        const int64_t ra = cx.a > kp.reg_first ? cx.a : kp.reg_first;
        const int64_t rb = cx.b < kp.reg_last ? cx.b : kp.reg_last;
        if (rb > ra) {
            const int Ls = kp.Lseg;
            const int total = (int)(rb - ra) * Ls;
            double* out = kp.vals + kp.seg_base + (ra - kp.reg_first) * (int64_t)Ls;
            const int sl0 = (int)(ra - cx.lo);
            int s = tid / Ls, k = tid - s * Ls;
            const int ds = nthr / Ls, dk = nthr - ds * Ls;
            for (int e = tid; e < total; e += nthr) {
                const uint32_t code = kp.tmpl[k];
                const double* rc = cx.rec + (int64_t)(sl0 + s - code_crec(code)) * R.stride;
                const double* rd = cx.rec + (int64_t)(sl0 + s - code_drec(code)) * R.stride;
                out[e] = eval_code(kp, rc, rd, code);
                s += ds; k += dk;
                if (k >= Ls) { k -= Ls; s += 1; }
            }
        }
    }
    // (C) the tile's slice of every V column
    if (kp.vr > 0) {
        const int vr = kp.vr;
        const int total = nsteps * vr;
        for (int kk = 0; kk < P::NV; ++kk) {
            double* out = kp.vals + kp.vcol_base[kk] + cx.a * (int64_t)vr;
            const uint32_t* codes = kp.vtmpl + kk * vr;
            int s = tid / vr, k = tid - s * vr;
            const int ds = nthr / vr, dk = nthr - ds * vr;
            for (int e = tid; e < total; e += nthr) {
                const double* rr = cx.rec + (int64_t)(slot0 + s) * R.stride;
                out[e] = eval_code(kp, rr, rr, codes[k]);
                s += ds; k += dk;
                if (k >= vr) { k -= vr; s += 1; }
            }
        }
    }
}

}  // namespace ctd
