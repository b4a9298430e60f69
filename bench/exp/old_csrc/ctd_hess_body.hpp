// ctd_hess_body.hpp -- Hessian of the Lagrangian, written as phase functions (see ctd_hess.hpp for the decomposition).
//
// One workgroup handles a TILE of consecutive time steps:
//   load   the tile's slice of xu, the multipliers y of its constraint rows and the normalized times go to LDS
//   eval   one lane per (step, evaluation point, outer direction p, chunk of inner directions): the OCP functions are
//          pushed through second-order forward numbers (ctd::Dual2) and the lane stores its K entries of the point's
//          dense Hessian HD (row p) into the step's LDS record; one lane per step fills the chain-rule coefficients
//   emit   lane e owns entry e of the step-periodic CSC segment and sums its terms  C[c1] C[c2] rec[di]  for every
//          step of the tile (coalesced 8-byte stores); a few lanes add up the tile's share of the V x V block
// Workgroup 0 is the EDGE block: first / last step columns, final-state columns, final-time path point and the
// boundary + Mayer point (explicit entry list).  hess_finish sums the V x V partials in a fixed order.
//
// Reference semantics: hess_coord!(nlp, x, y, vals; obj_weight) of ADNLPModels over the closures built at
// src/collocation.jl:137-149 -- objective src/DOCP_functions.jl:23-54 (+ integral(): trapeze.jl:78-110, midpoint.jl:79-97,
// irk.jl:179-228, irk_stagewise.jl:344-384), constraints src/DOCP_functions.jl:80-115 (+ stepStateConstraints!:
// trapeze.jl:118-142, midpoint.jl:124-140, irk.jl:236-308, irk_stagewise.jl:394-460).
#pragma once
#include "ctd_hess.hpp"
#include "ctd_kernel_body.hpp"

namespace ctd {

// inner directions per second-order eval lane (-DCTD_HESSK_OVERRIDE=k for tuning experiments)
#ifdef CTD_HESSK_OVERRIDE
template <class P> struct HessK { static constexpr int value = CTD_HESSK_OVERRIDE; };
#else
// Four for small OCPs; one for state dimension >= 8, where every further inner direction costs 2 doubles for each of the
// ~3 n + m second-order numbers a lane holds: with two, the 12-state quadrotor needed 356 registers per lane (one wave per
// SIMD) and ran 1.6x slower than with one (216 registers, two waves per SIMD); the 8-state one gains 4 %.
template <class P> struct HessK { static constexpr int value = (P::NX >= 8) ? 1 : 4; };
#endif

template <class P, int SC, int S> struct HRL {
    static constexpr HessRecLayout R =
        make_hess_layout(P::NX, P::NU, P::NV, P::NPATH, SC, (SC == SC_IRK || (SC == SC_MIDPOINT && S > 1)) ? S : 0,
                         (P::IT0 >= 0) || (P::ITF >= 0));      // (midpoint, S > 1: controls per step, one stage-type point each)
};

struct HBlockCtx {
    int is_edge, nslots, in_stride;
    int edge_part;  // edge blocks: which share of the edge entries
    int64_t a, b, lo;
    double* in;     // staged xu slice
    double* ly;     // multipliers: tile: (nslots + 1) * cb, block k + 1 = rows of step lo + k (block 0: step lo - 1);
                    // edge: per slot [previous step | own step], then the final-path rows and the boundary rows
    double* v;
    double* tau;    // tile: tau[e] = tau_{lo - 1 + e}; edge: 3 per slot (previous, own, next), then tau_N
    double* rec;
    double* cp;     // coefficient products: npairs per slot (hess_pair), slot k at cp + k * npairs
    double* red;    // tile: nvv * T per-step V x V contributions (summed in step order by hess_phase_vvsum)
    // term / task tables: LDS copies when small (staged by hess_phase_load), else the global tables
    const uint32_t *tptr, *terms, *cpos, *zpos, *vptr, *vterms;
    const uint32_t *tasks, *ptasks;
    // LDS copies of the coefficient-pair tables (factor kinds from the kernel arguments, where a lane-dependent index
    // would cost a global load per use; constants from hp.pair_c)
    const uint32_t* pairs;
    const double* pc;
};

// doubles at the head of every workgroup's LDS: pair factor kinds (kMaxPairs words) | pair constants (kMaxPairs)
constexpr int kHessCoefDoubles = kMaxPairs / 2 + kMaxPairs;

// words (uint32) of table data a tile stages in LDS: tptr | terms | cpos | zpos | vptr | vterms | tasks | ptasks
constexpr int kMaxStagedHessWords = 3072;
CTD_HD int hess_table_words(const HParams& hp) {
    return (hp.nc + 1) + hp.nterms + (hp.compact ? hp.nc : 0) + hp.nz + (hp.nvv + 1) + hp.nvterms + hp.ntask + hp.nptask;
}
CTD_HD bool hess_tables_staged(const HParams& hp) { return hess_table_words(hp) <= kMaxStagedHessWords; }
CTD_HD int hess_table_doubles(const HParams& hp) { return hess_tables_staged(hp) ? (hess_table_words(hp) + 1) / 2 : 0; }

CTD_HD int64_t hslot_step(const HParams& hp, const HBlockCtx& cx, int k) { return cx.is_edge ? hp.edge_steps[k] : cx.lo + k; }
CTD_HD const double* hslot_y(const HParams& hp, const HBlockCtx& cx, int k) { return cx.ly + (cx.is_edge ? 2 * k + 1 : k + 1) * hp.L.cb; }
CTD_HD const double* hslot_yprev(const HParams& hp, const HBlockCtx& cx, int k) { return cx.ly + (cx.is_edge ? 2 * k : k) * hp.L.cb; }
CTD_HD double hslot_tau(const HBlockCtx& cx, int k, int d) { return cx.is_edge ? cx.tau[3 * k + 1 + d] : cx.tau[k + 1 + d]; }

CTD_HD HBlockCtx make_hctx(const HParams& hp, int block, double* lds) {
    HBlockCtx cx;
    const Layout& L = hp.L;
    cx.tptr = hp.tptr; cx.terms = hp.terms; cx.cpos = hp.cpos; cx.zpos = hp.zpos; cx.vptr = hp.vptr; cx.vterms = hp.vterms;
    cx.tasks = hp.tasks; cx.ptasks = hp.ptasks;
    cx.pairs = reinterpret_cast<const uint32_t*>(lds);
    cx.pc = lds + kMaxPairs / 2;
    lds += kHessCoefDoubles;
    cx.edge_part = block;
    if (block < hp.n_edge_blocks) {
        cx.is_edge = 1;
        cx.nslots = hp.n_edge_slots;
        cx.in_stride = edge_in_stride(L);
        cx.a = cx.b = cx.lo = 0;
        cx.in = lds;
        cx.ly = cx.in + cx.nslots * cx.in_stride;
        cx.v = cx.ly + 2 * cx.nslots * L.cb + L.p + L.bc;
        cx.tau = cx.v + kMaxNV;
        cx.rec = cx.tau + 3 * kMaxHessEdgeSlots + 1;
        cx.cp = cx.rec + (cx.nslots + 2) * hp.R.stride;
        cx.red = cx.rec;
    } else {
        const int tile = hp.xcd_remap ? xcd_tile(block - hp.n_edge_blocks, hp.ntiles) : block - hp.n_edge_blocks;
        const int cap = hp.T + hp.HL + hp.HH;
        if (hess_tables_staged(hp)) {
            const uint32_t* w = reinterpret_cast<const uint32_t*>(lds);
            cx.tptr = w; w += hp.nc + 1;
            cx.terms = w; w += hp.nterms;
            if (hp.compact) { cx.cpos = w; w += hp.nc; }
            cx.zpos = w; w += hp.nz;
            cx.vptr = w; w += hp.nvv + 1;
            cx.vterms = w; w += hp.nvterms;
            cx.tasks = w;
            cx.ptasks = cx.tasks + hp.ntask;
            lds += hess_table_doubles(hp);
        }
        cx.is_edge = 0;
        cx.a = hp.step_begin + (int64_t)tile * hp.T;                   // (trapeze tiles walk nodes 0..N-1, node N is edge)
        cx.b = cx.a + hp.T < hp.step_end ? cx.a + hp.T : hp.step_end;
        cx.lo = cx.a - hp.HL;
        cx.nslots = (int)(cx.b - cx.a) + hp.HL + hp.HH;
        cx.in_stride = L.blk;
        cx.in = lds;
        cx.ly = cx.in + (cap + 1) * L.blk + L.n + L.m;
        cx.v = cx.ly + (cap + 1) * L.cb;
        cx.tau = cx.v + kMaxNV;
        cx.rec = cx.tau + cap + 3;
        cx.cp = cx.rec + cap * hp.R.stride;
        cx.red = cx.cp + cap * hp.npairs;
    }
    return cx;
}

// LDS (doubles) of an edge workgroup / of any workgroup of the tile kernel
inline int64_t hess_edge_lds_doubles(const HParams& hp) {
    const Layout& L = hp.L;
    return kHessCoefDoubles + (int64_t)hp.n_edge_slots * edge_in_stride(L) + 2 * hp.n_edge_slots * L.cb + L.p + L.bc + kMaxNV +
           3 * kMaxHessEdgeSlots + 1 + (int64_t)(hp.n_edge_slots + 2) * (hp.R.stride + hp.npairs);
}
inline int64_t hess_lds_doubles(const HParams& hp) {
    const Layout& L = hp.L;
    const int64_t cap = hp.T + hp.HL + hp.HH;
    const int64_t tile = kHessCoefDoubles + hess_table_doubles(hp) + (cap + 1) * L.blk + L.n + L.m + (cap + 1) * L.cb + kMaxNV + cap + 3 +
                         cap * (hp.R.stride + hp.npairs) + (int64_t)hp.nvv * hp.T;
    const int64_t edge = hess_edge_lds_doubles(hp);
    return tile > edge ? tile : edge;
}

CTD_HD double htau_global(const HParams& hp, int64_t i) {
    if (i < 0) i = 0;
    if (i > hp.L.N) i = hp.L.N;
    return hp.tau ? hp.tau[i] : (double)i / (double)hp.L.N;
}
template <class P> CTD_HD double htime_of(const HParams& hp, const double* v, double tau) {
    const double t0 = (P::IT0 >= 0) ? v[P::IT0 >= 0 ? P::IT0 : 0] : hp.L.t0;
    const double tf = (P::ITF >= 0) ? v[P::ITF >= 0 ? P::ITF : 0] : hp.L.tf;
    return t0 + tau * (tf - t0);
}

// multiplier of local row r of step s (rows of node N: only the final-time path rows exist)
CTD_HD double hess_y_of(const HParams& hp, const double* __restrict__ y, int64_t s, int r) {
    const Layout& L = hp.L;
    if (s >= 0 && s < L.N) return y[s * L.cb + r];
    if (s == L.N && r >= L.eqs) return y[L.N * L.cb + (r - L.eqs)];
    return 0.0;
}

// ------------------------------------------------------------------------------------------------------
// phase: load
// ------------------------------------------------------------------------------------------------------
// cf / pk: entry `tid` of the two pair tables, loaded by the caller ahead of its other copies (workgroups narrower than the
// tables, which only the emulator uses, fetch the rest here)
CTD_HD void hess_stage_coefs(const HParams& hp, const HBlockCtx& cx, double cf, uint32_t pk, int tid, int nthr) {
    if (tid < kMaxPairs) {
        const_cast<double*>(cx.pc)[tid] = cf;
        const_cast<uint32_t*>(cx.pairs)[tid] = pk;
    }
    for (int e = tid + nthr; e < hp.npairs; e += nthr) {
        const_cast<double*>(cx.pc)[e] = hp.pair_c[e];
        const_cast<uint32_t*>(cx.pairs)[e] = hp.pairs[e];
    }
}

template <class P>
CTD_HD void hess_phase_load(const HParams& hp, const HBlockCtx& cx, const double* __restrict__ xu,
                            const double* __restrict__ y, int tid, int nthr) {
    const Layout& L = hp.L;
    // pair tables (issued first: their latency overlaps the copies below)
    const double cf = tid < hp.npairs ? hp.pair_c[tid] : 0.0;
    const uint32_t pc = tid < hp.npairs ? hp.pairs[tid] : 0u;
    if (cx.is_edge) {
        hess_stage_coefs(hp, cx, cf, pc, tid, nthr);
        const int per = cx.in_stride;
        for (int e = tid; e < cx.nslots * per; e += nthr) {
            const int k = e / per, o = e - k * per;
            int64_t g = hp.edge_steps[k] * L.blk + o;
            if (o >= L.blk + L.n + L.m)        // control of the previous step (own step for step 0): implicit Euler's path control
                g = (hp.edge_steps[k] >= 1 ? hp.edge_steps[k] - 1 : 0) * (int64_t)L.blk + L.n + (o - (L.blk + L.n + L.m));
            cx.in[e] = (g < L.v_off) ? (hp.halo ? xnear(hp.near, xu, g) : xu)[g] : 0.0;
        }
        for (int e = tid; e < 2 * cx.nslots * L.cb; e += nthr) {
            const int blk2 = e / L.cb, r = e - blk2 * L.cb;
            cx.ly[e] = hess_y_of(hp, y, hp.edge_steps[blk2 >> 1] - 1 + (blk2 & 1), r);
        }
        for (int e = tid; e < L.p + L.bc; e += nthr) cx.ly[2 * cx.nslots * L.cb + e] = y[L.N * L.cb + e];
        for (int e = tid; e <= 3 * cx.nslots; e += nthr)
            cx.tau[e] = (e == 3 * cx.nslots) ? htau_global(hp, L.N) : htau_global(hp, hp.edge_steps[e / 3] - 1 + (e % 3));
    } else {
        const int64_t g0 = (cx.lo < 0 ? 0 : cx.lo) * (int64_t)L.blk;
        int64_t g1 = (cx.lo + cx.nslots) * (int64_t)L.blk + L.n + L.m;
        if (g1 > L.v_off) g1 = L.v_off;
        const double* __restrict__ src = xu + g0;
        double* dst = cx.in + (int)(g0 - cx.lo * (int64_t)L.blk);
        const int cnt = (int)(g1 - g0);
        // rows of steps lo-1 .. lo+nslots-1 are contiguous in y; tiles never hold node N (its rows sit after N * cb)
        const int ny = (cx.nslots + 1) * L.cb;
        const int64_t yb = (cx.lo - 1) * (int64_t)L.cb, ylim = L.N * (int64_t)L.cb;
        auto yval = [&](int e) -> double { const int64_t g = yb + e; return (g >= 0 && g < ylim) ? y[g] : 0.0; };
        // A lane first ISSUES its global loads of every stream (xu slice, multipliers, variable, term / task tables), then
        // stores them to LDS: one exposed memory latency instead of one per copy loop
        const bool st = hess_tables_staged(hp);
        const double x0 = tid < cnt ? src[tid] : 0.0, x1 = tid + nthr < cnt ? src[tid + nthr] : 0.0;
        const double y0 = tid < ny ? yval(tid) : 0.0, y1 = tid + nthr < ny ? yval(tid + nthr) : 0.0;
        const double vv = tid < P::NV ? xu[L.v_off + tid] : 0.0;
        const uint32_t w0 = (st && tid <= hp.nc) ? hp.tptr[tid] : 0u, w1 = (st && tid < hp.nterms) ? hp.terms[tid] : 0u;
        const uint32_t w6 = (st && hp.compact && tid < hp.nc) ? hp.cpos[tid] : 0u, w7 = (st && tid < hp.nz) ? hp.zpos[tid] : 0u;
        const uint32_t w2 = (st && tid <= hp.nvv) ? hp.vptr[tid] : 0u, w3 = (st && tid < hp.nvterms) ? hp.vterms[tid] : 0u;
        const uint32_t w4 = (st && tid < hp.ntask) ? hp.tasks[tid] : 0u, w5 = (st && tid < hp.nptask) ? hp.ptasks[tid] : 0u;
        const double tau_e = tid <= cx.nslots + 2 ? htau_global(hp, cx.lo - 1 + tid) : 0.0;     // (table load: issued with the rest)
        if (tid < cnt) dst[tid] = x0;
        if (tid + nthr < cnt) dst[tid + nthr] = x1;
        if (tid < ny) cx.ly[tid] = y0;
        if (tid + nthr < ny) cx.ly[tid + nthr] = y1;
        if (tid < kMaxNV) cx.v[tid] = vv;
        hess_stage_coefs(hp, cx, cf, pc, tid, nthr);
        if (st) {
            uint32_t* d;
            d = const_cast<uint32_t*>(cx.tptr);   if (tid <= hp.nc) d[tid] = w0;     for (int e = tid + nthr; e <= hp.nc; e += nthr) d[e] = hp.tptr[e];
            d = const_cast<uint32_t*>(cx.terms);  if (tid < hp.nterms) d[tid] = w1;  for (int e = tid + nthr; e < hp.nterms; e += nthr) d[e] = hp.terms[e];
            if (hp.compact) { d = const_cast<uint32_t*>(cx.cpos); if (tid < hp.nc) d[tid] = w6; for (int e = tid + nthr; e < hp.nc; e += nthr) d[e] = hp.cpos[e]; }
            d = const_cast<uint32_t*>(cx.zpos);   if (tid < hp.nz) d[tid] = w7;     for (int e = tid + nthr; e < hp.nz; e += nthr) d[e] = hp.zpos[e];
            d = const_cast<uint32_t*>(cx.vptr);   if (tid <= hp.nvv) d[tid] = w2;    for (int e = tid + nthr; e <= hp.nvv; e += nthr) d[e] = hp.vptr[e];
            d = const_cast<uint32_t*>(cx.vterms); if (tid < hp.nvterms) d[tid] = w3; for (int e = tid + nthr; e < hp.nvterms; e += nthr) d[e] = hp.vterms[e];
            d = const_cast<uint32_t*>(cx.tasks);  if (tid < hp.ntask) d[tid] = w4;   for (int e = tid + nthr; e < hp.ntask; e += nthr) d[e] = hp.tasks[e];
            if (tid < hp.nptask) d[hp.ntask + tid] = w5;
            for (int e = tid + nthr; e < hp.nptask; e += nthr) d[hp.ntask + e] = hp.ptasks[e];
        }
        for (int e = tid + 2 * nthr; e < cnt; e += nthr) dst[e] = src[e];
        for (int e = tid + 2 * nthr; e < ny; e += nthr) cx.ly[e] = yval(e);
        if (tid <= cx.nslots + 2) cx.tau[tid] = tau_e;
        for (int e = tid + nthr; e <= cx.nslots + 2; e += nthr) cx.tau[e] = htau_global(hp, cx.lo - 1 + e);
        // sharded iterate: the entries of the slice other shards own (a boundary tile: a handful) once more, from the owners' buffers
        // (same lane, same LDS word: ordered behind the copy above)
        if (hp.halo && (g0 < hp.own_lo || g1 > hp.own_hi))
            for (int e = tid; e < cnt; e += nthr) {
                const int64_t g = g0 + e;
                if (g < hp.own_lo || g >= hp.own_hi) dst[e] = xnear(hp.near, xu, g)[g];
            }
        return;
    }
    if (tid < kMaxNV) cx.v[tid] = (tid < P::NV) ? xu[L.v_off + tid] : 0.0;
}

// ------------------------------------------------------------------------------------------------------
// phase: eval
// ------------------------------------------------------------------------------------------------------
// step-dependent factor of a chain-rule coefficient (HF_* in ctd_hess.hpp)
template <class P>
CTD_HD double hess_factor(int kind, double h, double tau0, double tau1) {
    double f = kind == HF_H ? h : 1.0;
    if constexpr (Dirs<P>::FREE) {
#pragma unroll
        for (int k = 0; k < P::NV; ++k)
            if (kind == HF_DH + k) f = dtime_of<P>(tau1, k) - dtime_of<P>(tau0, k);
    } else {
        if (kind >= HF_DH) f = 0.0;
    }
    return f;
}

// coefficient products of one record (slot k, pair id) + the state-equation multipliers the K x V terms need
template <class P, int SC, int S>
CTD_HD void hess_pair(const HParams& hp, const HBlockCtx& cx, int k, int pid) {
    constexpr int n = P::NX;
    constexpr HessRecLayout R = HRL<P, SC, S>::R;
    double* rec = cx.rec + k * R.stride;
    const double tau0 = hslot_tau(cx, k, 0), tau1 = hslot_tau(cx, k, 1);
    const double h = htime_of<P>(hp, cx.v, tau1) - htime_of<P>(hp, cx.v, tau0);
    const int kinds = (int)cx.pairs[pid];
    cx.cp[k * hp.npairs + pid] = cx.pc[pid] * hess_factor<P>(kinds & 0xFF, h, tau0, tau1) * hess_factor<P>(kinds >> 8, h, tau0, tau1);
    if (pid == 0) rec[R.oZero] = 0.0;
    if (SC == SC_IRK && Dirs<P>::FREE && pid == 0) {
        const double* y = hslot_y(hp, cx, k);
#pragma unroll
        for (int r = 0; r < n; ++r) rec[R.oYX + r] = y[r];
    }
}

// seeds of one direction d (0..md-1; anything else: no direction) -- compare-and-select, no indexed registers
template <int K> CTD_HD Dual2<K> hess_seed(double val, double sa, const double* sb) {
    Dual2<K> r; r.v = val; r.a = sa;
#pragma unroll
    for (int i = 0; i < K; ++i) { r.b[i] = sb[i]; r.ab[i] = 0.0; }
    return r;
}

// One (p, chunk) lane of a stage-type point: Gauss-Legendre stage j of step s, the midpoint of step s, or trapeze node s.
template <class P, int SC, int S>
CTD_HD void hess_eval_stage(const HParams& hp, const HBlockCtx& cx, int k, int j, uint32_t task) {
    constexpr int n = P::NX, m = P::NU, nv = P::NV, np = P::NPATH, K = HessK<P>::value;
    constexpr bool FREE = Dirs<P>::FREE;
    constexpr HessRecLayout R = HRL<P, SC, S>::R;
    constexpr int md = R.md, vd = n + m;
    using T = Dual2<K>;
    const Layout& L = hp.L;
    const int64_t s = hslot_step(hp, cx, k);
    if (s < 0 || (SC == SC_TRAPEZE ? s > L.N : s >= L.N)) return;
    const double* base = cx.in + k * cx.in_stride;
    const double* y = hslot_y(hp, cx, k);
    double* rec = cx.rec + k * R.stride;
    const double taum = hslot_tau(cx, k, -1), tau0 = hslot_tau(cx, k, 0), tau1 = hslot_tau(cx, k, 1);
    const double tA = htime_of<P>(hp, cx.v, tau0), tB = htime_of<P>(hp, cx.v, tau1);
    const double h = tB - tA;
    const int p = (int)(task & 31u);
    int q[K];
#pragma unroll
    for (int i = 0; i < K; ++i) q[i] = (int)((task >> (5 + 5 * i)) & 31u);      // 31: no direction (>= md)
    // d(tau-dependent quantity)/d(direction): only the V directions move the time grid
    auto dt_of = [&](int d, double tau) -> double {
        const int kx = d - vd;
        return (FREE && kx >= 0 && d < md) ? dtime_of<P>(tau, kx) : 0.0;
    };
    auto unit = [&](int d, int target) -> double { return d == target ? 1.0 : 0.0; };
    double sb[K];

    // step length as a second-order number (its V-derivatives are constants)
#pragma unroll
    for (int i = 0; i < K; ++i) sb[i] = dt_of(q[i], tau1) - dt_of(q[i], tau0);
    const double hda = dt_of(p, tau1) - dt_of(p, tau0);
    double hdb[K];
#pragma unroll
    for (int i = 0; i < K; ++i) hdb[i] = sb[i];
    // midpoint with S > 1 controls per step: point j integrates over h / S (midpoint.jl:106,134)
    constexpr double hs = (SC == SC_MIDPOINT && S > 1) ? 1.0 / (double)S : 1.0;
    if (SC == SC_MIDPOINT && S > 1) {
#pragma unroll
        for (int i = 0; i < K; ++i) sb[i] = hs * hdb[i];
    }
    const T hh = hess_seed<K>(hs * h, hs * hda, sb);

    // evaluation time
    T t;
    if (SC == SC_IRK) {
        const double cj = butcher_c<S>(L, j);
#pragma unroll
        for (int i = 0; i < K; ++i) sb[i] = dt_of(q[i], tau0) + cj * hdb[i];
        t = hess_seed<K>(tA + cj * h, dt_of(p, tau0) + cj * hda, sb);
    } else if (SC == SC_MIDPOINT && L.euler == 0) {
#pragma unroll
        for (int i = 0; i < K; ++i) sb[i] = 0.5 * (dt_of(q[i], tau0) + dt_of(q[i], tau1));
        t = hess_seed<K>(0.5 * (tA + tB), 0.5 * (dt_of(p, tau0) + dt_of(p, tau1)), sb);
    } else if (SC == SC_MIDPOINT && L.euler == 2) {       // implicit Euler: (t_{i+1}, X_{i+1})
#pragma unroll
        for (int i = 0; i < K; ++i) sb[i] = dt_of(q[i], tau1);
        t = hess_seed<K>(tB, dt_of(p, tau1), sb);
    } else {                                              // trapeze node, explicit Euler: (t_i, X_i)
#pragma unroll
        for (int i = 0; i < K; ++i) sb[i] = dt_of(q[i], tau0);
        t = hess_seed<K>(tA, dt_of(p, tau0), sb);
    }

    // state at the evaluation point
    T x[n > 0 ? n : 1];
#pragma unroll
    for (int r = 0; r < n; ++r) {
        if (SC == SC_IRK) {
            const double* Kv = base + n + L.cu;
            double kap = 0.0;
#pragma unroll
            for (int l = 0; l < S; ++l) kap = kap + butcher_a<S>(L, j, l) * Kv[l * n + r];
#pragma unroll
            for (int i = 0; i < K; ++i) sb[i] = unit(q[i], r) + hdb[i] * kap;
            x[r] = hess_seed<K>(base[r] + h * kap, unit(p, r) + hda * kap, sb);
        } else {
#pragma unroll
            for (int i = 0; i < K; ++i) sb[i] = unit(q[i], r);
            const double xv = (SC == SC_MIDPOINT && L.euler == 0) ? 0.5 * (base[r] + base[L.blk + r])
                                                                  : ((SC == SC_MIDPOINT && L.euler == 2) ? base[L.blk + r] : base[r]);
            x[r] = hess_seed<K>(xv, unit(p, r), sb);
        }
    }
    T u[m > 0 ? m : 1];
#pragma unroll
    for (int b = 0; b < m; ++b) {
        const double uv = ((SC == SC_IRK && L.stagewise) || (SC == SC_MIDPOINT && S > 1)) ? base[n + j * m + b] : base[n + b];
#pragma unroll
        for (int i = 0; i < K; ++i) sb[i] = unit(q[i], n + b);
        u[b] = hess_seed<K>(uv, unit(p, n + b), sb);
    }
    T v[nv > 0 ? nv : 1];
#pragma unroll
    for (int kk = 0; kk < nv; ++kk) {
#pragma unroll
        for (int i = 0; i < K; ++i) sb[i] = unit(q[i], vd + kk);
        v[kk] = hess_seed<K>(cx.v[kk], unit(p, vd + kk), sb);
    }

    T f[n > 0 ? n : 1];
    P::template dynamics<T>(f, t, x, u, v);
    T phi(0.0);
    if (SC == SC_IRK) {
#pragma unroll
        for (int r = 0; r < n; ++r) phi = phi + f[r] * (-y[n + j * n + r]);
        if (P::HAS_LAGRANGE) phi = phi + (hh * P::template lagrange<T>(t, x, u, v)) * (hp.obj_weight * butcher_b<S>(L, j));
    } else if (SC == SC_MIDPOINT) {
        T inner(0.0);
#pragma unroll
        for (int r = 0; r < n; ++r) inner = inner + f[r] * (-y[r]);
        if (P::HAS_LAGRANGE) {
            if (SC == SC_MIDPOINT && S > 1) {
                // the quadrature point of control j sits at t_i + (j - 1/2) h / S (midpoint.jl:110); the dynamics keep the step's midpoint (:57)
                const double wj = ((double)j + 0.5) * hs;
#pragma unroll
                for (int i = 0; i < K; ++i) sb[i] = dt_of(q[i], tau0) + wj * hdb[i];
                const T tl = hess_seed<K>(tA + wj * h, dt_of(p, tau0) + wj * hda, sb);
                inner = inner + P::template lagrange<T>(tl, x, u, v) * hp.obj_weight;
            } else {
                inner = inner + P::template lagrange<T>(t, x, u, v) * hp.obj_weight;
            }
        }
        phi = hh * inner;
    } else {
        // node s is shared by step s-1 (length hm) and step s (length hh); the clamped tau makes the missing one 0
        const double tM = htime_of<P>(hp, cx.v, taum);
#pragma unroll
        for (int i = 0; i < K; ++i) sb[i] = dt_of(q[i], tau0) - dt_of(q[i], taum);
        const T hm = hess_seed<K>(tA - tM, dt_of(p, tau0) - dt_of(p, taum), sb);
        const double* yp = hslot_yprev(hp, cx, k);
#pragma unroll
        for (int r = 0; r < n; ++r) phi = phi + f[r] * ((hm * yp[r] + hh * y[r]) * (-0.5));
        if (P::HAS_LAGRANGE) phi = phi + P::template lagrange<T>(t, x, u, v) * ((hm + hh) * (0.5 * hp.obj_weight));
        if (np > 0) {
            T g[np > 0 ? np : 1];
            P::template path<T>(g, t, x, u, v);
#pragma unroll
            for (int r = 0; r < np; ++r) phi = phi + g[r] * y[L.eqs + r];
        }
    }
    double* HD = rec + R.oStage + j * R.stage_sz;
#pragma unroll
    for (int i = 0; i < K; ++i)
        if (q[i] < md) HD[hess_tri(md, p, q[i])] = phi.ab[i];
    if (SC == SC_IRK && FREE) {
        // RK[k][a] = h d2Phi/dx_a dV_k + dh/dv_k dPhi/dx_a, from whichever of the two directions is the outer one
#pragma unroll
        for (int i = 0; i < K; ++i) {
            if (p < n && q[i] >= vd && q[i] < md) HD[R.oRK + (q[i] - vd) * n + p] = h * phi.ab[i] + hdb[i] * phi.a;
            if (p >= vd && q[i] < n) HD[R.oRK + (p - vd) * n + q[i]] = h * phi.ab[i] + hda * phi.b[i];
        }
    }
}

// Symbolically differentiated stage functions of an OCP: members of the generated functor for run-time OCPs (HAS_SYM,
// ctd_jit.cpp), explicit specialisations generated at build time for the registry problems (ctd_sym_registry.hpp)
template <class P> struct SymStage {
    static constexpr bool value = P::HAS_SYM;
    CTD_HD static void irk(const double* p, double* HD) { if constexpr (P::HAS_SYM) P::stage_sym_irk(p, HD); }
    CTD_HD static void mid(const double* p, double* HD) { if constexpr (P::HAS_SYM) P::stage_sym_mid(p, HD); }
    CTD_HD static void trap(const double* p, double* HD) { if constexpr (P::HAS_SYM) P::stage_sym_trap(p, HD); }
};

// Whether the stage-type points of (P, SC, S) run the symbolic functions.  Not with several controls per step AND a Lagrange
// cost that reads the time: that quadrature point has a time of its own (midpoint.jl:110) where the generated function has
// one evaluation time.
template <class P, int SC, int S> constexpr bool hess_uses_sym() {
    return SymStage<P>::value && !(SC == SC_MIDPOINT && S > 1 && P::HAS_LAGRANGE && P::LAG_T);
}

// The same stage-type point for an OCP that carries symbolically differentiated stage functions (run-time OCPs,
// ctd_sym.hpp / ctd_jit.cpp): ONE lane fills the parameters of the point (SymPrm, ctd_hess.hpp) and the generated
// straight-line code writes every second derivative of the record (and the RK block) -- no second-order number types.
template <class P, int SC, int S>
CTD_HD void hess_eval_stage_sym(const HParams& hp, const HBlockCtx& cx, int k, int j) {
    if constexpr (SymStage<P>::value) {
        constexpr int n = P::NX, m = P::NU, nv = P::NV;
        constexpr bool FREE = Dirs<P>::FREE;
        constexpr HessRecLayout R = HRL<P, SC, S>::R;
        constexpr SymPrm Q = sym_prm(n, m, nv, P::NPATH);
        const Layout& L = hp.L;
        const int64_t s = hslot_step(hp, cx, k);
        if (s < 0 || (SC == SC_TRAPEZE ? s > L.N : s >= L.N)) return;
        const double* base = cx.in + k * cx.in_stride;
        const double* y = hslot_y(hp, cx, k);
        double* HD = cx.rec + k * R.stride + R.oStage + j * R.stage_sz;
        const double tau0 = hslot_tau(cx, k, 0), tau1 = hslot_tau(cx, k, 1);
        const double tA = htime_of<P>(hp, cx.v, tau0), tB = htime_of<P>(hp, cx.v, tau1), h = tB - tA;
        double prm[Q.count];
        double d0[nv > 0 ? nv : 1], d1[nv > 0 ? nv : 1];
#pragma unroll
        for (int kk = 0; kk < nv; ++kk) {
            d0[kk] = FREE ? dtime_of<P>(tau0, kk) : 0.0;
            d1[kk] = FREE ? dtime_of<P>(tau1, kk) : 0.0;
            prm[Q.HD + kk] = d1[kk] - d0[kk];
            prm[Q.V0 + kk] = cx.v[kk];
        }
        prm[Q.H0] = h;
        if (SC == SC_IRK) {
            const double cj = butcher_c<S>(L, j);
            prm[Q.T0] = tA + cj * h;
#pragma unroll
            for (int kk = 0; kk < nv; ++kk) prm[Q.TD + kk] = d0[kk] + cj * (d1[kk] - d0[kk]);
            const double* Kv = base + n + L.cu;
#pragma unroll
            for (int r = 0; r < n; ++r) {
                double kap = 0.0;
#pragma unroll
                for (int l = 0; l < S; ++l) kap = kap + butcher_a<S>(L, j, l) * Kv[l * n + r];
                prm[Q.X0 + r] = base[r] + h * kap;
                prm[Q.KAP + r] = kap;
                prm[Q.W + r] = -y[n + j * n + r];
            }
#pragma unroll
            for (int b = 0; b < m; ++b) prm[Q.U0 + b] = L.stagewise ? base[n + j * m + b] : base[n + b];
            prm[Q.CL] = P::HAS_LAGRANGE ? hp.obj_weight * butcher_b<S>(L, j) : 0.0;
            SymStage<P>::irk(prm, HD);
        } else if (SC == SC_TRAPEZE) {
            // node s between step s-1 (length hm) and step s (length h); the clamped tau makes the missing one 0
            const double taum = hslot_tau(cx, k, -1);
            const double tM = htime_of<P>(hp, cx.v, taum);
            const double* yp = hslot_yprev(hp, cx, k);
            prm[Q.T0] = tA;
            prm[Q.HM0] = tA - tM;
#pragma unroll
            for (int kk = 0; kk < nv; ++kk) {
                prm[Q.TD + kk] = d0[kk];
                prm[Q.HMD + kk] = d0[kk] - (FREE ? dtime_of<P>(taum, kk) : 0.0);
            }
#pragma unroll
            for (int r = 0; r < n; ++r) {
                prm[Q.X0 + r] = base[r];
                prm[Q.KAP + r] = 0.0;
                prm[Q.W + r] = y[r];
                prm[Q.WP + r] = yp[r];
            }
#pragma unroll
            for (int b = 0; b < m; ++b) prm[Q.U0 + b] = base[n + b];
#pragma unroll
            for (int r = 0; r < P::NPATH; ++r) prm[Q.WG + r] = y[L.eqs + r];
            prm[Q.CL] = P::HAS_LAGRANGE ? 0.5 * hp.obj_weight : 0.0;
            SymStage<P>::trap(prm, HD);
        } else {
            const double wa = L.euler == 0 ? 0.5 : (L.euler == 1 ? 1.0 : 0.0), wb = 1.0 - wa;   // weights of (t_i, X_i) / (t_i+1, X_i+1)
            if (S > 1) {       // several controls per step: point j integrates over h / S with the control U^j (midpoint.jl:134,146-153)
                prm[Q.H0] = h / (double)S;
#pragma unroll
                for (int kk = 0; kk < nv; ++kk) prm[Q.HD + kk] = (d1[kk] - d0[kk]) / (double)S;
            }
            prm[Q.T0] = wa * tA + wb * tB;
#pragma unroll
            for (int kk = 0; kk < nv; ++kk) prm[Q.TD + kk] = wa * d0[kk] + wb * d1[kk];
#pragma unroll
            for (int r = 0; r < n; ++r) {
                prm[Q.X0 + r] = L.euler == 0 ? 0.5 * (base[r] + base[L.blk + r]) : (L.euler == 1 ? base[r] : base[L.blk + r]);
                prm[Q.KAP + r] = 0.0;
                prm[Q.W + r] = -y[r];
            }
#pragma unroll
            for (int b = 0; b < m; ++b) prm[Q.U0 + b] = S > 1 ? base[n + j * m + b] : base[n + b];
            prm[Q.CL] = P::HAS_LAGRANGE ? hp.obj_weight : 0.0;
            SymStage<P>::mid(prm, HD);
        }
    }
}

// path point: x = X_s (or X_N), u = control of the step (stagewise: sum_l b_l U^l), t = t_s; `yrow` = multipliers of the rows
template <class P, int SC, int S>
CTD_HD void hess_eval_path(const HParams& hp, const double* xs, const double* ub, const double* vv, double tau,
                           const double* yrow, double* HP, uint32_t task) {
    constexpr int n = P::NX, m = P::NU, nv = P::NV, np = P::NPATH, K = HessK<P>::value;
    constexpr bool FREE = Dirs<P>::FREE;
    constexpr HessRecLayout R = HRL<P, SC, S>::R;
    constexpr int md = R.md, vd = n + m;
    using T = Dual2<K>;
    const Layout& L = hp.L;
    const int p = (int)(task & 31u);
    int q[K];
#pragma unroll
    for (int i = 0; i < K; ++i) q[i] = (int)((task >> (5 + 5 * i)) & 31u);
    auto dt_of = [&](int d) -> double {
        const int kx = d - vd;
        return (FREE && kx >= 0 && d < md) ? dtime_of<P>(tau, kx) : 0.0;
    };
    auto unit = [&](int d, int target) -> double { return d == target ? 1.0 : 0.0; };
    double sb[K];
#pragma unroll
    for (int i = 0; i < K; ++i) sb[i] = dt_of(q[i]);
    const T t = hess_seed<K>(htime_of<P>(hp, vv, tau), dt_of(p), sb);
    T x[n > 0 ? n : 1];
#pragma unroll
    for (int r = 0; r < n; ++r) {
#pragma unroll
        for (int i = 0; i < K; ++i) sb[i] = unit(q[i], r);
        x[r] = hess_seed<K>(xs[r], unit(p, r), sb);
    }
    T u[m > 0 ? m : 1];
#pragma unroll
    for (int b = 0; b < m; ++b) {
        double uv;
        if (SC == SC_IRK && L.stagewise) {
            uv = L.b[0] * ub[b];
#pragma unroll
            for (int l = 1; l < S; ++l) uv = uv + L.b[l] * ub[l * m + b];
        } else {
            uv = ub[b];
        }
#pragma unroll
        for (int i = 0; i < K; ++i) sb[i] = unit(q[i], n + b);
        u[b] = hess_seed<K>(uv, unit(p, n + b), sb);
    }
    T v[nv > 0 ? nv : 1];
#pragma unroll
    for (int kk = 0; kk < nv; ++kk) {
#pragma unroll
        for (int i = 0; i < K; ++i) sb[i] = unit(q[i], vd + kk);
        v[kk] = hess_seed<K>(vv[kk], unit(p, vd + kk), sb);
    }
    T g[np > 0 ? np : 1];
    P::template path<T>(g, t, x, u, v);
    T phi(0.0);
#pragma unroll
    for (int r = 0; r < np; ++r) phi = phi + g[r] * yrow[r];
#pragma unroll
    for (int i = 0; i < K; ++i)
        if (q[i] < md) HP[hess_tri(md, p, q[i])] = phi.ab[i];
}

// boundary + Mayer point: directions x0 | xf | v
template <class P, int SC, int S>
CTD_HD void hess_eval_boundary(const HParams& hp, const double* x0p, const double* xfp, const double* vv,
                               const double* yrow, double* HB, uint32_t task) {
    constexpr int n = P::NX, nv = P::NV, nb = P::NBC, K = HessK<P>::value;
    constexpr HessRecLayout R = HRL<P, SC, S>::R;
    constexpr int mdb = R.mdb;
    using T = Dual2<K>;
    const int p = (int)(task & 31u);
    int q[K];
#pragma unroll
    for (int i = 0; i < K; ++i) q[i] = (int)((task >> (5 + 5 * i)) & 31u);
    auto unit = [&](int d, int target) -> double { return d == target ? 1.0 : 0.0; };
    double sb[K];
    T x0[n > 0 ? n : 1], xf[n > 0 ? n : 1], v[nv > 0 ? nv : 1];
#pragma unroll
    for (int r = 0; r < n; ++r) {
#pragma unroll
        for (int i = 0; i < K; ++i) sb[i] = unit(q[i], r);
        x0[r] = hess_seed<K>(x0p[r], unit(p, r), sb);
#pragma unroll
        for (int i = 0; i < K; ++i) sb[i] = unit(q[i], n + r);
        xf[r] = hess_seed<K>(xfp[r], unit(p, n + r), sb);
    }
#pragma unroll
    for (int kk = 0; kk < nv; ++kk) {
#pragma unroll
        for (int i = 0; i < K; ++i) sb[i] = unit(q[i], 2 * n + kk);
        v[kk] = hess_seed<K>(vv[kk], unit(p, 2 * n + kk), sb);
    }
    T phi(0.0);
    if (nb > 0) {
        T r_[nb > 0 ? nb : 1];
        P::template boundary<T>(r_, x0, xf, v);
#pragma unroll
        for (int r = 0; r < nb; ++r) phi = phi + r_[r] * yrow[r];
    }
    if (P::HAS_MAYER) phi = phi + P::template mayer<T>(x0, xf, v) * hp.obj_weight;
#pragma unroll
    for (int i = 0; i < K; ++i)
        if (q[i] < mdb) HB[hess_tri(mdb, p, q[i])] = phi.ab[i];
}

// Compact segments (HParams::compact): the tile's part of vals is zero-filled before the emit phase stores the entries that
// have terms -- by the waves the point evaluations leave idle, at the start of the eval phase: the stores have long been
// acknowledged when the barrier after that phase (which waits for them) is reached, so the entries written again in the
// emit phase are ordered after them.
CTD_HD void hess_zero_fill(const HParams& hp, const HBlockCtx& cx, int tid, int nthr) {
    if (hp.compact != 1 || cx.is_edge) return;
    const int64_t i0 = cx.a > hp.reg_first ? cx.a : hp.reg_first;
    const int64_t i1 = cx.b < hp.reg_last ? cx.b : hp.reg_last;
    if (i1 <= i0) return;
    double* out0 = hp.vals + hp.seg_base + (i0 - hp.reg_first) * (int64_t)hp.Lseg;
    const int n = (int)(i1 - i0) * hp.Lseg, z0 = nthr > 64 ? 64 : 0;
    if (tid >= z0)
        for (int j = tid - z0; j < n; j += nthr - z0) emit_store(&out0[j], 0.0, hp.wt_store);
}

template <class P, int SC, int S>
CTD_HD void hess_phase_eval(const HParams& hp, const HBlockCtx& cx, int tid, int nthr) {
    constexpr int n = P::NX, np = P::NPATH;
    constexpr HessRecLayout R = HRL<P, SC, S>::R;
    constexpr bool PATH_PT = np > 0 && SC != SC_TRAPEZE;
    constexpr int PT = R.S + (PATH_PT ? 1 : 0);
    const Layout& L = hp.L;
    hess_zero_fill(hp, cx, tid, nthr);
    // (the coefficient products start at the LAST lane: the waves the point evaluations below leave idle take them)
    for (int w = nthr - 1 - tid; w < cx.nslots * hp.npairs; w += nthr) {
        const int k = (int)fast_div((uint32_t)w, hp.div_npairs);
        hess_pair<P, SC, S>(hp, cx, k, w - k * hp.npairs);
    }
    // stage-type points: item (slot k, stage j, task) on lanes 0, 1, ...; path points: item (slot k, task) from the next wave
    // boundary on (wrapping around): the two kinds run different code, lanes of one wave would take turns
    const int nstage = cx.nslots * R.S * hp.ntask;
    for (int w = tid; w < nstage; w += nthr) {
        const int k = (int)fast_div((uint32_t)w, hp.div_stage_tasks);
        const int r = w - k * R.S * hp.ntask;
        const int j = (int)fast_div((uint32_t)r, hp.div_ntask);
        if constexpr (hess_uses_sym<P, SC, S>()) hess_eval_stage_sym<P, SC, S>(hp, cx, k, j);
        else hess_eval_stage<P, SC, S>(hp, cx, k, j, cx.tasks[r - j * hp.ntask]);
    }
    if (PATH_PT) {
        const int npath = cx.nslots * hp.nptask;
        const int pofs = ((nstage + 63) & ~63) % nthr;
        for (int w = tid >= pofs ? tid - pofs : tid - pofs + nthr; w < npath; w += nthr) {
            const int k = (int)fast_div((uint32_t)w, hp.div_nptask);
            const uint32_t code = cx.ptasks[w - k * hp.nptask];
            const int64_t s = hslot_step(hp, cx, k);
            if (s >= 0 && s < L.N) {
                const double* base = cx.in + k * cx.in_stride;
                const double* ub = base + n;
                if (L.euler == 2 && s >= 1)     // implicit Euler: u(t_i) = U_{i-1} (euler.jl:59-72)
                    ub = cx.is_edge ? base + L.blk + n + P::NU : (k >= 1 ? base - L.blk + n : base + n);
                hess_eval_path<P, SC, S>(hp, base, ub, cx.v, hslot_tau(cx, k, 0), hslot_y(hp, cx, k) + L.eqs,
                                         cx.rec + k * R.stride + R.oHP, code);
            }
        }
    }
    if (cx.is_edge) {
        // final-time path point (DOCP_functions.jl:100): X_N with the controls of step N-1; boundary + Mayer point
        int kl = 0, kf = 0;
        for (int k = 0; k < cx.nslots; ++k) {
            if (hp.edge_steps[k] == L.N - 1) kl = k;
            if (hp.edge_steps[k] == 0) kf = k;
        }
        const double* last = cx.in + kl * cx.in_stride;
        const double* yfp = cx.ly + 2 * cx.nslots * L.cb;
        if (PATH_PT)
            for (int w = tid; w < hp.nptask; w += nthr) {
                hess_eval_path<P, SC, S>(hp, last + L.blk, last + n, cx.v, cx.tau[3 * cx.nslots], yfp,
                                         cx.rec + hp.edge_fp * R.stride + R.oHP, hp.ptasks[w]);
            }
        if (P::NBC > 0 || P::HAS_MAYER)
            for (int w = tid; w < hp.nbtask; w += nthr) {
                hess_eval_boundary<P, SC, S>(hp, cx.in + kf * cx.in_stride, last + L.blk, cx.v, yfp + L.p,
                                             cx.rec + hp.edge_b * R.stride, hp.btasks[w]);
            }
        // coefficient products of the two extra records: no step length (only ONE, HALF and the b_l can occur)
        for (int e = tid; e < 2 * hp.npairs; e += nthr) {
            const int which = e / hp.npairs, pid = e - which * hp.npairs;
            cx.cp[(hp.edge_fp + which) * hp.npairs + pid] = cx.pairs[pid] == 0u ? cx.pc[pid] : 0.0;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// phase: sum of the stage-type points of a step (midpoint, more than 3 controls per step: hess_sums_stages, ctd_hess.hpp)
// ------------------------------------------------------------------------------------------------------
// For every record and every pair of directions without a control, the block of point 0 receives the sum over the points
// (fixed order 0, 1, .. S-1: reproducible); the control pairs stay with their point.
template <class P, int SC, int S>
CTD_HD void hess_phase_stage_sum(const HParams& hp, const HBlockCtx& cx, int tid, int nthr) {
    if constexpr (hess_sums_stages(SC, S)) {
        constexpr int n = P::NX, m = P::NU, nd = P::NX + P::NV;
        constexpr HessRecLayout R = HRL<P, SC, S>::R;
        for (int w = tid; w < cx.nslots * nd * nd; w += nthr) {
            const int k = w / (nd * nd), r = w - k * nd * nd, a = r / nd, b = r - a * nd;
            if (a > b) continue;
            double* HD = cx.rec + k * R.stride + R.oStage + hess_tri(R.md, a < n ? a : a + m, b < n ? b : b + m);
            double sum = HD[0];
#pragma unroll 1
            for (int j = 1; j < S; ++j) sum = sum + HD[j * R.stage_sz];
            HD[0] = sum;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// phase: emit
// ------------------------------------------------------------------------------------------------------
CTD_HD double hess_term(const double* rec, int stride, const double* cp, int npairs, uint32_t code, int slot) {
    return cp[slot * npairs + term_pair(code)] * rec[slot * stride + term_di(code)];
}

// Number of terms the lanes of one wave run in a pass over the segment: the largest count among them -- wave-uniform, so
// the switch on it below does not diverge (w0 .. w0 + nwl - 1: the wave's segment positions, for the host build)
CTD_HD int hess_wave_max(int own, int w0, int wend, const uint32_t* tptr, FastDiv div_nc, int nc, int nwl) {
#if defined(__HIP_DEVICE_COMPILE__)
    int m = 0;
#pragma unroll
    for (int t = 0; t < kMaxTerms; ++t) m += __any(own > t) ? 1 : 0;      // (a compare and a scalar test per term; no LDS)
    return m;
#else
    int m = 0;
    for (int w = w0; w < w0 + nwl && w < wend; ++w) {
        const int g = (int)fast_div((uint32_t)w, div_nc), e = w - g * nc;
        const int nt = (int)(tptr[e + 1] - tptr[e]);
        m = nt > m ? nt : m;
    }
    return m;
#endif
}

// One lane, one segment entry with NT (possibly padded) terms: rounds u = 0 .. nu - 1 (uniform) are the steps of the lane,
// G apart; the lane stores in rounds u < n_own.  pa / pb: the coefficient products and the record of the lane's first step;
// da / db / dout: what one round adds to the two LDS offsets and to the output position (uniform).
template <int NT>
CTD_HD void hess_emit_steps(const double* pa, const double* pb, const uint32_t* codes, int nt, int oZero, int da, int db, double* out,
                            int64_t dout, int nu, int n_own, int wt) {
    constexpr int M = NT > 0 ? NT : 1;
    const double *qa[M], *qb[M];
#pragma unroll
    for (int t = 0; t < NT; ++t) {        // absent terms read 1.0 * rec[oZero] = 0
        const uint32_t code = t < nt ? codes[t] : pack_tile_term(0, oZero);
        qa[t] = pa + tile_term_a(code);
        qb[t] = pb + tile_term_b(code);
    }
    for (int u = 0; u < nu; ++u) {
        if (u < n_own) {
            double a[M], b[M];
#pragma unroll
            for (int t = 0; t < NT; ++t) { a[t] = qa[t][u * da]; b[t] = qb[t][u * db]; }
            double acc = 0.0;
#pragma unroll
            for (int t = 0; t < NT; ++t) acc = acc + a[t] * b[t];
            emit_store(&out[u * dout], acc, wt);
        }
    }
}

template <class P, int SC, int S>
CTD_HD void hess_phase_emit(const HParams& hp, const HBlockCtx& cx, int block, int tid, int nthr) {
    constexpr HessRecLayout R = HRL<P, SC, S>::R;
    if (cx.is_edge) {
        // (the edge blocks share the entries: an entry is a chain of dependent global loads -- index, term range, codes)
        const int n1 = hp.edge_end - hp.edge_begin, ntot = n1 + (hp.edge2_end - hp.edge2_begin);
        const int chunk = (ntot + hp.n_edge_blocks - 1) / hp.n_edge_blocks;
        const int wend = (cx.edge_part + 1) * chunk < ntot ? (cx.edge_part + 1) * chunk : ntot;
        for (int w = cx.edge_part * chunk + tid; w < wend; w += nthr) {
            const int e = w < n1 ? hp.edge_begin + w : hp.edge2_begin + (w - n1);
            double acc = 0.0;
            for (uint32_t t = hp.eptr[e]; t < hp.eptr[e + 1]; ++t) {
                const uint32_t code = hp.eterms[t];
                acc = acc + hess_term(cx.rec, R.stride, cx.cp, hp.npairs, code, term_slot(code));
            }
            hp.vals[hp.edge_idx[e]] = acc;
        }
        for (int e = tid; e < hp.nvv; e += nthr) {
            double acc = 0.0;
            if (hp.edge_vv && cx.edge_part == 0)
                for (uint32_t t = hp.evptr[e]; t < hp.evptr[e + 1]; ++t) {
                    const uint32_t code = hp.eterms[t];
                    acc = acc + hess_term(cx.rec, R.stride, cx.cp, hp.npairs, code, term_slot(code));
                }
            hp.partials[(int64_t)cx.edge_part * hp.nvv + e] = acc;
        }
        return;
    }
    const int64_t i0 = cx.a > hp.reg_first ? cx.a : hp.reg_first;
    const int64_t i1 = cx.b < hp.reg_last ? cx.b : hp.reg_last;
    // lane (k, g) owns entry k of the segment for the steps i0 + g, i0 + g + G, ...: its (few) term codes are read once,
    // then it walks its steps; short segments are replicated G times across the workgroup.  Inside a wave every lane
    // runs the same number of terms (the wave's maximum) so that all LDS reads of a step are in flight together instead
    // of one exec-masked read-wait-fma chain per term.  The walk itself is address arithmetic on uniform strides: the
    // output position of lane (k, g) in round u is  vals[seg_base + (i0 - reg_first + g + u * G) * Lseg + cpos[k]].  Only the nc
    // entries that have terms are walked; the positions of the others (structural zeros of the pattern) are zero-filled.
    const int nreg = (int)(i1 - i0);
    if (nreg > 0 && hp.Lseg > 0) {
        double* out0 = hp.vals + hp.seg_base + (i0 - hp.reg_first) * (int64_t)hp.Lseg;
        const int nc = hp.nc;
        const int G = (nc > 0 && nc < nthr) ? nthr / nc : 1;
        const int nu = (nreg + G - 1) / G;
        const int k0 = (int)(i0 - cx.lo);
        const int64_t dout = (int64_t)G * hp.Lseg;
        const int da = G * hp.npairs, db = G * R.stride;
        const int wave_lo = tid & ~63, nwl = nthr - wave_lo < 64 ? nthr - wave_lo : 64;
        for (int w0 = wave_lo; w0 < nc * G; w0 += nthr) {
            const int w = w0 + (tid - wave_lo);
            const bool live = w < nc * G;
            const int g = live ? (int)fast_div((uint32_t)w, hp.div_nc) : 0;
            const int k = live ? w - g * nc : 0;
            const uint32_t t0 = cx.tptr[k];
            const int nt = live ? (int)(cx.tptr[k + 1] - t0) : 0;
            const int e = hp.compact ? (int)cx.cpos[k] : k;
            const int n_own = live ? nu - (g + (nu - 1) * G >= nreg ? 1 : 0) : 0;
            const int wmax = hess_wave_max(nt, w0, nc * G, cx.tptr, hp.div_nc, nc, nwl);
            const double* pa = cx.cp + (k0 + g) * hp.npairs;
            const double* pb = cx.rec + (k0 + g) * R.stride;
            const uint32_t* codes = cx.terms + t0;
            double* out = out0 + (int64_t)g * hp.Lseg + e;
            switch (wmax) {
                case 0: hess_emit_steps<0>(pa, pb, codes, nt, R.oZero, da, db, out, dout, nu, n_own, hp.wt_store); break;
                case 1: hess_emit_steps<1>(pa, pb, codes, nt, R.oZero, da, db, out, dout, nu, n_own, hp.wt_store); break;
                case 2: hess_emit_steps<2>(pa, pb, codes, nt, R.oZero, da, db, out, dout, nu, n_own, hp.wt_store); break;
                case 3: hess_emit_steps<3>(pa, pb, codes, nt, R.oZero, da, db, out, dout, nu, n_own, hp.wt_store); break;
                case 4: hess_emit_steps<4>(pa, pb, codes, nt, R.oZero, da, db, out, dout, nu, n_own, hp.wt_store); break;
                case 5: hess_emit_steps<5>(pa, pb, codes, nt, R.oZero, da, db, out, dout, nu, n_own, hp.wt_store); break;
                case 6: hess_emit_steps<6>(pa, pb, codes, nt, R.oZero, da, db, out, dout, nu, n_own, hp.wt_store); break;
                default: hess_emit_steps<kMaxTerms>(pa, pb, codes, nt, R.oZero, da, db, out, dout, nu, n_own, hp.wt_store); break;
            }
        }
        // compact = 2: zeros at the listed positions of every step (item j = step * nz + position index)
        for (int j = tid; j < hp.nz * nreg; j += nthr) {
            const int u = (int)fast_div((uint32_t)j, hp.div_nz);
            emit_store(&out0[u * (int64_t)hp.Lseg + cx.zpos[j - u * hp.nz]], 0.0, hp.wt_store);
        }
    }
    // V x V entries: one lane per (step of the tile, entry) adds up that step's terms; hess_phase_vvsum then sums the steps
    const int ns = (int)(cx.b - cx.a);
    for (int w = tid; w < ns * hp.nvv; w += nthr) {
        const int si = w / hp.nvv, e = w - si * hp.nvv;
        const int k = si + (int)(cx.a - cx.lo);
        double acc = 0.0;
        uint32_t t = cx.vptr[e];
        const uint32_t t1 = cx.vptr[e + 1];
        for (; t + 4 <= t1; t += 4) {      // four terms in flight (a term is a code load -> two LDS reads -> multiply-add chain)
            const uint32_t c0 = cx.vterms[t], c1 = cx.vterms[t + 1], c2 = cx.vterms[t + 2], c3 = cx.vterms[t + 3];
            const double p0 = hess_term(cx.rec, R.stride, cx.cp, hp.npairs, c0, k), p1 = hess_term(cx.rec, R.stride, cx.cp, hp.npairs, c1, k);
            const double p2 = hess_term(cx.rec, R.stride, cx.cp, hp.npairs, c2, k), p3 = hess_term(cx.rec, R.stride, cx.cp, hp.npairs, c3, k);
            acc = (((acc + p0) + p1) + p2) + p3;
        }
        for (; t < t1; ++t) acc = acc + hess_term(cx.rec, R.stride, cx.cp, hp.npairs, cx.vterms[t], k);
        cx.red[e * hp.T + si] = acc;
    }
}

// the tile's share of the V x V entries: its steps in step order (fixed summation order)
CTD_HD void hess_phase_vvsum(const HParams& hp, const HBlockCtx& cx, int block, int tid, int nthr) {
    if (cx.is_edge) return;
    const int ns = (int)(cx.b - cx.a);
    for (int e = tid; e < hp.nvv; e += nthr) {
        double acc = 0.0;
        int si = 0;
        for (; si + 4 <= ns; si += 4) {
            const double r0 = cx.red[e * hp.T + si], r1 = cx.red[e * hp.T + si + 1], r2 = cx.red[e * hp.T + si + 2], r3 = cx.red[e * hp.T + si + 3];
            acc = (((acc + r0) + r1) + r2) + r3;
        }
        for (; si < ns; ++si) acc = acc + cx.red[e * hp.T + si];
        hp.partials[(int64_t)block * hp.nvv + e] = acc;
    }
}

// V x V entries: sum of the per-workgroup partials in a fixed order (lane t takes workgroups t, t + nthr, ...; then a tree)
CTD_HD double hess_finish_partial(const HParams& hp, int e, int tid, int nthr) {
    double acc = 0.0;
    const volatile double* part = hp.partials;       // written by other workgroups of the same launch (last-workgroup finish)
    for (int b = tid; b < hp.ntiles + hp.n_edge_blocks; b += nthr) acc = acc + part[(int64_t)b * hp.nvv + e];
    return acc;
}

}  // namespace ctd
