// ctd_kernel_body.hpp -- the fused constraints + sparse-Jacobian evaluation, written as phase functions.
//
// One workgroup evaluates a TILE of consecutive time steps of the collocation grid (reference loop:
// `for i in 1:docp.time.steps`, src/DOCP_functions.jl:92-98), in phases separated by workgroup barriers:
//
//   load   the tile's slice of the NLP vector xu (step-major, external layout) is copied once into LDS with
//          coalesced loads, together with the normalized times tau_i of its grid points; every later read of
//          X_i, U_i^j, K_i^j, X_{i+1} comes from LDS
//   eval   one lane per (step, eval point, direction chunk): the OCP dynamics (and path constraints) are evaluated
//          on forward duals in registers -> df/dx, df/du, df/dt, df/dv and the values land in the step's LDS record
//          (replaces setWorkArray + stepStateConstraints! + stepPathConstraints! AND the ncolors Dual passes of
//          ADNLPModels: trapeze.jl:50-71,118-142, midpoint.jl:47-72,124-140, irk.jl:236-308,
//          irk_stagewise.jl:394-460, DOCP_functions.jl:122-140).  When all directions of a function fit one chunk the
//          same lane also finishes its part of the scheme's chain rule (no extra phase);
//   fin    only for OCPs whose directions need several chunks: one lane per (step, stage) combines the chunks
//          (d/dv through the free time grid of get_time_grid, DOCP_data.jl:437-458, residual rows, coefficients)
//   fin2   trapeze only: the step residual needs the dynamics of two nodes
//   emit   all lanes stream the outputs in their final external order with coalesced 8-byte stores:
//          c rows of the tile, the tile's contiguous range of CSC values (one 32-bit code per entry of the
//          step-periodic pattern: value = coef * record[di] + beta), and the tile's slice of every V column
//
// The first workgroup is the EDGE block: boundary constraints (DOCP_functions.jl:103-111), path constraints at
// the final time (:100) and the few CSC entries whose layout is not step-periodic (first and last step columns,
// final-state columns, tails of the V columns), driven by an explicit (index, code) list.
//
// The phase functions are plain templates over (OCP functor, scheme class, stage count); `tid`/`nthr` are the lane id
// and workgroup size.  ctd_kernels.hpp wraps them in the __global__ kernel; tests/emu/ steps them serially on the CPU
// (test infrastructure only -- the C ABI never takes that path).
#pragma once
#include "ctd_layout.hpp"


namespace ctd {

// Experiment builds only (make EXTRA=-DCTD_SUBSTAMPS): cycle stamps INSIDE the evaluation phase, taken by lane 0 of waves 0
// and 1 after draining the wave's memory counters (so each segment is timed serialised: an upper bound of its share).
// Words [grid * 12, grid * 28) of the ctd_debug_stamps buffer; never compiled into the shipped library.
#if defined(CTD_SUBSTAMPS) && defined(__HIP_DEVICE_COMPILE__)
#ifndef CTD_SUB_WAVE0
#define CTD_SUB_WAVE0 0          /* first of the two waves whose stamps are kept */
#endif
#define CTD_SUB(kp, id)                                                                                              \
    do {                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        __builtin_amdgcn_s_waitcnt(0);                                                                               \
        if ((kp).stamps && (threadIdx.x & 63) == 0 && threadIdx.x >= 64 * CTD_SUB_WAVE0 && threadIdx.x < 64 * CTD_SUB_WAVE0 + 128) \
            (kp).stamps[(size_t)gridDim.x * 12 + ((size_t)blockIdx.x * 2 + (threadIdx.x >> 6) - CTD_SUB_WAVE0) * 8 + (id)] = clock64(); \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    } while (0)
#else
#define CTD_SUB(kp, id) do { } while (0)
#endif
// the same inside the EMIT phase (make EXTRA=-DCTD_SUBSTAMPS_EMIT; the memory counters are NOT drained: issue times)
#if defined(CTD_SUBSTAMPS_EMIT) && defined(__HIP_DEVICE_COMPILE__)
#define CTD_SUBE(kp, id)                                                                                             \
    do {                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        if ((kp).stamps && (threadIdx.x & 63) == 0 && threadIdx.x < 128)                                             \
            (kp).stamps[(size_t)gridDim.x * 12 + ((size_t)blockIdx.x * 2 + (threadIdx.x >> 6)) * 8 + (id)] = clock64(); \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    } while (0)
#else
#define CTD_SUBE(kp, id) do { } while (0)
#endif

struct BlockCtx {
    int is_edge;
    int edge_part;     // edge blocks only: which of the kp.has_edge edge workgroups this is (each evaluates the edge records and emits its
                       // share of the explicit (index, code) list)
    int direct;        // 1: in / v point into xu (global memory), tau is null (make_direct_ctx)
    int nslots;        // records held by this block (step / node records)
    int in_stride;     // doubles between the inputs of consecutive slots
    int64_t a, b;      // steps [a, b) whose outputs this tile emits
    int64_t lo;        // step / node index of slot 0 (tile)
    double* in;        // staged slice of xu
    double* v;         // optimisation variables
    double* tau;       // normalized times: tile: tau[k] = tau_{lo+k}, k <= nslots+1; edge: tau[2k], tau[2k+1], tau[2 ns] = tau_N
    double* rec;       // records
    const uint32_t* codes;   // emit template of the step segment (Lseg codes): LDS copy when small, else the global table
    const uint32_t* vcodes;  // V-column templates (nv * vr codes)
    const XHalo* halo;       // sharded iterate read in place: non-null in the blocks that read other shards' variables (first /
                             // last tile of the shard, edge block); null everywhere else and whenever xu holds all the block reads
    const double* xu;        // the kernel's own iterate buffer (direct blocks with a halo table)
};

// blocks of a shard whose reads leave the shard's own variables (kp.halo set): the tiles at either end, the edge block
CTD_HD const XHalo* block_halo(const KParams& kp, const BlockCtx& cx) {
    if (!kp.halo) return nullptr;
    if (cx.is_edge) return kp.halo;
    return (cx.lo < kp.step_begin || cx.b + kp.HH >= kp.step_end) ? kp.halo : nullptr;
}

// doubles reserved at the start of the LDS for the emit templates (staged once per workgroup by load_codes)
// (only small templates are staged: copying thousands of codes per workgroup costs more than the latency it hides)
constexpr int kMaxStagedCodes = 1024;
CTD_HD bool codes_staged(const KParams& kp) { return kp.stage_codes != 0; }
// LDS is handed out in granules of 320 dwords on gfx950 (160 KiB per CU): workgroups of `lds_bytes` that fit one CU
inline int wgs_per_cu(int64_t lds_bytes) {
    const int64_t g = (lds_bytes + 1279) / 1280 * 1280;
    return g > 0 ? (int)((160 * 1024) / g) : 32;
}
CTD_HD int code_doubles(const KParams& kp) { return codes_staged(kp) ? (kp.Lseg + kp.L.nv * kp.vr + 1) / 2 : 0; }

CTD_HD int64_t slot_index(const KParams& kp, const BlockCtx& cx, int k) {
    return cx.is_edge ? kp.edge_steps[k] : cx.lo + k;
}

CTD_HD BlockCtx make_ctx(const KParams& kp, int block, double* lds) {
    BlockCtx cx;
    cx.direct = 0;
    const Layout& L = kp.L;
    cx.codes = codes_staged(kp) ? reinterpret_cast<const uint32_t*>(lds) : kp.tmpl;
    cx.vcodes = codes_staged(kp) ? cx.codes + kp.Lseg : kp.vtmpl;
    lds += code_doubles(kp);
    cx.edge_part = 0;
    if (block < kp.has_edge) {
        cx.is_edge = 1;
        cx.edge_part = block;
        cx.nslots = kp.n_edge_slots;
        cx.in_stride = edge_in_stride(L);
        cx.a = cx.b = cx.lo = 0;
        cx.in = lds;
        cx.v = cx.in + cx.nslots * cx.in_stride;
        cx.tau = cx.v + kMaxNV;
        cx.rec = cx.tau + 2 * kMaxEdgeSlots + 2;
    } else {
        const int tb = block - kp.has_edge;
        const int tile = kp.xcd_remap ? xcd_tile(tb, kp.ntiles) : tb;
        const int cap = kp.T + kp.HL + kp.HH;
        cx.is_edge = 0;
        cx.a = kp.step_begin + (int64_t)tile * kp.T;
        cx.b = cx.a + kp.T < kp.step_end ? cx.a + kp.T : kp.step_end;
        cx.lo = cx.a - kp.HL;
        cx.nslots = (int)(cx.b - cx.a) + kp.HL + kp.HH;
        cx.in_stride = tile_in_stride(L);
        cx.in = lds;
        cx.v = cx.in + (cap + 1) * tile_in_stride(L) + L.n + L.m;
        cx.tau = cx.v + kMaxNV;
        cx.rec = cx.tau + cap + 2;
    }
    cx.xu = nullptr;
    cx.halo = block_halo(kp, cx);
    return cx;
}

// Direct tiles: OCPs whose functions are differentiated in one pass (Dirs<P>::FUSED) on the one-point schemes and the
// Gauss-Legendre schemes skip the staging of xu in LDS.  An evaluating lane reads the handful of doubles of its own step
// straight from global memory (L2-resident: x was just written by the solver) into registers, so the load -> barrier ->
// evaluate hop of the staged driver disappears; the only workgroup barrier left sits between the evaluation and the emission.
// `in`, `v` point into xu (global address space: this context is built on a code path of its own so the compiler emits
// global loads, not flat ones), `tau` is null: slot_tau computes the times of the lane's own grid points.
CTD_HD BlockCtx make_direct_ctx(const KParams& kp, int block, double* lds, const double* xu) {
    BlockCtx cx;
    const Layout& L = kp.L;
    cx.direct = 1;
    cx.codes = kp.tmpl;
    cx.vcodes = kp.vtmpl;
    cx.in_stride = L.blk;
    cx.v = const_cast<double*>(xu) + L.v_off;
    cx.tau = nullptr;
    cx.rec = lds + code_doubles(kp);
    cx.xu = xu;
    cx.halo = nullptr;
    cx.edge_part = 0;
    if (block < kp.has_edge) {
        // edge block: slot k holds step kp.edge_steps[k] (slot_base); X_{i+1} (and U_{i-1} for implicit Euler) are where
        // the global layout has them
        cx.is_edge = 1;
        cx.edge_part = block;
        cx.nslots = kp.n_edge_slots;
        cx.a = cx.b = cx.lo = 0;
        cx.in = const_cast<double*>(xu);
        cx.halo = kp.halo;
        return cx;
    }
    const int tile = block - kp.has_edge;      // (the XCD-aware tile order of the staged driver measured neutral: not offered here)
    cx.is_edge = 0;
    cx.a = kp.step_begin + (int64_t)tile * kp.T;
    cx.b = cx.a + kp.T < kp.step_end ? cx.a + kp.T : kp.step_end;
    cx.lo = cx.a - kp.HL;
    cx.nslots = (int)(cx.b - cx.a) + kp.HL + kp.HH;
    cx.in = const_cast<double*>(xu) + cx.lo * (int64_t)L.blk;
    cx.halo = block_halo(kp, cx);
    return cx;
}

// inputs of slot k: staged copy in LDS, or (direct) the step's own block of xu -- of the owner's buffer when the iterate is
// sharded and the step belongs to a neighbour
CTD_HD const double* slot_base(const KParams& kp, const BlockCtx& cx, int k) {
    if (cx.direct && cx.halo) {
        const int64_t g = slot_index(kp, cx, k) * (int64_t)cx.in_stride;
        return xnear(kp.near, cx.xu, g) + g;
    }
    if (cx.direct && cx.is_edge) return cx.in + kp.edge_steps[k] * (int64_t)cx.in_stride;
    return cx.in + k * cx.in_stride;
}
// X_{i+1} (and U_{i+1}) of slot k: behind the step's block, in the staged copy and in every buffer that holds both -- the
// next shard's buffer for the last step of a shard
CTD_HD const double* slot_next(const KParams& kp, const BlockCtx& cx, int k) {
    if (cx.direct && cx.halo) {
        const int64_t g = (slot_index(kp, cx, k) + 1) * (int64_t)cx.in_stride;
        return xnear(kp.near, cx.xu, g) + g;
    }
    return slot_base(kp, cx, k) + ((cx.direct || cx.is_edge) ? kp.L.blk : cx.in_stride);     // (staged tile: the next slot)
}
// block of step i - 1 (implicit Euler's path control U_{i-1}); direct blocks only
CTD_HD const double* slot_prev(const KParams& kp, const BlockCtx& cx, int k) {
    if (cx.direct && cx.halo) {
        const int64_t g = (slot_index(kp, cx, k) - 1) * (int64_t)cx.in_stride;
        return xnear(kp.near, cx.xu, g) + g;
    }
    return slot_base(kp, cx, k) - ((cx.direct || cx.is_edge) ? kp.L.blk : cx.in_stride);
}

// LDS doubles a block needs (host uses this to size the launch)
inline int64_t lds_doubles(const KParams& kp) {
    const Layout& L = kp.L;
    const int64_t cap = kp.T + kp.HL + kp.HH;
    int64_t tile = code_doubles(kp) + (cap + 1) * tile_in_stride(L) + L.n + L.m + kMaxNV + cap + 2 + cap * kp.R.stride;
    int64_t edge = code_doubles(kp) + (int64_t)kp.n_edge_slots * edge_in_stride(L) + kMaxNV + 2 * kMaxEdgeSlots + 2 +
                   (int64_t)(kp.n_edge_slots + 1) * kp.R.stride + kp.R.bsize;      // step slots, final-path record, boundary record
    return tile > edge ? tile : edge;
}

// normalized time of grid point i: collect(LinRange(0, 1, N+1))[i+1] = i / N (src/DOCP_data.jl:179-183), or the
// user grid normalised on the host (:191-199).  Evaluated once per slot in phase_load (one FP64 division per slot).
CTD_HD double tau_global(const KParams& kp, int64_t i) {
    if (i < 0) i = 0;
    if (i > kp.L.N) i = kp.L.N;
    return kp.tau ? kp.tau[i] : (double)i / (double)kp.L.N;
}
// direct blocks hold no staged times: the lane reads (or computes) tau of its own grid points
CTD_HD double slot_tau(const KParams& kp, const BlockCtx& cx, int k, int d) {
    if (cx.direct) return tau_global(kp, slot_index(kp, cx, k) + d);
    return cx.is_edge ? cx.tau[2 * k + d] : cx.tau[k + d];
}
CTD_HD double final_tau(const KParams& kp, const BlockCtx& cx) {       // edge block only
    return cx.direct ? tau_global(kp, kp.L.N) : cx.tau[2 * cx.nslots];
}

// get_time_grid (src/DOCP_data.jl:437-458): t_i = t0 + tau_i (tf - t0), t0/tf fixed or components of v
template <class P> CTD_HD double time_of(const KParams& kp, const double* v, double tau) {
    const double t0 = (P::IT0 >= 0) ? v[P::IT0 >= 0 ? P::IT0 : 0] : kp.L.t0;
    const double tf = (P::ITF >= 0) ? v[P::ITF >= 0 ? P::ITF : 0] : kp.L.tf;
    return t0 + tau * (tf - t0);
}
// d t / d v_k, following the dual arithmetic of the same expression
template <class P> CTD_HD double dtime_of(double tau, int k) {
    const double dt0 = (P::IT0 == k) ? 1.0 : 0.0;
    const double dtf = (P::ITF == k) ? 1.0 : 0.0;
    return dt0 + tau * (dtf - dt0);
}

// record layout of (OCP, scheme class, stages): a compile-time constant, so every LDS access of the kernels uses an
// immediate offset (the host builds its emit codes from the same function, ctd_host.cpp)
// S: stages of a Gauss-Legendre scheme; for the midpoint scheme the number of controls per step (control_steps, 1 in collocation)
// Structural nonzeros of the dynamics' first partials: slot of d f_r / d x_c inside the F block (fx) and of d f_r / d u_c inside a
// G block (gu), or -1 where the derivative is identically zero.  Dense (every pair has its n x ldx / n x ldg place) for OCPs
// evaluated with forward duals; OCPs with generated dynamics code (ctd_sym_registry.hpp at build time, the functor's text for
// run-time OCPs) specialise it: the generator only stores -- and the records only hold -- the nonzeros.
template <class P> struct DynNZ {
    static constexpr bool sparse = false;
    static constexpr int nF = -1, nG = -1;
    CTD_HD static constexpr int fx(int, int) { return -1; }
    CTD_HD static constexpr int gu(int, int) { return -1; }
};
template <class P, int SC, int S> struct RL {
    static constexpr int cb = (SC == SC_IRK ? P::NX * (1 + S) : P::NX) + P::NPATH;
    static constexpr RecLayout R = make_rec_layout(P::NX, P::NU, P::NV, P::NPATH, P::NBC, SC == SC_IRK ? S : 0, cb,
                                                   SC == SC_MIDPOINT ? P::NU * S : P::NU, DynNZ<P>::nF, DynNZ<P>::nG);
    // offset of d f_r / d x_c (d f_r / d u_c of control block jb) inside an eval block, or -1: structurally zero (sparse blocks only)
    CTD_HD static constexpr int F(int r, int c) {
        if (DynNZ<P>::sparse) { const int s = DynNZ<P>::fx(r, c); return s < 0 ? -1 : R.oF + s; }
        return R.oF + r * R.ldx + c;
    }
    CTD_HD static constexpr int G(int r, int c, int jb = 0) {
        if (DynNZ<P>::sparse) { const int s = DynNZ<P>::gu(r, c); return s < 0 ? -1 : R.oG + jb * DynNZ<P>::nG + s; }
        return R.oG + r * R.ldg + jb * P::NU + c;
    }
};
// evaluation points per step that get a lane (and an eval block) of their own: the stages; the control sub-steps of the midpoint
// scheme are walked by ONE lane (their partials are summed)
template <int SC, int S> struct StagePoints { static constexpr int value = SC == SC_IRK ? S : 1; };

template <class P> struct Dirs {
    static constexpr int DYN = P::NX + P::NU + (P::DYN_T ? 1 : 0) + (P::DYN_V ? P::NV : 0);
    static constexpr int PATH = P::NX + P::NU + (P::PATH_T ? 1 : 0) + (P::PATH_V ? P::NV : 0);
    static constexpr int BND = 2 * P::NX + P::NV;
    static constexpr int DC = P::DC;
    static constexpr int NCH_DYN = (DYN + DC - 1) / DC;
    static constexpr int NCH_PATH = (PATH + DC - 1) / DC;
    static constexpr int NCH_BND = (BND + DC - 1) / DC;
    // every function of the OCP is differentiated in a single pass: the evaluating lane finishes the chain rule
    static constexpr bool FUSED = (NCH_DYN == 1) && (P::NPATH == 0 || NCH_PATH == 1);
    static constexpr bool FREE = (P::IT0 >= 0) || (P::ITF >= 0);
};

// tiles of this (OCP, scheme class) run the direct driver (make_direct_ctx); the trapeze residual needs the records of two
// nodes (phase_fin2) and keeps the staged one
template <class P, int SC> struct DirectTile { static constexpr bool value = Dirs<P>::FUSED && SC != SC_TRAPEZE; };

// ------------------------------------------------------------------------------------------------------
// phase: load
// ------------------------------------------------------------------------------------------------------
// stage the emit templates in LDS (once per workgroup): the emit loops then start from an LDS read instead of a
// dependent global load
CTD_HD void load_codes(const KParams& kp, const BlockCtx& cx, int tid, int nthr) {
    if (!codes_staged(kp)) return;
    uint32_t* dst = const_cast<uint32_t*>(cx.codes);
    for (int e = tid; e < kp.Lseg; e += nthr) dst[e] = kp.tmpl[e];
    const int nvc = kp.L.nv * kp.vr;
    for (int e = tid; e < nvc; e += nthr) dst[kp.Lseg + e] = kp.vtmpl[e];
}

template <class P, int SC, int S, bool LOAD_V = true>
CTD_HD void phase_load(const KParams& kp, const BlockCtx& cx, const double* __restrict__ xu, int tid, int nthr) {
    const Layout& L = kp.L;
    if (cx.is_edge) {
        const int per = cx.in_stride;
        for (int e = tid; e < cx.nslots * per; e += nthr) {
            const int k = e / per, o = e - k * per;
            int64_t g = kp.edge_steps[k] * L.blk + o;
            if (o >= L.blk + L.n + L.m)        // control of the previous step (own step for step 0): implicit Euler's path control
                g = (kp.edge_steps[k] >= 1 ? kp.edge_steps[k] - 1 : 0) * (int64_t)L.blk + L.n + (o - (L.blk + L.n + L.m));
            cx.in[e] = (g < L.v_off) ? (cx.halo ? xnear(kp.near, xu, g) : xu)[g] : 0.0;
        }
        for (int e = tid; e <= 2 * cx.nslots; e += nthr)
            cx.tau[e] = (e == 2 * cx.nslots) ? tau_global(kp, L.N) : tau_global(kp, kp.edge_steps[e >> 1] + (e & 1));
    } else {
        const int64_t g0 = (cx.lo < 0 ? 0 : cx.lo) * (int64_t)L.blk;
        int64_t g1 = (cx.lo + cx.nslots) * (int64_t)L.blk + L.n + L.m;
        if (g1 > L.v_off) g1 = L.v_off;
        // 32-bit lane-relative indices: the 64-bit part of the addresses is wave-uniform
        const double* __restrict__ src = xu + g0;
        const int cnt = (int)(g1 - g0);
        // element e of the slice (offset eo from the first slot's block) -> slot eo / blk of the staged copy, pitch in_stride
        const int eo0 = (int)(g0 - cx.lo * (int64_t)L.blk), padw = cx.in_stride - L.blk;
        auto at = [&](int e) -> double& { const int eo = eo0 + e; return cx.in[eo + (int)fast_div((uint32_t)eo, kp.div_blk) * padw]; };
        // A lane first ISSUES its global loads of every stream (two elements of the xu slice, its optimisation variable, its
        // share of the emit templates), then stores them to LDS: one exposed memory latency instead of one per copy loop
        const bool codes = LOAD_V && codes_staged(kp);
        const int nvc = kp.L.nv * kp.vr;
        if (cx.halo) {
            // first / last tile of a shard with the iterate sharded: every element from the buffer of the shard that owns it
            for (int e = tid; e < cnt; e += nthr) at(e) = xnear(kp.near, xu, g0 + e)[g0 + e];
            if (LOAD_V && tid < kMaxNV) cx.v[tid] = (tid < P::NV) ? xu[L.v_off + tid] : 0.0;
            if (codes) {
                uint32_t* cd = const_cast<uint32_t*>(cx.codes);
                for (int e = tid; e < kp.Lseg; e += nthr) cd[e] = kp.tmpl[e];
                for (int e = tid; e < nvc; e += nthr) cd[kp.Lseg + e] = kp.vtmpl[e];
            }
            for (int e = tid; e <= cx.nslots + 1; e += nthr) cx.tau[e] = tau_global(kp, cx.lo + e);
            return;
        }
        const double x0 = tid < cnt ? src[tid] : 0.0;
        const double x1 = tid + nthr < cnt ? src[tid + nthr] : 0.0;
        const double vv = (LOAD_V && tid < P::NV) ? xu[L.v_off + tid] : 0.0;
        const uint32_t c0 = (codes && tid < kp.Lseg) ? kp.tmpl[tid] : 0u;
        const uint32_t c1 = (codes && tid < nvc) ? kp.vtmpl[tid] : 0u;
        const double tau_e = tid <= cx.nslots + 1 ? tau_global(kp, cx.lo + tid) : 0.0;      // (table load: issued with the rest)
        if (tid < cnt) at(tid) = x0;
        if (tid + nthr < cnt) at(tid + nthr) = x1;
        if (LOAD_V && tid < kMaxNV) cx.v[tid] = vv;
        if (codes) {
            uint32_t* cd = const_cast<uint32_t*>(cx.codes);
            if (tid < kp.Lseg) cd[tid] = c0;
            if (tid < nvc) cd[kp.Lseg + tid] = c1;
            for (int e = tid + nthr; e < kp.Lseg; e += nthr) cd[e] = kp.tmpl[e];
            for (int e = tid + nthr; e < nvc; e += nthr) cd[kp.Lseg + e] = kp.vtmpl[e];
        }
        for (int e = tid + 2 * nthr; e < cnt; e += nthr) at(e) = src[e];
        if (tid <= cx.nslots + 1) cx.tau[tid] = tau_e;
        for (int e = tid + nthr; e <= cx.nslots + 1; e += nthr) cx.tau[e] = tau_global(kp, cx.lo + e);
        return;
    }
    if (LOAD_V && tid < kMaxNV) cx.v[tid] = (tid < P::NV) ? xu[L.v_off + tid] : 0.0;
}

// The x slice of a tile in two halves (multi-tile workgroups, KParams::wg_stride): load_issue reads the lane's elements of the
// NEXT tile into registers before the current tile is emitted, load_commit stores them to the staged copy afterwards -- the
// global-memory latency of the slice hides behind the emission.  Tiles at a shard boundary of a sharded iterate (cx.halo) and
// slices longer than two elements per lane fall back to plain loads in load_commit.
struct TileIn { double x0, x1, tau; };
template <class P>
CTD_HD TileIn load_issue(const KParams& kp, const BlockCtx& cx, const double* __restrict__ xu, int tid, int nthr) {
    TileIn t{0.0, 0.0, 0.0};
    if (cx.halo) return t;
    const Layout& L = kp.L;
    const int64_t g0 = (cx.lo < 0 ? 0 : cx.lo) * (int64_t)L.blk;
    int64_t g1 = (cx.lo + cx.nslots) * (int64_t)L.blk + L.n + L.m;
    if (g1 > L.v_off) g1 = L.v_off;
    const double* __restrict__ src = xu + g0;
    const int cnt = (int)(g1 - g0);
    if (tid < cnt) t.x0 = src[tid];
    if (tid + nthr < cnt) t.x1 = src[tid + nthr];
    if (tid <= cx.nslots + 1) t.tau = tau_global(kp, cx.lo + tid);
    return t;
}
template <class P, int SC, int S>
CTD_HD void load_commit(const KParams& kp, const BlockCtx& cx, const double* __restrict__ xu, const TileIn& t, int tid, int nthr) {
    if (cx.halo) { phase_load<P, SC, S, false>(kp, cx, xu, tid, nthr); return; }
    const Layout& L = kp.L;
    const int64_t g0 = (cx.lo < 0 ? 0 : cx.lo) * (int64_t)L.blk;
    int64_t g1 = (cx.lo + cx.nslots) * (int64_t)L.blk + L.n + L.m;
    if (g1 > L.v_off) g1 = L.v_off;
    const double* __restrict__ src = xu + g0;
    const int cnt = (int)(g1 - g0);
    const int eo0 = (int)(g0 - cx.lo * (int64_t)L.blk), padw = cx.in_stride - L.blk;
    auto at = [&](int e) -> double& { const int eo = eo0 + e; return cx.in[eo + (int)fast_div((uint32_t)eo, kp.div_blk) * padw]; };
    if (tid < cnt) at(tid) = t.x0;
    if (tid + nthr < cnt) at(tid + nthr) = t.x1;
    for (int e = tid + 2 * nthr; e < cnt; e += nthr) at(e) = src[e];
    if (tid <= cx.nslots + 1) cx.tau[tid] = t.tau;
    for (int e = tid + nthr; e <= cx.nslots + 1; e += nthr) cx.tau[e] = tau_global(kp, cx.lo + e);
}

// ------------------------------------------------------------------------------------------------------
// finishing pieces (called from the eval lanes when Dirs<P>::FUSED, from phase_fin otherwise)
// ------------------------------------------------------------------------------------------------------
template <class P> CTD_HD void fill_const_coefs(const KParams& kp, double* C) {
#pragma unroll
    for (int e = 0; e < kNC; ++e) C[e] = 0.0;
    C[C_ONE] = 1.0; C[C_NEG1] = -1.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) C[C_B + j] = kp.L.b[j];
}

// per-record coefficients + (IRK) the state-equation rows, which depend on the inputs only
// `row` < 0: the whole lead role of slot k; `row` = r >= 0 (Gauss-Legendre schemes, wide states): state row r only, the
// coefficients with row 0 -- one lane per (step, row) instead of a serial walk over the n rows
template <class P, int SC, int S>
CTD_HD void fin_lead(const KParams& kp, const BlockCtx& cx, int k, int row = -1) {
    constexpr int n = P::NX, nv = P::NV;
    const Layout& L = kp.L;
    constexpr RecLayout R = RL<P, SC, S>::R;
    const int64_t i = slot_index(kp, cx, k);
    double* rec = cx.rec + k * R.stride;
    double* C = rec + R.oC;
    if (row <= 0) fill_const_coefs<P>(kp, C);
    if (i < 0 || i >= L.N) return;
    const double tau0 = slot_tau(kp, cx, k, 0), tau1 = slot_tau(kp, cx, k, 1);
    const double h = time_of<P>(kp, cx.v, tau1) - time_of<P>(kp, cx.v, tau0);
    if (SC == SC_IRK) {
        const double* base = slot_base(kp, cx, k);
        const double* nxt = slot_next(kp, cx, k);
        const double* K = base + n + L.cu;
        if (row <= 0) {
#pragma unroll
            for (int j = 0; j < S; ++j) {
#pragma unroll
                for (int l = 0; l < S; ++l) C[C_HA + 3 * j + l] = -(h * L.a[3 * j + l]);
                C[C_HB + j] = -(h * L.b[j]);
            }
        }
        if (row >= 0) {          // one state row, r a runtime value (addresses only)
            const int r = row;
            double sumbk = L.b[0] * K[r];
#pragma unroll
            for (int j = 1; j < S; ++j) sumbk = sumbk + L.b[j] * K[j * n + r];
            rec[R.oR + r] = nxt[r] - (base[r] + h * sumbk);
#pragma unroll
            for (int kk = 0; kk < nv; ++kk) {
                const double dh = Dirs<P>::FREE ? dtime_of<P>(tau1, kk) - dtime_of<P>(tau0, kk) : 0.0;
                rec[R.oSv + r * nv + kk] = -(dh * sumbk);
            }
            return;
        }
        // state rows: X_{i+1} - (X_i + h sum_j b_j K^j)   (irk_stagewise.jl:456-457, irk.jl:304-306)
#pragma unroll
        for (int r = 0; r < n; ++r) {
            double sumbk = L.b[0] * K[r];
#pragma unroll
            for (int j = 1; j < S; ++j) sumbk = sumbk + L.b[j] * K[j * n + r];
            rec[R.oR + r] = nxt[r] - (base[r] + h * sumbk);
#pragma unroll
            for (int kk = 0; kk < nv; ++kk) {
                const double dh = Dirs<P>::FREE ? dtime_of<P>(tau1, kk) - dtime_of<P>(tau0, kk) : 0.0;
                rec[R.oSv + r * nv + kk] = -(dh * sumbk);
            }
        }
    } else if (SC == SC_MIDPOINT) {
        C[C_NHH] = -(0.5 * (h / (double)S));      // h_i = (t_{i+1} - t_i) / control_steps   (midpoint.jl:134)
        C[C_NH] = -(h / (double)S);
    } else {
        C[C_NHH] = -(0.5 * h);
    }
}

// the part of the chain rule that needs all partials of one eval point
// `ev`: the eval block of (slot k, point j) -- its place in the LDS record, or a register copy the caller stores afterwards
// (r0, rstep): the rows r0, r0 + rstep, ... only -- the lane that evaluated those rows of the dynamics (split evaluation: one
// part per wave, so the row tests are wave-uniform branches) finishes them itself
template <class P, int SC, int S>
CTD_HD void fin_stage(const KParams& kp, const BlockCtx& cx, int k, int j, double* ev, int r0 = 0, int rstep = 1) {
    constexpr int n = P::NX, nv = P::NV;
    auto mine = [&](int r) { return rstep == 1 || (r % rstep) == r0; };
    constexpr bool FREE = Dirs<P>::FREE;
    const Layout& L = kp.L;
    constexpr RecLayout R = RL<P, SC, S>::R;
    const int64_t i = slot_index(kp, cx, k);
    if (i < 0) return;
    if (SC == SC_TRAPEZE ? (i > L.N) : (i >= L.N)) return;
    double* rec = cx.rec + k * R.stride;
    const double* base = slot_base(kp, cx, k);
    const double tau0 = slot_tau(kp, cx, k, 0), tau1 = slot_tau(kp, cx, k, 1);
    double dti[nv > 0 ? nv : 1], dh[nv > 0 ? nv : 1];
#pragma unroll
    for (int kk = 0; kk < nv; ++kk) {
        dti[kk] = FREE ? dtime_of<P>(tau0, kk) : 0.0;
        dh[kk] = FREE ? dtime_of<P>(tau1, kk) - dti[kk] : 0.0;
    }
    if (SC == SC_IRK) {
        const double* K = base + n + L.cu;
        // stage rows: K_i^j - f(...)   (irk_stagewise.jl:448-451)
#pragma unroll
        for (int r = 0; r < n; ++r)
            if (mine(r)) rec[R.oR + n + j * n + r] = K[j * n + r] - ev[R.of + r];
#pragma unroll
        for (int kk = 0; kk < nv; ++kk) {
            // d x_ij / d v_kk = dh * sum_l a_jl K^l  (x_i itself does not depend on v)
            double dx[n > 0 ? n : 1];
#pragma unroll
            for (int c = 0; c < n; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int l = 0; l < S; ++l) acc = acc + (dh[kk] * butcher_a<S>(L, j, l)) * K[l * n + c];
                dx[c] = acc;
            }
            const double dtij = dti[kk] + butcher_c<S>(L, j) * dh[kk];
#pragma unroll
            for (int r = 0; r < n; ++r) {
                if (!mine(r)) continue;
                double w = P::DYN_V ? ev[R.oW + r * nv + kk] : 0.0;
                if (P::DYN_T && FREE) w = w + ev[R.oft + r] * dtij;
                if (FREE) {
#pragma unroll
                    for (int c = 0; c < n; ++c)
                        if (RL<P, SC, S>::F(r, c) >= 0) w = w + ev[RL<P, SC, S>::F(r, c)] * dx[c];
                }
                ev[R.oW + r * nv + kk] = w;
            }
        }
    } else if (SC == SC_MIDPOINT) {
        // S = control_steps: f, W, ft hold the sums over the sub-steps (same t_s, x_s for all of them, midpoint.jl:53-69), so
        // x_{i+1} - (x_i + h_i sum_j f_j) with h_i = (t_{i+1} - t_i) / S  (:134-153; S = 1: :139)
        const double h = (time_of<P>(kp, cx.v, tau1) - time_of<P>(kp, cx.v, tau0)) / (double)S;
        const double* nxt = slot_next(kp, cx, k);
#pragma unroll
        for (int r = 0; r < n; ++r) {
            if (!mine(r)) continue;
            const double f = ev[R.of + r];
            rec[R.oR + r] = nxt[r] - (base[r] + h * f);    // midpoint.jl:139
#pragma unroll
            for (int kk = 0; kk < nv; ++kk) {
                double w = P::DYN_V ? ev[R.oW + r * nv + kk] : 0.0;
                if (P::DYN_T && FREE)
                    w = w + ev[R.oft + r] * (L.euler == 0 ? 0.5 * (dti[kk] + (dti[kk] + dh[kk])) : (L.euler == 1 ? dti[kk] : dti[kk] + dh[kk]));
                ev[R.oW + r * nv + kk] = w;
                rec[R.oSv + r * nv + kk] = -((S > 1 ? dh[kk] / (double)S : dh[kk]) * f + h * w);
            }
        }
    } else {  // SC_TRAPEZE: node-level total d f / d v; the step residual needs the next node (fin_trapeze_step)
#pragma unroll
        for (int r = 0; r < n; ++r)
#pragma unroll
            for (int kk = 0; kk < nv; ++kk) {
                if (!mine(r)) continue;
                double w = P::DYN_V ? ev[R.oW + r * nv + kk] : 0.0;
                if (P::DYN_T && FREE) w = w + ev[R.oft + r] * dti[kk];
                ev[R.oW + r * nv + kk] = w;
            }
    }
}

// One ROW r of fin_stage for Gauss-Legendre schemes, r a runtime lane parameter (LDS addresses only, no per-row code):
// wide-state OCPs spread the n rows of a stage over n lanes instead of walking them serially.
template <class P, int S>
CTD_HD void fin_stage_row(const KParams& kp, const BlockCtx& cx, int k, int j, int r) {
    constexpr int n = P::NX, nv = P::NV;
    constexpr bool FREE = Dirs<P>::FREE;
    constexpr RecLayout R = RL<P, SC_IRK, S>::R;
    const Layout& L = kp.L;
    const int64_t i = slot_index(kp, cx, k);
    if (i < 0 || i >= L.N) return;
    double* rec = cx.rec + k * R.stride;
    double* ev = rec + R.oEval + j * R.eval_sz;
    const double* K = slot_base(kp, cx, k) + n + L.cu;
    const double tau0 = slot_tau(kp, cx, k, 0), tau1 = slot_tau(kp, cx, k, 1);
    rec[R.oR + n + j * n + r] = K[j * n + r] - ev[R.of + r];
#pragma unroll
    for (int kk = 0; kk < nv; ++kk) {
        const double dti = FREE ? dtime_of<P>(tau0, kk) : 0.0;
        const double dh = FREE ? dtime_of<P>(tau1, kk) - dti : 0.0;
        double w = P::DYN_V ? ev[R.oW + r * nv + kk] : 0.0;
        if (P::DYN_T && FREE) w = w + ev[R.oft + r] * (dti + butcher_c<S>(L, j) * dh);
        if (FREE) {
#pragma unroll
            for (int c = 0; c < n; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int l = 0; l < S; ++l) acc = acc + (dh * butcher_a<S>(L, j, l)) * K[l * n + c];
                const int sl = RL<P, SC_IRK, S>::F(r, c);      // (r is a lane parameter: a table lookup for sparse blocks)
                if (sl >= 0) w = w + ev[sl] * acc;
            }
        }
        ev[R.oW + r * nv + kk] = w;
    }
}

// path rows: total d/dv = explicit + dg/dt * dt/dv
template <class P, int SC, int S>
CTD_HD void fin_path(const KParams& kp, double* rec, double tau) {
    constexpr int nv = P::NV, np = P::NPATH;
    constexpr RecLayout R = RL<P, SC, S>::R;
#pragma unroll
    for (int r = 0; r < np; ++r)
#pragma unroll
        for (int kk = 0; kk < nv; ++kk) {
            double pv = P::PATH_V ? rec[R.oPv + r * nv + kk] : 0.0;
            if (P::PATH_T && Dirs<P>::FREE) pv = pv + rec[R.oPt + r] * dtime_of<P>(tau, kk);
            rec[R.oPv + r * nv + kk] = pv;
        }
}

// trapeze: X_{i+1} - (X_i + h/2 (f_i + f_{i+1}))  (trapeze.jl:128-140); needs the record of node i+1
template <class P>
CTD_HD void fin_trapeze_step(const KParams& kp, const BlockCtx& cx, int k) {
    constexpr int n = P::NX, nv = P::NV;
    const Layout& L = kp.L;
    constexpr RecLayout R = RL<P, SC_TRAPEZE, 1>::R;
    const int64_t i = slot_index(kp, cx, k);
    if (i < 0 || i >= L.N || k + 1 >= cx.nslots) return;
    if (slot_index(kp, cx, k + 1) != i + 1) return;
    double* rec = cx.rec + k * R.stride;
    const double* nxt = rec + R.stride;
    const double* base = slot_base(kp, cx, k);
    const double* xnext = slot_next(kp, cx, k);
    const double tau0 = slot_tau(kp, cx, k, 0), tau1 = slot_tau(kp, cx, k, 1);
    const double half_h = 0.5 * (time_of<P>(kp, cx.v, tau1) - time_of<P>(kp, cx.v, tau0));
    const double* e0 = rec + R.oEval;
    const double* e1 = nxt + R.oEval;
#pragma unroll
    for (int r = 0; r < n; ++r) {
        const double fs = e0[R.of + r] + e1[R.of + r];
        rec[R.oR + r] = xnext[r] - (base[r] + half_h * fs);
#pragma unroll
        for (int kk = 0; kk < nv; ++kk) {
            const double dhalf = Dirs<P>::FREE ? 0.5 * (dtime_of<P>(tau1, kk) - dtime_of<P>(tau0, kk)) : 0.0;
            rec[R.oSv + r * nv + kk] = -(dhalf * fs + half_h * (e0[R.oW + r * nv + kk] + e1[R.oW + r * nv + kk]));
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// phase: eval (dual evaluation of the OCP functions)
// ------------------------------------------------------------------------------------------------------
// the control seen by path constraints: U_i, or for stagewise schemes the b-weighted stage average
// (get_OCP_control_at_time_step, src/ode/common.jl:140-155 / irk_stagewise.jl:197-205)
template <class P, int S> CTD_HD void node_control(const KParams& kp, const double* base, double* u) {
    const Layout& L = kp.L;
    if (L.stagewise) {
#pragma unroll
        for (int c = 0; c < P::NU; ++c) u[c] = L.b[0] * base[P::NX + c];
#pragma unroll
        for (int j = 1; j < S; ++j)
#pragma unroll
            for (int c = 0; c < P::NU; ++c) u[c] = u[c] + L.b[j] * base[P::NX + j * P::NU + c];
    } else {
#pragma unroll
        for (int c = 0; c < P::NU; ++c) u[c] = base[P::NX + c];
    }
}

// control of the path constraints of node i held by slot k: node_control, except for implicit Euler where
// u(t_i) = U_{i-1} for i >= 1 (get_OCP_control_at_time_step, euler.jl:59-72): previous block of a tile / extra field of an edge input
template <class P, int S> CTD_HD void path_control(const KParams& kp, const BlockCtx& cx, int k, int64_t i, double* u) {
    const Layout& L = kp.L;
    const double* base = slot_base(kp, cx, k);
    if (L.euler == 2 && i >= 1 && (cx.is_edge || k >= 1)) {
        const double* up = (cx.is_edge && !cx.direct) ? base + L.blk + P::NX + P::NU : slot_prev(kp, cx, k) + P::NX;
#pragma unroll
        for (int c = 0; c < P::NU; ++c) u[c] = up[c];
    } else {
        node_control<P, S>(kp, base, u);
    }
}

// Symbolic first derivatives of the dynamics (ctd_sym.hpp): member of the generated functor of a run-time OCP, explicit
// specialisation generated at build time for a registry problem (ctd_sym_registry.hpp, included at the end of this header)
// (this primary template serves the generated functors of run-time OCPs, which carry DYN_PARTS and, when > 1, dyn_sym_part: the same
// split by rows as the registry's wide problems -- a 12-state run-time OCP then runs one part per wave like the built-in one)
template <class P> struct SymDyn {
    static constexpr bool value = P::HAS_SYM_DYN;
    static constexpr int parts = P::DYN_PARTS;       // lanes the generated code of one evaluation point is split over (by rows)
    CTD_HD static void eval(const double* p, double* ev) { if constexpr (P::HAS_SYM_DYN) P::dyn_sym(p, ev); }
    CTD_HD static void eval_part(int q, const double* p, double* ev) {
        if constexpr (P::HAS_SYM_DYN && P::DYN_PARTS > 1) P::dyn_sym_part(q, p, ev);
        else eval(p, ev);
    }
};
template <class P> struct SymPath {
    static constexpr bool value = P::HAS_SYM_PATH;
    CTD_HD static void eval(const double* p, double* px, double* val) { if constexpr (P::HAS_SYM_PATH) P::path_sym(p, px, val); }
};
template <class P> struct SymLag {       // Lagrange cost: out = [value | l_x | l_u | l_t | l_v]
    static constexpr bool value = P::HAS_SYM_LAG;
    CTD_HD static void eval(const double* p, double* out) { if constexpr (P::HAS_SYM_LAG) P::lag_sym(p, out); }
};
template <class P> struct SymStage;
// symbolic second derivatives of the path point (lane-per-step Hessian kernel, ctd_hess_step.hpp)
template <class P> struct SymPathH { static constexpr bool value = false; CTD_HD static void eval(const double*, double*) {} };

// The OCP dynamics and all their first partials at ONE point (t, x, u, v) into the eval block `ev` (layout RLT::R, offsets
// relative to the block): generated straight-line code when the OCP has it, forward duals (chunk q of the directions) otherwise.
// SPLIT: the caller runs the NCH_DYN lanes of this point in different waves (uniform q per wave), so the generated code may be
// split over them by rows; otherwise the lane of chunk 0 evaluates all of it
template <class P, class RLT, bool SPLIT>
CTD_HD void eval_point(const KParams& kp, const double* vv, int q, double t, const double* xv, const double* uv, double* ev) {
    constexpr int n = P::NX, m = P::NU, nv = P::NV, DC = P::DC;
    using D = Dual<DC>;
    constexpr RecLayout R = RLT::R;
    struct { const double* v; } cx{vv};
    if constexpr (SymDyn<P>::value) {
        // every partial of the point by the generated straight-line code: on the lane of the first direction chunk, or split
        // by rows over the lanes of all chunks when the generator provides the parts
        constexpr bool split = SPLIT && SymDyn<P>::parts > 1 && SymDyn<P>::parts == Dirs<P>::NCH_DYN;
        if (!split && q != 0) return;
        CTD_SUB(kp, 6);
        double prm[1 + n + m + nv];
        prm[0] = t;
#pragma unroll
        for (int c = 0; c < n; ++c) prm[1 + c] = xv[c];
#pragma unroll
        for (int c = 0; c < m; ++c) prm[1 + n + c] = uv[c];
#pragma unroll
        for (int c = 0; c < nv; ++c) prm[1 + n + m + c] = cx.v[c];
#ifdef CTD_ABL_NOEVAL      /* EXPERIMENT build only (profiles/r04_experiments.md): no arithmetic in the evaluation -- the bound of anything a
                              restructured evaluation (row split, ...) could gain.  Outputs are garbage. */
#pragma unroll
        for (int e = 0; e < R.eval_sz; ++e) ev[e] = prm[e % (1 + n + m + nv)];
        return;
#endif
        if constexpr (split) SymDyn<P>::eval_part(q, prm, ev);
        else SymDyn<P>::eval(prm, ev);
        return;
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
#pragma unroll
    for (int d = 0; d < DC; ++d) {
        const int g = g0 + d;
        if (g < n) {
#pragma unroll
            for (int r = 0; r < n; ++r) ev[R.oF + r * R.ldx + g] = out[r].d[d];
        } else if (g < n + m) {
#pragma unroll
            for (int r = 0; r < n; ++r) ev[R.oG + r * R.ldg + (g - n)] = out[r].d[d];
        } else if (P::DYN_T && g == n + m) {
#pragma unroll
            for (int r = 0; r < n; ++r) ev[R.oft + r] = out[r].d[d];
        } else if (P::DYN_V && g < Dirs<P>::DYN) {
            const int kk = g - n - m - (P::DYN_T ? 1 : 0);
#pragma unroll
            for (int r = 0; r < n; ++r) ev[R.oW + r * nv + kk] = out[r].d[d];
        }
    }
    if (q == 0) {
#pragma unroll
        for (int r = 0; r < n; ++r) ev[R.of + r] = out[r].v;
    }
}

// one dynamics evaluation on duals: slot k (step or node i), eval point j, direction chunk q
template <class P, int SC, int S, bool SPLIT = false>
CTD_HD void eval_dynamics(const KParams& kp, const BlockCtx& cx, int k, int j, int q, double* ev) {
    constexpr int n = P::NX, m = P::NU, nv = P::NV;
    const Layout& L = kp.L;
    constexpr RecLayout R = RL<P, SC, S>::R;
    const int64_t i = slot_index(kp, cx, k);
    if (i < 0) return;
    if (SC == SC_TRAPEZE ? (i > L.N) : (i >= L.N)) return;
    const double* base = slot_base(kp, cx, k);
    const double ti = time_of<P>(kp, cx.v, slot_tau(kp, cx, k, 0));
    double xv[n > 0 ? n : 1], uv[m > 0 ? m : 1];
    double t;
    if (SC == SC_TRAPEZE) {                       // f(t_i, X_i, U_i, v): trapeze.jl:60-69
        t = ti;
#pragma unroll
        for (int c = 0; c < n; ++c) xv[c] = base[c];
#pragma unroll
        for (int c = 0; c < m; ++c) uv[c] = base[n + c];
    } else if (SC == SC_MIDPOINT) {               // f(0.5(t_i+t_{i+1}), 0.5(X_i+X_{i+1}), U_i, v): midpoint.jl:53-66
        const double tip1 = time_of<P>(kp, cx.v, slot_tau(kp, cx, k, 1));
        const double* nxt = slot_next(kp, cx, k);
        if (L.euler == 0) {
            t = 0.5 * (ti + tip1);
#pragma unroll
            for (int c = 0; c < n; ++c) xv[c] = 0.5 * (base[c] + nxt[c]);
        } else {                                  // Euler: f(t_i, X_i, U_i, v) or f(t_{i+1}, X_{i+1}, U_i, v): euler.jl:86-102
            t = (L.euler == 1) ? ti : tip1;
#pragma unroll
            for (int c = 0; c < n; ++c) xv[c] = (L.euler == 1) ? base[c] : nxt[c];
        }
#pragma unroll
        for (int c = 0; c < m; ++c) uv[c] = base[n + c];
    } else {                                      // f(t_i + c_j h, X_i + h sum_l a_jl K^l, U_i^j | U_i, v): irk_stagewise.jl:424-446
        const double h = time_of<P>(kp, cx.v, slot_tau(kp, cx, k, 1)) - ti;
        t = ti + butcher_c<S>(L, j) * h;
        const double* K = base + n + L.cu;
#pragma unroll
        for (int c = 0; c < n; ++c) {
            double x = base[c];
#pragma unroll
            for (int l = 0; l < S; ++l) x = x + h * butcher_a<S>(L, j, l) * K[l * n + c];
            xv[c] = x;
        }
        const double* U = base + n + (L.stagewise ? j * m : 0);
#pragma unroll
        for (int c = 0; c < m; ++c) uv[c] = U[c];
    }
    if constexpr (SC == SC_MIDPOINT && S > 1) {
        // control_steps = S > 1 (direct shooting, midpoint.jl:47-72): the dynamics at the SAME (t_s, x_s) once per control U_i^jj of
        // the step.  The step residual and its Jacobian only need the sums over the sub-steps of F, W, f, f_t and the blocks G_jj
        // side by side (n x m S), so this lane walks the sub-steps and adds its share: everything (generated code, chunk 0), its
        // rows (generated code split by rows), or its direction columns (forward duals, chunk q).
        using RL1 = RL<P, SC_MIDPOINT, 1>;
        constexpr RecLayout R1 = RL1::R;
        constexpr int DC = P::DC, NP = Dirs<P>::NCH_DYN;
        constexpr bool sym = SymDyn<P>::value;
        constexpr bool symsplit = sym && SPLIT && SymDyn<P>::parts > 1 && SymDyn<P>::parts == NP;
        constexpr int gT = n + m, gV = n + m + (P::DYN_T ? 1 : 0);
        if (sym && !symsplit && q != 0) return;
        auto dir = [&](int g) { return sym || (g >= q * DC && g < (q + 1) * DC); };
        for (int jj = 0; jj < S; ++jj) {
            double tmp[R1.eval_sz];
#pragma unroll
            for (int e = 0; e < R1.eval_sz; ++e) tmp[e] = 0.0;
#pragma unroll
            for (int c = 0; c < m; ++c) uv[c] = base[n + jj * m + c];
            eval_point<P, RL1, SPLIT>(kp, cx.v, q, t, xv, uv, tmp);
#pragma unroll
            for (int r = 0; r < n; ++r) {
                if (symsplit && (r % NP) != q) continue;
#pragma unroll
                for (int c = 0; c < n; ++c)
                        if (dir(c) && RL1::F(r, c) >= 0)
                        ev[RL<P, SC, S>::F(r, c)] = (jj == 0 ? 0.0 : ev[RL<P, SC, S>::F(r, c)]) + tmp[RL1::F(r, c)];
#pragma unroll
                for (int c = 0; c < m; ++c)
                    if (dir(n + c) && RL1::G(r, c) >= 0) ev[RL<P, SC, S>::G(r, c, jj)] = tmp[RL1::G(r, c)];
                if (P::DYN_T && dir(gT)) ev[R.oft + r] = (jj == 0 ? 0.0 : ev[R.oft + r]) + tmp[R1.oft + r];
                if (P::DYN_V) {
#pragma unroll
                    for (int kk = 0; kk < nv; ++kk)
                        if (dir(gV + kk)) ev[R.oW + r * nv + kk] = (jj == 0 ? 0.0 : ev[R.oW + r * nv + kk]) + tmp[R1.oW + r * nv + kk];
                }
                if (sym || q == 0) ev[R.of + r] = (jj == 0 ? 0.0 : ev[R.of + r]) + tmp[R1.of + r];
            }
        }
        return;
    }
    eval_point<P, RL<P, SC, S>, SPLIT>(kp, cx.v, q, t, xv, uv, ev);
}

// path constraints g(t, x, u, v) on duals into record `rec`: stepPathConstraints!, DOCP_functions.jl:122-140
template <class P, int SC, int S>
CTD_HD void eval_path(const KParams& kp, double* rec, double t, const double* xv, const double* uv, const double* vv, int q,
                      int value_off) {
    constexpr int n = P::NX, m = P::NU, nv = P::NV, np = P::NPATH, DC = P::DC;
    using D = Dual<DC>;
    constexpr RecLayout R = RL<P, SC, S>::R;
    if constexpr (SymPath<P>::value) {
        if (q != 0) return;
        double prm[1 + n + m + nv];
        prm[0] = t;
#pragma unroll
        for (int c = 0; c < n; ++c) prm[1 + c] = xv[c];
#pragma unroll
        for (int c = 0; c < m; ++c) prm[1 + n + c] = uv[c];
#pragma unroll
        for (int c = 0; c < nv; ++c) prm[1 + n + m + c] = vv[c];
        SymPath<P>::eval(prm, rec + R.oPx, rec + value_off);
        return;
    }
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
        if (g < n) {
#pragma unroll
            for (int r = 0; r < np; ++r) rec[R.oPx + r * R.ldx + g] = out[r].d[d];
        } else if (g < n + m) {
#pragma unroll
            for (int r = 0; r < np; ++r) rec[R.oPu + r * R.ldu + (g - n)] = out[r].d[d];
        } else if (P::PATH_T && g == n + m) {
#pragma unroll
            for (int r = 0; r < np; ++r) rec[R.oPt + r] = out[r].d[d];
        } else if (P::PATH_V && g < Dirs<P>::PATH) {
            const int kk = g - n - m - (P::PATH_T ? 1 : 0);
#pragma unroll
            for (int r = 0; r < np; ++r) rec[R.oPv + r * nv + kk] = out[r].d[d];
        }
    }
    if (q == 0) {
#pragma unroll
        for (int r = 0; r < np; ++r) rec[value_off + r] = out[r].v;
    }
}

template <class P, int SC, int S, bool REG = false>
CTD_HD void eval_step_path(const KParams& kp, const BlockCtx& cx, int k, int q) {
    constexpr int n = P::NX, m = P::NU, np = P::NPATH;
    constexpr RecLayout R = RL<P, SC, S>::R;
    constexpr int eqs = RL<P, SC, S>::cb - P::NPATH;      // = L.eqs
    const Layout& L = kp.L;
    const int64_t i = slot_index(kp, cx, k);
    if (i < 0 || i >= L.N) return;
    const double* base = slot_base(kp, cx, k);
    double uv[m > 0 ? m : 1];
    path_control<P, S>(kp, cx, k, i, uv);
    double xv[n > 0 ? n : 1];
#pragma unroll
    for (int c = 0; c < n; ++c) xv[c] = base[c];
    double* rec = cx.rec + k * R.stride;
    const double tau = slot_tau(kp, cx, k, 0);
    if constexpr (REG) {
        // the path block and the path values are composed in registers (a private copy of the record's fields, every index a
        // compile-time constant) and stored once: no read-modify-write through LDS
        double lrec[R.oR + eqs + (np > 0 ? np : 1)];
#pragma unroll
        for (int e = R.oPx; e < R.oR; ++e) lrec[e] = 0.0;
        eval_path<P, SC, S>(kp, lrec, time_of<P>(kp, cx.v, tau), xv, uv, cx.v, q, R.oR + eqs);
        if (Dirs<P>::FUSED) fin_path<P, SC, S>(kp, lrec, tau);
#pragma unroll
        for (int e = R.oPx; e < R.oR; ++e) rec[e] = lrec[e];
#pragma unroll
        for (int r = 0; r < np; ++r) rec[R.oR + eqs + r] = lrec[R.oR + eqs + r];
    } else {
        eval_path<P, SC, S>(kp, rec, time_of<P>(kp, cx.v, tau), xv, uv, cx.v, q, R.oR + L.eqs);
        if (Dirs<P>::FUSED) fin_path<P, SC, S>(kp, rec, tau);
    }
}

// path constraints at the final time (DOCP_functions.jl:100) with the convention u(tf) = U_N unless U_{N+1} exists
template <class P, int SC, int S>
CTD_HD void eval_final_path(const KParams& kp, const BlockCtx& cx, int q) {
    constexpr int n = P::NX, m = P::NU;
    constexpr RecLayout R = RL<P, SC, S>::R;
    const Layout& L = kp.L;
    const double* base = slot_base(kp, cx, kp.edge_slot_last);
    const double* nxt = slot_next(kp, cx, kp.edge_slot_last);
    double xv[n > 0 ? n : 1], uv[m > 0 ? m : 1];
#pragma unroll
    for (int c = 0; c < n; ++c) xv[c] = nxt[c];
    if (SC == SC_TRAPEZE) {
#pragma unroll
        for (int c = 0; c < m; ++c) uv[c] = nxt[n + c];
    } else node_control<P, S>(kp, base, uv);
    double* rec = cx.rec + kp.edge_fp * R.stride;
    const double tau = final_tau(kp, cx);
    eval_path<P, SC, S>(kp, rec, time_of<P>(kp, cx.v, tau), xv, uv, cx.v, q, R.oR);
    if (Dirs<P>::FUSED) fin_path<P, SC, S>(kp, rec, tau);
}

// boundary constraints phi(x0, xf, v) on duals: DOCP_functions.jl:103-111
template <class P, int SC, int S>
CTD_HD void eval_boundary(const KParams& kp, const BlockCtx& cx, int q) {
    constexpr int n = P::NX, nv = P::NV, nb = P::NBC, DC = P::DC;
    using D = Dual<DC>;
    const Layout& L = kp.L;
    constexpr RecLayout R = RL<P, SC, S>::R;
    const double* b0 = slot_base(kp, cx, kp.edge_slot_first);
    const double* bf = slot_next(kp, cx, kp.edge_slot_last);
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
    double* rec = cx.rec + kp.edge_b * R.stride;
#pragma unroll
    for (int d = 0; d < DC; ++d) {
        const int g = g0 + d;
        if (g < n) {
#pragma unroll
            for (int r = 0; r < nb; ++r) rec[R.oB0 + r * R.ldx + g] = out[r].d[d];
        } else if (g < 2 * n) {
#pragma unroll
            for (int r = 0; r < nb; ++r) rec[R.oBf + r * R.ldx + (g - n)] = out[r].d[d];
        } else if (g < 2 * n + nv) {
#pragma unroll
            for (int r = 0; r < nb; ++r) rec[R.oBv + r * nv + (g - 2 * n)] = out[r].d[d];
        }
    }
    if (q == 0) {
#pragma unroll
        for (int r = 0; r < nb; ++r) rec[R.oBval + r] = out[r].v;
    }
}

// REG (direct tiles of small FUSED problems): a lane composes its eval block in registers and stores it to the LDS record
// once, instead of writing the partials to LDS and finishing the chain rule with reads and writes of the same words
template <class P, int SC, int S = 1> struct RegEval {
    static constexpr bool value = DirectTile<P, SC>::value && (P::NX * (P::NX + P::NU * (SC == SC_MIDPOINT ? S : 1) + P::NV + 2) <= 64);
};

// Wide OCPs: the symbolic code of one evaluation point is long (12 states: ~1300 instructions) and only S * ns lanes would run
// it.  It comes split by rows into NCH_DYN parts: part q of every point runs in wave q (uniform code per wave, the parts side by
// side on different SIMDs); the path passes and the lead role (per-step coefficients, state rows: inputs only) take the next
// lanes of the same waves, so the fin phase that follows is ONE pass of one task per lane.
// FOLDED FIN: OCPs whose dynamics come as generated straight-line code (every partial of a point, or of its rows, on ONE lane) and
// whose path rows need one pass finish the chain rule on the evaluating lanes -- stage rows and total d/dv behind the dynamics,
// the path rows' d/dv behind the path pass, the lead role on a lane of its own -- so a tile has no fin phase and one barrier
// less (12-state quadrotor, Gauss-Legendre 3: the fin phase was 1.8 of the tile's 10.9 us; midpoint 1.5 of 11.5).  Edge blocks
// keep the fin phase (final-time path record, one lane per kind of task).
// (Run-time OCPs whose generated dynamics code is LONG -- the swimmer of the reference's problem folder: 213 statements -- do not
// come here with generated code at all: inlined into this kernel it compiled into spills and, on some trees, wrong or faulting code
// whatever launch bound and fin variant was chosen; ctd_jit.cpp routes them to the forward-dual path, profiles/r04_experiments.md.)
template <class P, int SC, int S>
CTD_HD bool fin_folded(const BlockCtx& cx) {
#ifdef CTD_NO_FOLD
    return false;
#else
    constexpr bool path_ok = P::NPATH == 0 || SymPath<P>::value || Dirs<P>::NCH_PATH == 1;
    return !Dirs<P>::FUSED && SymDyn<P>::value && path_ok && !cx.is_edge;
#endif
}

// EDGE BLOCK of an OCP with generated dynamics code (one workgroup of every launch: first / last steps, final-time path rows,
// boundary rows; it is the longest workgroup of the light kernels).  One lane per (stage, slot) runs the generated code -- split
// by rows over one wave per part where the generator provides the parts -- and finishes its rows of the chain rule; the other
// kinds of task (path passes, final-time path passes, coefficient records, boundary passes) sit on lanes 32.. of the waves, the
// heavy kinds on different waves.  (In index order, as the fallback below deals them, the 12-state quadrotor's edge block ran all
// four parts of every point on ONE lane and every kind in the same two waves: 17 - 23 us of evaluation + 10 us of fin.)
// generated dynamics code that comes in at least this many parts (= direction chunks) runs one part per wave.  Four: with three
// (the 8-state quadrotor, 12 directions) the split measured SLOWER in both rounds -- round 2, 7-step tiles: +4 %; round 3, 16-step
// tiles (48 lanes per part-wave): cfg 5' 38.6 -> 40.2 us, optimized 16.9 -> 18.5, trapeze 10.9 -> 11.7 (profiles/r03_experiments.md)
#ifndef CTD_SPLIT_MIN_PARTS
#define CTD_SPLIT_MIN_PARTS 4
#endif
constexpr int kSplitMinParts = CTD_SPLIT_MIN_PARTS;

template <class P, int SC, int S>
CTD_HD bool edge_sym_layout(const BlockCtx& cx, int nthr) {
#ifdef CTD_NO_EDGE_SYM
    return false;
#else
    constexpr int r_path = (P::NPATH > 0) ? Dirs<P>::NCH_PATH : 0;
    constexpr int n_b = (P::NBC > 0) ? Dirs<P>::NCH_BND : 0;
    constexpr bool parts = SymDyn<P>::parts >= kSplitMinParts && SymDyn<P>::parts == Dirs<P>::NCH_DYN;      // then one wave per part
    return SymDyn<P>::value && !Dirs<P>::FUSED && cx.is_edge && cx.nslots <= 8 && nthr >= 256 && r_path <= 4 && n_b <= 32 &&
           StagePoints<SC, S>::value * 8 <= 32 && (!parts || Dirs<P>::NCH_DYN * 64 <= nthr);
#endif
}

template <class P, int SC, int S>
CTD_HD bool split_eval(const BlockCtx& cx, int nthr) {
#ifdef CTD_NO_SPLIT
    return false;
#else
    // (four parts and more: measured +2 % for the 12-state quadrotor, -4 % for the 8-state one with three, profiles/r02_tile_sweeps.log)
    constexpr bool ok = SymDyn<P>::value && SymDyn<P>::parts >= kSplitMinParts && SymDyn<P>::parts == Dirs<P>::NCH_DYN && !Dirs<P>::FUSED;
    constexpr int NP = Dirs<P>::NCH_DYN;
    constexpr int r_path = (P::NPATH > 0) ? Dirs<P>::NCH_PATH : 0;
    // lanes of a wave: S ns dynamics points (the lead tasks -- Gauss-Legendre: (step, state row) pairs dealt round robin over the
    // dynamics lanes of all waves; one-point schemes: one task per step on the last wave -- run BEHIND the dynamics on the same
    // lanes) | ns path points
    return ok && !cx.is_edge && StagePoints<SC, S>::value * cx.nslots + ((P::NPATH > 0) ? cx.nslots : 0) <= 64 && r_path <= NP && NP * 64 <= nthr;
#endif
}

// stores of the emit phase: outputs are written once and not read again by this kernel.  CTD_NT_STORE=1: nontemporal stores
// (they bypass the L2's allocation, so the next evaluation finds x still cached) -- an experiment knob, see DESIGN.md
#ifndef CTD_NT_STORE
#define CTD_NT_STORE 0
#endif
CTD_HD void emit_store(double* p, double v, int wt = 0) {
#if defined(__HIP_DEVICE_COMPILE__)
#if CTD_NT_STORE
    __builtin_nontemporal_store(v, p);
    return;
#endif
    // wt (KParams::wt_store, wave-uniform): WRITE-THROUGH store (sc1) -- the line leaves the XCD's L2 at once instead of staying dirty
    // until the end of the kernel, where the write-back of what is left is serial with the next launch (MI355X_MICROARCH.md: a
    // dependent kernel boundary costs + B / 6 TB/s for B dirty bytes).  Small launches gain (one round of workgroups: their stores
    // are latency-, not throughput-bound), large ones lose (8-byte sc1 stores cost more per byte): the engine decides per handle
    if (wt) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
#endif
    *p = v;
}

// NB > 1 (long periods, e.g. 2904 codes per step for the 12-state quadrotor on Gauss-Legendre 3): more[q] = the code of position
// tid + (q + 1) nthr -- ALL the positions a lane walks, fetched before the evaluation.  Read one position ahead inside the emit
// loop instead, every code load queued behind the workgroups' own stores (microseconds under a full store queue): the emit phase
// of that kernel was bound by those dependent loads, not by bandwidth.
// kpos: the position of the period the lane owns behind the barrier (early emission: an entry of KParams::pos); eb / ek: code and
// position (or row / V entry) of the early output the lane stores when it sits in the lead wave
// have: the fields are filled (a caller that does not prefetch passes a zeroed struct, NOT a null pointer chosen at run time: a
// struct whose address is selected against nullptr lives in scratch memory)
template <int NB> struct EmitPreT { uint32_t b; uint32_t v[kMaxNV]; int64_t eidx; uint32_t more[NB > 1 ? NB - 1 : 1]; int kpos; uint32_t eb; int ek; int have; };
using EmitPre = EmitPreT<1>;     // edge block: b = the code of edge entry `tid`, eidx its index
// Early emission (KParams::pos): the lead wave of a Gauss-Legendre tile stores the outputs that only read what its own lead tasks
// wrote -- lane l owns one early output of every step of the tile -- while the other waves still evaluate the dynamics
template <class P, int SC, int S, int NB>
CTD_HD void early_emit(const KParams& kp, const BlockCtx& cx, int l, const EmitPreT<NB>& pre) {
    constexpr RecLayout R = RL<P, SC, S>::R;
    const int stride = R.stride;
    const int nsteps = (int)(cx.b - cx.a), slot0 = (int)(cx.a - cx.lo);
    const int ne = kp.n_early, nc = kp.c_early, vre = kp.vr_early;
    if (l < ne) {                                   // a position of the step-periodic CSC segment
        if (!kp.vals) return;
        const int64_t ra = cx.a > kp.reg_first ? cx.a : kp.reg_first, rb = cx.b < kp.reg_last ? cx.b : kp.reg_last;
        const int nreg = (int)(rb - ra);
        if (nreg <= 0) return;
        const uint32_t code = pre.eb;
        const int bt = code_beta(code);
        const double beta = bt == 0 ? 0.0 : (bt == 1 ? 1.0 : -1.0);
        const int sl0 = (int)(ra - cx.lo);
        const double* pc = cx.rec + sl0 * stride + R.oC + code_ci(code);
        const double* pd = cx.rec + sl0 * stride + code_di(code);
        double* out = kp.vals + kp.seg_base + (ra - kp.reg_first) * (int64_t)kp.Lseg + pre.ek;
        const int last = nreg - 1;
        for (int s0 = 0; s0 < nreg; s0 += 4) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = s0 + u < last ? s0 + u : last;
                a[u] = pc[s * stride];
                b[u] = pd[s * stride];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (s0 + u < nreg) emit_store(&out[(s0 + u) * kp.Lseg], a[u] * b[u] + beta, kp.wt_store);
        }
    } else if (l < ne + nc) {                       // a state row of c
        if (!kp.c) return;
        const int r = l - ne;
        double* out = kp.c + cx.a * (int64_t)kp.L.cb + r;
        const double* src = cx.rec + slot0 * stride + R.oR + r;
        for (int s = 0; s < nsteps; ++s) emit_store(&out[s * kp.L.cb], src[s * stride], kp.wt_store);
    } else if (l < ne + nc + P::NV * vre) {         // d(state row) / dv of a V column
        if (!kp.vals) return;
        const int e = l - ne - nc, kk = e / vre;
        const uint32_t code = pre.eb;
        const double* pc = cx.rec + slot0 * stride + R.oC + code_ci(code);
        const double* pd = cx.rec + slot0 * stride + code_di(code);
        double* out = kp.vals + kp.vcol_base[kk] + cx.a * (int64_t)kp.vr + pre.ek;
        for (int s = 0; s < nsteps; ++s) emit_store(&out[s * kp.vr], pc[s * stride] * pd[s * stride], kp.wt_store);
    }
}

// first lane of the lead wave of an early-emission tile: behind the dynamics / path lanes of a FULL tile, at a wave boundary
template <class P, int SC, int S> CTD_HD int early_leadbase(const KParams& kp) {
    constexpr int r_dyn = StagePoints<SC, S>::value * Dirs<P>::NCH_DYN, r_path = (P::NPATH > 0) ? Dirs<P>::NCH_PATH : 0;
    int lgT = 0;
    while ((1 << lgT) < kp.T) ++lgT;
    return (((r_dyn + r_path) << lgT) + 63) & ~63;
}

template <class P, int SC, int S, bool REG = false, int NB = 1>
CTD_HD void phase_eval(const KParams& kp, const BlockCtx& cx, int tid, int nthr, const EmitPreT<NB>* epre = nullptr) {
    constexpr bool FUSED = Dirs<P>::FUSED;
    constexpr RecLayout R = RL<P, SC, S>::R;
    const int ns = cx.nslots;
    constexpr int r_dyn = StagePoints<SC, S>::value * Dirs<P>::NCH_DYN;
    constexpr int r_path = (P::NPATH > 0) ? Dirs<P>::NCH_PATH : 0;
    constexpr int r_lead = FUSED ? 1 : 0;
    if (cx.is_edge) {
        // Edge block: a handful of slots, but every KIND of task (dynamics, path, lead, final-time path, boundary).  Lanes of
        // one wave that run different kinds execute them one after the other, so each kind gets waves of its own:
        //   dynamics passes | path passes + final-time path | lead + coefficient records | boundary passes
        // (8 lanes per role; segment starts are rounded up to a wave).  Falls back to one task per lane in index order when the
        // workgroup has too few waves.
        constexpr int n_fp = (P::NPATH > 0) ? Dirs<P>::NCH_PATH : 0;
        constexpr int n_b = (P::NBC > 0) ? Dirs<P>::NCH_BND : 0;
        if constexpr (SymDyn<P>::value && !FUSED) {
            if (edge_sym_layout<P, SC, S>(cx, nthr)) {
                constexpr int NP = Dirs<P>::NCH_DYN;
                constexpr bool parts = SymDyn<P>::parts >= kSplitMinParts && SymDyn<P>::parts == NP;      // one wave per part of the generated code
                constexpr int NPW = parts ? NP : 1;
                const int wave = tid >> 6, l = tid & 63;
                if (wave < NPW && l < StagePoints<SC, S>::value * 8) {
                    const int j = l >> 3, k = l & 7;
                    if (k < ns) {
                        double* ev = cx.rec + k * R.stride + R.oEval + j * R.eval_sz;
                        if constexpr (parts) {
                            eval_dynamics<P, SC, S, true>(kp, cx, k, j, wave, ev);
                            fin_stage<P, SC, S>(kp, cx, k, j, ev, wave, NP);
                        } else {
                            eval_dynamics<P, SC, S>(kp, cx, k, j, 0, ev);
                            fin_stage<P, SC, S>(kp, cx, k, j, ev);
                        }
                    }
                } else if (l >= 32) {
                    // aux kinds on lanes 32.. : path passes | final-time path passes + coefficient records | boundary passes, on the
                    // waves 0, 1, 2 when the dynamics parts fill the first lanes of all waves, else on the waves 1, 2, 3
                    const int a = wave - (parts ? 0 : 1), t = l - 32;
                    if (a == 0) { if (t < r_path * 8 && (t & 7) < ns) eval_step_path<P, SC, S>(kp, cx, t & 7, t >> 3); }
                    else if (a == 1) {
                        if (t < n_fp) eval_final_path<P, SC, S>(kp, cx, t);
                        else if (t >= 8 && t < 10) fill_const_coefs<P>(kp, cx.rec + (t == 8 ? kp.edge_fp : kp.edge_b) * R.stride + R.oC);
                    } else if (a == 2) { if (t < n_b) eval_boundary<P, SC, S>(kp, cx, t); }
                }
                for (int k = tid; k < ns + 2; k += nthr) cx.rec[k * R.stride] = 1.0;
                return;
            }
        }
        constexpr int seg0 = 0;
        constexpr int seg1 = seg0 + ((r_dyn * 8 + 63) & ~63);
        constexpr int seg2 = seg1 + ((r_path * 8 + n_fp + 63) & ~63);
        constexpr int seg3 = seg2 + ((r_lead * 8 + 2 + 63) & ~63);
        constexpr int segE = seg3 + ((n_b + 63) & ~63);
        const bool wide = ns <= 8 && segE <= nthr;
        const int total = wide ? segE : (r_dyn + r_path + r_lead) * ns + n_fp + n_b + 2;
        for (int task = tid; task < total; task += nthr) {
            int kind, k = 0, role = 0;          // kind: 0 dyn, 1 path, 2 lead, 3 final path, 4 boundary, 5 coefficient records
            if (wide) {
                if (task < seg1) { const int l = task - seg0; kind = 0; role = l >> 3; k = l & 7; if (role >= r_dyn) continue; }
                else if (task < seg2) {
                    const int l = task - seg1;
                    if (l < r_path * 8) { kind = 1; role = l >> 3; k = l & 7; }
                    else if (l < r_path * 8 + n_fp) { kind = 3; role = l - r_path * 8; }
                    else continue;
                } else if (task < seg3) {
                    const int l = task - seg2;
                    if (l < r_lead * 8) { kind = 2; k = l & 7; }
                    else if (l < r_lead * 8 + 2) { kind = 5; role = l - r_lead * 8; }
                    else continue;
                } else { const int l = task - seg3; if (l >= n_b) continue; kind = 4; role = l; }
                if (k >= ns) continue;
            } else {
                int t = task;
                if (t < r_dyn * ns) { kind = 0; role = t / ns; k = t - role * ns; }
                else if ((t -= r_dyn * ns) < r_path * ns) { kind = 1; role = t / ns; k = t - role * ns; }
                else if ((t -= r_path * ns) < r_lead * ns) { kind = 2; k = t; }
                else if ((t -= r_lead * ns) < n_fp) { kind = 3; role = t; }
                else if ((t -= n_fp) < n_b) { kind = 4; role = t; }
                else { kind = 5; role = t - n_b; }
            }
            if (kind == 0) {
                const int j = role / Dirs<P>::NCH_DYN, q = role % Dirs<P>::NCH_DYN;
                double* ev = cx.rec + k * R.stride + R.oEval + j * R.eval_sz;
                eval_dynamics<P, SC, S>(kp, cx, k, j, q, ev);
                if (FUSED) fin_stage<P, SC, S>(kp, cx, k, j, ev);
            } else if (kind == 1) eval_step_path<P, SC, S>(kp, cx, k, role);
            else if (kind == 2) fin_lead<P, SC, S>(kp, cx, k);
            else if (kind == 3) eval_final_path<P, SC, S>(kp, cx, role);
            else if (kind == 4) eval_boundary<P, SC, S>(kp, cx, role);
            else fill_const_coefs<P>(kp, cx.rec + (role == 0 ? kp.edge_fp : kp.edge_b) * R.stride + R.oC);
        }
        for (int k = tid; k < ns + 2; k += nthr) cx.rec[k * R.stride] = 1.0;
        return;
    }
    // Tiles: task = (role << lg) | slot with the slot count rounded up to a power of two: decoding is a shift and a mask, and
    // neighbouring lanes run the same role on neighbouring steps.  Roles: S * NCH_DYN dynamics passes, NCH_PATH path
    // passes, one lead role (coefficients + state rows, fused mode).
    CTD_SUB(kp, 0);
    if constexpr (SymDyn<P>::value && SymDyn<P>::parts >= kSplitMinParts && SymDyn<P>::parts == Dirs<P>::NCH_DYN && !FUSED) {
        if (split_eval<P, SC, S>(cx, nthr)) {
            constexpr int NP = Dirs<P>::NCH_DYN;
            const int nd = StagePoints<SC, S>::value * ns;
            const int wave = tid >> 6, l = tid & 63;
            const bool fold = fin_folded<P, SC, S>(cx);
            auto fin_path_slot = [&](int k) {          // total d/dv of the path rows of slot k (phase_fin's path task)
                const int64_t i = slot_index(kp, cx, k);
                if (i >= 0 && i < kp.L.N) fin_path<P, SC, S>(kp, cx.rec + k * R.stride, slot_tau(kp, cx, k, 0));
            };
            if (wave < NP) {
#if !defined(CTD_ABL) || CTD_ABL != 2          /* (ablation builds, never shipped: 1 no path rows, 2 no dynamics, 3 no lead) */
                if (l < nd) {
                    const int j = l / ns, k = l - j * ns;
                    double* ev = cx.rec + k * R.stride + R.oEval + j * R.eval_sz;
                    eval_dynamics<P, SC, S, true>(kp, cx, k, j, wave, ev);
                    if (fold) fin_stage<P, SC, S>(kp, cx, k, j, ev, wave, NP);      // the rows this wave's part evaluated
                }
#endif
                CTD_SUB(kp, 2);
#if !defined(CTD_ABL) || CTD_ABL != 3
                // lead role behind the dynamics, on the same lanes (short tasks that read the inputs only): Gauss-Legendre schemes by
                // (step, state row) over the dynamics lanes of all waves, one-point schemes one task per step on the last wave
                if (SC == SC_IRK) {
                    if (l < nd)
                        for (int t = l * NP + wave; t < ns * P::NX; t += nd * NP) fin_lead<P, SC, S>(kp, cx, t / P::NX, t % P::NX);
                } else if (l < ns && wave == NP - 1) {
                    fin_lead<P, SC, S>(kp, cx, l);
                }
#endif
                CTD_SUB(kp, 3);
#if !defined(CTD_ABL) || CTD_ABL != 1
                if (l >= nd && l < nd + ns) {
                    // symbolic path rows: ONE pass per point (chunk 0), on the wave whose part of the dynamics is the lightest
                    // (part 1 of the 12-state quadrotor: 4100 cycles against 5100; the pass costs 3000); forward duals: chunk q on wave q
                    constexpr int PW = NP > 1 ? 1 : 0;
                    if (SymPath<P>::value ? wave == PW : wave < r_path) {
                        eval_step_path<P, SC, S>(kp, cx, l - nd, SymPath<P>::value ? 0 : wave);
                        if (fold) fin_path_slot(l - nd);
                    }
                }
#endif
                CTD_SUB(kp, 4);
            }
            for (int k = tid; k < ns; k += nthr) cx.rec[k * R.stride] = 1.0;
            CTD_SUB(kp, 5);
            return;
        }
    }
    const int lg = ns <= 1 ? 0 : 32 - __builtin_clz((unsigned)(ns - 1));
    const int mask = (1 << lg) - 1;
    if constexpr (FUSED && SC == SC_IRK) {
        if (kp.n_early > 0 && epre != nullptr) {
            // EARLY EMISSION: the lead tasks sit in a wave of their own (behind the dynamics / path lanes, at a wave boundary); that
            // wave then stores the outputs which only read its records while the other waves still evaluate
            int lgT = 0;                                    // (lanes as for a full tile: the host sized the workgroup with kp.T)
            while ((1 << lgT) < kp.T) ++lgT;
            const int nb = (r_dyn + r_path) << lgT, leadbase = (nb + 63) & ~63;
            const int tmask = (1 << lgT) - 1;
            if (tid < nb) {
                const int k = tid & tmask, role = tid >> lgT;
                if (k < ns) {
                    if (role < r_dyn) {
                        const int j = role / Dirs<P>::NCH_DYN, q = role % Dirs<P>::NCH_DYN;
                        double* ev = cx.rec + k * R.stride + R.oEval + j * R.eval_sz;
                        if constexpr (REG) {
                            double evr[R.eval_sz];
#pragma unroll
                            for (int e = 0; e < R.eval_sz; ++e) evr[e] = 0.0;
                            eval_dynamics<P, SC, S>(kp, cx, k, j, q, evr);
                            fin_stage<P, SC, S>(kp, cx, k, j, evr);
                            const int64_t i = slot_index(kp, cx, k);
                            if (i >= 0 && i < kp.L.N) {
#pragma unroll
                                for (int e = 0; e < R.eval_sz; ++e) ev[e] = evr[e];
                            }
                        } else {
                            eval_dynamics<P, SC, S>(kp, cx, k, j, q, ev);
                            fin_stage<P, SC, S>(kp, cx, k, j, ev);
                        }
                    } else {
                        eval_step_path<P, SC, S, REG>(kp, cx, k, role - r_dyn);
                    }
                }
            } else if ((tid >> 6) == (leadbase >> 6)) {
                const int k = tid - leadbase;
                if (k < ns) {
                    fin_lead<P, SC, S>(kp, cx, k);
                    cx.rec[k * R.stride] = 1.0;
                }
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the wave's LDS writes before its LDS reads below
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                early_emit<P, SC, S, NB>(kp, cx, tid & 63, *epre);
#endif
                // (the serial emulator of tests/emu steps the lanes one after the other: it calls early_emit for the lead wave
                // once every lane has run this phase)
            }
            for (int k = tid; k < ns; k += nthr) cx.rec[k * R.stride] = 1.0;
            return;
        }
    }
    const bool fold_g = fin_folded<P, SC, S>(cx);          // (then with a lead role, as in fused mode)
    const int total = (r_dyn + r_path + ((FUSED || fold_g) ? 1 : 0)) << lg;
    for (int task = tid; task < total; task += nthr) {
        const int k = task & mask, role = task >> lg;
        if (k >= ns) continue;
        if (role < r_dyn) {
            const int j = role / Dirs<P>::NCH_DYN, q = role % Dirs<P>::NCH_DYN;
            double* ev = cx.rec + k * R.stride + R.oEval + j * R.eval_sz;
            if constexpr (REG) {
                double evr[R.eval_sz];
#pragma unroll
                for (int e = 0; e < R.eval_sz; ++e) evr[e] = 0.0;
                CTD_SUB(kp, 1);
                eval_dynamics<P, SC, S>(kp, cx, k, j, q, evr);
                CTD_SUB(kp, 2);
                fin_stage<P, SC, S>(kp, cx, k, j, evr);
                CTD_SUB(kp, 3);
                const int64_t i = slot_index(kp, cx, k);
                if (i >= 0 && i < kp.L.N) {
#pragma unroll
                    for (int e = 0; e < R.eval_sz; ++e) ev[e] = evr[e];
                }
                CTD_SUB(kp, 4);
            } else {
                eval_dynamics<P, SC, S>(kp, cx, k, j, q, ev);
                // (folded fin: the generated code put every partial of the point on the lane of chunk 0)
                if (FUSED || (fold_g && q == 0)) fin_stage<P, SC, S>(kp, cx, k, j, ev);
            }
        } else if (role < r_dyn + r_path) {
            eval_step_path<P, SC, S, REG>(kp, cx, k, role - r_dyn);
            if (fold_g) {
                const int64_t i = slot_index(kp, cx, k);
                if (i >= 0 && i < kp.L.N) fin_path<P, SC, S>(kp, cx.rec + k * R.stride, slot_tau(kp, cx, k, 0));
            }
        } else {
            CTD_SUB(kp, 1);
            fin_lead<P, SC, S>(kp, cx, k);
            CTD_SUB(kp, 4);
        }
    }
    // record header: [0] = 1.0 (every record of the block, used by constant entries of the pattern)
    for (int k = tid; k < ns; k += nthr) cx.rec[k * R.stride] = 1.0;
    CTD_SUB(kp, 5);
}

// ------------------------------------------------------------------------------------------------------
// phase: fin (only when the OCP needs several direction chunks) and fin2 (trapeze)
// ------------------------------------------------------------------------------------------------------
template <class P, int SC, int S>
CTD_HD void phase_fin(const KParams& kp, const BlockCtx& cx, int tid, int nthr) {
    if (Dirs<P>::FUSED) return;
    if (fin_folded<P, SC, S>(cx)) return;                  // (the evaluating lanes did it)
    constexpr RecLayout R = RL<P, SC, S>::R;
    const int ns = cx.nslots;
    constexpr int rows = (SC == SC_IRK && P::NX > 4) ? P::NX : 1;    // rows of a stage per lane: 1 row each for wide states
    // after a split evaluation the lead role is done, and the path rows' total d/dv rides with the first stage task of its step:
    // 12-state quadrotor, 7 steps x 3 stages x 12 rows = 252 tasks: one pass of 256 lanes (266 tasks before: two)
    const bool split = split_eval<P, SC, S>(cx, nthr);
    const bool esl = edge_sym_layout<P, SC, S>(cx, nthr);          // (edge block: the stage rows were finished by the evaluating lanes)
    const int n_stage = esl ? 0 : StagePoints<SC, S>::value * ns * rows, n_lead = split ? 0 : ns, n_path = (P::NPATH > 0 && !split) ? ns : 0;
    const int n_fp = (cx.is_edge && P::NPATH > 0) ? 1 : 0;
    for (int task = tid; task < n_stage + n_lead + n_path + n_fp; task += nthr) {
        int t = task;
        if (t < n_stage) {
            const int k = t % ns, jr = t / ns;
            if (rows > 1) fin_stage_row<P, S>(kp, cx, k, jr / rows, jr % rows);
            else fin_stage<P, SC, S>(kp, cx, k, jr, cx.rec + k * R.stride + R.oEval + jr * R.eval_sz);
            if (split && P::NPATH > 0 && jr == 0) {
                const int64_t i = slot_index(kp, cx, k);
                if (i >= 0 && i < kp.L.N) fin_path<P, SC, S>(kp, cx.rec + k * R.stride, slot_tau(kp, cx, k, 0));
            }
            continue;
        }
        t -= n_stage;
        if (t < n_lead) { fin_lead<P, SC, S>(kp, cx, t); continue; }
        t -= n_lead;
        if (t < n_path) {
            const int64_t i = slot_index(kp, cx, t);
            if (i >= 0 && i < kp.L.N) fin_path<P, SC, S>(kp, cx.rec + t * R.stride, slot_tau(kp, cx, t, 0));
            continue;
        }
        fin_path<P, SC, S>(kp, cx.rec + kp.edge_fp * R.stride, final_tau(kp, cx));
    }
}

template <class P, int SC, int S>
CTD_HD void phase_fin2(const KParams& kp, const BlockCtx& cx, int tid, int nthr) {
    if (SC == SC_TRAPEZE && !cx.is_edge)
        for (int k = tid; k < cx.nslots; k += nthr) fin_trapeze_step<P>(kp, cx, k);
}

// ------------------------------------------------------------------------------------------------------
// phase: emit
// ------------------------------------------------------------------------------------------------------
CTD_HD double eval_code(int oC, const double* rec_c, const double* rec_d, uint32_t code) {
    const double coef = rec_c[oC + code_ci(code)];
    const double data = rec_d[code_di(code)];
    const int bt = code_beta(code);
    const double beta = bt == 0 ? 0.0 : (bt == 1 ? 1.0 : -1.0);
    return coef * data + beta;
}

// The explicit edge entries [0, ntot) are dealt over the kp.has_edge edge workgroups in contiguous shares: a 12-state quadrotor on
// Gauss-Legendre 3 has 6432 of them (boundary rows x (x0, xf, v), final-time path rows, first / last step columns), each a dependent
// chain global load -> LDS reads -> scattered store queued behind the whole chip's stores: ONE workgroup needed ~80 us for them
// (the Hessian kernel has shared its edge list over up to 16 workgroups since round 2, ctd_hess_host.cpp).
// Every edge workgroup evaluates the edge records itself (a few points; they run side by side on different CUs).
CTD_HD int edge_share_begin(const KParams& kp, const BlockCtx& cx, int ntot) {
    const int E = kp.has_edge > 1 ? kp.has_edge : 1, per = (ntot + E - 1) / E;
    const int lo = cx.edge_part * per;
    return lo < ntot ? lo : ntot;
}
CTD_HD int edge_share_end(const KParams& kp, const BlockCtx& cx, int ntot) {
    const int E = kp.has_edge > 1 ? kp.has_edge : 1, per = (ntot + E - 1) / E;
    const int hi = (cx.edge_part + 1) * per;
    return hi < ntot ? hi : ntot;
}

// The codes a lane needs in phase_emit when it owns ONE position of each period (period <= workgroup size): read from the
// global tables before the evaluation starts, so their latency hides behind it and the emission starts from registers.
// upper bound of the CSC period of (OCP, scheme class, stages) -- the reference's dense-block patterns (Appendix A.4 of SURVEY.md)
// -- in units of 256 positions: the codes a lane of a 256-lane workgroup may have to hold
template <class P, int SC, int S> struct EmitN {
    static constexpr int n = P::NX, m = P::NU, nv = P::NV, p = P::NPATH, s = SC == SC_IRK ? S : 0;
    static constexpr int seg = SC == SC_IRK ? n * (2 * n + s * n + nv) + s * n * (n + s * m + s * n + nv) + p * (n + s * m + nv)
                                            : n * (2 * n + 2 * m * (SC == SC_MIDPOINT ? S : 1) + nv) + p * (n + 2 * m + nv);
#ifndef CTD_PRE_MAX
#define CTD_PRE_MAX 1      /* measured on MI355X: holding all codes of a long period in registers is SLOWER (12-state quadrotor, Gauss-Legendre 3: +7 %) */
#endif
    static constexpr int value = (seg + 255) / 256 < 1 ? 1 : ((seg + 255) / 256 > CTD_PRE_MAX ? CTD_PRE_MAX : (seg + 255) / 256);
};
template <class P, int NB = 1>
CTD_HD EmitPreT<NB> emit_prefetch(const KParams& kp, const BlockCtx& cx, int tid, int nthr) {
    EmitPreT<NB> pre;
    pre.b = 0u;
    pre.eidx = 0;
    pre.have = 1;
    pre.kpos = 0; pre.eb = 0u; pre.ek = 0;
#pragma unroll
    for (int q = 0; q < (NB > 1 ? NB - 1 : 1); ++q) pre.more[q] = 0u;
#pragma unroll
    for (int kk = 0; kk < kMaxNV; ++kk) pre.v[kk] = 0u;
    if (cx.is_edge) {
        const int n1 = kp.edge_end - kp.edge_begin, ntot = n1 + (kp.edge2_end - kp.edge2_begin);
        const int w = edge_share_begin(kp, cx, ntot) + tid;          // first entry of this edge workgroup's share that the lane owns
        if (w < edge_share_end(kp, cx, ntot)) {
            const int e = w < n1 ? kp.edge_begin + w : kp.edge2_begin + (w - n1);
            pre.b = kp.edge_code[e];
            pre.eidx = kp.edge_idx[e];
        }
        return pre;
    }
    const int Ls = kp.Lseg;
    pre.kpos = 0; pre.eb = 0u; pre.ek = 0;
    if (kp.n_early > 0) {
        // early emission: the lane's late position (two dependent loads, hidden behind the evaluation) and, for the lead wave, the
        // early output of lane l = tid & 63: [n_early positions | c_early rows of c | nv * vr_early V entries]
        const int q = tid - (int)fast_div((uint32_t)tid, kp.div_late) * kp.n_late;
        pre.kpos = kp.pos[q];
        pre.b = kp.tmpl[pre.kpos];
        const int l = tid & 63;
        if (l < kp.n_early) { pre.ek = kp.pos[kp.n_late + l]; pre.eb = kp.tmpl[pre.ek]; }
        else if (l >= kp.n_early + kp.c_early && l < kp.n_early + kp.c_early + P::NV * kp.vr_early) {
            const int e = l - kp.n_early - kp.c_early, kk = e / kp.vr_early;
            pre.ek = e - kk * kp.vr_early;
            pre.eb = kp.vtmpl[kk * kp.vr + pre.ek];
        }
    } else if (Ls > 0 && Ls <= nthr) {
        const int k = tid - (int)fast_div((uint32_t)tid, kp.div_Lseg) * Ls;
        pre.b = kp.tmpl[k];
    } else if (Ls > nthr) {
        pre.b = kp.tmpl[tid];              // first of the positions tid, tid + nthr, ... this lane owns
        if constexpr (NB > 1) {
#pragma unroll
            for (int q = 0; q < NB - 1; ++q)
                if (tid + (q + 1) * nthr < Ls) pre.more[q] = kp.tmpl[tid + (q + 1) * nthr];
        }
    }
    const int vr = kp.vr;
#pragma unroll
    for (int kk = 0; kk < kMaxNV; ++kk) {
        if (kk < P::NV && vr > 0 && vr <= nthr) {
            const int k = tid - (int)fast_div((uint32_t)tid, kp.div_vr) * vr;
            pre.v[kk] = kp.vtmpl[kk * vr + k];
        }
    }
    return pre;
}

template <class P, int SC, int S, int NB = 1>
CTD_HD void phase_emit_impl(const KParams& kp, const BlockCtx& cx, int tid, int nthr, const EmitPreT<NB> pre_v, const bool hp) {
    const Layout& L = kp.L;
    constexpr RecLayout R = RL<P, SC, S>::R;
    const EmitPreT<NB>* pre_ = &pre_v;                     // hp: the lane's codes were prefetched (pre_v holds them)
    if (cx.is_edge) {
        const int n1 = kp.edge_end - kp.edge_begin, nall = n1 + (kp.edge2_end - kp.edge2_begin);
        const int wlo = edge_share_begin(kp, cx, nall), ntot = edge_share_end(kp, cx, nall);      // this edge workgroup's share
        // four entries per round: their (code, index) loads are in flight together -- one dependent global load per entry, queued
        // behind the whole chip's stores, made the 6432 edge entries of the 12-state quadrotor (Gauss-Legendre 3) a 48 us phase
        for (int w0 = wlo + tid; w0 < ntot; w0 += 4 * nthr) {
            uint32_t code[4];
            int64_t idx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int w = w0 + u * nthr;
                const int e = w < n1 ? kp.edge_begin + w : kp.edge2_begin + (w - n1);
                const bool have = hp && w == wlo + tid;          // first entry: prefetched before the evaluation
                code[u] = have ? pre_->b : (w < ntot ? kp.edge_code[e] : 0u);
                idx[u] = have ? pre_->eidx : (w < ntot ? kp.edge_idx[e] : 0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (w0 + u * nthr >= ntot) break;
                const double val = eval_code(R.oC, cx.rec + code_crec(code[u]) * R.stride, cx.rec + code_drec_raw(code[u]) * R.stride, code[u]);
                if (idx[u] & kEdgeCBit) { if (kp.c) kp.c[idx[u] & ~kEdgeCBit] = val; }
                else if (kp.vals) kp.vals[idx[u]] = val;
            }
        }
        return;
    }
    const int nsteps = (int)(cx.b - cx.a);
    const int slot0 = (int)(cx.a - cx.lo);
    CTD_SUBE(kp, 0);
    const int stride = R.stride;
    // Every output stream of the tile is step-periodic with a small period (cb rows, Lseg CSC entries, vr entries of a
    // V column).  A lane owns ONE position k of the period (its code is decoded once, into registers) and walks the
    // steps; with period <= nthr, floor(nthr / period) steps are in flight at a time and lane -> address is the identity
    // inside each pass (fully coalesced 8-byte stores); with period > nthr a lane owns positions k, k + nthr, ...
    // (A) constraint rows of the tile: c[a*cb .. b*cb)
    if (kp.c) {
        const int cb = L.cb;
        double* out = kp.c + cx.a * (int64_t)cb;
        const int par = (int)fast_div((uint32_t)nthr, kp.div_cb);
        if (par >= 1) {
            if (tid < par * cb) {
                const int g = (int)fast_div((uint32_t)tid, kp.div_cb), r = tid - g * cb;
                const double* src = cx.rec + (slot0 + g) * stride + R.oR + r;
                if (r >= kp.c_early)          // (early emission: the leading rows were stored by the lead wave)
                    for (int s = g; s < nsteps; s += par, src += par * stride) emit_store(&out[s * cb + r], *src, kp.wt_store);
            }
        } else {
            for (int r = tid; r < cb; r += nthr) {
                const double* src = cx.rec + slot0 * stride + R.oR + r;
                for (int s = 0; s < nsteps; ++s, src += stride) emit_store(&out[s * cb + r], *src, kp.wt_store);
            }
        }
    }
    CTD_SUBE(kp, 1);
    if (!kp.vals) return;
    // (B) step-periodic CSC segments of the regular steps of the tile
    {
        const int64_t ra = cx.a > kp.reg_first ? cx.a : kp.reg_first;
        const int64_t rb = cx.b < kp.reg_last ? cx.b : kp.reg_last;
        if (rb > ra) {
            const int Ls = kp.Lseg;
            const int nreg = (int)(rb - ra);
            double* out = kp.vals + kp.seg_base + (ra - kp.reg_first) * (int64_t)Ls;
            const int sl0 = (int)(ra - cx.lo);
            // early emission: only the late positions are left (kp.pos[0 .. n_late)), more steps in flight per pass
            const bool late_only = kp.n_early > 0;
            const int Lw = late_only ? kp.n_late : Ls;          // positions walked here
            const int par = (int)fast_div((uint32_t)nthr, late_only ? kp.div_late : kp.div_Lseg);
            // inner loops: uniform trip count and batches of 4 steps, so the 8 LDS reads of a batch are independent
            // and in flight together (reads past the last step are clamped, only the store is predicated)
            if (par >= 1) {
                if (tid < par * Lw) {
                    const int g = (int)fast_div((uint32_t)tid, late_only ? kp.div_late : kp.div_Lseg);
                    const int k = late_only ? (hp ? pre_->kpos : (int)kp.pos[tid - g * Lw]) : tid - g * Ls;
                    const uint32_t code = hp ? pre_->b : cx.codes[k];
                    const int bt = code_beta(code);
                    const double beta = bt == 0 ? 0.0 : (bt == 1 ? 1.0 : -1.0);
                    const double* pc = cx.rec + (sl0 - code_crec(code)) * stride + R.oC + code_ci(code);
                    const double* pd = cx.rec + (sl0 - code_drec(code)) * stride + code_di(code);
                    const int last = nreg - 1;
                    for (int s0 = g; s0 < nreg; s0 += 4 * par) {
                        double a[4], b[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int s = s0 + u * par < last ? s0 + u * par : last;
                            a[u] = pc[s * stride];
                            b[u] = pd[s * stride];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int s = s0 + u * par;
                            if (s < nreg) emit_store(&out[s * Ls + k], a[u] * b[u] + beta, kp.wt_store);
                        }
                    }
                }
            } else {
                // long periods (more positions than lanes): a lane owns positions tid, tid + nthr, ...; the code of the NEXT
                // position is fetched while the current one is streamed out (the table is read from global memory / L2 here:
                // a dependent load at the top of every position would expose its latency a dozen times per tile)
                uint32_t code = (hp && tid < Ls) ? pre_->b : (tid < Ls ? cx.codes[tid] : 0u);
                int q = 0;
                for (int k = tid; k < Ls; k += nthr, ++q) {
                    uint32_t nxt = 0u;
                    bool have = false;
                    if constexpr (NB > 1) {              // codes fetched before the evaluation (register array: constant indices)
                        if (hp && q < NB - 1) {
                            have = true;
#pragma unroll
                            for (int e = 0; e < NB - 1; ++e)
                                if (e == q) nxt = pre_->more[e];
                        }
                    }
                    if (!have) nxt = k + nthr < Ls ? cx.codes[k + nthr] : 0u;
                    const int bt = code_beta(code);
                    const double beta = bt == 0 ? 0.0 : (bt == 1 ? 1.0 : -1.0);
                    const double* pc = cx.rec + (sl0 - code_crec(code)) * stride + R.oC + code_ci(code);
                    const double* pd = cx.rec + (sl0 - code_drec(code)) * stride + code_di(code);
                    const int last = nreg - 1;
                    for (int s0 = 0; s0 < nreg; s0 += 4) {
                        double a[4], b[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int s = s0 + u < last ? s0 + u : last;
                            a[u] = pc[s * stride];
                            b[u] = pd[s * stride];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (s0 + u < nreg) emit_store(&out[(s0 + u) * Ls + k], a[u] * b[u] + beta, kp.wt_store);
                    }
                    code = nxt;
                }
            }
        }
    }
    CTD_SUBE(kp, 2);
    // (C) the tile's slice of every V column
    if (kp.vr > 0) {
        const int vr = kp.vr;
        const int par = (int)fast_div((uint32_t)nthr, kp.div_vr);
#pragma unroll
        for (int kk = 0; kk < P::NV; ++kk) {
            double* out = kp.vals + kp.vcol_base[kk] + cx.a * (int64_t)vr;
            const uint32_t* codes = cx.vcodes + kk * vr;
            if (par >= 1) {
                if (tid < par * vr) {
                    const int g = (int)fast_div((uint32_t)tid, kp.div_vr), k = tid - g * vr;
                    const uint32_t code = hp ? pre_->v[kk] : codes[k];
                    const double* pc = cx.rec + (slot0 + g) * stride + R.oC + code_ci(code);
                    const double* pd = cx.rec + (slot0 + g) * stride + code_di(code);
                    const int adv = par * stride;
                    if (k >= kp.vr_early)      // (early emission: the leading entries were stored by the lead wave)
                        for (int s = g; s < nsteps; s += par, pc += adv, pd += adv) emit_store(&out[s * vr + k], (*pc) * (*pd), kp.wt_store);
                }
            } else {
                for (int k = tid; k < vr; k += nthr) {
                    const uint32_t code = codes[k];
                    const double* pc = cx.rec + slot0 * stride + R.oC + code_ci(code);
                    const double* pd = cx.rec + slot0 * stride + code_di(code);
                    for (int s = 0; s < nsteps; ++s, pc += stride, pd += stride) emit_store(&out[s * vr + k], (*pc) * (*pd), kp.wt_store);
                }
            }
        }
    }
    CTD_SUBE(kp, 3);
}
// (pointer form: the drivers that always prefetch, and the emulator)
template <class P, int SC, int S, int NB = 1>
CTD_HD void phase_emit(const KParams& kp, const BlockCtx& cx, int tid, int nthr, const EmitPreT<NB>* pre = nullptr) {
    if (pre != nullptr) phase_emit_impl<P, SC, S, NB>(kp, cx, tid, nthr, *pre, pre->have != 0);
    else phase_emit_impl<P, SC, S, NB>(kp, cx, tid, nthr, EmitPreT<NB>{}, false);
}

}  // namespace ctd

// symbolic functions of the registry problems (generated at build time; run-time OCPs carry theirs in the functor)
#if !defined(__HIPCC_RTC__)
#include "ctd_problems.hpp"
#include "ctd_sym_registry.hpp"
#endif
