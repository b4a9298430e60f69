// ctd_hess_step.hpp -- Hessian of the Lagrangian, one LANE per time step (Gauss-Legendre schemes with 2 and 3 stages of the
// registry OCPs: SymAsm<P>, ctd_asm_registry.hpp).
//
// The tile kernel of ctd_hess_body.hpp is bound by instruction issue on partly filled waves: a tile's stage points keep 20-80
// lanes busy, its walk over the step-periodic segment 100-200, and every wave instruction costs four cycles of its SIMD
// whatever the number of active lanes.  Here every lane owns a whole step, in registers: it reads the step's
// variables and multipliers, runs the symbolically differentiated stage function (SymStage<P>::irk, the same code the tile
// kernel runs) for each of the S stage points and the symbolic path-point Hessian (SymPathH<P>) into a register copy R[] of
// the step record, and then the step's ASSEMBLY
// function: straight-line code, generated at build time from the term tables of the host model (ctd_gen_asm.cpp), that forms
// every structurally nonzero entry of the step's segment in CSC order as  sum of (constant) x (1 | h | dh/dv_k) x R[..].
// After every 32 entries the wave (= the workgroup: 64 steps) flushes them: the entries sit transposed in LDS, lane l takes
// position l of the current 64-position window of the segment and walks the 64 steps, so every store instruction writes
// 512 contiguous bytes; positions of the pattern that no entry feeds (structural zeros) are written as 0.0 by the same
// stores -- every position is written exactly once.  The irregular entries (first / last step columns, final state, boundary
// and Mayer terms) stay with the edge path of the tile kernel: the first n_edge_blocks workgroups of the same launch run
// hess_body's edge block (on 64 lanes, next to the step workgroups instead of before them); the V x V contributions are
// summed per workgroup in step order and go through the same partials + finish kernel.
//
// Reference semantics: as ctd_hess_body.hpp (hess_coord! over src/DOCP_functions.jl:23-115, irk_stagewise.jl:344-460).
#pragma once
#include "ctd_hess_kernels.hpp"

namespace ctd {

// primary template: OCPs without generated assembly functions (run-time OCPs, OCPs with path constraints)
template <class P> struct SymAsm {
    static constexpr bool value = false;
    static constexpr int nout_irk2_sw = -1, nout_irk2_cc = -1, nout_irk3_sw = -1, nout_irk3_cc = -1;
    static constexpr short pr_irk2_sw[2] = {0, 0}, pr_irk2_cc[2] = {0, 0}, pr_irk3_sw[2] = {0, 0}, pr_irk3_cc[2] = {0, 0};
    template <class F> CTD_HD static void irk2_sw(const double*, const double*, double, const double*, double*, double*, F&&) {}
    template <class F> CTD_HD static void irk2_cc(const double*, const double*, double, const double*, double*, double*, F&&) {}
    template <class F> CTD_HD static void irk3_sw(const double*, const double*, double, const double*, double*, double*, F&&) {}
    template <class F> CTD_HD static void irk3_cc(const double*, const double*, double, const double*, double*, double*, F&&) {}
};

}  // namespace ctd
#include "ctd_asm_registry.hpp"
namespace ctd {

// The step kernel is taken for light OCPs only: a lane holds the used part of the step record in registers -- up to 8 directions
// per evaluation point (Goddard: 255 registers, no scratch).  The quadrotors (13 / 17 directions) spill (276 B .. 2 KB of scratch
// per lane) and run 1.3-4x slower than the tile kernel (50 vs 40 us and 168 vs 89 us at N = 20 000): they keep the tile kernel.
template <class P> struct StepOK { static constexpr bool value = SymAsm<P>::value && (P::NX + P::NU + P::NV <= 8); };

// (stages, stagewise controls) -> the generated variant
template <class P, int S, bool SW> struct StepFn {
    static constexpr int nout = !SymAsm<P>::value ? -1
                                : (S == 2 ? (SW ? SymAsm<P>::nout_irk2_sw : SymAsm<P>::nout_irk2_cc)
                                          : (S == 3 ? (SW ? SymAsm<P>::nout_irk3_sw : SymAsm<P>::nout_irk3_cc) : -1));
    static const short* pairs() {
        if (S == 2) return SW ? SymAsm<P>::pr_irk2_sw : SymAsm<P>::pr_irk2_cc;
        return SW ? SymAsm<P>::pr_irk3_sw : SymAsm<P>::pr_irk3_cc;
    }
    template <class F> CTD_HD static void run(const double* R, const double* K, double h, const double* dh, double* buf, double* vv, F&& flush) {
        if constexpr (S == 2 && SW) SymAsm<P>::irk2_sw(R, K, h, dh, buf, vv, flush);
        else if constexpr (S == 2) SymAsm<P>::irk2_cc(R, K, h, dh, buf, vv, flush);
        else if constexpr (S == 3 && SW) SymAsm<P>::irk3_sw(R, K, h, dh, buf, vv, flush);
        else if constexpr (S == 3) SymAsm<P>::irk3_cc(R, K, h, dh, buf, vv, flush);
    }
};

constexpr int kStepBlock = 64;        // one wave: the flush needs no workgroup barrier beyond the wave's own ordering
// LDS row of a step: the outputs of one chunk (all of them when the step has at most kSymStepChunk), odd stride
constexpr int step_buf_stride(int nout) { return ((nout < kSymStepChunk ? (nout > 0 ? nout : 1) : kSymStepChunk) | 1); }

struct SParams {
    Layout L;
    const double* tau;              // normalized grid on device (N+1) or nullptr (uniform)
    int64_t step_begin, step_end;   // steps this launch covers (the handle's shard): all of them add to the V x V entries,
    int64_t reg_lo, reg_hi;         // those in [reg_lo, reg_hi) write their segment (the others' entries are edge entries)
    int64_t seg_base, reg_first;    // vals[seg_base + (i - reg_first) * Lseg + e]: entry e of step i's segment
    int32_t Lseg, nout, nchunk, nvv;
    const int32_t* src;             // Lseg: the output (0 .. nout-1) that feeds position e, -1: structural zero of the pattern
    const int32_t* chunk_pos;       // nchunk + 1: flush k covers the positions [chunk_pos[k], chunk_pos[k+1])
    const double* ck;               // kHC x kHC: constant parts of the chain-rule coefficient pairs, ck[c1 * kHC + c2]
    double obj_weight;
    double* vals;
    double* partials;               // V x V partial sums per workgroup: this launch writes rows part_base + blockIdx
    int32_t part_base;
    int32_t wt_store;               // write-through value stores (HParams::wt_store)
};

// The workgroup is ONE wave: its LDS operations execute in program order, so the transposition through LDS needs no workgroup
// barrier -- only the compiler must keep the order and the LDS counter must be drained (__syncthreads() would also wait for
// the wave's outstanding global stores; measured equal on MI355X, kept for the weaker requirement).
__device__ __forceinline__ void step_lds_order() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0) only
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
// two waves per SIMD: 255 registers per lane without scratch (three: 168 registers, 68-420 B of scratch, 42 vs 28 us at cfg 4)
#ifndef CTD_STEP_WAVES
#define CTD_STEP_WAVES 2
#endif
template <class P, int S, bool SW>
__global__ __launch_bounds__(kStepBlock) __attribute__((amdgpu_waves_per_eu(CTD_STEP_WAVES))) void hess_step_kernel(const HParams hp, const SParams sp, const double* __restrict__ xu,
                                                                const double* __restrict__ y) {
    constexpr int n = P::NX, m = P::NU, nv = P::NV;
    constexpr bool FREE = Dirs<P>::FREE;
    constexpr HessRecLayout R_ = HRL<P, SC_IRK, S>::R;
    constexpr SymPrm Q = sym_prm(n, m, nv, P::NPATH);
    constexpr int nvv = nv * (nv + 1) / 2;
    constexpr int kStepBufStride = step_buf_stride(StepFn<P, S, SW>::nout);
    extern __shared__ double lds[];       // max(edge block of the tile kernel, kStepBlock * (kStepBufStride + nvv) doubles)
    if ((int)blockIdx.x < hp.n_edge_blocks) {
        hess_body<P, SC_IRK, S, false>(hp, xu, y, (int)blockIdx.x, lds);
        return;
    }
    const int wg = (int)blockIdx.x - hp.n_edge_blocks;
    const Layout& L = sp.L;
    const int lane = threadIdx.x;
    const int64_t i0 = sp.step_begin + (int64_t)wg * kStepBlock;
    const int nsw = (int)(sp.step_end - i0 < kStepBlock ? sp.step_end - i0 : kStepBlock);      // steps of this wave (>= 1)
    const int64_t i = i0 + (lane < nsw ? lane : nsw - 1);       // (idle lanes repeat the last step: same code, nothing stored)

    // the step's variables, multipliers and times; then one symbolic stage function per stage point into the register record
    double v[nv > 0 ? nv : 1];
#pragma unroll
    for (int k = 0; k < nv; ++k) v[k] = xu[L.v_off + k];
    const double t0 = (P::IT0 >= 0) ? v[P::IT0 >= 0 ? P::IT0 : 0] : L.t0;
    const double tf = (P::ITF >= 0) ? v[P::ITF >= 0 ? P::ITF : 0] : L.tf;
    const double tau0 = sp.tau ? sp.tau[i] : (double)i / (double)L.N;
    const double tau1 = sp.tau ? sp.tau[i + 1] : (double)(i + 1) / (double)L.N;
    const double tA = t0 + tau0 * (tf - t0), tB = t0 + tau1 * (tf - t0), h = tB - tA;
    double d0[nv > 0 ? nv : 1], dh[nv > 0 ? nv : 1];
#pragma unroll
    for (int k = 0; k < nv; ++k) {
        d0[k] = FREE ? dtime_of<P>(tau0, k) : 0.0;
        dh[k] = FREE ? dtime_of<P>(tau1, k) - d0[k] : 0.0;
    }
    const double* base = xu + i * (int64_t)L.blk;
    const double* yr = y + i * (int64_t)L.cb;
    double X[n], Kv[S * n];
#pragma unroll
    for (int r = 0; r < n; ++r) X[r] = base[r];
#pragma unroll
    for (int e = 0; e < S * n; ++e) Kv[e] = base[n + L.cu + e];
    double R[R_.stride];
#pragma unroll
    for (int j = 0; j < S; ++j) {
        double prm[Q.count];
#pragma unroll
        for (int k = 0; k < nv; ++k) {
            prm[Q.HD + k] = dh[k];
            prm[Q.V0 + k] = v[k];
            prm[Q.TD + k] = d0[k] + butcher_c<S>(L, j) * dh[k];
        }
        prm[Q.H0] = h;
        prm[Q.T0] = tA + butcher_c<S>(L, j) * h;
#pragma unroll
        for (int r = 0; r < n; ++r) {
            double kap = 0.0;
#pragma unroll
            for (int l = 0; l < S; ++l) kap = kap + butcher_a<S>(L, j, l) * Kv[l * n + r];
            prm[Q.X0 + r] = X[r] + h * kap;
            prm[Q.KAP + r] = kap;
            prm[Q.W + r] = -yr[n + j * n + r];
        }
#pragma unroll
        for (int b = 0; b < m; ++b) prm[Q.U0 + b] = SW ? base[n + j * m + b] : base[n + b];
        prm[Q.CL] = P::HAS_LAGRANGE ? sp.obj_weight * butcher_b<S>(L, j) : 0.0;
        SymStage<P>::irk(prm, R + R_.oStage + j * R_.stage_sz);
    }
    if (FREE) {
#pragma unroll
        for (int r = 0; r < n; ++r) R[R_.oYX + r] = yr[r];
    }
    if constexpr (P::NPATH > 0) {      // path point of node i: (t_i, X_i, u_i, v), u_i = sum_l b_l U_i^l for stagewise controls
        constexpr SymPathPrm QP = sym_path_prm(n, m, nv, P::NPATH);
        double pp[QP.count];
        pp[QP.T0] = tA;
#pragma unroll
        for (int k = 0; k < nv; ++k) { pp[QP.TD + k] = d0[k]; pp[QP.V0 + k] = v[k]; }
#pragma unroll
        for (int r = 0; r < n; ++r) pp[QP.X0 + r] = X[r];
#pragma unroll
        for (int b = 0; b < m; ++b) {
            double uv;
            if (SW) {
                uv = L.b[0] * base[n + b];
#pragma unroll
                for (int l = 1; l < S; ++l) uv = uv + L.b[l] * base[n + l * m + b];
            } else {
                uv = base[n + b];
            }
            pp[QP.U0 + b] = uv;
        }
#pragma unroll
        for (int r = 0; r < P::NPATH; ++r) pp[QP.WG + r] = yr[L.eqs + r];
        SymPathH<P>::eval(pp, R + R_.oHP);
    }

    double* buf = lds + lane * kStepBufStride;
    // the position table of the segment in LDS (the flushes read it per 64-position window: a global load there would expose
    // its latency once per window and batch of steps)
    int* lsrc = reinterpret_cast<int*>(lds + kStepBlock * kStepBufStride + kStepBlock * (nvv > 0 ? nvv : 1));
    for (int e = lane; e < sp.Lseg; e += kStepBlock) lsrc[e] = sp.src[e];
    double* out0 = sp.vals + sp.seg_base + (i0 - sp.reg_first) * (int64_t)sp.Lseg;
    // steps of this wave that write their segment (the regular ones)
    const int st_lo = (int)(sp.reg_lo > i0 ? (sp.reg_lo - i0 < nsw ? sp.reg_lo - i0 : nsw) : 0);
    const int st_hi = (int)(sp.reg_hi < i0 + nsw ? (sp.reg_hi > i0 ? sp.reg_hi - i0 : 0) : nsw);
    auto flush = [&](int k) {
        step_lds_order();                                  // the chunk's outputs of all 64 steps are in LDS
        // Lane l takes position l of the current 64-position window and walks the steps eight at a time (eight LDS reads in
        // flight, then eight stores).  No per-element predicates: a position without an output reads any valid LDS address and
        // selects 0.0 (exec-mask juggling per element cost 4300 scalar instructions per wave, 29 % of its time --
        // rocprofv3 SQ_INSTS_SALU / SQ_ACTIVE_INST_SCA); only the lanes beyond the chunk's positions are masked, once per window.
        const int e0 = sp.chunk_pos[k], e1 = sp.chunk_pos[k + 1];
        for (int eb = e0; eb < e1; eb += kStepBlock) {
            const int e = eb + lane;
            const bool live = e < e1;
            const int sidx = lsrc[live ? e : e0];
            const bool has = sidx >= 0;
            const double* col = lds + (has ? sidx - k * kSymStepChunk : 0);
            double* o = out0 + e;
            int st = st_lo;
            for (; st + 8 <= st_hi; st += 8) {
                double t[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) t[q] = col[(st + q) * kStepBufStride];
#pragma unroll
                for (int q = 0; q < 8; ++q) t[q] = has ? t[q] : 0.0;
                if (live) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) emit_store(&o[(st + q) * (int64_t)sp.Lseg], t[q], sp.wt_store);
                }
            }
            for (; st < st_hi; ++st) {
                const double t = has ? col[st * kStepBufStride] : 0.0;
                if (live) emit_store(&o[st * (int64_t)sp.Lseg], t, sp.wt_store);
            }
        }
        step_lds_order();                                  // before the next chunk overwrites the buffer
    };
    double vv[nvv > 0 ? nvv : 1];
    // (the constants K[] are read with scalar loads: a copy in LDS measured 33 vs 29 us at cfg 4)
    StepFn<P, S, SW>::run(R, sp.ck, h, dh, buf, vv, flush);
    if (StepFn<P, S, SW>::nout == 0) {                     // no outputs at all: the segment is all zeros
        for (int e = lane; e < sp.Lseg; e += kStepBlock)
            for (int st = st_lo; st < st_hi; ++st) out0[st * (int64_t)sp.Lseg + e] = 0.0;
    }
    if (nvv > 0) {
        double* vb = lds + kStepBlock * kStepBufStride;
#pragma unroll
        for (int e = 0; e < nvv; ++e) vb[lane * nvv + e] = vv[e];
        step_lds_order();
        if (lane < nvv) {
            double acc = 0.0;
            for (int st = 0; st < nsw; ++st) acc = acc + vb[st * nvv + lane];      // step order: fixed summation order
            sp.partials[((int64_t)sp.part_base + wg) * nvv + lane] = acc;
        }
    }
}
#endif

#if !defined(__HIPCC_RTC__)
inline size_t hess_step_lds_bytes(int nout, int nv, int Lseg) {
    return sizeof(double) * kStepBlock * (step_buf_stride(nout) + (nv * (nv + 1) / 2 > 0 ? nv * (nv + 1) / 2 : 1)) + sizeof(int) * (size_t)(Lseg + 2);
}

template <class P, int S, bool SW>
hipError_t launch_hess_step_variant(const HParams& hp, const SParams& sp, const double* xu, const double* y, size_t lds_bytes, hipStream_t st,
                                    hipEvent_t e0, hipEvent_t e1) {
    const int grid = hp.n_edge_blocks + (int)((sp.step_end - sp.step_begin + kStepBlock - 1) / kStepBlock);
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)hess_step_kernel<P, S, SW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    if (e0 || e1) hipExtLaunchKernelGGL((hess_step_kernel<P, S, SW>), dim3(grid), dim3(kStepBlock), lds_bytes, st, e0, e1, 0, hp, sp, xu, y);
    else hess_step_kernel<P, S, SW><<<grid, kStepBlock, lds_bytes, st>>>(hp, sp, xu, y);
    if (hp.nvv > 0) hess_finish_kernel<P><<<1, kHessBlock, 0, st>>>(hp);
    return hipGetLastError();
}
// hp: the tile kernel's parameters with n_edge_blocks edge workgroups of kStepBlock lanes and ntiles = the number of step
// workgroups (the finish kernel adds that many partials)
template <class P>
hipError_t launch_hess_step(const HParams& hp, const SParams& sp, const double* xu, const double* y, size_t lds_bytes, hipStream_t st,
                            hipEvent_t e0, hipEvent_t e1) {
    if constexpr (StepOK<P>::value) {
        const bool sw = sp.L.stagewise != 0;
        if (sp.L.s == 2) return sw ? launch_hess_step_variant<P, 2, true>(hp, sp, xu, y, lds_bytes, st, e0, e1) : launch_hess_step_variant<P, 2, false>(hp, sp, xu, y, lds_bytes, st, e0, e1);
        if (sp.L.s == 3) return sw ? launch_hess_step_variant<P, 3, true>(hp, sp, xu, y, lds_bytes, st, e0, e1) : launch_hess_step_variant<P, 3, false>(hp, sp, xu, y, lds_bytes, st, e0, e1);
    }
    return hipErrorInvalidValue;
}
// (row, column) pairs of the outputs of the variant (host side of the position tables); nullptr / -1: no such variant
template <class P> const short* hess_step_pairs(int s, bool stagewise, int* nout) {
    *nout = -1;
    if constexpr (StepOK<P>::value) {
        if (s == 2) { if (stagewise) { *nout = StepFn<P, 2, true>::nout; return StepFn<P, 2, true>::pairs(); } *nout = StepFn<P, 2, false>::nout; return StepFn<P, 2, false>::pairs(); }
        if (s == 3) { if (stagewise) { *nout = StepFn<P, 3, true>::nout; return StepFn<P, 3, true>::pairs(); } *nout = StepFn<P, 3, false>::nout; return StepFn<P, 3, false>::pairs(); }
    }
    return nullptr;
}
#define CTD_INSTANTIATE_HESS_STEP(P) \
    template hipError_t launch_hess_step<P>(const HParams&, const SParams&, const double*, const double*, size_t, hipStream_t, hipEvent_t, hipEvent_t); \
    template const short* hess_step_pairs<P>(int, bool, int*);
#define CTD_EXTERN_HESS_STEP(P) \
    extern template hipError_t launch_hess_step<P>(const HParams&, const SParams&, const double*, const double*, size_t, hipStream_t, hipEvent_t, hipEvent_t); \
    extern template const short* hess_step_pairs<P>(int, bool, int*);
#endif

}  // namespace ctd
