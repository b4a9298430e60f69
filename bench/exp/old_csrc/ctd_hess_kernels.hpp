// ctd_hess_kernels.hpp -- __global__ wrappers and launchers of the Hessian-of-the-Lagrangian kernels
// (phases: ctd_hess_body.hpp).  Instantiated per OCP in ctd_hkern_*.hip.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#endif
#include "ctd_hess_body.hpp"
#if !defined(__HIPCC_RTC__)
#include "ctd_problems.hpp"
#endif

namespace ctd {

constexpr int kHessBlock = 256;

template <bool DBG>
__device__ __forceinline__ void hess_stamp(const HParams& hp, int slot) {
    if constexpr (DBG) {
        if (hp.stamps && threadIdx.x == 0) {
            unsigned long long* p = hp.stamps + ((size_t)blockIdx.x * 5 + slot) * 2;
            p[0] = wall_clock64();
            p[1] = clock64();
        }
    }
}

// All kernel arguments the tile path reads, named as inputs of empty asm statements: their scalar loads are issued back to
// back at the top of the kernel (one scalar-cache miss instead of a dozen dependent ones, ~1.5 us before the first load
// of x was issued); the last statement touches one word of every remaining 64-byte line of the argument block so that the
// loads the compiler still places later (register pressure) hit the scalar cache.
__device__ __forceinline__ void hess_pin_kernargs(const HParams& hp, const double* xu, const double* y) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"s"(hp.T), "s"(hp.HL), "s"(hp.HH), "s"(hp.ntiles), "s"(hp.n_edge_blocks), "s"(hp.step_begin), "s"(hp.step_end), "s"(hp.L.blk),
                 "s"(hp.L.cb), "s"(hp.L.N), "s"(hp.L.v_off), "s"(hp.L.n), "s"(hp.L.m), "s"(hp.L.eqs), "s"(hp.L.cu), "s"(hp.tau),
                 "s"(xu), "s"(y), "s"(blockDim.x), "s"(hp.R.stride), "s"(hp.npairs));
    asm volatile("" ::"s"(hp.Lseg), "s"(hp.nc), "s"(hp.compact), "s"(hp.nz), "s"(hp.cpos), "s"(hp.zpos), "s"(hp.tptr), "s"(hp.terms), "s"(hp.nterms), "s"(hp.nvv), "s"(hp.vptr), "s"(hp.vterms),
                 "s"(hp.nvterms), "s"(hp.tasks), "s"(hp.ptasks), "s"(hp.ntask), "s"(hp.nptask), "s"(hp.slot_tasks),
                 "s"(hp.seg_base), "s"(hp.reg_first), "s"(hp.reg_last), "s"(hp.vals));
    const uint32_t* w = reinterpret_cast<const uint32_t*>(&hp);
    constexpr int nline = (int)(sizeof(HParams) / 64);
    static_assert(nline <= 24, "HParams grew: extend the line touch below");
    asm volatile("" ::"s"(w[0]), "s"(w[16 * (1 < nline ? 1 : 0)]), "s"(w[16 * (2 < nline ? 2 : 0)]), "s"(w[16 * (3 < nline ? 3 : 0)]),
                 "s"(w[16 * (4 < nline ? 4 : 0)]), "s"(w[16 * (5 < nline ? 5 : 0)]), "s"(w[16 * (6 < nline ? 6 : 0)]),
                 "s"(w[16 * (7 < nline ? 7 : 0)]), "s"(w[16 * (8 < nline ? 8 : 0)]), "s"(w[16 * (9 < nline ? 9 : 0)]),
                 "s"(w[16 * (10 < nline ? 10 : 0)]), "s"(w[16 * (11 < nline ? 11 : 0)]), "s"(w[16 * (12 < nline ? 12 : 0)]),
                 "s"(w[16 * (13 < nline ? 13 : 0)]), "s"(w[16 * (14 < nline ? 14 : 0)]), "s"(w[16 * (15 < nline ? 15 : 0)]));
#endif
}

// at least two waves per SIMD: an instance a few registers over 256 per lane spills them instead of halving its occupancy
#ifdef CTD_HESS_NO_CAP          // ablation
#define CTD_HESS_CAP
#elif defined(CTD_HESS_WAVES)   // experiment: more waves per SIMD (fewer registers per lane)
#define CTD_HESS_CAP __attribute__((amdgpu_waves_per_eu(CTD_HESS_WAVES)))
#else
#define CTD_HESS_CAP __attribute__((amdgpu_waves_per_eu(2)))
#endif
// DBG = true: diagnostics instantiation (ctd_hess_debug_stamps, env CTD_HESS_STOP); the default one holds no stamp / stop code
template <class P, int SC, int S, bool DBG>
__device__ __forceinline__ void hess_body(const HParams& hp, const double* __restrict__ xu, const double* __restrict__ y, int block,
                                          double* hess_lds) {
    hess_pin_kernargs(hp, xu, y);
    const int tid = threadIdx.x, nthr = blockDim.x;
    hess_stamp<DBG>(hp, 0);
    const HBlockCtx cx = make_hctx(hp, block, hess_lds);
    hess_phase_load<P>(hp, cx, xu, y, tid, nthr);
    __syncthreads();
    hess_stamp<DBG>(hp, 1);
    if (DBG && hp.debug_stop == 1) return;
    hess_phase_eval<P, SC, S>(hp, cx, tid, nthr);
    if constexpr (hess_sums_stages(SC, S)) {
        __syncthreads();
        hess_phase_stage_sum<P, SC, S>(hp, cx, tid, nthr);
    }
    __syncthreads();
    hess_stamp<DBG>(hp, 2);
    if (DBG && hp.debug_stop == 2) return;
    hess_phase_emit<P, SC, S>(hp, cx, block, tid, nthr);
    if (hp.nvv > 0) {
        __syncthreads();
        hess_phase_vvsum(hp, cx, block, tid, nthr);
    }
    hess_stamp<DBG>(hp, 3);
    if (DBG && hp.stamps) {      // diagnostics: time until this workgroup's stores have left the CU
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        hess_stamp<DBG>(hp, 4);
    }
}

template <class P, int SC, int S, bool DBG>
__global__ __launch_bounds__(kHessBlock) CTD_HESS_CAP void hess_kernel(const HParams hp, const double* __restrict__ xu,
                                                          const double* __restrict__ y) {
    extern __shared__ double hess_lds[];
    hess_body<P, SC, S, DBG>(hp, xu, y, (int)blockIdx.x, hess_lds);
}

// V x V entries: fixed-order sum of the per-workgroup partials (one workgroup)
__device__ __forceinline__ void hess_finish_body(const HParams& hp, double* red) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    for (int e = 0; e < hp.nvv; ++e) {
        red[tid] = hess_finish_partial(hp, e, tid, nthr);
        __syncthreads();
        for (int off = nthr >> 1; off > 0; off >>= 1) {
            if (tid < off) red[tid] = red[tid] + red[tid + off];
            __syncthreads();
        }
        if (tid == 0 && hp.vv_idx[e] >= 0) hp.vals[hp.vv_idx[e]] = red[0];
        __syncthreads();
    }
}
template <class P>
__global__ __launch_bounds__(kHessBlock) void hess_finish_kernel(const HParams hp) {
    __shared__ double red[kHessBlock];
    hess_finish_body(hp, red);
}

#if !defined(__HIPCC_RTC__)
template <class P, int SC, int S, bool DBG>
hipError_t launch_hess_variant_dbg(const HParams& hp, const double* xu, const double* y, size_t lds_bytes, hipStream_t st,
                                   hipEvent_t e0, hipEvent_t e1) {
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)hess_kernel<P, SC, S, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    const int grid = hp.ntiles + hp.n_edge_blocks;
    if (e0 || e1) hipExtLaunchKernelGGL((hess_kernel<P, SC, S, DBG>), dim3(grid), dim3(kHessBlock), lds_bytes, st, e0, e1, 0, hp, xu, y);
    else hess_kernel<P, SC, S, DBG><<<grid, kHessBlock, lds_bytes, st>>>(hp, xu, y);
    if (hp.nvv > 0) hess_finish_kernel<P><<<1, kHessBlock, 0, st>>>(hp);
    return hipGetLastError();
}
template <class P, int SC, int S>
hipError_t launch_hess_variant(const HParams& hp, const double* xu, const double* y, size_t lds_bytes, hipStream_t st,
                               hipEvent_t e0, hipEvent_t e1) {
    if (hp.stamps || hp.debug_stop) return launch_hess_variant_dbg<P, SC, S, true>(hp, xu, y, lds_bytes, st, e0, e1);
    return launch_hess_variant_dbg<P, SC, S, false>(hp, xu, y, lds_bytes, st, e0, e1);
}

template <class P>
hipError_t launch_hess(const HParams& hp, const double* xu, const double* y, size_t lds_bytes, hipStream_t st, hipEvent_t e0,
                       hipEvent_t e1) {
    const int sc = hp.L.sc;
    if (sc == SC_TRAPEZE) return launch_hess_variant<P, SC_TRAPEZE, 1>(hp, xu, y, lds_bytes, st, e0, e1);
    if (sc == SC_MIDPOINT) {      // S: controls per step (registry problems 1 - 3, as the constraint / Jacobian kernels)
        if (hp.L.cs == 2) return launch_hess_variant<P, SC_MIDPOINT, 2>(hp, xu, y, lds_bytes, st, e0, e1);
        if (hp.L.cs == 3) return launch_hess_variant<P, SC_MIDPOINT, 3>(hp, xu, y, lds_bytes, st, e0, e1);
        if (hp.L.cs > 3) return hipErrorInvalidValue;
        return launch_hess_variant<P, SC_MIDPOINT, 1>(hp, xu, y, lds_bytes, st, e0, e1);
    }
    if (hp.L.s == 1) return launch_hess_variant<P, SC_IRK, 1>(hp, xu, y, lds_bytes, st, e0, e1);
    if (hp.L.s == 2) return launch_hess_variant<P, SC_IRK, 2>(hp, xu, y, lds_bytes, st, e0, e1);
    return launch_hess_variant<P, SC_IRK, 3>(hp, xu, y, lds_bytes, st, e0, e1);
}

#define CTD_INSTANTIATE_HESS(P) \
    template hipError_t launch_hess<P>(const HParams&, const double*, const double*, size_t, hipStream_t, hipEvent_t, hipEvent_t);
#define CTD_EXTERN_HESS(P) \
    extern template hipError_t launch_hess<P>(const HParams&, const double*, const double*, size_t, hipStream_t, hipEvent_t, hipEvent_t);

#endif  // !__HIPCC_RTC__

}  // namespace ctd
