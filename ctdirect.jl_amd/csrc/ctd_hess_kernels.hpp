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

__device__ __forceinline__ void hess_stamp(const HParams& hp, int slot) {
    if (hp.stamps && threadIdx.x == 0) {
        unsigned long long* p = hp.stamps + ((size_t)blockIdx.x * 5 + slot) * 2;
        p[0] = wall_clock64();
        p[1] = clock64();
    }
}

// at least two waves per SIMD: an instance a few registers over 256 per lane spills them instead of halving its occupancy
#ifdef CTD_HESS_NO_CAP          // ablation
#define CTD_HESS_CAP
#else
#define CTD_HESS_CAP __attribute__((amdgpu_waves_per_eu(2)))
#endif
template <class P, int SC, int S>
__global__ __launch_bounds__(kHessBlock) CTD_HESS_CAP void hess_kernel(const HParams hp, const double* __restrict__ xu,
                                                          const double* __restrict__ y) {
    extern __shared__ double hess_lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    hess_stamp(hp, 0);
    const HBlockCtx cx = make_hctx(hp, blockIdx.x, hess_lds);
    hess_phase_load<P>(hp, cx, xu, y, tid, nthr);
    __syncthreads();
    hess_stamp(hp, 1);
    if (hp.debug_stop == 1) return;
    hess_phase_eval<P, SC, S>(hp, cx, tid, nthr);
    __syncthreads();
    hess_stamp(hp, 2);
    if (hp.debug_stop == 2) return;
    hess_phase_emit<P, SC, S>(hp, cx, blockIdx.x, tid, nthr);
    if (hp.nvv > 0) {
        __syncthreads();
        hess_phase_vvsum(hp, cx, blockIdx.x, tid, nthr);
        if (hp.done_counter) {
            // V x V entries without a second launch: every workgroup publishes its partial (release at device scope), counts
            // itself, and the last one to arrive (acquire) adds all partials in the same fixed order as hess_finish_kernel
            __shared__ int hess_is_last;
            __shared__ double hess_red[kHessBlock];
            __threadfence();
            __syncthreads();
            if (tid == 0) hess_is_last = (atomicAdd(hp.done_counter, 1u) == gridDim.x - 1) ? 1 : 0;
            __syncthreads();
            if (hess_is_last) {
                __threadfence();
                for (int e = 0; e < hp.nvv; ++e) {
                    hess_red[tid] = hess_finish_partial(hp, e, tid, nthr);
                    __syncthreads();
                    for (int off = nthr >> 1; off > 0; off >>= 1) {
                        if (tid < off) hess_red[tid] = hess_red[tid] + hess_red[tid + off];
                        __syncthreads();
                    }
                    if (tid == 0) hp.vals[hp.vv_idx[e]] = hess_red[0];
                    __syncthreads();
                }
                if (tid == 0) *hp.done_counter = 0u;
            }
        }
    }
    hess_stamp(hp, 3);
    if (hp.stamps) {             // diagnostics: time until this workgroup's stores have left the CU
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        hess_stamp(hp, 4);
    }
}

// V x V entries: fixed-order sum of the per-workgroup partials (one workgroup)
template <class P>
__global__ __launch_bounds__(kHessBlock) void hess_finish_kernel(const HParams hp) {
    __shared__ double red[kHessBlock];
    const int tid = threadIdx.x, nthr = blockDim.x;
    for (int e = 0; e < hp.nvv; ++e) {
        red[tid] = hess_finish_partial(hp, e, tid, nthr);
        __syncthreads();
        for (int off = nthr >> 1; off > 0; off >>= 1) {
            if (tid < off) red[tid] = red[tid] + red[tid + off];
            __syncthreads();
        }
        if (tid == 0) hp.vals[hp.vv_idx[e]] = red[0];
        __syncthreads();
    }
}

#if !defined(__HIPCC_RTC__)
template <class P, int SC, int S>
hipError_t launch_hess_variant(const HParams& hp, const double* xu, const double* y, size_t lds_bytes, hipStream_t st,
                               hipEvent_t e0, hipEvent_t e1) {
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)hess_kernel<P, SC, S>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    const int grid = hp.ntiles + 1;
    if (e0 || e1) hipExtLaunchKernelGGL((hess_kernel<P, SC, S>), dim3(grid), dim3(kHessBlock), lds_bytes, st, e0, e1, 0, hp, xu, y);
    else hess_kernel<P, SC, S><<<grid, kHessBlock, lds_bytes, st>>>(hp, xu, y);
    if (hp.nvv > 0 && !hp.done_counter) hess_finish_kernel<P><<<1, kHessBlock, 0, st>>>(hp);
    return hipGetLastError();
}

template <class P>
hipError_t launch_hess(const HParams& hp, const double* xu, const double* y, size_t lds_bytes, hipStream_t st, hipEvent_t e0,
                       hipEvent_t e1) {
    const int sc = hp.L.sc;
    if (sc == SC_TRAPEZE) return launch_hess_variant<P, SC_TRAPEZE, 1>(hp, xu, y, lds_bytes, st, e0, e1);
    if (sc == SC_MIDPOINT) return launch_hess_variant<P, SC_MIDPOINT, 1>(hp, xu, y, lds_bytes, st, e0, e1);
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
