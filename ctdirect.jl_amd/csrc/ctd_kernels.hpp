// ctd_kernels.hpp -- __global__ wrappers around the phase functions of ctd_kernel_body.hpp, plus the objective kernels.
// Included by the per-problem translation units (ctd_kern_*.hip), which explicitly instantiate launch_* for one OCP
// so the registry compiles in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "ctd_kernel_body.hpp"
#include "ctd_problems.hpp"

namespace ctd {

// Fused constraints + sparse Jacobian values.  One workgroup = one tile of time steps (block 0 = edge block when
// kp.has_edge).  Dynamic LDS = lds_doubles(kp) * 8 bytes.
__device__ __forceinline__ void ctd_stamp(const KParams& kp, int slot) {
    if (kp.stamps && threadIdx.x == 0) {
        unsigned long long* p = kp.stamps + ((size_t)blockIdx.x * 6 + slot) * 2;
        p[0] = wall_clock64();   // constant 100 MHz counter, comparable across workgroups
        p[1] = clock64();        // shader cycles
    }
}

template <class P, int SC, int S>
__global__ void __launch_bounds__(P::MAXB) cons_jac_kernel(const KParams kp, const double* __restrict__ xu) {
    extern __shared__ double ctd_lds[];
    ctd_stamp(kp, 0);
    if (kp.debug_stop == 1) return;
    const BlockCtx cx = make_ctx(kp, (int)blockIdx.x, ctd_lds);
    const int tid = (int)threadIdx.x, nthr = (int)blockDim.x;
    phase_load<P, SC, S>(kp, cx, xu, tid, nthr);
    __syncthreads();
    ctd_stamp(kp, 1);
    if (kp.debug_stop == 2) return;
    phase_eval<P, SC, S>(kp, cx, tid, nthr);
    __syncthreads();
    ctd_stamp(kp, 2);
    if (kp.debug_stop == 3) return;
    if (!Dirs<P>::FUSED) {
        phase_fin<P, SC, S>(kp, cx, tid, nthr);
        __syncthreads();
    }
    if (SC == SC_TRAPEZE) {
        phase_fin2<P, SC, S>(kp, cx, tid, nthr);
        __syncthreads();
    }
    ctd_stamp(kp, 3);
    if (kp.debug_stop == 4) return;
    phase_emit<P, SC, S>(kp, cx, tid, nthr);
    ctd_stamp(kp, 4);
    if (kp.stamps) {             // diagnostics: time until this workgroup's stores have left the CU
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        ctd_stamp(kp, 5);
    }
}

// ---- pipelined driver -------------------------------------------------------------------------------------------
// Co-resident workgroups of the classic driver run in lock-step (all evaluate, then all store), so HBM idles during
// every load/eval phase.  Here a workgroup owns a chunk of consecutive steps and software-pipelines it in sub-tiles:
// in iteration q wave 0 (producer) evaluates sub-tile q+1 into the other record buffer while waves 1.. (consumers)
// prefetch the inputs of sub-tile q+2 and stream sub-tile q out; one workgroup barrier per iteration.  The grid is sized
// to what is resident at once, so the store stream only waits for the prologue.
__device__ __forceinline__ void ctd_wave_sync() {
    // LDS results of this wave's earlier instructions become visible to all its lanes (program order + drained counters)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <class P, int SC, int S>
__device__ __forceinline__ void ctd_produce(const KParams& kp, const BlockCtx& cx, int lane) {
    // the producer's dependent FP64 chain is the pole of every iteration: let it win issue arbitration against the
    // co-resident consumer waves (whose stores are in flight anyway)
    __builtin_amdgcn_s_setprio(3);
    phase_eval<P, SC, S>(kp, cx, lane, 64);
    if (!Dirs<P>::FUSED) {
        ctd_wave_sync();
        phase_fin<P, SC, S>(kp, cx, lane, 64);
    }
    if (SC == SC_TRAPEZE) {
        ctd_wave_sync();
        phase_fin2<P, SC, S>(kp, cx, lane, 64);
    }
    __builtin_amdgcn_s_setprio(0);
}

template <class P, int SC, int S>
__global__ void __launch_bounds__(P::MAXB) cons_jac_pipe_kernel(const KParams kp, const double* __restrict__ xu) {
    extern __shared__ double ctd_lds[];
    const int tid = (int)threadIdx.x, nthr = (int)blockDim.x;
    if (kp.has_edge && blockIdx.x == 0) {          // edge block: same phases as the classic driver
        const BlockCtx cx = make_ctx(kp, 0, ctd_lds);
        phase_load<P, SC, S>(kp, cx, xu, tid, nthr);
        __syncthreads();
        phase_eval<P, SC, S>(kp, cx, tid, nthr);
        __syncthreads();
        if (!Dirs<P>::FUSED) {
            phase_fin<P, SC, S>(kp, cx, tid, nthr);
            __syncthreads();
        }
        phase_emit<P, SC, S>(kp, cx, tid, nthr);
        return;
    }
    const int chunk = (int)blockIdx.x - (kp.has_edge ? 1 : 0);
    const int64_t A = kp.step_begin + (int64_t)chunk * kp.pipe_chunk;
    const int64_t B = A + kp.pipe_chunk < kp.step_end ? A + kp.pipe_chunk : kp.step_end;
    if (A >= B) return;
    const int Q = (int)((B - A + kp.pipe_Ts - 1) / kp.pipe_Ts);
    const int wave = tid >> 6, lane = tid & 63;
    const int ctid = tid - 64, cthr = nthr - 64;       // consumer lane id / count
    ctd_stamp(kp, 0);
    // prologue: inputs of sub-tiles 0 and 1, records of sub-tile 0
    {
        const BlockCtx c0 = make_sub_ctx(kp, ctd_lds, A, B, 0);
        phase_load<P, SC, S, true>(kp, c0, xu, tid, nthr);
        if (Q > 1) {
            const BlockCtx c1 = make_sub_ctx(kp, ctd_lds, A, B, 1);
            phase_load<P, SC, S, false>(kp, c1, xu, tid, nthr);
        }
        __syncthreads();
        ctd_stamp(kp, 1);
        if (wave == 0) ctd_produce<P, SC, S>(kp, c0, lane);
        __syncthreads();
        ctd_stamp(kp, 2);
    }
    for (int q = 0; q < Q; ++q) {
        if (q == 1) ctd_stamp(kp, 3);          // diagnostics: end of iteration 0
        if (q == Q - 1) ctd_stamp(kp, 4);      // start of the last iteration
        if (wave == 0) {
            if (q + 1 < Q) ctd_produce<P, SC, S>(kp, make_sub_ctx(kp, ctd_lds, A, B, q + 1), lane);
        } else {
            // emit first (its stores are fire-and-forget), then fetch the inputs of sub-tile q+2: the load latency then
            // overlaps the producer's evaluation instead of delaying this iteration's stores
            phase_emit<P, SC, S>(kp, make_sub_ctx(kp, ctd_lds, A, B, q), ctid, cthr);
            if (q + 2 < Q) phase_load<P, SC, S, false>(kp, make_sub_ctx(kp, ctd_lds, A, B, q + 2), xu, ctid, cthr);
        }
        __syncthreads();
    }
    ctd_stamp(kp, 5);
}

// ---- objective: Mayer + Lagrange quadrature (src/DOCP_functions.jl:23-54) ------------------------------------
// One lane per quadrature unit (trapeze: node, otherwise: step) of the shard; per-workgroup partial sums are
// reduced with wave shuffles and written to partial[blockIdx]; obj_finish_kernel adds them in index order
// (deterministic) together with the Mayer term.
struct ObjParams {
    Layout L;
    const double* tau;
    int64_t unit_begin, unit_end;   // nodes (trapeze) or steps
    int32_t add_mayer;
    double* partial;
    double* out;
    int32_t nblocks;
};

template <class P> __device__ double obj_time(const ObjParams& op, const double* v, int64_t i) {
    const double t0 = (P::IT0 >= 0) ? v[P::IT0 >= 0 ? P::IT0 : 0] : op.L.t0;
    const double tf = (P::ITF >= 0) ? v[P::ITF >= 0 ? P::ITF : 0] : op.L.tf;
    const double tau = op.tau ? op.tau[i] : (double)i / (double)op.L.N;
    return t0 + tau * (tf - t0);
}

template <class P, int SC>
__device__ double lagrange_unit(const ObjParams& op, const double* __restrict__ xu, const double* v, int64_t i) {
    constexpr int n = P::NX, m = P::NU;
    const Layout& L = op.L;
    const double* base = xu + i * (int64_t)L.blk;
    double x[n > 0 ? n : 1], u[m > 0 ? m : 1];
    if (SC == SC_TRAPEZE) {            // trapeze.jl:78-110: node weights h_1/2, (t_{i+1}-t_{i-1})/2, h_N/2
        double w;
        if (i == 0) w = (obj_time<P>(op, v, 1) - obj_time<P>(op, v, 0)) / 2.0;
        else if (i == L.N) w = (obj_time<P>(op, v, L.N) - obj_time<P>(op, v, L.N - 1)) / 2.0;
        else w = (obj_time<P>(op, v, i + 1) - obj_time<P>(op, v, i - 1)) / 2.0;
        for (int c = 0; c < n; ++c) x[c] = base[c];
        for (int c = 0; c < m; ++c) u[c] = base[n + c];
        return w * P::template lagrange<double>(obj_time<P>(op, v, i), x, u, v);
    }
    const double ti = obj_time<P>(op, v, i), tip1 = obj_time<P>(op, v, i + 1);
    const double h = tip1 - ti;
    if (SC == SC_MIDPOINT) {           // midpoint.jl:87-97
        for (int c = 0; c < n; ++c) x[c] = 0.5 * (base[c] + base[L.blk + c]);
        for (int c = 0; c < m; ++c) u[c] = base[n + c];
        return h * P::template lagrange<double>(0.5 * (ti + tip1), x, u, v);
    }
    // irk.jl:179-228 / irk_stagewise.jl:344-384
    const double* K = base + n + L.cu;
    double local = 0.0;
    for (int j = 0; j < L.s; ++j) {
        for (int c = 0; c < n; ++c) {
            double xc = base[c];
            for (int l = 0; l < L.s; ++l) xc = xc + h * L.a[3 * j + l] * K[l * n + c];
            x[c] = xc;
        }
        const double* U = base + n + (L.stagewise ? j * m : 0);
        for (int c = 0; c < m; ++c) u[c] = U[c];
        const double term = L.b[j] * P::template lagrange<double>(ti + L.c[j] * h, x, u, v);
        local = (j == 0) ? term : local + term;
    }
    return h * local;
}

template <class P, int SC>
__global__ void __launch_bounds__(256) obj_partial_kernel(const ObjParams op, const double* __restrict__ xu) {
    __shared__ double wsum[4];
    double v[P::NV > 0 ? P::NV : 1];
    for (int k = 0; k < P::NV; ++k) v[k] = xu[op.L.v_off + k];
    double acc = 0.0;
    if (P::HAS_LAGRANGE) {
        for (int64_t i = op.unit_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < op.unit_end;
             i += (int64_t)gridDim.x * blockDim.x)
            acc += lagrange_unit<P, SC>(op, xu, v, i);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wsum[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += wsum[w];
        op.partial[blockIdx.x] = s;
    }
}

template <class P>
__global__ void obj_finish_kernel(const ObjParams op, const double* __restrict__ xu) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int b = 0; b < op.nblocks; ++b) s += op.partial[b];
    double mayer = 0.0;
    if (P::HAS_MAYER && op.add_mayer) {
        constexpr int n = P::NX, nv = P::NV;
        double x0[n > 0 ? n : 1], xf[n > 0 ? n : 1], v[nv > 0 ? nv : 1];
        for (int c = 0; c < n; ++c) { x0[c] = xu[c]; xf[c] = xu[op.L.N * (int64_t)op.L.blk + c]; }
        for (int k = 0; k < nv; ++k) v[k] = xu[op.L.v_off + k];
        mayer = P::template mayer<double>(x0, xf, v);
    }
    op.out[0] = mayer + s;
}

// ---- launchers ---------------------------------------------------------------------------------------------------
// Defined here as templates; each per-problem translation unit (ctd_kern_*.hip) explicitly instantiates them for one
// OCP so the registry compiles in parallel, and ctd_engine.hip only sees `extern template` declarations.
// Five kernel variants per OCP: (trapeze), (midpoint), (Gauss-Legendre s = 1, 2, 3); the stage count is a template
// parameter so every loop over stages unrolls.
template <class P, int SC, int S>
hipError_t launch_variant(const KParams& kp, const double* xu, int grid, int block, size_t lds_bytes, hipStream_t st,
                          hipEvent_t e0, hipEvent_t e1) {
    if (kp.pipe_Ts > 0) {
        if (lds_bytes > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)cons_jac_pipe_kernel<P, SC, S>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return e;
        }
        if (e0 || e1) hipExtLaunchKernelGGL((cons_jac_pipe_kernel<P, SC, S>), dim3(grid), dim3(block), lds_bytes, st, e0, e1, 0, kp, xu);
        else cons_jac_pipe_kernel<P, SC, S><<<grid, block, lds_bytes, st>>>(kp, xu);
        return hipGetLastError();
    }
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)cons_jac_kernel<P, SC, S>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    // e0/e1 (optional): events recorded by the dispatch itself right before / after THIS kernel, so
    // hipEventElapsedTime(e0, e1) is the kernel's own duration on the stream it was launched on
    if (e0 || e1) hipExtLaunchKernelGGL((cons_jac_kernel<P, SC, S>), dim3(grid), dim3(block), lds_bytes, st, e0, e1, 0, kp, xu);
    else cons_jac_kernel<P, SC, S><<<grid, block, lds_bytes, st>>>(kp, xu);
    return hipGetLastError();
}

// resident workgroups per CU of the pipelined kernel for this geometry (occupancy API; advisory)
template <class P, int SC, int S>
int pipe_blocks_per_cu(int block, size_t lds_bytes) {
    int nb = 0;
    if (lds_bytes > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)cons_jac_pipe_kernel<P, SC, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, cons_jac_pipe_kernel<P, SC, S>, block, lds_bytes) != hipSuccess) return 0;
    return nb;
}
template <class P>
int pipe_occupancy(int sc, int s, int block, size_t lds_bytes) {
    if (sc == SC_TRAPEZE) return pipe_blocks_per_cu<P, SC_TRAPEZE, 1>(block, lds_bytes);
    if (sc == SC_MIDPOINT) return pipe_blocks_per_cu<P, SC_MIDPOINT, 1>(block, lds_bytes);
    if (s == 1) return pipe_blocks_per_cu<P, SC_IRK, 1>(block, lds_bytes);
    if (s == 2) return pipe_blocks_per_cu<P, SC_IRK, 2>(block, lds_bytes);
    return pipe_blocks_per_cu<P, SC_IRK, 3>(block, lds_bytes);
}

template <class P>
hipError_t launch_cons_jac(int sc, const KParams& kp, const double* xu, int grid, int block, size_t lds_bytes, hipStream_t st,
                           hipEvent_t e0, hipEvent_t e1) {
    if (sc == SC_TRAPEZE) return launch_variant<P, SC_TRAPEZE, 1>(kp, xu, grid, block, lds_bytes, st, e0, e1);
    if (sc == SC_MIDPOINT) return launch_variant<P, SC_MIDPOINT, 1>(kp, xu, grid, block, lds_bytes, st, e0, e1);
    if (kp.L.s == 1) return launch_variant<P, SC_IRK, 1>(kp, xu, grid, block, lds_bytes, st, e0, e1);
    if (kp.L.s == 2) return launch_variant<P, SC_IRK, 2>(kp, xu, grid, block, lds_bytes, st, e0, e1);
    return launch_variant<P, SC_IRK, 3>(kp, xu, grid, block, lds_bytes, st, e0, e1);
}

template <class P>
hipError_t launch_obj(int sc, const ObjParams& op, const double* xu, int grid, int block, hipStream_t st) {
    if (sc == SC_TRAPEZE) obj_partial_kernel<P, SC_TRAPEZE><<<grid, block, 0, st>>>(op, xu);
    else if (sc == SC_MIDPOINT) obj_partial_kernel<P, SC_MIDPOINT><<<grid, block, 0, st>>>(op, xu);
    else obj_partial_kernel<P, SC_IRK><<<grid, block, 0, st>>>(op, xu);
    obj_finish_kernel<P><<<1, 64, 0, st>>>(op, xu);
    return hipGetLastError();
}

#define CTD_INSTANTIATE_LAUNCHERS(P)                                                                                       \
    template hipError_t launch_cons_jac<P>(int, const KParams&, const double*, int, int, size_t, hipStream_t, hipEvent_t, \
                                           hipEvent_t);                                                                    \
    template hipError_t launch_obj<P>(int, const ObjParams&, const double*, int, int, hipStream_t);                       \
    template int pipe_occupancy<P>(int, int, int, size_t);
#define CTD_EXTERN_LAUNCHERS(P)                                                                                            \
    extern template hipError_t launch_cons_jac<P>(int, const KParams&, const double*, int, int, size_t, hipStream_t,      \
                                                  hipEvent_t, hipEvent_t);                                                 \
    extern template hipError_t launch_obj<P>(int, const ObjParams&, const double*, int, int, hipStream_t);               \
    extern template int pipe_occupancy<P>(int, int, int, size_t);

}  // namespace ctd
