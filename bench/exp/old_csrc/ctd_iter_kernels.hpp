// ctd_iter_kernels.hpp -- one solver iteration in two launches.
//
// An interior-point iteration calls obj, grad!, cons! + jac_coord! and hess_coord! of the ADNLPModel built at
// src/collocation.jl:137-149 at one (x, y).  As separate launches on one stream they serialise: at solver-typical sizes every
// one of them is a single round of workgroups that costs 5-12 us, ~2 us of it launch overhead, and the host needs ~4 us to
// enqueue each (six launches: ~29 us per iteration at 10 000 Goddard steps).  Forking the stream into side streams is slower
// still on this runtime (49 us: each cross-stream event costs more than the kernels it orders, profiles/r02_iteration.md).
// iter_main_kernel is the HORIZONTAL fusion of the four evaluation kernels: one grid whose workgroups take, by index
// range, the body of the Hessian kernel, of the constraint / Jacobian kernel, of the gradient pass and of the objective
// quadrature (the bodies are the ones the single-purpose kernels wrap, unchanged).  All of them are resident together, so the
// launch costs about what its longest body costs.  iter_finish_kernel then adds the few cross-workgroup sums in a fixed
// order (V x V entries of the Hessian; dg/dv and the objective).
#pragma once
#include "ctd_kernels.hpp"
#include "ctd_hess_kernels.hpp"

namespace ctd {

struct IterParams {
    HParams hp;
    KParams kp;
    GradParams gp;
    ObjParams op;
    int32_t nb_h, nb_cj, nb_g, nb_o;     // workgroups per body (0 = body not requested); the Hessian's come first: longest first
    int32_t zero_g;                       // gradient: 1 = Mayer-only cost (zero-fill all of g), 0 = Lagrange (zero the tail only)
    int32_t pad_;
};

constexpr int kIterBlock = kHessBlock;

template <class P, int SC, int S>
__global__ __launch_bounds__(kIterBlock) CTD_HESS_CAP void iter_main_kernel(const IterParams ip, const double* __restrict__ xu,
                                                              const double* __restrict__ y) {
    extern __shared__ double iter_lds[];
    int b = (int)blockIdx.x;
    if (b < ip.nb_h) { hess_body<P, SC, S, false>(ip.hp, xu, y, b, iter_lds); return; }
    b -= ip.nb_h;
    if (b < ip.nb_cj) { cons_jac_body<P, SC, S, false>(ip.kp, xu, b, iter_lds); return; }
    b -= ip.nb_cj;
    if (b < ip.nb_g) {
        const Layout& L = ip.gp.L;
        if (ip.zero_g) {               // Mayer-only cost: the gradient is zero except at x_0, x_f, v (added by the finish kernel)
            for (int64_t i = (int64_t)b * blockDim.x + threadIdx.x; i < L.nvar; i += (int64_t)ip.nb_g * blockDim.x) ip.gp.g[i] = 0.0;
            return;
        }
        // Lagrange cost: the per-step pass writes every entry of the step blocks; this workgroup's share of the tail is zeroed
        for (int64_t i = L.N * (int64_t)L.blk + (int64_t)b * blockDim.x + threadIdx.x; i < L.nvar; i += (int64_t)ip.nb_g * blockDim.x)
            ip.gp.g[i] = 0.0;
        grad_units_body<P, SC, S>(ip.gp, xu, b, reinterpret_cast<double(*)[kMaxNV]>(iter_lds));
        return;
    }
    b -= ip.nb_g;
    if (b < ip.nb_o) obj_partial_body<P, SC>(ip.op, xu, b, ip.nb_o, iter_lds);
}

template <class P>
__global__ __launch_bounds__(kIterBlock) void iter_finish_kernel(const IterParams ip, const double* __restrict__ xu) {
    __shared__ double red[kIterBlock];
    if (blockIdx.x == 0) {
        if (ip.nb_h > 0 && ip.hp.nvv > 0) hess_finish_body(ip.hp, red);
        return;
    }
    if (ip.nb_g > 0) grad_finish_body<P>(ip.gp, xu);
    if (ip.op.out) obj_finish_body<P>(ip.op, xu);
}

#if !defined(__HIPCC_RTC__)
template <class P, int SC, int S>
hipError_t launch_iter_variant(const IterParams& ip, const double* xu, const double* y, size_t lds_bytes, hipStream_t st) {
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)iter_main_kernel<P, SC, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    const int grid = ip.nb_h + ip.nb_cj + ip.nb_g + ip.nb_o;
    if (grid > 0) iter_main_kernel<P, SC, S><<<grid, kIterBlock, lds_bytes, st>>>(ip, xu, y);
    iter_finish_kernel<P><<<2, kIterBlock, 0, st>>>(ip, xu);
    return hipGetLastError();
}
template <class P>
hipError_t launch_iter(const IterParams& ip, const double* xu, const double* y, size_t lds_bytes, hipStream_t st) {
    const int sc = ip.kp.L.sc, s = ip.kp.L.s;
    if (sc == SC_TRAPEZE) return launch_iter_variant<P, SC_TRAPEZE, 1>(ip, xu, y, lds_bytes, st);
    if (sc == SC_MIDPOINT) return launch_iter_variant<P, SC_MIDPOINT, 1>(ip, xu, y, lds_bytes, st);
    if (s == 1) return launch_iter_variant<P, SC_IRK, 1>(ip, xu, y, lds_bytes, st);
    if (s == 2) return launch_iter_variant<P, SC_IRK, 2>(ip, xu, y, lds_bytes, st);
    return launch_iter_variant<P, SC_IRK, 3>(ip, xu, y, lds_bytes, st);
}
#define CTD_INSTANTIATE_ITER(P) template hipError_t launch_iter<P>(const IterParams&, const double*, const double*, size_t, hipStream_t);
#define CTD_EXTERN_ITER(P) extern template hipError_t launch_iter<P>(const IterParams&, const double*, const double*, size_t, hipStream_t);
#endif

}  // namespace ctd
