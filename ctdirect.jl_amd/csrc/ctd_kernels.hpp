// ctd_kernels.hpp -- __global__ wrappers around the phase functions of ctd_kernel_body.hpp, plus the objective kernels.
// Included by the per-problem translation units (ctd_kern_*.hip), which explicitly instantiate launch_* for one OCP
// so the registry compiles in parallel.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <cstring>
#endif
#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#endif

#include "ctd_kernel_body.hpp"
#if !defined(__HIPCC_RTC__)
#include "ctd_problems.hpp"
#endif

namespace ctd {

// Fused constraints + sparse Jacobian values.  One workgroup = one tile of time steps (block 0 = edge block when
// kp.has_edge).  Dynamic LDS = lds_doubles(kp) * 8 bytes.
// DBG = true is the diagnostics instantiation (ctd_debug_stamps, env CTD_DEBUG_STOP): phase stamps and early returns.  The
// default instantiation holds neither, so no launch pays for their branches or their kernel-argument loads.
#ifndef CTD_MULTI_TILE_LOOP
#define CTD_MULTI_TILE_LOOP 0
#endif
constexpr bool kMultiTileLoop = CTD_MULTI_TILE_LOOP != 0;

template <bool DBG>
__device__ __forceinline__ void ctd_stamp(const KParams& kp, int slot) {
    if constexpr (DBG) {
        if (kp.stamps && threadIdx.x == 0) {
            unsigned long long* p = kp.stamps + ((size_t)blockIdx.x * 6 + slot) * 2;
            p[0] = wall_clock64();   // constant 100 MHz counter, comparable across workgroups
            p[1] = clock64();        // shader cycles
        }
    }
}

// The kernel arguments are read with scalar loads where they are first used; behind branches that is one dependent
// scalar-cache miss after the other (nine rounds in the first version of this kernel, ~1 us before the first load of x was
// issued).  Naming every field the tile path needs as an input of one empty asm statement makes the compiler issue all their
// loads back to back at the top of the kernel: one miss latency, then everything sits in scalar registers.
template <int S>
__device__ __forceinline__ void ctd_pin_kernargs(const KParams& kp, const double* xu) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"s"(kp.has_edge), "s"(kp.ntiles), "s"(kp.T), "s"(kp.HL), "s"(kp.HH), "s"(kp.step_begin),
                 "s"(kp.step_end), "s"(kp.L.blk), "s"(kp.L.N), "s"(kp.L.v_off), "s"(kp.L.cu), "s"(kp.L.stagewise), "s"(kp.L.nv),
                 "s"(kp.L.cb), "s"(kp.L.eqs), "s"(kp.L.euler), "s"(kp.tau), "s"(kp.L.t0), "s"(kp.L.tf), "s"(xu), "s"(blockDim.x));
    asm volatile("" ::"s"(kp.tmpl), "s"(kp.vtmpl), "s"(kp.Lseg), "s"(kp.vr), "s"(kp.div_cb.M), "s"(kp.div_Lseg.M), "s"(kp.div_vr.M),
                 "s"(kp.div_cb.d), "s"(kp.div_Lseg.d), "s"(kp.div_vr.d), "s"(kp.seg_base), "s"(kp.reg_first), "s"(kp.reg_last),
                 "s"(kp.vcol_base[0]), "s"(kp.c), "s"(kp.vals), "s"(kp.halo), "s"(kp.pos), "s"(kp.n_early), "s"(kp.n_late));
    if (S >= 1) asm volatile("" ::"s"(kp.L.a[0]), "s"(kp.L.b[0]), "s"(kp.L.c[0]));
    if (S >= 2) asm volatile("" ::"s"(kp.L.a[1]), "s"(kp.L.a[3]), "s"(kp.L.a[4]), "s"(kp.L.b[1]), "s"(kp.L.c[1]));
    if (S >= 3) asm volatile("" ::"s"(kp.L.a[2]), "s"(kp.L.a[5]), "s"(kp.L.a[6]), "s"(kp.L.a[7]), "s"(kp.L.a[8]), "s"(kp.L.b[2]), "s"(kp.L.c[2]));
#endif
}

// The same for the emit phase.  The evaluation needs every scalar register it can get, so the values pinned at the top of the
// kernel do not survive it: the compiler re-loads each kernel argument where the emit phase first uses it -- behind a branch
// each, i.e. one scalar-cache round trip after the other again (a dozen `s_load; s_waitcnt lgkmcnt(0)` pairs on the critical path
// of a 1.4 us phase).  Named once more in front of the barrier, they arrive together while the waves wait for each other.
__device__ __forceinline__ void ctd_pin_emit_kernargs(const KParams& kp) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"s"(kp.c), "s"(kp.vals), "s"(kp.L.cb), "s"(kp.Lseg), "s"(kp.vr), "s"(kp.div_cb.M), "s"(kp.div_Lseg.M), "s"(kp.div_vr.M),
                 "s"(kp.div_cb.d), "s"(kp.div_Lseg.d), "s"(kp.div_vr.d), "s"(kp.seg_base), "s"(kp.reg_first), "s"(kp.reg_last),
                 "s"(kp.vcol_base[0]), "s"(kp.n_early), "s"(kp.c_early), "s"(kp.vr_early), "s"(kp.n_late), "s"(kp.div_late.M), "s"(kp.div_late.d));
#endif
}

// workgroup `block` of the evaluation (the kernel below; also a branch of the fused iteration kernel, ctd_iter_kernels.hpp)
template <class P, int SC, int S, bool DBG>
__device__ __forceinline__ void cons_jac_body(const KParams& kp, const double* __restrict__ xu, int block, double* ctd_lds) {
    ctd_pin_kernargs<SC == SC_IRK ? S : 0>(kp, xu);
    ctd_stamp<DBG>(kp, 0);
    if (DBG && kp.debug_stop == 1) return;
    const int tid = (int)threadIdx.x, nthr = (int)blockDim.x;
    if constexpr (DirectTile<P, SC>::value) {
        // direct driver: no staging of xu, one barrier (see make_direct_ctx); the staged phases below are not instantiated
        const BlockCtx cx = make_direct_ctx(kp, block, ctd_lds, xu);
        const EmitPre pre = emit_prefetch<P>(kp, cx, tid, nthr);
        ctd_stamp<DBG>(kp, 1);
        phase_eval<P, SC, S, RegEval<P, SC, S>::value, 1>(kp, cx, tid, nthr, &pre);
        ctd_pin_emit_kernargs(kp);
        __syncthreads();
        ctd_stamp<DBG>(kp, 2);
        ctd_stamp<DBG>(kp, 3);
        if (DBG && kp.debug_stop >= 2 && kp.debug_stop <= 4) return;
        phase_emit<P, SC, S>(kp, cx, tid, nthr, &pre);
        ctd_stamp<DBG>(kp, 4);
        if (DBG && kp.stamps) {
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            ctd_stamp<DBG>(kp, 5);
        }
        return;
    }
    BlockCtx cx = make_ctx(kp, block, ctd_lds);
    // codes not staged in LDS (long periods): every code the lane will need is fetched now, the latency hidden behind load + eval
    constexpr int NB = EmitN<P, SC, S>::value;
    EmitPreT<NB> pre = {};
    const bool use_pre = !cx.is_edge && !codes_staged(kp);
    if (use_pre) pre = emit_prefetch<P, NB>(kp, cx, tid, nthr);
    phase_load<P, SC, S>(kp, cx, xu, tid, nthr);
    // multi-tile workgroups (kp.wg_stride > 0; EXPERIMENT, compiled with -DCTD_MULTI_TILE_LOOP=1 only): this workgroup goes on with
    // block + wg_stride, ... -- the templates, v and the lane's codes stay where they are (the period of the tables is the step:
    // the same for every tile), the x slice of the next tile travels while this one is emitted.  One barrier per tile boundary: it
    // orders the emission's reads of the records and the staged copy of the next slice before the next evaluation.
    // Measured on MI355X (profiles/r03_experiments.md): SLOWER than one tile per workgroup -- 12-state quadrotor GL3 124 us against
    // 106, optimized pattern 55.5 against 42.4.  A resident round of workgroups that all start together stays in lock-step (every
    // workgroup evaluates, then every workgroup stores: the memory system idles, then saturates), while the hardware dispatcher
    // starts the next tile whenever a slot frees and so spreads the phases; the loop also costs 30 - 55 registers.
    const int nblk = kp.ntiles + kp.has_edge;
    for (;;) {
        __syncthreads();
        ctd_stamp<DBG>(kp, 1);
        if (DBG && kp.debug_stop == 2) return;
        phase_eval<P, SC, S>(kp, cx, tid, nthr);
        ctd_pin_emit_kernargs(kp);
        __syncthreads();
        ctd_stamp<DBG>(kp, 2);
        if (DBG && kp.debug_stop == 3) return;
        if (!Dirs<P>::FUSED && !fin_folded<P, SC, S>(cx)) {     // (block-uniform)
            phase_fin<P, SC, S>(kp, cx, tid, nthr);
            __syncthreads();
        }
        if (SC == SC_TRAPEZE) {
            phase_fin2<P, SC, S>(kp, cx, tid, nthr);
            __syncthreads();
        }
        ctd_stamp<DBG>(kp, 3);
        if (DBG && kp.debug_stop == 4) return;
        block += kp.wg_stride;
        const bool more = kMultiTileLoop && kp.wg_stride > 0 && !cx.is_edge && block < nblk;
        BlockCtx nx = cx;
        TileIn tin{0.0, 0.0, 0.0};
        if (more) {
            nx = make_ctx(kp, block, ctd_lds);
            tin = load_issue<P>(kp, nx, xu, tid, nthr);
        }
        phase_emit_impl<P, SC, S, NB>(kp, cx, tid, nthr, pre, use_pre);
        ctd_stamp<DBG>(kp, 4);
        if (!more) break;
        load_commit<P, SC, S>(kp, nx, xu, tin, tid, nthr);
        cx = nx;
    }
    if (DBG && kp.stamps) {      // diagnostics: time until this workgroup's stores have left the CU
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        ctd_stamp<DBG>(kp, 5);
    }
}

// Waves per SIMD the kernel is compiled for (the second launch bound caps the registers per lane at 512 / value).  Four: the
// 12-state quadrotor's Gauss-Legendre 3 instantiation needed 129 registers -- ONE more than four waves per SIMD allow -- and ran
// three workgroups per CU where its 15 KiB of LDS would let ten in.  The midpoint kernels with several controls per step
// (195 - 245 registers) and OCPs wider than the registry's keep the compiler's own choice.
template <class P, int SC, int S> struct MinWaves {
    static constexpr int value = (P::NX <= 12 && !(SC == SC_MIDPOINT && S > 1)) ? 4 : 1;
};

template <class P, int SC, int S, bool DBG>
__global__ void __launch_bounds__(P::MAXB, (MinWaves<P, SC, S>::value)) cons_jac_kernel(const KParams kp, const double* __restrict__ xu) {
    extern __shared__ double ctd_lds[];
    cons_jac_body<P, SC, S, DBG>(kp, xu, (int)blockIdx.x, ctd_lds);
}

// EXPERIMENT build only (make EXTRA=-DCTD_KP_INDIRECT; VERDICT r03 item 7 ii): the 640-byte parameter block lives in device memory
// and the kernel takes a 16-byte argument list (pointer to it, xu).  Measured on MI355X (profiles/r04_experiments.md): no gain --
// back-to-back launches of this kernel are GPU-bound (5.9 us per kernel against 3.3 us of host time per launch), and the scalar
// loads of the parameters now hang off one more dependent load.
#ifdef CTD_KP_INDIRECT
template <class P, int SC, int S, bool DBG>
__global__ void __launch_bounds__(P::MAXB, (MinWaves<P, SC, S>::value)) cons_jac_kernel_ind(const KParams* __restrict__ kpp, const double* __restrict__ xu) {
    extern __shared__ double ctd_lds[];
    cons_jac_body<P, SC, S, DBG>(*kpp, xu, (int)blockIdx.x, ctd_lds);
}
#endif

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
    // sharded iterate read in place (ctd_set_x_shards): the next shard's first node (midpoint / Euler units of a shard's last step)
    // and X_1 / X_{N+1} of the Mayer term come from the owners' buffers; null: xu holds everything
    const XHalo* halo;
    XNear near;             // (the same buffers in the kernel arguments: one memory round trip per remote entry)
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
    // X_{i+1} sits behind the step's block -- in the next shard's buffer for the last step of a shard of a sharded iterate
    const int64_t gn = (i + 1) * (int64_t)L.blk;
    const double* nxt = ((op.halo && i + 1 >= op.unit_end) ? xnear(op.near, xu, gn) : xu) + gn;      // (only the shard's last unit looks at the table)
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
    if (SC == SC_MIDPOINT) {           // midpoint.jl:87-97; Euler (euler.jl:112-134): (t_i, X_i, U_i) or (t_{i+1}, X_{i+1}, U_i)
        if (L.cs > 1) {                // control_steps > 1 (midpoint.jl:99-116): h_i = h / cs, l(t_i + (j - 1/2) h_i, x_s, U_i^j, v), j = 1..cs
            const double hi = h / (double)L.cs;
            for (int c = 0; c < n; ++c) x[c] = 0.5 * (base[c] + nxt[c]);
            double val = 0.0;
            for (int j = 1; j <= L.cs; ++j) {
                for (int c = 0; c < m; ++c) u[c] = base[n + (j - 1) * m + c];
                const double term = hi * P::template lagrange<double>(ti + ((double)j - 0.5) * hi, x, u, v);
                val = (j == 1) ? term : val + term;
            }
            return val;
        }
        for (int c = 0; c < m; ++c) u[c] = base[n + c];
        if (L.euler == 0) {
            for (int c = 0; c < n; ++c) x[c] = 0.5 * (base[c] + nxt[c]);
            return h * P::template lagrange<double>(0.5 * (ti + tip1), x, u, v);
        }
        for (int c = 0; c < n; ++c) x[c] = (L.euler == 1) ? base[c] : nxt[c];
        return h * P::template lagrange<double>(L.euler == 1 ? ti : tip1, x, u, v);
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

// body of the quadrature pass for workgroup `block` of `nblocks` (wsum: 4+ doubles of LDS); also a branch of the fused
// iteration kernel (iter_main_kernel)
template <class P, int SC>
__device__ __forceinline__ void obj_partial_body(const ObjParams& op, const double* __restrict__ xu, int block, int nblocks, double* wsum) {
    double v[P::NV > 0 ? P::NV : 1];
    for (int k = 0; k < P::NV; ++k) v[k] = xu[op.L.v_off + k];
    double acc = 0.0;
    if (P::HAS_LAGRANGE) {
        for (int64_t i = op.unit_begin + (int64_t)block * blockDim.x + threadIdx.x; i < op.unit_end;
             i += (int64_t)nblocks * blockDim.x)
            acc += lagrange_unit<P, SC>(op, xu, v, i);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wsum[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += wsum[w];
        op.partial[block] = s;
    }
}
template <class P, int SC>
__global__ void __launch_bounds__(256) obj_partial_kernel(const ObjParams op, const double* __restrict__ xu) {
    __shared__ double wsum[4];
    obj_partial_body<P, SC>(op, xu, (int)blockIdx.x, (int)gridDim.x, wsum);
}

template <class P>
__device__ __forceinline__ void obj_finish_body(const ObjParams& op, const double* __restrict__ xu) {
    if (threadIdx.x >= 64) return;
    // one wave: lane l adds the partials l, l + 64, ... in index order, then a fixed shuffle tree (deterministic; the
    // loads of the 64 lanes are in flight together instead of one dependent chain of nblocks loads)
    double s = 0.0;
    for (int b = threadIdx.x; b < op.nblocks; b += 64) s += op.partial[b];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (threadIdx.x != 0) return;
    double mayer = 0.0;
    if (P::HAS_MAYER && op.add_mayer) {
        constexpr int n = P::NX, nv = P::NV;
        double x0[n > 0 ? n : 1], xf[n > 0 ? n : 1], v[nv > 0 ? nv : 1];
        const int64_t gf = op.L.N * (int64_t)op.L.blk;
        const double* xa = op.halo ? xnear(op.near, xu, 0) : xu;
        const double* xb = op.halo ? xnear(op.near, xu, gf) : xu;
        for (int c = 0; c < n; ++c) { x0[c] = xa[c]; xf[c] = xb[gf + c]; }
        for (int k = 0; k < nv; ++k) v[k] = xu[op.L.v_off + k];
        mayer = P::template mayer<double>(x0, xf, v);
    }
    op.out[0] = mayer + s;
}
template <class P>
__global__ void obj_finish_kernel(const ObjParams op, const double* __restrict__ xu) {
    if (blockIdx.x != 0) return;
    obj_finish_body<P>(op, xu);
}

// ---- objective gradient: grad!(nlp, x, g) -----------------------------------------------------------------------------
// In the reference the gradient is ReverseDiff over __objective (gradient_backend = ReverseDiffADGradient,
// src/collocation.jl:127).  Here: forward duals over the Lagrange / Mayer functions + the quadrature's chain rule.
// Owner-computes, no atomics: one lane per quadrature unit writes the gradient entries of its own variables (trapeze:
// node; midpoint: node, gathering the two adjacent steps; Gauss-Legendre: step); the partials with respect to v are
// reduced per workgroup and summed in index order by grad_finish_kernel, which also adds the Mayer term.
struct GradParams {
    Layout L;
    const double* tau;
    double* g;          // nvar, zero-filled before the launch
    double* partial;    // nblocks * kMaxNV
    int32_t nblocks;
    // quadrature units [unit_begin, unit_end) this launch evaluates: all of them (ctd_grad*: the whole objective's gradient), or the
    // units of a shard of the grid (ctd_grad_shard_dev_async: the shard's own entries of g + its partial sums of d/dv)
    int64_t unit_begin, unit_end;
    int32_t owns_first, owns_last;      // the Mayer term's d/dx0 goes to the shard that owns X_1, d/dxf and d/dv to the one that owns X_{N+1}
    // sharded iterate read in place (ctd_set_x_shards): blocks of other shards come from their owners' buffers; null: xu holds everything
    const XHalo* halo;
    XNear near;
};
// start of the variable block of step / node `st` in the buffer that owns it
CTD_HD const double* grad_block(const GradParams& gp, const double* xu, int64_t st) {
    const int64_t g = st * (int64_t)gp.L.blk;
    return (gp.halo ? xnear(gp.near, xu, g) : xu) + g;
}

template <class P> struct LagDirs {
    static constexpr int N = P::NX + P::NU + (P::LAG_T ? 1 : 0) + (P::LAG_V ? P::NV : 0);
    static constexpr int NCH = (N + P::DC - 1) / P::DC;
    static constexpr int NMAY = 2 * P::NX + P::NV;
    static constexpr int NCH_MAY = (NMAY + P::DC - 1) / P::DC;
};

// d lagrange / d (x, u, t, v) at one point, DC directions per pass, chunk index compile-time (no runtime-indexed arrays)
template <class P, int Q>
__device__ __forceinline__ void lagrange_partials(double t, const double* x, const double* u, const double* v, double& val,
                                                  double* lx, double* lu, double& lt, double* lv) {
    constexpr int n = P::NX, m = P::NU, nv = P::NV, DC = P::DC;
    if constexpr (SymLag<P>::value) {          // generated straight-line code (symbolic first derivatives, ctd_sym.hpp)
        double prm[1 + n + m + nv], out[2 + n + m + nv];
        prm[0] = t;
#pragma unroll
        for (int c = 0; c < n; ++c) prm[1 + c] = x[c];
#pragma unroll
        for (int c = 0; c < m; ++c) prm[1 + n + c] = u[c];
#pragma unroll
        for (int c = 0; c < nv; ++c) prm[1 + n + m + c] = v[c];
        SymLag<P>::eval(prm, out);
        val = out[0];
#pragma unroll
        for (int c = 0; c < n; ++c) lx[c] = out[1 + c];
#pragma unroll
        for (int c = 0; c < m; ++c) lu[c] = out[1 + n + c];
        if (P::LAG_T) lt = out[1 + n + m];
        if (P::LAG_V) {
#pragma unroll
            for (int c = 0; c < nv; ++c) lv[c] = out[2 + n + m + c];
        }
        return;
    }
    using D = Dual<DC>;
    constexpr int g0 = Q * DC, gT = n + m, gV = n + m + (P::LAG_T ? 1 : 0);
    D X[n > 0 ? n : 1], U[m > 0 ? m : 1], V[nv > 0 ? nv : 1], Tt;
#pragma unroll
    for (int c = 0; c < n; ++c) {
        X[c].v = x[c];
#pragma unroll
        for (int d = 0; d < DC; ++d) X[c].d[d] = (g0 + d == c) ? 1.0 : 0.0;
    }
#pragma unroll
    for (int c = 0; c < m; ++c) {
        U[c].v = u[c];
#pragma unroll
        for (int d = 0; d < DC; ++d) U[c].d[d] = (g0 + d == n + c) ? 1.0 : 0.0;
    }
    Tt.v = t;
#pragma unroll
    for (int d = 0; d < DC; ++d) Tt.d[d] = (P::LAG_T && g0 + d == gT) ? 1.0 : 0.0;
#pragma unroll
    for (int c = 0; c < nv; ++c) {
        V[c].v = v[c];
#pragma unroll
        for (int d = 0; d < DC; ++d) V[c].d[d] = (P::LAG_V && g0 + d == gV + c) ? 1.0 : 0.0;
    }
    const D r = P::template lagrange<D>(Tt, X, U, V);
#pragma unroll
    for (int d = 0; d < DC; ++d) {
        constexpr int dummy = 0; (void)dummy;
        const int g = g0 + d;
        if (g < n) lx[g] = r.d[d];
        else if (g < n + m) lu[g - n] = r.d[d];
        else if (P::LAG_T && g == gT) lt = r.d[d];
        else if (P::LAG_V && g < LagDirs<P>::N) lv[g - gV] = r.d[d];
    }
    if (Q == 0) val = r.v;
    if constexpr (Q + 1 < LagDirs<P>::NCH) lagrange_partials<P, Q + 1>(t, x, u, v, val, lx, lu, lt, lv);
}

template <class P, int Q>
__device__ __forceinline__ void mayer_partials(const double* x0, const double* xf, const double* v, double* g0x, double* gfx, double* gv) {
    constexpr int n = P::NX, nv = P::NV, DC = P::DC;
    using D = Dual<DC>;
    constexpr int g0 = Q * DC;
    D X0[n > 0 ? n : 1], XF[n > 0 ? n : 1], V[nv > 0 ? nv : 1];
#pragma unroll
    for (int c = 0; c < n; ++c) {
        X0[c].v = x0[c]; XF[c].v = xf[c];
#pragma unroll
        for (int d = 0; d < DC; ++d) { X0[c].d[d] = (g0 + d == c) ? 1.0 : 0.0; XF[c].d[d] = (g0 + d == n + c) ? 1.0 : 0.0; }
    }
#pragma unroll
    for (int c = 0; c < nv; ++c) {
        V[c].v = v[c];
#pragma unroll
        for (int d = 0; d < DC; ++d) V[c].d[d] = (g0 + d == 2 * n + c) ? 1.0 : 0.0;
    }
    const D r = P::template mayer<D>(X0, XF, V);
#pragma unroll
    for (int d = 0; d < DC; ++d) {
        const int g = g0 + d;
        if (g < n) g0x[g] = r.d[d];
        else if (g < 2 * n) gfx[g - n] = r.d[d];
        else if (g < 2 * n + nv) gv[g - 2 * n] = r.d[d];
    }
    if constexpr (Q + 1 < LagDirs<P>::NCH_MAY) mayer_partials<P, Q + 1>(x0, xf, v, g0x, gfx, gv);
}

template <class P> __device__ __forceinline__ double grad_tau(const GradParams& gp, int64_t i) {
    return gp.tau ? gp.tau[i] : (double)i / (double)gp.L.N;
}
template <class P> __device__ __forceinline__ double grad_time(const GradParams& gp, const double* v, double tau) {
    const double t0 = (P::IT0 >= 0) ? v[P::IT0 >= 0 ? P::IT0 : 0] : gp.L.t0;
    const double tf = (P::ITF >= 0) ? v[P::ITF >= 0 ? P::ITF : 0] : gp.L.tf;
    return t0 + tau * (tf - t0);
}

// body of the gradient pass for workgroup `block` (wsum: 4 * kMaxNV doubles of LDS); also a branch of iter_main_kernel
template <class P, int SC, int S>
__device__ __forceinline__ void grad_units_body(const GradParams& gp, const double* __restrict__ xu, int block, double (*wsum)[kMaxNV]) {
    constexpr int n = P::NX, m = P::NU, nv = P::NV;
    constexpr bool FREE = (P::IT0 >= 0) || (P::ITF >= 0);
    const Layout& L = gp.L;
    double v[nv > 0 ? nv : 1], gv[nv > 0 ? nv : 1];
    for (int k = 0; k < nv; ++k) { v[k] = xu[L.v_off + k]; gv[k] = 0.0; }
    const int64_t i = gp.unit_begin + (int64_t)block * blockDim.x + threadIdx.x;      // step, or node for trapeze / midpoint
    if (P::HAS_LAGRANGE && i < gp.unit_end) {
        double* g = gp.g;
        const double* base = grad_block(gp, xu, i);
        const double* nextb = (SC == SC_MIDPOINT && i < L.N) ? grad_block(gp, xu, i + 1) : base;      // X_{i+1}: the next shard's for a shard's last node
        double x[n > 0 ? n : 1], u[m > 0 ? m : 1], lx[n > 0 ? n : 1], lu[m > 0 ? m : 1], lv[nv > 0 ? nv : 1];
        double lt = 0.0, val = 0.0;
        for (int k = 0; k < nv; ++k) lv[k] = 0.0;
        if (SC == SC_TRAPEZE) {
            // node weight w_i = h_1/2, (t_{i+1}-t_{i-1})/2, h_N/2   (trapeze.jl:78-110)
            const int64_t ia = i == 0 ? 0 : i - 1, ib = i == L.N ? L.N : i + 1;
            const double ta = grad_tau<P>(gp, ia), tb = grad_tau<P>(gp, ib), ti = grad_tau<P>(gp, i);
            const double w = (grad_time<P>(gp, v, tb) - grad_time<P>(gp, v, ta)) / 2.0;
            for (int c = 0; c < n; ++c) x[c] = base[c];
            for (int c = 0; c < m; ++c) u[c] = base[n + c];
            lagrange_partials<P, 0>(grad_time<P>(gp, v, ti), x, u, v, val, lx, lu, lt, lv);
            for (int c = 0; c < n; ++c) g[i * L.blk + c] = w * lx[c];
            for (int c = 0; c < m; ++c) g[i * L.blk + n + c] = w * lu[c];
            for (int k = 0; k < nv; ++k) {
                const double dw = FREE ? (dtime_of<P>(tb, k) - dtime_of<P>(ta, k)) / 2.0 : 0.0;
                gv[k] = dw * val + w * ((P::LAG_V ? lv[k] : 0.0) + ((P::LAG_T && FREE) ? lt * dtime_of<P>(ti, k) : 0.0));
            }
        } else if (SC == SC_MIDPOINT) {
            // node i gathers step i (x_i, u_i, v) and step i-1 (x_i): h * l(0.5(t_i+t_{i+1}), 0.5(x_i+x_{i+1}), u_i, v)   (midpoint.jl:87-97).
            // Euler (euler.jl:112-134) evaluates at (t_i, x_i) [explicit: only the own step reaches x_i] or (t_{i+1}, x_{i+1})
            // [implicit: only the previous step does]; wa / wb are d x_eval / d x_i of the own / previous step's point
            const double wa = L.euler == 0 ? 0.5 : (L.euler == 1 ? 1.0 : 0.0), wb = L.euler == 0 ? 0.5 : (L.euler == 1 ? 0.0 : 1.0);
            double gx[n > 0 ? n : 1];
            for (int c = 0; c < n; ++c) gx[c] = 0.0;
            if (L.cs > 1) {
                // control_steps > 1 (midpoint.jl:99-116): the step's cost is sum_j h_i l(t_ij, x_s, U_i^j, v), h_i = h / cs,
                // t_ij = t_i + (j - 1/2) h_i; node i owns x_i (half of x_s of its own step and of the previous one) and its U_i^j
                const double cs = (double)L.cs;
                for (int side = 0; side < 2; ++side) {           // 0: own step i, 1: previous step i - 1
                    const int64_t st = side == 0 ? i : i - 1;
                    if (st < 0 || st >= L.N) continue;
                    const double* sb = grad_block(gp, xu, st);
                    const double* sn = grad_block(gp, xu, st + 1);
                    const double t0 = grad_tau<P>(gp, st), t1 = grad_tau<P>(gp, st + 1);
                    const double ta = grad_time<P>(gp, v, t0), hi = (grad_time<P>(gp, v, t1) - ta) / cs;
                    for (int c = 0; c < n; ++c) x[c] = 0.5 * (sb[c] + sn[c]);
                    for (int j = 1; j <= L.cs; ++j) {
                        for (int c = 0; c < m; ++c) u[c] = sb[n + (j - 1) * m + c];
                        const double w = (double)j - 0.5;
                        lagrange_partials<P, 0>(ta + w * hi, x, u, v, val, lx, lu, lt, lv);
                        for (int c = 0; c < n; ++c) gx[c] = gx[c] + hi * (0.5 * lx[c]);
                        if (side == 0) {
                            for (int c = 0; c < m; ++c) g[i * L.blk + n + (j - 1) * m + c] = hi * lu[c];
                            for (int k = 0; k < nv; ++k) {
                                const double d0 = FREE ? dtime_of<P>(t0, k) : 0.0, d1 = FREE ? dtime_of<P>(t1, k) : 0.0;
                                const double dhi = (d1 - d0) / cs;
                                gv[k] = gv[k] + dhi * val + hi * ((P::LAG_V ? lv[k] : 0.0) + ((P::LAG_T && FREE) ? lt * (d0 + w * dhi) : 0.0));
                            }
                        }
                    }
                }
            } else {
            if (i < L.N) {
                const double t0 = grad_tau<P>(gp, i), t1 = grad_tau<P>(gp, i + 1);
                const double ta = grad_time<P>(gp, v, t0), tb = grad_time<P>(gp, v, t1), h = tb - ta;
                for (int c = 0; c < n; ++c) x[c] = L.euler == 0 ? 0.5 * (base[c] + nextb[c]) : (L.euler == 1 ? base[c] : nextb[c]);
                for (int c = 0; c < m; ++c) u[c] = base[n + c];
                lagrange_partials<P, 0>(L.euler == 0 ? 0.5 * (ta + tb) : (L.euler == 1 ? ta : tb), x, u, v, val, lx, lu, lt, lv);
                if (wa != 0.0) for (int c = 0; c < n; ++c) gx[c] = h * (wa * lx[c]);
                for (int c = 0; c < m; ++c) g[i * L.blk + n + c] = h * lu[c];
                for (int k = 0; k < nv; ++k) {
                    const double d0 = FREE ? dtime_of<P>(t0, k) : 0.0, d1 = FREE ? dtime_of<P>(t1, k) : 0.0;
                    const double dts = L.euler == 0 ? 0.5 * (d0 + d1) : (L.euler == 1 ? d0 : d1);
                    gv[k] = (d1 - d0) * val + h * ((P::LAG_V ? lv[k] : 0.0) + ((P::LAG_T && FREE) ? lt * dts : 0.0));
                }
            }
            if (i >= 1 && wb != 0.0) {
                const double* pb = grad_block(gp, xu, i - 1);      // (the previous shard's last block for a shard's first node)
                const double ta = grad_time<P>(gp, v, grad_tau<P>(gp, i - 1)), tb = grad_time<P>(gp, v, grad_tau<P>(gp, i));
                const double h = tb - ta;
                for (int c = 0; c < n; ++c) x[c] = L.euler == 0 ? 0.5 * (pb[c] + base[c]) : base[c];
                for (int c = 0; c < m; ++c) u[c] = pb[n + c];
                double val2 = 0.0, lt2 = 0.0, lu2[m > 0 ? m : 1], lv2[nv > 0 ? nv : 1];
                lagrange_partials<P, 0>(L.euler == 0 ? 0.5 * (ta + tb) : tb, x, u, v, val2, lx, lu2, lt2, lv2);
                for (int c = 0; c < n; ++c) gx[c] = gx[c] + h * (wb * lx[c]);
            }
            }
            for (int c = 0; c < n; ++c) g[i * L.blk + c] = gx[c];
        } else {
            // step i: h sum_j b_j l(t_ij, x_ij, u_ij, v)   (irk.jl:179-228, irk_stagewise.jl:344-384)
            const double t0 = grad_tau<P>(gp, i), t1 = grad_tau<P>(gp, i + 1);
            const double ti = grad_time<P>(gp, v, t0), h = grad_time<P>(gp, v, t1) - ti;
            const double* K = base + n + L.cu;
            double gx[n > 0 ? n : 1], gK[S][n > 0 ? n : 1], gu[m > 0 ? m : 1];
            for (int c = 0; c < n; ++c) { gx[c] = 0.0; for (int l = 0; l < S; ++l) gK[l][c] = 0.0; }
            for (int c = 0; c < m; ++c) gu[c] = 0.0;
            double sum_bl = 0.0;
            // not unrolled: with the stages unrolled the compiler keeps every stage's x_ij, sum a K and l_x live at once
            // (364 registers per lane for the 12-state quadrotor at S = 3, one wave per SIMD); j is wave-uniform
#pragma unroll 1
            for (int j = 0; j < S; ++j) {
                double sa[n > 0 ? n : 1];              // sum_l a_jl K^l (for d x_ij / d v)
                for (int c = 0; c < n; ++c) {
                    double xc = base[c], acc = 0.0;
#pragma unroll
                    for (int l = 0; l < S; ++l) { xc = xc + h * L.a[3 * j + l] * K[l * n + c]; acc = acc + L.a[3 * j + l] * K[l * n + c]; }
                    x[c] = xc; sa[c] = acc;
                }
                const double* U = base + n + (L.stagewise ? j * m : 0);
                for (int c = 0; c < m; ++c) u[c] = U[c];
                lagrange_partials<P, 0>(ti + L.c[j] * h, x, u, v, val, lx, lu, lt, lv);
                const double hb = h * L.b[j];
                for (int c = 0; c < n; ++c) {
                    gx[c] = gx[c] + hb * lx[c];
#pragma unroll
                    for (int l = 0; l < S; ++l) gK[l][c] = gK[l][c] + hb * (h * L.a[3 * j + l]) * lx[c];
                }
                if (L.stagewise) { for (int c = 0; c < m; ++c) g[i * L.blk + n + j * m + c] = hb * lu[c]; }
                else { for (int c = 0; c < m; ++c) gu[c] = gu[c] + hb * lu[c]; }
                sum_bl = sum_bl + L.b[j] * val;
                for (int k = 0; k < nv; ++k) {
                    const double d0 = FREE ? dtime_of<P>(t0, k) : 0.0, dh = FREE ? dtime_of<P>(t1, k) - d0 : 0.0;
                    double e = P::LAG_V ? lv[k] : 0.0;
                    if (P::LAG_T && FREE) e = e + lt * (d0 + L.c[j] * dh);
                    if (FREE) for (int c = 0; c < n; ++c) e = e + lx[c] * (dh * sa[c]);
                    gv[k] = gv[k] + hb * e;
                }
            }
            for (int k = 0; k < nv; ++k) {
                const double dh = FREE ? dtime_of<P>(t1, k) - dtime_of<P>(t0, k) : 0.0;
                gv[k] = gv[k] + dh * sum_bl;
            }
            for (int c = 0; c < n; ++c) g[i * L.blk + c] = gx[c];
            if (!L.stagewise) for (int c = 0; c < m; ++c) g[i * L.blk + n + c] = gu[c];
#pragma unroll
            for (int l = 0; l < S; ++l)
                for (int c = 0; c < n; ++c) g[i * L.blk + n + L.cu + l * n + c] = gK[l][c];
        }
    }
    // deterministic reduction of the v-partials: wave shuffles, then one value per workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < nv; ++k) {
        double a = gv[k];
        for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
        if (lane == 0) wsum[wave][k] = a;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int k = 0; k < nv; ++k) {
            double sacc = 0.0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) sacc += wsum[w][k];
            gp.partial[(size_t)block * kMaxNV + k] = sacc;
        }
}
template <class P, int SC, int S>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) grad_units_kernel(const GradParams gp, const double* __restrict__ xu) {
    __shared__ double wsum[4][kMaxNV];
    grad_units_body<P, SC, S>(gp, xu, (int)blockIdx.x, wsum);
}

template <class P>
__device__ __forceinline__ void grad_finish_body(const GradParams& gp, const double* __restrict__ xu) {
    if (threadIdx.x >= 64) return;
    constexpr int n = P::NX, nv = P::NV;
    const Layout& L = gp.L;
    double gvs[nv > 0 ? nv : 1];
    for (int k = 0; k < nv; ++k) {          // one wave, fixed order (see obj_finish_kernel)
        double sacc = 0.0;
        for (int b = threadIdx.x; b < gp.nblocks; b += 64) sacc += gp.partial[(size_t)b * kMaxNV + k];
        for (int off = 32; off > 0; off >>= 1) sacc += __shfl_down(sacc, off, 64);
        gvs[k] = sacc;
    }
    if (threadIdx.x != 0) return;
    if (P::HAS_MAYER && (gp.owns_first || gp.owns_last)) {
        double x0[n > 0 ? n : 1], xf[n > 0 ? n : 1], v[nv > 0 ? nv : 1], g0x[n > 0 ? n : 1], gfx[n > 0 ? n : 1], gmv[nv > 0 ? nv : 1];
        const double* b0 = grad_block(gp, xu, 0);
        const double* bf = grad_block(gp, xu, L.N);
        for (int c = 0; c < n; ++c) { x0[c] = b0[c]; xf[c] = bf[c]; g0x[c] = 0.0; gfx[c] = 0.0; }
        for (int k = 0; k < nv; ++k) { v[k] = xu[L.v_off + k]; gmv[k] = 0.0; }
        mayer_partials<P, 0>(x0, xf, v, g0x, gfx, gmv);
        for (int c = 0; c < n; ++c) {
            if (gp.owns_first) gp.g[c] += g0x[c];
            if (gp.owns_last) gp.g[L.N * (int64_t)L.blk + c] += gfx[c];
        }
        if (gp.owns_last) for (int k = 0; k < nv; ++k) gvs[k] += gmv[k];
    }
    for (int k = 0; k < nv; ++k) gp.g[L.v_off + k] = gvs[k];
}
template <class P>
__global__ void grad_finish_kernel(const GradParams gp, const double* __restrict__ xu) {
    if (blockIdx.x != 0) return;
    grad_finish_body<P>(gp, xu);
}

#if !defined(__HIPCC_RTC__)
template <class P>
hipError_t launch_grad(int sc, int s, const GradParams& gp, const double* xu, int grid, hipStream_t st) {
    if (grid <= 0) {}      // Mayer-only cost: nothing to integrate
    else if (sc == SC_TRAPEZE) grad_units_kernel<P, SC_TRAPEZE, 1><<<grid, 256, 0, st>>>(gp, xu);
    else if (sc == SC_MIDPOINT) grad_units_kernel<P, SC_MIDPOINT, 1><<<grid, 256, 0, st>>>(gp, xu);
    else if (s == 1) grad_units_kernel<P, SC_IRK, 1><<<grid, 256, 0, st>>>(gp, xu);
    else if (s == 2) grad_units_kernel<P, SC_IRK, 2><<<grid, 256, 0, st>>>(gp, xu);
    else grad_units_kernel<P, SC_IRK, 3><<<grid, 256, 0, st>>>(gp, xu);
    grad_finish_kernel<P><<<1, 64, 0, st>>>(gp, xu);
    return hipGetLastError();
}

#endif

#if !defined(__HIPCC_RTC__)
// ---- launchers ---------------------------------------------------------------------------------------------------
// Defined here as templates; each per-problem translation unit (ctd_kern_*.hip) explicitly instantiates them for one
// OCP so the registry compiles in parallel, and ctd_engine.hip only sees `extern template` declarations.
// Five kernel variants per OCP: (trapeze), (midpoint), (Gauss-Legendre s = 1, 2, 3); the stage count is a template
// parameter so every loop over stages unrolls.
template <class P, int SC, int S, bool DBG>
hipError_t launch_variant_dbg(const KParams& kp, const double* xu, int grid, int block, size_t lds_bytes, hipStream_t st,
                              hipEvent_t e0, hipEvent_t e1) {
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)cons_jac_kernel<P, SC, S, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
#ifdef CTD_KP_INDIRECT
    {
        static KParams* d_kp = nullptr;          // (one block per kernel instantiation: an experiment, one handle at a time)
        static KParams last;
        static bool have = false;
        if (!d_kp && hipMalloc((void**)&d_kp, sizeof(KParams)) != hipSuccess) return hipGetLastError();
        if (!have || std::memcmp(&last, &kp, sizeof(KParams)) != 0) {
            hipError_t e = hipMemcpyAsync(d_kp, &kp, sizeof(KParams), hipMemcpyHostToDevice, st);
            if (e != hipSuccess) return e;
            (void)hipStreamSynchronize(st);
            last = kp; have = true;
        }
        if (e0 || e1) hipExtLaunchKernelGGL((cons_jac_kernel_ind<P, SC, S, DBG>), dim3(grid), dim3(block), lds_bytes, st, e0, e1, 0, (const KParams*)d_kp, xu);
        else cons_jac_kernel_ind<P, SC, S, DBG><<<grid, block, lds_bytes, st>>>((const KParams*)d_kp, xu);
        return hipGetLastError();
    }
#endif
    // e0/e1 (optional): events recorded by the dispatch itself right before / after THIS kernel, so
    // hipEventElapsedTime(e0, e1) is the kernel's own duration on the stream it was launched on
    if (e0 || e1) hipExtLaunchKernelGGL((cons_jac_kernel<P, SC, S, DBG>), dim3(grid), dim3(block), lds_bytes, st, e0, e1, 0, kp, xu);
    else cons_jac_kernel<P, SC, S, DBG><<<grid, block, lds_bytes, st>>>(kp, xu);
    return hipGetLastError();
}
template <class P, int SC, int S>
hipError_t launch_variant(const KParams& kp, const double* xu, int grid, int block, size_t lds_bytes, hipStream_t st,
                          hipEvent_t e0, hipEvent_t e1) {
    if (kp.stamps || kp.debug_stop) return launch_variant_dbg<P, SC, S, true>(kp, xu, grid, block, lds_bytes, st, e0, e1);
    return launch_variant_dbg<P, SC, S, false>(kp, xu, grid, block, lds_bytes, st, e0, e1);
}

template <class P>
hipError_t launch_cons_jac(int sc, const KParams& kp, const double* xu, int grid, int block, size_t lds_bytes, hipStream_t st,
                           hipEvent_t e0, hipEvent_t e1) {
    if (sc == SC_TRAPEZE) return launch_variant<P, SC_TRAPEZE, 1>(kp, xu, grid, block, lds_bytes, st, e0, e1);
    if (sc == SC_MIDPOINT) {       // (midpoint: the template's stage count is control_steps -- 1 in collocation, up to 3 compiled in)
        if (kp.L.cs == 2) return launch_variant<P, SC_MIDPOINT, 2>(kp, xu, grid, block, lds_bytes, st, e0, e1);
        if (kp.L.cs == 3) return launch_variant<P, SC_MIDPOINT, 3>(kp, xu, grid, block, lds_bytes, st, e0, e1);
        if (kp.L.cs > 3) return hipErrorInvalidValue;
        return launch_variant<P, SC_MIDPOINT, 1>(kp, xu, grid, block, lds_bytes, st, e0, e1);
    }
    if (kp.L.s == 1) return launch_variant<P, SC_IRK, 1>(kp, xu, grid, block, lds_bytes, st, e0, e1);
    if (kp.L.s == 2) return launch_variant<P, SC_IRK, 2>(kp, xu, grid, block, lds_bytes, st, e0, e1);
    return launch_variant<P, SC_IRK, 3>(kp, xu, grid, block, lds_bytes, st, e0, e1);
}

// resident workgroups per CU of the kernel launch_cons_jac would run (registers and LDS): sizes the multi-tile grid
template <class P, int SC, int S>
int occupancy_variant(int block, size_t lds_bytes) {
    int nb = 0;
    if (lds_bytes > 64 * 1024 &&
        hipFuncSetAttribute((const void*)cons_jac_kernel<P, SC, S, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
        return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, cons_jac_kernel<P, SC, S, false>, block, lds_bytes) != hipSuccess) return 0;
    return nb;
}
template <class P>
int occupancy_cons_jac(int sc, const KParams& kp, int block, size_t lds_bytes) {
    if (sc == SC_TRAPEZE) return occupancy_variant<P, SC_TRAPEZE, 1>(block, lds_bytes);
    if (sc == SC_MIDPOINT) {
        if (kp.L.cs == 2) return occupancy_variant<P, SC_MIDPOINT, 2>(block, lds_bytes);
        if (kp.L.cs == 3) return occupancy_variant<P, SC_MIDPOINT, 3>(block, lds_bytes);
        return occupancy_variant<P, SC_MIDPOINT, 1>(block, lds_bytes);
    }
    if (kp.L.s == 1) return occupancy_variant<P, SC_IRK, 1>(block, lds_bytes);
    if (kp.L.s == 2) return occupancy_variant<P, SC_IRK, 2>(block, lds_bytes);
    return occupancy_variant<P, SC_IRK, 3>(block, lds_bytes);
}

template <class P>
hipError_t launch_obj(int sc, const ObjParams& op, const double* xu, int grid, int block, hipStream_t st) {
    if (grid <= 0) {}      // Mayer-only cost: nothing to integrate
    else if (sc == SC_TRAPEZE) obj_partial_kernel<P, SC_TRAPEZE><<<grid, block, 0, st>>>(op, xu);
    else if (sc == SC_MIDPOINT) obj_partial_kernel<P, SC_MIDPOINT><<<grid, block, 0, st>>>(op, xu);
    else obj_partial_kernel<P, SC_IRK><<<grid, block, 0, st>>>(op, xu);
    obj_finish_kernel<P><<<1, 64, 0, st>>>(op, xu);
    return hipGetLastError();
}

#define CTD_INSTANTIATE_LAUNCHERS(P)                                                                                       \
    template hipError_t launch_cons_jac<P>(int, const KParams&, const double*, int, int, size_t, hipStream_t, hipEvent_t, \
                                           hipEvent_t);                                                                    \
    template hipError_t launch_obj<P>(int, const ObjParams&, const double*, int, int, hipStream_t);                       \
    template int occupancy_cons_jac<P>(int, const KParams&, int, size_t);                                                 \
    template hipError_t launch_grad<P>(int, int, const GradParams&, const double*, int, hipStream_t);
#define CTD_EXTERN_LAUNCHERS(P)                                                                                            \
    extern template hipError_t launch_cons_jac<P>(int, const KParams&, const double*, int, int, size_t, hipStream_t,      \
                                                  hipEvent_t, hipEvent_t);                                                 \
    extern template hipError_t launch_obj<P>(int, const ObjParams&, const double*, int, int, hipStream_t);               \
    extern template int occupancy_cons_jac<P>(int, const KParams&, int, size_t);                                         \
    extern template hipError_t launch_grad<P>(int, int, const GradParams&, const double*, int, hipStream_t);

#endif  // !__HIPCC_RTC__

}  // namespace ctd
