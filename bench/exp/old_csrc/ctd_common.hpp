// ctd_common.hpp -- shared host/device definitions of the collocation engine.
#pragma once
// __HIPCC_RTC__: the same headers are compiled at run time by hiprtc for OCPs registered through ctd_register_ocp
// (ctd_jit.cpp); hiprtc pre-includes the HIP runtime and has no C++ standard library headers
#if !defined(__HIPCC_RTC__)
#include <cmath>
#include <cstdint>
#include <cstddef>
#endif

#if defined(__HIPCC_RTC__)
using __hip_internal::int32_t;
using __hip_internal::uint32_t;
using __hip_internal::int64_t;
using __hip_internal::uint64_t;
using __hip_internal::uint16_t;
using __hip_internal::uint8_t;
typedef unsigned long uintptr_t;
#define CTD_HD __host__ __device__ __forceinline__
#define CTD_STORE2(p, a, b) (*reinterpret_cast<double2*>(p) = make_double2((a), (b)))
#elif defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CTD_HD __host__ __device__ __forceinline__
// one 16-byte store of two consecutive doubles (p is 16-byte aligned)
#define CTD_STORE2(p, a, b) (*reinterpret_cast<double2*>(p) = make_double2((a), (b)))
#else
// The kernel bodies are plain C++ templates so that the test-suite can compile them with g++ and step them
// serially with bounds checking (tests/emu/, test infrastructure only).  The shipped library always builds
// them with hipcc for gfx950 and the C ABI only ever launches the HIP kernels.
#define CTD_HD inline
#define CTD_STORE2(p, a, b) do { (p)[0] = (a); (p)[1] = (b); } while (0)
#endif

namespace ctd {

// ---- forward-mode dual number with K directions, evaluated in registers ---------------------------------
// The reference obtains Jacobian values by pushing ForwardDiff.Dual numbers through its generic callbacks
// (ADNLPModels.SparseADJacobian, call site src/collocation.jl:116-120).  The engine differentiates the
// user functions (dynamics / path / boundary / costs) only, K directions at a time, and applies the scheme's
// chain rule in closed form.
template <int K>
struct Dual {
    double v;
    double d[K];
    CTD_HD Dual() {}
    CTD_HD Dual(double x) : v(x) {
#pragma unroll
        for (int i = 0; i < K; ++i) d[i] = 0.0;
    }
};

template <int K> CTD_HD Dual<K> operator+(const Dual<K>& a, const Dual<K>& b) {
    Dual<K> r; r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = a.d[i] + b.d[i];
    return r;
}
template <int K> CTD_HD Dual<K> operator-(const Dual<K>& a, const Dual<K>& b) {
    Dual<K> r; r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = a.d[i] - b.d[i];
    return r;
}
template <int K> CTD_HD Dual<K> operator-(const Dual<K>& a) {
    Dual<K> r; r.v = -a.v;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = -a.d[i];
    return r;
}
template <int K> CTD_HD Dual<K> operator*(const Dual<K>& a, const Dual<K>& b) {
    Dual<K> r; r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r;
}
// quotient: the value is the exact IEEE quotient; the partials use one reciprocal instead of K divisions
template <int K> CTD_HD Dual<K> operator/(const Dual<K>& a, const Dual<K>& b) {
    Dual<K> r; const double q = a.v / b.v; const double inv = 1.0 / b.v; r.v = q;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = (a.d[i] - q * b.d[i]) * inv;
    return r;
}
template <int K> CTD_HD Dual<K> operator+(const Dual<K>& a, double b) { Dual<K> r = a; r.v = a.v + b; return r; }
template <int K> CTD_HD Dual<K> operator+(double a, const Dual<K>& b) { Dual<K> r = b; r.v = a + b.v; return r; }
template <int K> CTD_HD Dual<K> operator-(const Dual<K>& a, double b) { Dual<K> r = a; r.v = a.v - b; return r; }
template <int K> CTD_HD Dual<K> operator-(double a, const Dual<K>& b) {
    Dual<K> r; r.v = a - b.v;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = -b.d[i];
    return r;
}
template <int K> CTD_HD Dual<K> operator*(const Dual<K>& a, double b) {
    Dual<K> r; r.v = a.v * b;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = a.d[i] * b;
    return r;
}
template <int K> CTD_HD Dual<K> operator*(double a, const Dual<K>& b) { return b * a; }
template <int K> CTD_HD Dual<K> operator/(const Dual<K>& a, double b) {
    Dual<K> r; r.v = a.v / b; const double inv = 1.0 / b;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = a.d[i] * inv;
    return r;
}
template <int K> CTD_HD Dual<K> operator/(double a, const Dual<K>& b) {
    Dual<K> r; const double q = a / b.v; const double w = -q * (1.0 / b.v); r.v = q;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = w * b.d[i];
    return r;
}

CTD_HD double d_exp(double x) { return ::exp(x); }

// sin and cos of one argument together.  The library functions cost ~100 dependent instructions EACH on gfx950 (range reduction
// included) and an evaluating lane of a rigid-body OCP needs six of them per point: 80 % of the evaluation phase of the 12-state
// quadrotor kernels.  |x| <= 1e6 (every angle an OCP meets): k = rint(x 2/pi), three-constant Cody-Waite reduction with fused
// multiply-adds (k pio2_1 and k pio2_2 are exact products: 33-bit constants, |k| < 2^20), fdlibm's minimax polynomials on
// [-pi/4, pi/4] (errors below 1 ulp), quadrant selection -- about 30 instructions for the pair, results within ~1 ulp of the
// library's.  Larger arguments, NaN and Inf take the library functions.  Both d_sin and d_cos inline this: asked for the same
// argument in one block, the common part is computed once (common-subexpression elimination).
CTD_HD void d_sincos(double x, double& sn, double& cs) {
    const double fn = __builtin_rint(x * 6.36619772367581382433e-01);
    double r = __builtin_fma(-fn, 1.57079632673412561417e+00, x);
    r = __builtin_fma(-fn, 6.07710050630396597660e-11, r);
    r = __builtin_fma(-fn, 2.02226624871116645580e-21, r);
    const double z = r * r;
    const double ps = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
                      z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    const double sr = r + (z * r) * (-1.66666666666666324348e-01 + z * ps);
    const double pc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                      z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    const double cr = 1.0 - (0.5 * z - z * pc);
    const int q = (int)fn & 3;
    sn = (q & 1) ? cr : sr;
    cs = (q & 1) ? sr : cr;
    if (q & 2) sn = -sn;
    if ((q + 1) & 2) cs = -cs;
    if (!(__builtin_fabs(x) <= 1.0e6)) { sn = ::sin(x); cs = ::cos(x); }
}
CTD_HD double d_sin(double x) { double s, c; d_sincos(x, s, c); return s; }
CTD_HD double d_cos(double x) { double s, c; d_sincos(x, s, c); return c; }
CTD_HD double d_sqr(double x) { return x * x; }
template <int K> CTD_HD Dual<K> d_exp(const Dual<K>& a) {
    Dual<K> r; const double e = ::exp(a.v); r.v = e;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = e * a.d[i];
    return r;
}
template <int K> CTD_HD Dual<K> d_sin(const Dual<K>& a) {
    Dual<K> r; double s, c; d_sincos(a.v, s, c); r.v = s;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = c * a.d[i];
    return r;
}
template <int K> CTD_HD Dual<K> d_cos(const Dual<K>& a) {
    Dual<K> r; double s, c; d_sincos(a.v, s, c); r.v = c;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = -s * a.d[i];
    return r;
}
template <int K> CTD_HD Dual<K> d_sqr(const Dual<K>& a) {
    Dual<K> r; r.v = a.v * a.v; const double t = 2.0 * a.v;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = t * a.d[i];
    return r;
}
CTD_HD double d_sqrt(double x) { return ::sqrt(x); }
template <int K> CTD_HD Dual<K> d_sqrt(const Dual<K>& a) {
    Dual<K> r; const double s = ::sqrt(a.v); r.v = s; const double w = 0.5 / s;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = w * a.d[i];
    return r;
}
// further unary functions of run-time defined OCPs (ctd_jit.cpp): value f0 and derivative f1 at a.v
template <int K> CTD_HD Dual<K> d1_chain(const Dual<K>& a, double f0, double f1) {
    Dual<K> r; r.v = f0;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = f1 * a.d[i];
    return r;
}
CTD_HD double d_log(double x) { return ::log(x); }
CTD_HD double d_tan(double x) { return ::tan(x); }
CTD_HD double d_atan(double x) { return ::atan(x); }
CTD_HD double d_tanh(double x) { return ::tanh(x); }
CTD_HD double d_abs(double x) { return ::fabs(x); }
CTD_HD double d_sgn(double x) { return x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : 0.0); }
// asin acos sinh cosh floor, real powers and max / min: the rest of what the reference's problem folder uses (test/problems/bioreactor.jl:19
// max(0, sin)^2 and floor, parametric.jl:4; swimmer / action use sqrt and trigonometry only)
CTD_HD double d_asin(double x) { return ::asin(x); }
CTD_HD double d_acos(double x) { return ::acos(x); }
CTD_HD double d_sinh(double x) { return ::sinh(x); }
CTD_HD double d_cosh(double x) { return ::cosh(x); }
CTD_HD double d_floor(double x) { return ::floor(x); }
CTD_HD double d_powr(double x, double p) { return ::pow(x, p); }
// 1.0 where a > b, else 0.0: the selector of max / min.  ForwardDiff's rules (DiffRules: max -> (x > y ? 1 : 0, x > y ? 0 : 1),
// min -> (x > y ? 0 : 1, x > y ? 1 : 0)): at a tie max follows its SECOND argument and min its FIRST
CTD_HD double d_gt(double a, double b) { return a > b ? 1.0 : 0.0; }
CTD_HD double d_max(double a, double b) { return a > b ? a : b; }
CTD_HD double d_min(double a, double b) { return a > b ? b : a; }
template <int K> CTD_HD Dual<K> d_log(const Dual<K>& a) { return d1_chain(a, ::log(a.v), 1.0 / a.v); }
template <int K> CTD_HD Dual<K> d_asin(const Dual<K>& a) { return d1_chain(a, ::asin(a.v), 1.0 / ::sqrt(1.0 - a.v * a.v)); }
template <int K> CTD_HD Dual<K> d_acos(const Dual<K>& a) { return d1_chain(a, ::acos(a.v), -1.0 / ::sqrt(1.0 - a.v * a.v)); }
template <int K> CTD_HD Dual<K> d_sinh(const Dual<K>& a) { return d1_chain(a, ::sinh(a.v), ::cosh(a.v)); }
template <int K> CTD_HD Dual<K> d_cosh(const Dual<K>& a) { return d1_chain(a, ::cosh(a.v), ::sinh(a.v)); }
template <int K> CTD_HD Dual<K> d_floor(const Dual<K>& a) { return d1_chain(a, ::floor(a.v), 0.0); }
template <int K> CTD_HD Dual<K> d_powr(const Dual<K>& a, double p) { return d1_chain(a, ::pow(a.v, p), p * ::pow(a.v, p - 1.0)); }
template <int K> CTD_HD Dual<K> d_max(const Dual<K>& a, const Dual<K>& b) { return a.v > b.v ? a : b; }
template <int K> CTD_HD Dual<K> d_min(const Dual<K>& a, const Dual<K>& b) { return a.v > b.v ? b : a; }
// (generated functors call these with the scalar type named: d_max2<T>(0.0, e) converts a constant operand)
template <class T> CTD_HD T d_max2(const T& a, const T& b) { return d_max(a, b); }
template <class T> CTD_HD T d_min2(const T& a, const T& b) { return d_min(a, b); }
template <int K> CTD_HD Dual<K> d_tan(const Dual<K>& a) { const double t = ::tan(a.v); return d1_chain(a, t, 1.0 + t * t); }
template <int K> CTD_HD Dual<K> d_atan(const Dual<K>& a) { return d1_chain(a, ::atan(a.v), 1.0 / (1.0 + a.v * a.v)); }
template <int K> CTD_HD Dual<K> d_tanh(const Dual<K>& a) { const double t = ::tanh(a.v); return d1_chain(a, t, 1.0 - t * t); }
template <int K> CTD_HD Dual<K> d_abs(const Dual<K>& a) { return d1_chain(a, ::fabs(a.v), d_sgn(a.v)); }
// x^k for a small non-negative integer k by repeated multiplication (run-time defined OCPs, ctd_jit.cpp)
template <class T> CTD_HD T d_powi(const T& x, int k) {
    T r = x;
    for (int i = 1; i < k; ++i) r = r * x;
    return r;
}
CTD_HD double d_val(double x) { return x; }
template <int K> CTD_HD double d_val(const Dual<K>& a) { return a.v; }

// ---- second-order forward number: one "outer" direction a, K "inner" directions b_k, and the mixed second
// derivatives ab_k = d2/(da db_k).  One lane of the Hessian kernel pushes it through an OCP function to obtain K
// entries of one row of that function's Hessian (plus the first derivative along a).  ADNLPModels gets the same
// numbers from nested ForwardDiff duals over the whole Lagrangian (backend selection src/collocation.jl:121-125).
template <int K>
struct Dual2 {
    double v, a;
    double b[K], ab[K];
    CTD_HD Dual2() {}
    CTD_HD Dual2(double x) : v(x), a(0.0) {
#pragma unroll
        for (int i = 0; i < K; ++i) { b[i] = 0.0; ab[i] = 0.0; }
    }
};
// r = f(x) given f0 = f(x.v), f1 = f'(x.v), f2 = f''(x.v)
template <int K> CTD_HD Dual2<K> d2_chain(const Dual2<K>& x, double f0, double f1, double f2) {
    Dual2<K> r; r.v = f0; r.a = f1 * x.a;
    const double t = f2 * x.a;
#pragma unroll
    for (int i = 0; i < K; ++i) { r.b[i] = f1 * x.b[i]; r.ab[i] = f1 * x.ab[i] + t * x.b[i]; }
    return r;
}
template <int K> CTD_HD Dual2<K> operator+(const Dual2<K>& x, const Dual2<K>& y) {
    Dual2<K> r; r.v = x.v + y.v; r.a = x.a + y.a;
#pragma unroll
    for (int i = 0; i < K; ++i) { r.b[i] = x.b[i] + y.b[i]; r.ab[i] = x.ab[i] + y.ab[i]; }
    return r;
}
template <int K> CTD_HD Dual2<K> operator-(const Dual2<K>& x, const Dual2<K>& y) {
    Dual2<K> r; r.v = x.v - y.v; r.a = x.a - y.a;
#pragma unroll
    for (int i = 0; i < K; ++i) { r.b[i] = x.b[i] - y.b[i]; r.ab[i] = x.ab[i] - y.ab[i]; }
    return r;
}
template <int K> CTD_HD Dual2<K> operator-(const Dual2<K>& x) {
    Dual2<K> r; r.v = -x.v; r.a = -x.a;
#pragma unroll
    for (int i = 0; i < K; ++i) { r.b[i] = -x.b[i]; r.ab[i] = -x.ab[i]; }
    return r;
}
template <int K> CTD_HD Dual2<K> operator*(const Dual2<K>& x, const Dual2<K>& y) {
    Dual2<K> r; r.v = x.v * y.v; r.a = x.a * y.v + x.v * y.a;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        r.b[i] = x.b[i] * y.v + x.v * y.b[i];
        r.ab[i] = (x.ab[i] * y.v + x.v * y.ab[i]) + (x.a * y.b[i] + x.b[i] * y.a);
    }
    return r;
}
template <int K> CTD_HD Dual2<K> d2_recip(const Dual2<K>& y) {
    const double q = 1.0 / y.v;
    return d2_chain(y, q, -(q * q), 2.0 * (q * q) * q);
}
// quotients go through one reciprocal: the VALUE of a Dual2 is never an output of the Hessian kernel (only second
// derivatives are), so it may differ from the IEEE quotient in the last bit
template <int K> CTD_HD Dual2<K> operator/(const Dual2<K>& x, const Dual2<K>& y) { return x * d2_recip(y); }
template <int K> CTD_HD Dual2<K> operator+(const Dual2<K>& x, double y) { Dual2<K> r = x; r.v = x.v + y; return r; }
template <int K> CTD_HD Dual2<K> operator+(double x, const Dual2<K>& y) { Dual2<K> r = y; r.v = x + y.v; return r; }
template <int K> CTD_HD Dual2<K> operator-(const Dual2<K>& x, double y) { Dual2<K> r = x; r.v = x.v - y; return r; }
template <int K> CTD_HD Dual2<K> operator-(double x, const Dual2<K>& y) { Dual2<K> r = -y; r.v = x - y.v; return r; }
template <int K> CTD_HD Dual2<K> operator*(const Dual2<K>& x, double y) {
    Dual2<K> r; r.v = x.v * y; r.a = x.a * y;
#pragma unroll
    for (int i = 0; i < K; ++i) { r.b[i] = x.b[i] * y; r.ab[i] = x.ab[i] * y; }
    return r;
}
template <int K> CTD_HD Dual2<K> operator*(double x, const Dual2<K>& y) { return y * x; }
template <int K> CTD_HD Dual2<K> operator/(const Dual2<K>& x, double y) { return x * (1.0 / y); }
template <int K> CTD_HD Dual2<K> operator/(double x, const Dual2<K>& y) { return d2_recip(y) * x; }
template <int K> CTD_HD Dual2<K> d_exp(const Dual2<K>& x) { const double e = ::exp(x.v); return d2_chain(x, e, e, e); }
template <int K> CTD_HD Dual2<K> d_sin(const Dual2<K>& x) { double s, c; d_sincos(x.v, s, c); return d2_chain(x, s, c, -s); }
template <int K> CTD_HD Dual2<K> d_cos(const Dual2<K>& x) { double s, c; d_sincos(x.v, s, c); return d2_chain(x, c, -s, -c); }
template <int K> CTD_HD Dual2<K> d_sqr(const Dual2<K>& x) { return d2_chain(x, x.v * x.v, 2.0 * x.v, 2.0); }
template <int K> CTD_HD Dual2<K> d_sqrt(const Dual2<K>& x) { const double s = ::sqrt(x.v); return d2_chain(x, s, 0.5 / s, -0.25 / (s * x.v)); }
template <int K> CTD_HD Dual2<K> d_log(const Dual2<K>& x) { const double q = 1.0 / x.v; return d2_chain(x, ::log(x.v), q, -(q * q)); }
template <int K> CTD_HD Dual2<K> d_tan(const Dual2<K>& x) { const double t = ::tan(x.v), s = 1.0 + t * t; return d2_chain(x, t, s, 2.0 * t * s); }
template <int K> CTD_HD Dual2<K> d_atan(const Dual2<K>& x) { const double q = 1.0 / (1.0 + x.v * x.v); return d2_chain(x, ::atan(x.v), q, -2.0 * x.v * (q * q)); }
template <int K> CTD_HD Dual2<K> d_tanh(const Dual2<K>& x) { const double t = ::tanh(x.v), s = 1.0 - t * t; return d2_chain(x, t, s, -2.0 * t * s); }
template <int K> CTD_HD Dual2<K> d_abs(const Dual2<K>& x) { return d2_chain(x, ::fabs(x.v), d_sgn(x.v), 0.0); }
template <int K> CTD_HD Dual2<K> d_asin(const Dual2<K>& x) { const double q = 1.0 / (1.0 - x.v * x.v), r = ::sqrt(q); return d2_chain(x, ::asin(x.v), r, x.v * q * r); }
template <int K> CTD_HD Dual2<K> d_acos(const Dual2<K>& x) { const double q = 1.0 / (1.0 - x.v * x.v), r = ::sqrt(q); return d2_chain(x, ::acos(x.v), -r, -(x.v * q * r)); }
template <int K> CTD_HD Dual2<K> d_sinh(const Dual2<K>& x) { const double s = ::sinh(x.v), c = ::cosh(x.v); return d2_chain(x, s, c, s); }
template <int K> CTD_HD Dual2<K> d_cosh(const Dual2<K>& x) { const double s = ::sinh(x.v), c = ::cosh(x.v); return d2_chain(x, c, s, c); }
template <int K> CTD_HD Dual2<K> d_floor(const Dual2<K>& x) { return d2_chain(x, ::floor(x.v), 0.0, 0.0); }
template <int K> CTD_HD Dual2<K> d_powr(const Dual2<K>& x, double p) {
    return d2_chain(x, ::pow(x.v, p), p * ::pow(x.v, p - 1.0), p * (p - 1.0) * ::pow(x.v, p - 2.0));
}
template <int K> CTD_HD Dual2<K> d_max(const Dual2<K>& a, const Dual2<K>& b) { return a.v > b.v ? a : b; }
template <int K> CTD_HD Dual2<K> d_min(const Dual2<K>& a, const Dual2<K>& b) { return a.v > b.v ? b : a; }
template <int K> CTD_HD double d_val(const Dual2<K>& x) { return x.v; }

}  // namespace ctd
