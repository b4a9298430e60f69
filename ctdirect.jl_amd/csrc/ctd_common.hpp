// ctd_common.hpp -- shared host/device definitions of the collocation engine.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstddef>

#if defined(__HIPCC__)
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
CTD_HD double d_sin(double x) { return ::sin(x); }
CTD_HD double d_cos(double x) { return ::cos(x); }
CTD_HD double d_sqr(double x) { return x * x; }
template <int K> CTD_HD Dual<K> d_exp(const Dual<K>& a) {
    Dual<K> r; const double e = ::exp(a.v); r.v = e;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = e * a.d[i];
    return r;
}
template <int K> CTD_HD Dual<K> d_sin(const Dual<K>& a) {
    Dual<K> r; const double s = ::sin(a.v), c = ::cos(a.v); r.v = s;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = c * a.d[i];
    return r;
}
template <int K> CTD_HD Dual<K> d_cos(const Dual<K>& a) {
    Dual<K> r; const double s = ::sin(a.v), c = ::cos(a.v); r.v = c;
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
CTD_HD double d_val(double x) { return x; }
template <int K> CTD_HD double d_val(const Dual<K>& a) { return a.v; }

}  // namespace ctd
