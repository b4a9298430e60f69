// TEST INFRASTRUCTURE ONLY -- part of the CPU oracle (see oracle/ctd_oracle.cpp header).
//
// One-partial forward dual number, the CPU stand-in for ForwardDiff.Dual{Tag,Float64,1}
// that ADNLPModels.SparseADJacobian pushes through the reference's c!(c, x) closure
// (reference call site: src/collocation.jl:116-120; behaviour notes test/archives/AD_backend.md:3-5).
#pragma once
#include <cmath>

namespace orc {

struct D1 {
    double v;  // value
    double d;  // single partial
    D1() : v(0.0), d(0.0) {}
    D1(double v_) : v(v_), d(0.0) {}
    D1(double v_, double d_) : v(v_), d(d_) {}
};

inline D1 operator+(const D1& a, const D1& b) { return D1(a.v + b.v, a.d + b.d); }
inline D1 operator-(const D1& a, const D1& b) { return D1(a.v - b.v, a.d - b.d); }
inline D1 operator-(const D1& a) { return D1(-a.v, -a.d); }
inline D1 operator*(const D1& a, const D1& b) { return D1(a.v * b.v, a.d * b.v + a.v * b.d); }
inline D1 operator/(const D1& a, const D1& b) {
    double q = a.v / b.v;
    return D1(q, (a.d - q * b.d) / b.v);
}
inline D1 operator+(const D1& a, double b) { return D1(a.v + b, a.d); }
inline D1 operator+(double a, const D1& b) { return D1(a + b.v, b.d); }
inline D1 operator-(const D1& a, double b) { return D1(a.v - b, a.d); }
inline D1 operator-(double a, const D1& b) { return D1(a - b.v, -b.d); }
inline D1 operator*(const D1& a, double b) { return D1(a.v * b, a.d * b); }
inline D1 operator*(double a, const D1& b) { return D1(a * b.v, a * b.d); }
inline D1 operator/(const D1& a, double b) { return D1(a.v / b, a.d / b); }
inline D1 operator/(double a, const D1& b) {
    double q = a / b.v;
    return D1(q, -q * b.d / b.v);
}
inline D1& operator+=(D1& a, const D1& b) { a = a + b; return a; }

inline D1 exp(const D1& a) { double e = std::exp(a.v); return D1(e, e * a.d); }
inline D1 sin(const D1& a) { return D1(std::sin(a.v), std::cos(a.v) * a.d); }
inline D1 cos(const D1& a) { return D1(std::cos(a.v), -std::sin(a.v) * a.d); }
inline D1 sqrt(const D1& a) { double s = std::sqrt(a.v); return D1(s, a.d / (2.0 * s)); }
// x^2 as written in the problem files (Julia literal power): value x*x, partial 2*x*x'
inline D1 sq(const D1& a) { return D1(a.v * a.v, 2.0 * a.v * a.d); }

inline double exp(double a) { return std::exp(a); }
inline double sin(double a) { return std::sin(a); }
inline double cos(double a) { return std::cos(a); }
inline double sqrt(double a) { return std::sqrt(a); }
inline double sq(double a) { return a * a; }

inline double value(double a) { return a; }
inline double value(const D1& a) { return a.v; }

}  // namespace orc
