// TEST INFRASTRUCTURE ONLY -- part of the CPU oracle (see oracle/ctd_oracle.cpp header).
//
// Forward dual number with a block of partials, used by the oracle's BLOCK mode: one time step of the
// transcription is differentiated with respect to all of its own variables in a single pass (dense local forward AD),
// instead of one pass of the whole constraints function per colour.  Same arithmetic rules as dual.hpp's D1.
#pragma once
#include <cmath>

namespace orc {

// CAP = compile-time number of partials (the block's variable count rounded up to 16 / 32 / 64 / 128): fixed trip counts,
// so the compiler vectorises every loop
template <int CAP>
struct DNT {
    static constexpr int kCap = CAP;
    double v;
    double d[CAP];
    DNT() : v(0.0) { for (int i = 0; i < CAP; ++i) d[i] = 0.0; }
    DNT(double v_) : v(v_) { for (int i = 0; i < CAP; ++i) d[i] = 0.0; }
    static DNT variable(double v_, int k) { DNT r(v_); r.d[k] = 1.0; return r; }
};

#define ORC_DN_T template <int CAP> inline DNT<CAP>
#define ORC_DN_LOOP for (int i = 0; i < CAP; ++i)
ORC_DN_T operator+(const DNT<CAP>& a, const DNT<CAP>& b) { DNT<CAP> r; r.v = a.v + b.v; ORC_DN_LOOP r.d[i] = a.d[i] + b.d[i]; return r; }
ORC_DN_T operator-(const DNT<CAP>& a, const DNT<CAP>& b) { DNT<CAP> r; r.v = a.v - b.v; ORC_DN_LOOP r.d[i] = a.d[i] - b.d[i]; return r; }
ORC_DN_T operator-(const DNT<CAP>& a) { DNT<CAP> r; r.v = -a.v; ORC_DN_LOOP r.d[i] = -a.d[i]; return r; }
ORC_DN_T operator*(const DNT<CAP>& a, const DNT<CAP>& b) { DNT<CAP> r; r.v = a.v * b.v; ORC_DN_LOOP r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
ORC_DN_T operator/(const DNT<CAP>& a, const DNT<CAP>& b) {
    const double q = a.v / b.v;
    DNT<CAP> r; r.v = q; ORC_DN_LOOP r.d[i] = (a.d[i] - q * b.d[i]) / b.v; return r;
}
ORC_DN_T operator+(const DNT<CAP>& a, double b) { DNT<CAP> r(a); r.v = a.v + b; return r; }
ORC_DN_T operator+(double a, const DNT<CAP>& b) { DNT<CAP> r(b); r.v = a + b.v; return r; }
ORC_DN_T operator-(const DNT<CAP>& a, double b) { DNT<CAP> r(a); r.v = a.v - b; return r; }
ORC_DN_T operator-(double a, const DNT<CAP>& b) { DNT<CAP> r; r.v = a - b.v; ORC_DN_LOOP r.d[i] = -b.d[i]; return r; }
ORC_DN_T operator*(const DNT<CAP>& a, double b) { DNT<CAP> r; r.v = a.v * b; ORC_DN_LOOP r.d[i] = a.d[i] * b; return r; }
ORC_DN_T operator*(double a, const DNT<CAP>& b) { DNT<CAP> r; r.v = a * b.v; ORC_DN_LOOP r.d[i] = a * b.d[i]; return r; }
ORC_DN_T operator/(const DNT<CAP>& a, double b) { DNT<CAP> r; r.v = a.v / b; ORC_DN_LOOP r.d[i] = a.d[i] / b; return r; }
ORC_DN_T operator/(double a, const DNT<CAP>& b) {
    const double q = a / b.v;
    DNT<CAP> r; r.v = q; ORC_DN_LOOP r.d[i] = -q * b.d[i] / b.v; return r;
}
template <int CAP> inline DNT<CAP>& operator+=(DNT<CAP>& a, const DNT<CAP>& b) { a = a + b; return a; }

ORC_DN_T exp(const DNT<CAP>& a) { const double e = std::exp(a.v); DNT<CAP> r; r.v = e; ORC_DN_LOOP r.d[i] = e * a.d[i]; return r; }
ORC_DN_T sin(const DNT<CAP>& a) { const double c = std::cos(a.v); DNT<CAP> r; r.v = std::sin(a.v); ORC_DN_LOOP r.d[i] = c * a.d[i]; return r; }
ORC_DN_T cos(const DNT<CAP>& a) { const double s = std::sin(a.v); DNT<CAP> r; r.v = std::cos(a.v); ORC_DN_LOOP r.d[i] = -s * a.d[i]; return r; }
ORC_DN_T sqrt(const DNT<CAP>& a) { const double s = std::sqrt(a.v); DNT<CAP> r; r.v = s; ORC_DN_LOOP r.d[i] = a.d[i] / (2.0 * s); return r; }
ORC_DN_T sq(const DNT<CAP>& a) { DNT<CAP> r; r.v = a.v * a.v; ORC_DN_LOOP r.d[i] = 2.0 * a.v * a.d[i]; return r; }
template <int CAP> inline double value(const DNT<CAP>& a) { return a.v; }
#undef ORC_DN_LOOP
#undef ORC_DN_T

}  // namespace orc
