// TEST INFRASTRUCTURE ONLY -- part of the CPU oracle (see oracle/ctd_oracle.cpp header).
//
// Sparse second-order forward number: value, sparse gradient and sparse (lower-triangular) Hessian with respect to
// the NLP variables.  Pushing it through the restated callbacks `objective` / `constraints` gives the Hessian of the
// Lagrangian the way ADNLPModels' sparse Hessian backend defines it for the reference: generic AD over the closures
// f(x) and c!(c, x) handed over at src/collocation.jl:137-149 (backend selection :121-125), evaluated on the pattern
// of DOCP_Hessian_pattern.  Every intermediate of one time step depends on a few dozen variables only, so the sparse
// containers stay short; the cost grows with the support of the running Lagrange sum (fine for the test sizes).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>

namespace orc {

struct S2 {
    using GVec = std::vector<std::pair<int64_t, double>>;   // (variable, d/dx_i), sorted by variable
    using HVec = std::vector<std::pair<uint64_t, double>>;  // (key(i, j) with i >= j, d2/dx_i dx_j), sorted by key
    double v;
    GVec g;
    HVec h;
    S2() : v(0.0) {}
    S2(double v_) : v(v_) {}
    static S2 variable(double v_, int64_t index) { S2 r(v_); r.g.emplace_back(index, 1.0); return r; }
    static uint64_t key(int64_t i, int64_t j) { return i >= j ? ((uint64_t)i << 32) | (uint64_t)j : ((uint64_t)j << 32) | (uint64_t)i; }
};

namespace s2detail {
template <class V> static V axpby(double ca, const V& a, double cb, const V& b) {
    V r;
    r.reserve(a.size() + b.size());
    size_t i = 0, j = 0;
    while (i < a.size() || j < b.size()) {
        if (j == b.size() || (i < a.size() && a[i].first < b[j].first)) { r.emplace_back(a[i].first, ca * a[i].second); ++i; }
        else if (i == a.size() || b[j].first < a[i].first) { r.emplace_back(b[j].first, cb * b[j].second); ++j; }
        else { r.emplace_back(a[i].first, ca * a[i].second + cb * b[j].second); ++i; ++j; }
    }
    return r;
}
template <class V> static V scaled(double c, const V& a) {
    V r(a);
    for (auto& e : r) e.second *= c;
    return r;
}
// w * (ga (x) gb + gb (x) ga), lower triangle, appended to `out` (unsorted)
static inline void outer_sym(double w, const S2::GVec& ga, const S2::GVec& gb, S2::HVec& out) {
    for (auto& a : ga)
        for (auto& b : gb) {
            const double t = w * a.second * b.second;
            out.emplace_back(S2::key(a.first, b.first), a.first == b.first ? 2.0 * t : t);
        }
}
// w * (g (x) g)
static inline void outer_self(double w, const S2::GVec& g, S2::HVec& out) {
    for (size_t i = 0; i < g.size(); ++i)
        for (size_t j = 0; j <= i; ++j) out.emplace_back(S2::key(g[i].first, g[j].first), w * g[i].second * g[j].second);
}
static inline void compress(S2::HVec& h) {
    std::sort(h.begin(), h.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
    size_t w = 0;
    for (size_t r = 0; r < h.size(); ++r) {
        if (w > 0 && h[w - 1].first == h[r].first) h[w - 1].second += h[r].second;
        else h[w++] = h[r];
    }
    h.resize(w);
}
// f(a) with derivatives f1 = f'(a.v), f2 = f''(a.v)
static inline S2 chain(const S2& a, double f0, double f1, double f2) {
    S2 r(f0);
    r.g = scaled(f1, a.g);
    r.h = scaled(f1, a.h);
    if (f2 != 0.0 && !a.g.empty()) { outer_self(f2, a.g, r.h); compress(r.h); }
    return r;
}
}  // namespace s2detail

inline S2 operator+(const S2& a, const S2& b) { S2 r(a.v + b.v); r.g = s2detail::axpby(1.0, a.g, 1.0, b.g); r.h = s2detail::axpby(1.0, a.h, 1.0, b.h); return r; }
inline S2 operator-(const S2& a, const S2& b) { S2 r(a.v - b.v); r.g = s2detail::axpby(1.0, a.g, -1.0, b.g); r.h = s2detail::axpby(1.0, a.h, -1.0, b.h); return r; }
inline S2 operator-(const S2& a) { S2 r(-a.v); r.g = s2detail::scaled(-1.0, a.g); r.h = s2detail::scaled(-1.0, a.h); return r; }
inline S2 operator*(const S2& a, const S2& b) {
    S2 r(a.v * b.v);
    r.g = s2detail::axpby(b.v, a.g, a.v, b.g);
    r.h = s2detail::axpby(b.v, a.h, a.v, b.h);
    if (!a.g.empty() && !b.g.empty()) { s2detail::outer_sym(1.0, a.g, b.g, r.h); s2detail::compress(r.h); }
    return r;
}
inline S2 recip(const S2& b) { const double q = 1.0 / b.v; return s2detail::chain(b, q, -q * q, 2.0 * q * q * q); }
inline S2 operator/(const S2& a, const S2& b) { S2 r = a * recip(b); r.v = a.v / b.v; return r; }
inline S2 operator+(const S2& a, double b) { S2 r = a; r.v = a.v + b; return r; }
inline S2 operator+(double a, const S2& b) { S2 r = b; r.v = a + b.v; return r; }
inline S2 operator-(const S2& a, double b) { S2 r = a; r.v = a.v - b; return r; }
inline S2 operator-(double a, const S2& b) { S2 r = -b; r.v = a - b.v; return r; }
inline S2 operator*(const S2& a, double b) { S2 r(a.v * b); r.g = s2detail::scaled(b, a.g); r.h = s2detail::scaled(b, a.h); return r; }
inline S2 operator*(double a, const S2& b) { return b * a; }
inline S2 operator/(const S2& a, double b) { S2 r = a * (1.0 / b); r.v = a.v / b; return r; }
inline S2 operator/(double a, const S2& b) { S2 r = recip(b) * a; r.v = a / b.v; return r; }
inline S2& operator+=(S2& a, const S2& b) { a = a + b; return a; }

inline S2 exp(const S2& a) { const double e = std::exp(a.v); return s2detail::chain(a, e, e, e); }
inline S2 sin(const S2& a) { const double s = std::sin(a.v), c = std::cos(a.v); return s2detail::chain(a, s, c, -s); }
inline S2 cos(const S2& a) { const double s = std::sin(a.v), c = std::cos(a.v); return s2detail::chain(a, c, -s, -c); }
inline S2 sqrt(const S2& a) { const double s = std::sqrt(a.v); return s2detail::chain(a, s, 0.5 / s, -0.25 / (s * a.v)); }
inline S2 sq(const S2& a) { return s2detail::chain(a, a.v * a.v, 2.0 * a.v, 2.0); }
inline double value(const S2& a) { return a.v; }

}  // namespace orc
