// TEST INFRASTRUCTURE ONLY -- part of the CPU oracle (see oracle/ctd_oracle.cpp header).
//
// OCP definitions used by the oracle, restated from the reference's problem files
// (/root/reference/test/problems/*.jl, test/ci/test_discretization_stagewise.jl).  They are written
// generically over the scalar type (double or orc::D1), exactly as the reference's Julia closures
// are generic over Float64 / ForwardDiff.Dual.  The signatures follow the CTModels conventions
// quoted in src/DOCP_functions.jl:105-111,133-138 and src/ode/trapeze.jl:66:
//     dynamics!(r, t, x, u, v)   lagrange(t, x, u, v)   mayer(x0, xf, v)
//     path!(r, t, x, u, v)       boundary!(r, x0, xf, v)
// These definitions are deliberately NOT shared with the product (ctdirect.jl_amd/csrc/): the HIP
// engine has its own device functors, so a mistake in either shows up as a parity failure.
#pragma once
#include <cmath>
#include <limits>
#include <vector>
#include "dual.hpp"

namespace orc {

static const double INF = std::numeric_limits<double>::infinity();

struct BoxEntry { int index; double lb, ub; };  // (lb, index, ub) triplets, src/DOCP_variables.jl:88-98

// ---------------------------------------------------------------------------------------------
// id 0: Goddard, abstract (@def) form.   test/problems/goddard.jl:7-49
//   F0 = [v, -D/m - 1/r^2, 0], F1 = [0, Tmax/m, -b*Tmax], D = Cd v^2 exp(-beta (r-1))    :7-16
//   xdot = F0(x) + u F1(x) :44 ; tf free (variable) :27-28 ; maximise r(tf) :45
//   boxes :35-42 ; boundary x(0)==x0 (3 rows), m(tf)==mf :38-39  (row order as in goddard_all :134-139)
// ---------------------------------------------------------------------------------------------
struct Goddard {
    static constexpr int n = 3, m = 1, nv = 1, p = 0, bc = 4;
    static constexpr bool freet0 = false, freetf = true, has_lagrange = false, has_mayer = true, maximize = true;
    static constexpr double Cd = 310, beta = 500, b = 2, Tmax = 3.5, r0 = 1, v0 = 0, m0 = 1, mf = 0.6, vmax = 0.1;
    template <class T> static T t0(const T*) { return T(0.0); }
    template <class T> static T tf(const T* v) { return v[0]; }
    template <class T> static void dynamics(T* r, const T&, const T* x, const T* u, const T*) {
        const T& rr = x[0]; const T& vv = x[1]; const T& mm = x[2];
        T D = Cd * sq(vv) * exp(-beta * (rr - 1.0));
        T f0[3] = {vv, -D / mm - 1.0 / sq(rr), T(0.0)};
        T f1[3] = {T(0.0), Tmax / mm, T(-b * Tmax)};
        for (int i = 0; i < 3; ++i) r[i] = f0[i] + u[0] * f1[i];
    }
    template <class T> static T lagrange(const T&, const T*, const T*, const T*) { return T(0.0); }
    template <class T> static T mayer(const T*, const T* xf, const T*) { return xf[0]; }
    template <class T> static void path(T*, const T&, const T*, const T*, const T*) {}
    template <class T> static void boundary(T* r, const T* x0, const T* xf, const T*) {
        r[0] = x0[0]; r[1] = x0[1]; r[2] = x0[2]; r[3] = xf[2];
    }
    static void path_bounds(double*, double*) {}
    static void boundary_bounds(double* lb, double* ub) {
        const double b_[4] = {r0, v0, m0, mf};
        for (int i = 0; i < 4; ++i) lb[i] = ub[i] = b_[i];
    }
    static std::vector<BoxEntry> state_box() { return {{0, r0, r0 + 0.1}, {1, v0, vmax}, {2, mf, m0}}; }
    static std::vector<BoxEntry> control_box() { return {{0, 0.0, 1.0}}; }
    static std::vector<BoxEntry> variable_box() { return {{0, 0.01, INF}}; }
    // init=(state=[1.01, 0.05, 0.8],)  :48
    static bool init_state(double, double* x) { x[0] = 1.01; x[1] = 0.05; x[2] = 0.8; return true; }
    static bool init_control(double, double*) { return false; }
    static bool init_variable(double*) { return false; }
};

// ---------------------------------------------------------------------------------------------
// id 1: goddard_all -- functional in-place form with 3 nonlinear path constraints.
//   test/problems/goddard.jl:87-158 : f! :127-132, path! :117-121 (lb/ub :122-124), bc! :134-146,
//   boxes :104-113, mayer xf[1] maximised :125-126
// ---------------------------------------------------------------------------------------------
struct GoddardAll {
    static constexpr int n = 3, m = 1, nv = 1, p = 3, bc = 4;
    static constexpr bool freet0 = false, freetf = true, has_lagrange = false, has_mayer = true, maximize = true;
    static constexpr double Cd = 310, beta = 500, b = 2, Tmax = 3.5, r0 = 1, v0 = 0, m0 = 1, mf = 0.6, vmax = 0.1;
    template <class T> static T t0(const T*) { return T(0.0); }
    template <class T> static T tf(const T* v) { return v[0]; }
    template <class T> static void dynamics(T* r, const T&, const T* x, const T* u, const T*) {
        r[0] = x[1];
        T D = Cd * sq(x[1]) * exp(-beta * (x[0] - 1.0));
        r[1] = -D / x[2] - 1.0 / sq(x[0]) + u[0] * Tmax / x[2];
        r[2] = -b * Tmax * u[0];
    }
    template <class T> static T lagrange(const T&, const T*, const T*, const T*) { return T(0.0); }
    template <class T> static T mayer(const T*, const T* xf, const T*) { return xf[0]; }
    template <class T> static void path(T* r, const T&, const T* x, const T* u, const T* v) {
        r[0] = x[1];
        r[1] = u[0];
        r[2] = x[0] + x[1] + x[2] + u[0] + v[0];
    }
    template <class T> static void boundary(T* r, const T* x0, const T* xf, const T*) {
        r[0] = x0[0]; r[1] = x0[1]; r[2] = x0[2]; r[3] = xf[2];
    }
    static void path_bounds(double* lb, double* ub) {
        lb[0] = -INF; lb[1] = -INF; lb[2] = 0.0;
        ub[0] = vmax; ub[1] = 1.0; ub[2] = INF;
    }
    static void boundary_bounds(double* lb, double* ub) {
        const double b_[4] = {r0, v0, m0, mf};
        for (int i = 0; i < 4; ++i) lb[i] = ub[i] = b_[i];
    }
    static std::vector<BoxEntry> state_box() { return {{0, r0, INF}, {1, v0, INF}, {2, 0.0, m0}}; }
    static std::vector<BoxEntry> control_box() { return {{0, 0.0, INF}}; }
    static std::vector<BoxEntry> variable_box() { return {{0, 0.01, INF}}; }
    static bool init_state(double, double* x) { x[0] = 1.01; x[1] = 0.05; x[2] = 0.8; return true; }
    static bool init_control(double, double*) { return false; }
    static bool init_variable(double*) { return false; }
};

// ---------------------------------------------------------------------------------------------
// id 2: double integrator with one nonlinear path constraint (BASELINE config 3).
//   Base problem: double_integrator_minenergy, test/problems/double_integrator.jl:42-58
//   (T = 2, xdot = [v, u], q(0)=0, v(0)=0, q(T)=1, v(T)=0, min int u^2).
//   The reference has NO double-integrator variant with a nonlinear path constraint (SURVEY section 8
//   note); the build adds  g = q + 0.1 w^2 <= 1.05  and a control box -5 <= u <= 5.
// ---------------------------------------------------------------------------------------------
struct DoubleIntegratorPath {
    static constexpr int n = 2, m = 1, nv = 0, p = 1, bc = 4;
    static constexpr bool freet0 = false, freetf = false, has_lagrange = true, has_mayer = false, maximize = false;
    template <class T> static T t0(const T*) { return T(0.0); }
    template <class T> static T tf(const T*) { return T(2.0); }
    template <class T> static void dynamics(T* r, const T&, const T* x, const T* u, const T*) {
        r[0] = x[1];
        r[1] = u[0];
    }
    template <class T> static T lagrange(const T&, const T*, const T* u, const T*) { return sq(u[0]); }
    template <class T> static T mayer(const T*, const T*, const T*) { return T(0.0); }
    template <class T> static void path(T* r, const T&, const T* x, const T*, const T*) {
        r[0] = x[0] + 0.1 * sq(x[1]);
    }
    template <class T> static void boundary(T* r, const T* x0, const T* xf, const T*) {
        r[0] = x0[0]; r[1] = x0[1]; r[2] = xf[0]; r[3] = xf[1];
    }
    static void path_bounds(double* lb, double* ub) { lb[0] = -INF; ub[0] = 1.05; }
    static void boundary_bounds(double* lb, double* ub) {
        const double b_[4] = {0.0, 0.0, 1.0, 0.0};
        for (int i = 0; i < 4; ++i) lb[i] = ub[i] = b_[i];
    }
    static std::vector<BoxEntry> state_box() { return {}; }
    static std::vector<BoxEntry> control_box() { return {{0, -5.0, 5.0}}; }
    static std::vector<BoxEntry> variable_box() { return {}; }
    static bool init_state(double, double*) { return false; }
    static bool init_control(double, double*) { return false; }
    static bool init_variable(double*) { return false; }
};

// ---------------------------------------------------------------------------------------------
// id 3: quadrotor, the reference's 8-state / 4-control model.  test/problems/quadrotor.jl:7-105
//   dynamics :20-40 (rotation-matrix thrust), boxes :66-73, path cos(theta)cos(phi) >= cos(tiltmax) :75,
//   8 initial + 6 final conditions :77-92, cost tf + int(1e-8(phi^2+theta^2+psi^2+at^2) + 1e2 (psi-u0[3])^2) :98
// ---------------------------------------------------------------------------------------------
struct Quadrotor8 {
    static constexpr int n = 8, m = 4, nv = 1, p = 1, bc = 14;
    static constexpr bool freet0 = false, freetf = true, has_lagrange = true, has_mayer = true, maximize = false;
    static constexpr double g = 9.81, atmin = 0.0, atmax = 9.18 * 5, tiltmax = 1.1 / 2, dtiltmax = 6.0 / 2;
    template <class T> static T t0(const T*) { return T(0.0); }
    template <class T> static T tf(const T* v) { return v[0]; }
    template <class T> static void dynamics(T* r, const T&, const T* x, const T* u, const T*) {
        const T& v1 = x[3]; const T& v2 = x[4]; const T& v3 = x[5]; const T& phi = x[6]; const T& th = x[7];
        const T& at = u[0]; const T& phid = u[1]; const T& thd = u[2]; const T& psi = u[3];
        T cr = cos(phi), sr = sin(phi), cp = cos(th), sp = sin(th), cy = cos(psi), sy = sin(psi);
        // third column of R (the only one multiplied by a nonzero entry of [0;0;at])
        T R13 = cy * sp * cr + sy * sr;
        T R23 = sy * sp * cr - cy * sr;
        T R33 = cp * cr;
        r[0] = v1; r[1] = v2; r[2] = v3;
        r[3] = 0.0 + R13 * at;
        r[4] = 0.0 + R23 * at;
        r[5] = -g + R33 * at;
        r[6] = phid; r[7] = thd;
    }
    template <class T> static T lagrange(const T&, const T* x, const T* u, const T*) {
        return 1e-8 * (sq(x[6]) + sq(x[7]) + sq(u[3]) + sq(u[0])) + (1e2 * sq(u[3] - 0.0));
    }
    template <class T> static T mayer(const T*, const T*, const T* v) { return v[0]; }
    template <class T> static void path(T* r, const T&, const T* x, const T*, const T*) {
        r[0] = cos(x[7]) * cos(x[6]);
    }
    template <class T> static void boundary(T* r, const T* x0, const T* xf, const T*) {
        for (int i = 0; i < 8; ++i) r[i] = x0[i];
        for (int i = 0; i < 6; ++i) r[8 + i] = xf[i];
    }
    static void path_bounds(double* lb, double* ub) { lb[0] = std::cos(tiltmax); ub[0] = INF; }
    static void boundary_bounds(double* lb, double* ub) {
        const double b_[14] = {0.0, 0.0, 2.5, 0, 0, 0, 0, 0, 0.01, 5.0, 2.5, 0.0, 0.0, 0.0};
        for (int i = 0; i < 14; ++i) lb[i] = ub[i] = b_[i];
    }
    static std::vector<BoxEntry> state_box() { return {{6, -M_PI / 2, M_PI / 2}, {7, -M_PI / 2, M_PI / 2}}; }
    static std::vector<BoxEntry> control_box() {
        return {{0, atmin, atmax}, {1, -dtiltmax, dtiltmax}, {2, -dtiltmax, dtiltmax}};
    }
    static std::vector<BoxEntry> variable_box() { return {{0, 0.1, INF}}; }
    // x_init = t -> [p0 + (pf-p0) t; v0 + (vf-v0) t; u0[2:3]], control [10,0,0,0], variable [1.0]  :101-102
    static bool init_state(double t, double* x) {
        const double p0[3] = {0.0, 0.0, 2.5}, pf[3] = {0.01, 5.0, 2.5};
        for (int i = 0; i < 3; ++i) x[i] = p0[i] + (pf[i] - p0[i]) * t;
        for (int i = 3; i < 8; ++i) x[i] = 0.0;
        return true;
    }
    static bool init_control(double, double* u) { u[0] = 10.0; u[1] = u[2] = u[3] = 0.0; return true; }
    static bool init_variable(double* v) { v[0] = 1.0; return true; }
};

// ---------------------------------------------------------------------------------------------
// id 4: quadrotor, 12-state / 4-control rigid body (BASELINE config 5).  NOT in the reference
//   (its quadrotor is id 3); defined by the build, see DESIGN.md "quadrotor-12".
//   x = (p[3], v[3], phi, theta, psi, w[3]), u = (at, tau[3]), v = (tf)
// ---------------------------------------------------------------------------------------------
struct Quadrotor12 {
    static constexpr int n = 12, m = 4, nv = 1, p = 1, bc = 23;
    static constexpr bool freet0 = false, freetf = true, has_lagrange = true, has_mayer = true, maximize = false;
    static constexpr double g = 9.81, Jx = 0.03, Jy = 0.03, Jz = 0.06, tiltmax = 0.55;
    template <class T> static T t0(const T*) { return T(0.0); }
    template <class T> static T tf(const T* v) { return v[0]; }
    template <class T> static void dynamics(T* r, const T&, const T* x, const T* u, const T*) {
        const T& phi = x[6]; const T& th = x[7]; const T& psi = x[8];
        const T& w1 = x[9]; const T& w2 = x[10]; const T& w3 = x[11];
        const T& at = u[0];
        T cr = cos(phi), sr = sin(phi), cp = cos(th), sp = sin(th), cy = cos(psi), sy = sin(psi);
        T R13 = cy * sp * cr + sy * sr;
        T R23 = sy * sp * cr - cy * sr;
        T R33 = cp * cr;
        r[0] = x[3]; r[1] = x[4]; r[2] = x[5];
        r[3] = R13 * at;
        r[4] = R23 * at;
        r[5] = -g + R33 * at;
        T tp = sp / cp;
        r[6] = w1 + sr * tp * w2 + cr * tp * w3;
        r[7] = cr * w2 - sr * w3;
        r[8] = (sr * w2 + cr * w3) / cp;
        r[9] = ((Jy - Jz) * w2 * w3 + u[1]) / Jx;
        r[10] = ((Jz - Jx) * w3 * w1 + u[2]) / Jy;
        r[11] = ((Jx - Jy) * w1 * w2 + u[3]) / Jz;
    }
    template <class T> static T lagrange(const T&, const T* x, const T* u, const T*) {
        return 1e-8 * (sq(x[6]) + sq(x[7]) + sq(u[0])) + 1e-2 * (sq(u[1]) + sq(u[2]) + sq(u[3])) + 1e2 * sq(x[8]);
    }
    template <class T> static T mayer(const T*, const T*, const T* v) { return v[0]; }
    template <class T> static void path(T* r, const T&, const T* x, const T*, const T*) {
        r[0] = cos(x[7]) * cos(x[6]);
    }
    template <class T> static void boundary(T* r, const T* x0, const T* xf, const T*) {
        for (int i = 0; i < 12; ++i) r[i] = x0[i];
        for (int i = 0; i < 8; ++i) r[12 + i] = xf[i];          // p, v, phi, theta at tf
        for (int i = 0; i < 3; ++i) r[20 + i] = xf[9 + i];      // body rates at tf (psi(tf) free)
    }
    static void path_bounds(double* lb, double* ub) { lb[0] = std::cos(tiltmax); ub[0] = INF; }
    static void boundary_bounds(double* lb, double* ub) {
        double b_[23] = {0};
        b_[2] = 2.5;                             // p3(0)
        b_[12] = 0.01; b_[13] = 5.0; b_[14] = 2.5;  // p(tf)
        for (int i = 0; i < 23; ++i) lb[i] = ub[i] = b_[i];
    }
    static std::vector<BoxEntry> state_box() { return {{6, -M_PI / 2, M_PI / 2}, {7, -M_PI / 2, M_PI / 2}}; }
    static std::vector<BoxEntry> control_box() {
        return {{0, 0.0, 45.9}, {1, -1.0, 1.0}, {2, -1.0, 1.0}, {3, -1.0, 1.0}};
    }
    static std::vector<BoxEntry> variable_box() { return {{0, 0.1, INF}}; }
    static bool init_state(double t, double* x) {
        const double p0[3] = {0.0, 0.0, 2.5}, pf[3] = {0.01, 5.0, 2.5};
        for (int i = 0; i < 3; ++i) x[i] = p0[i] + (pf[i] - p0[i]) * t;
        for (int i = 3; i < 12; ++i) x[i] = 0.0;
        return true;
    }
    static bool init_control(double, double* u) { u[0] = 10.0; u[1] = u[2] = u[3] = 0.0; return true; }
    static bool init_variable(double* v) { v[0] = 1.0; return true; }
};

// ---------------------------------------------------------------------------------------------
// id 5: scalar stagewise test problem.  test/ci/test_discretization_stagewise.jl:1-14
//   t in [0,1], xdot = u, 0 <= u <= 2, x(0)=0, x(1)=1, min int u^2
// ---------------------------------------------------------------------------------------------
struct StagewiseScalar {
    static constexpr int n = 1, m = 1, nv = 0, p = 0, bc = 2;
    static constexpr bool freet0 = false, freetf = false, has_lagrange = true, has_mayer = false, maximize = false;
    template <class T> static T t0(const T*) { return T(0.0); }
    template <class T> static T tf(const T*) { return T(1.0); }
    template <class T> static void dynamics(T* r, const T&, const T*, const T* u, const T*) { r[0] = u[0]; }
    template <class T> static T lagrange(const T&, const T*, const T* u, const T*) { return sq(u[0]); }
    template <class T> static T mayer(const T*, const T*, const T*) { return T(0.0); }
    template <class T> static void path(T*, const T&, const T*, const T*, const T*) {}
    template <class T> static void boundary(T* r, const T* x0, const T* xf, const T*) { r[0] = x0[0]; r[1] = xf[0]; }
    static void path_bounds(double*, double*) {}
    static void boundary_bounds(double* lb, double* ub) { lb[0] = ub[0] = 0.0; lb[1] = ub[1] = 1.0; }
    static std::vector<BoxEntry> state_box() { return {}; }
    static std::vector<BoxEntry> control_box() { return {{0, 0.0, 2.0}}; }
    static std::vector<BoxEntry> variable_box() { return {}; }
    static bool init_state(double, double*) { return false; }
    static bool init_control(double, double*) { return false; }
    static bool init_variable(double*) { return false; }
};

// ---------------------------------------------------------------------------------------------
// id 6: estimate_initial_condition (zero control).  test/problems/autonomous_system.jl:6-43
//   t in [0, pi/2], xdot = [-x2, x1], v = x(0) free, boundary x0 - v = 0, mayer (xf1)^2 + (xf2-1)^2
// ---------------------------------------------------------------------------------------------
struct EstimateInitialCondition {
    static constexpr int n = 2, m = 0, nv = 2, p = 0, bc = 2;
    static constexpr bool freet0 = false, freetf = false, has_lagrange = false, has_mayer = true, maximize = false;
    template <class T> static T t0(const T*) { return T(0.0); }
    template <class T> static T tf(const T*) { return T(M_PI / 2); }
    template <class T> static void dynamics(T* r, const T&, const T* x, const T*, const T*) { r[0] = -x[1]; r[1] = x[0]; }
    template <class T> static T lagrange(const T&, const T*, const T*, const T*) { return T(0.0); }
    template <class T> static T mayer(const T*, const T* xf, const T*) { return sq(xf[0] - 0.0) + sq(xf[1] - 1.0); }
    template <class T> static void path(T*, const T&, const T*, const T*, const T*) {}
    template <class T> static void boundary(T* r, const T* x0, const T*, const T* v) { r[0] = x0[0] - v[0]; r[1] = x0[1] - v[1]; }
    static void path_bounds(double*, double*) {}
    static void boundary_bounds(double* lb, double* ub) { lb[0] = ub[0] = 0.0; lb[1] = ub[1] = 0.0; }
    static std::vector<BoxEntry> state_box() { return {}; }
    static std::vector<BoxEntry> control_box() { return {}; }
    static std::vector<BoxEntry> variable_box() { return {}; }
    static bool init_state(double, double*) { return false; }
    static bool init_control(double, double*) { return false; }
    static bool init_variable(double*) { return false; }
};

// ---------------------------------------------------------------------------------------------
// id 7: estimate_rotation_rate (zero control, dynamics depend on v).  autonomous_system.jl:46-87
//   t in [0,1], xdot = alpha [-x2, x1], 0 <= alpha <= 10, boundary x0 - [1,0] = 0,
//   mayer (xf1)^2 + (xf2-1)^2 + 0.01 alpha^2
// ---------------------------------------------------------------------------------------------
struct EstimateRotationRate {
    static constexpr int n = 2, m = 0, nv = 1, p = 0, bc = 2;
    static constexpr bool freet0 = false, freetf = false, has_lagrange = false, has_mayer = true, maximize = false;
    template <class T> static T t0(const T*) { return T(0.0); }
    template <class T> static T tf(const T*) { return T(1.0); }
    template <class T> static void dynamics(T* r, const T&, const T* x, const T*, const T* v) {
        r[0] = v[0] * (-x[1]);
        r[1] = v[0] * x[0];
    }
    template <class T> static T lagrange(const T&, const T*, const T*, const T*) { return T(0.0); }
    template <class T> static T mayer(const T*, const T* xf, const T* v) {
        return sq(xf[0] - 0.0) + sq(xf[1] - 1.0) + 0.01 * sq(v[0]);
    }
    template <class T> static void path(T*, const T&, const T*, const T*, const T*) {}
    template <class T> static void boundary(T* r, const T* x0, const T*, const T*) { r[0] = x0[0] - 1.0; r[1] = x0[1] - 0.0; }
    static void path_bounds(double*, double*) {}
    static void boundary_bounds(double* lb, double* ub) { lb[0] = ub[0] = 0.0; lb[1] = ub[1] = 0.0; }
    static std::vector<BoxEntry> state_box() { return {}; }
    static std::vector<BoxEntry> control_box() { return {}; }
    static std::vector<BoxEntry> variable_box() { return {{0, 0.0, 10.0}}; }
    static bool init_state(double, double*) { return false; }
    static bool init_control(double, double*) { return false; }
    static bool init_variable(double*) { return false; }
};

// ---------------------------------------------------------------------------------------------
// id 8: least_squares_with_constraint (zero control, time-dependent Lagrange, true :path constraint).
//   autonomous_system.jl:90-138: t in [0,1], xdot = [-x2, x1], boundary x0 - v = 0,
//   path x1^2 + x2^2 <= 2, cost int (t-0.5)^2((x1-0.7)^2+(x2-0.7)^2) + 0.01 (v1^2+v2^2)
// ---------------------------------------------------------------------------------------------
struct LeastSquaresConstraint {
    static constexpr int n = 2, m = 0, nv = 2, p = 1, bc = 2;
    static constexpr bool freet0 = false, freetf = false, has_lagrange = true, has_mayer = true, maximize = false;
    template <class T> static T t0(const T*) { return T(0.0); }
    template <class T> static T tf(const T*) { return T(1.0); }
    template <class T> static void dynamics(T* r, const T&, const T* x, const T*, const T*) { r[0] = -x[1]; r[1] = x[0]; }
    template <class T> static T lagrange(const T& t, const T* x, const T*, const T*) {
        return sq(t - 0.5) * (sq(x[0] - 0.7) + sq(x[1] - 0.7));
    }
    template <class T> static T mayer(const T*, const T*, const T* v) { return 0.01 * (sq(v[0]) + sq(v[1])); }
    template <class T> static void path(T* r, const T&, const T* x, const T*, const T*) { r[0] = sq(x[0]) + sq(x[1]); }
    template <class T> static void boundary(T* r, const T* x0, const T*, const T* v) { r[0] = x0[0] - v[0]; r[1] = x0[1] - v[1]; }
    static void path_bounds(double* lb, double* ub) { lb[0] = -INF; ub[0] = 2.0; }
    static void boundary_bounds(double* lb, double* ub) { lb[0] = ub[0] = 0.0; lb[1] = ub[1] = 0.0; }
    static std::vector<BoxEntry> state_box() { return {}; }
    static std::vector<BoxEntry> control_box() { return {}; }
    static std::vector<BoxEntry> variable_box() { return {}; }
    static bool init_state(double, double*) { return false; }
    static bool init_control(double, double*) { return false; }
    static bool init_variable(double*) { return false; }
};

// ---------------------------------------------------------------------------------------------
// id 9: double integrator, free t0 and tf (nv = 2).  test/problems/double_integrator.jl:79-99
//   t in [v1, v2], xdot = [x2, u], -1<=u<=1, x(t0)=[0,0], x(tf)=[1,0], 0.05<=t0,tf<=10, maximise t0.
//   The reference's extra linear constraint 0.01 <= tf - t0 is carried as a 5th boundary row
//   (how CTModels files it is not visible from the reference repo; build's choice).
// ---------------------------------------------------------------------------------------------
struct DoubleIntegratorFreeT0Tf {
    static constexpr int n = 2, m = 1, nv = 2, p = 0, bc = 5;
    static constexpr bool freet0 = true, freetf = true, has_lagrange = false, has_mayer = true, maximize = true;
    template <class T> static T t0(const T* v) { return v[0]; }
    template <class T> static T tf(const T* v) { return v[1]; }
    template <class T> static void dynamics(T* r, const T&, const T* x, const T* u, const T*) { r[0] = x[1]; r[1] = u[0]; }
    template <class T> static T lagrange(const T&, const T*, const T*, const T*) { return T(0.0); }
    template <class T> static T mayer(const T*, const T*, const T* v) { return v[0]; }
    template <class T> static void path(T*, const T&, const T*, const T*, const T*) {}
    template <class T> static void boundary(T* r, const T* x0, const T* xf, const T* v) {
        r[0] = x0[0]; r[1] = x0[1]; r[2] = xf[0]; r[3] = xf[1]; r[4] = v[1] - v[0];
    }
    static void path_bounds(double*, double*) {}
    static void boundary_bounds(double* lb, double* ub) {
        const double b_[4] = {0.0, 0.0, 1.0, 0.0};
        for (int i = 0; i < 4; ++i) lb[i] = ub[i] = b_[i];
        lb[4] = 0.01; ub[4] = INF;
    }
    static std::vector<BoxEntry> state_box() { return {}; }
    static std::vector<BoxEntry> control_box() { return {{0, -1.0, 1.0}}; }
    static std::vector<BoxEntry> variable_box() { return {{0, 0.05, 10.0}, {1, 0.05, 10.0}}; }
    static bool init_state(double, double*) { return false; }
    static bool init_control(double, double*) { return false; }
    static bool init_variable(double*) { return false; }
};

// ---------------------------------------------------------------------------------------------
// id 10 (oracle only; the engine takes it as a run-time OCP): goddard_all with the dynamics of the abstract form,
//   xdot = F0(x) + u F1(x)  (test/problems/goddard.jl:7-15,44) instead of the f! of goddard.jl:127-132.  The two forms
//   compute the same numbers; their TRACED patterns differ: `u * 0` in the r-row makes that row depend on the control at the
//   operator level, so a trapeze step has 28 Jacobian entries instead of 26 -- the count of the archived table
//   test/archives/AD_backend.md:63 (28011 at N = 1000, 280011 at N = 10000), which predates the current f!.
// ---------------------------------------------------------------------------------------------
struct GoddardAllF0F1 : GoddardAll {
    template <class T> static void dynamics(T* r, const T&, const T* x, const T* u, const T*) {
        const T& rr = x[0]; const T& vv = x[1]; const T& mm = x[2];
        T D = Cd * sq(vv) * exp(-beta * (rr - 1.0));
        T f0[3] = {vv, -D / mm - 1.0 / sq(rr), T(0.0)};
        T f1[3] = {T(0.0), Tmax / mm, T(-b * Tmax)};
        for (int i = 0; i < 3; ++i) r[i] = f0[i] + u[0] * f1[i];
    }
};

// ---------------------------------------------------------------------------------------------
// id 11 (oracle only; the engine takes it as a run-time OCP): algal-bacterial consortium.
//   test/problems/algal_bacterial.jl:3-52: parameters :6-15, phi / rho / mu :17-19, f :26-33, x(t0) == x0 :40,
//   x >= [0,0,0,qmin,0,0] :41, 0 <= u <= [1, dmax] :42, maximise x6(tf) :46; t in [0, 20].
//   The only problem the reference publishes per-configuration tables for (test/archives/jump_ctdirect.md:41-67).
// ---------------------------------------------------------------------------------------------
struct AlgalBacterial {
    static constexpr int n = 6, m = 2, nv = 0, p = 0, bc = 6;
    static constexpr bool freet0 = false, freetf = false, has_lagrange = false, has_mayer = true, maximize = true;
    static constexpr double s_in = 0.5, beta = 23e-3, gamma = 0.44, dmax = 1.5, phimax = 6.48, ks = 0.09, rhomax = 27.3e-3,
                            kv = 0.57e-3, mumax = 1.0211, qmin = 2.7628e-3;
    template <class T> static T t0(const T*) { return T(0.0); }
    template <class T> static T tf(const T*) { return T(20.0); }
    template <class T> static T phi(const T& s) { return phimax * s / (ks + s); }
    template <class T> static T rho(const T& v) { return rhomax * v / (kv + v); }
    template <class T> static T mu(const T& q) { return mumax * (1.0 - qmin / q); }
    template <class T> static void dynamics(T* r, const T&, const T* x, const T* u, const T*) {
        const T& al = u[0]; const T& d = u[1];
        r[0] = d * (s_in - x[0]) - phi(x[0]) * x[1] / gamma;
        r[1] = ((1.0 - al) * phi(x[0]) - d) * x[1];
        r[2] = al * beta * phi(x[0]) * x[1] - rho(x[2]) * x[4] - d * x[2];
        r[3] = rho(x[2]) - mu(x[3]) * x[3];
        r[4] = (mu(x[3]) - d) * x[4];
        r[5] = d * x[4];
    }
    template <class T> static T lagrange(const T&, const T*, const T*, const T*) { return T(0.0); }
    template <class T> static T mayer(const T*, const T* xf, const T*) { return xf[5]; }
    template <class T> static void path(T*, const T&, const T*, const T*, const T*) {}
    template <class T> static void boundary(T* r, const T* x0, const T*, const T*) { for (int i = 0; i < 6; ++i) r[i] = x0[i]; }
    static void path_bounds(double*, double*) {}
    static void boundary_bounds(double* lb, double* ub) {
        const double b_[6] = {0.1629, 0.0487, 0.0003, 0.0177, 0.035, 0.0};
        for (int i = 0; i < 6; ++i) lb[i] = ub[i] = b_[i];
    }
    static std::vector<BoxEntry> state_box() { return {{0, 0.0, INF}, {1, 0.0, INF}, {2, 0.0, INF}, {3, qmin, INF}, {4, 0.0, INF}, {5, 0.0, INF}}; }
    static std::vector<BoxEntry> control_box() { return {{0, 0.0, 1.0}, {1, 0.0, dmax}}; }
    static std::vector<BoxEntry> variable_box() { return {}; }
    static bool init_state(double, double*) { return false; }
    static bool init_control(double, double*) { return false; }
    static bool init_variable(double*) { return false; }
};

}  // namespace orc
