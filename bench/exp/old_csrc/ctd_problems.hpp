// ctd_problems.hpp -- compiled OCP registry of the engine (device functors + host-side static data).
//
// In the reference the OCP functions are arbitrary Julia closures obtained from CTModels:
//   CTModels.dynamics(ocp)(dx, t, x, u, v)                 called at src/ode/trapeze.jl:66, midpoint.jl:64,
//                                                          irk.jl:291, irk_stagewise.jl:441
//   CTModels.lagrange(ocp)(t, x, u, v), CTModels.mayer(ocp)(x0, xf, v)      src/DOCP_functions.jl:35-48
//   CTModels.path_constraints_nl(ocp)[2](c, t, x, u, v)                    src/DOCP_functions.jl:136-138
//   CTModels.boundary_constraints_nl(ocp)[2](c, x0, xf, v)                 src/DOCP_functions.jl:108-110
// Closures cannot cross a C ABI onto a GPU, so the engine ships a registry of problems as C++ functors that are
// generic over the scalar type (double or ctd::Dual<K>), each citing the problem file it restates.
//
// Per problem:  dims (DOCPdims, src/DOCP_data.jl:88-94), flags (DOCPFlags, :24-30), where t0/tf live
// (CTModels.initial_time / final_time, src/DOCP_data.jl:445-454), explicit-dependence traits used to size the
// dual passes, the five functions above, and host-only static data (boxes, bounds, init tuple).
#pragma once
#include <limits>
#include <vector>
#include "ctd_common.hpp"

// dual directions evaluated per pass by one lane (per-problem default; -DCTD_DC_OVERRIDE=k for tuning experiments)
#ifdef CTD_DC_OVERRIDE
#define CTD_DC(dflt) (CTD_DC_OVERRIDE)
#else
#define CTD_DC(dflt) (dflt)
#endif

namespace ctd {

struct BoxItem { int index; double lb, ub; };   // (lb, index, ub) triplets, src/DOCP_variables.jl:88-98

// host-side static description of a registry entry
struct ProblemInfo {
    const char* name;
    int n, m, nv, npath, nbc;
    int it0, itf;            // index of t0 / tf inside v (0-based), -1 when fixed
    double t0, tf;           // fixed values (used when the index is -1)
    bool lagrange, mayer, maximize;
    std::vector<BoxItem> state_box, control_box, variable_box;
    std::vector<double> path_lb, path_ub, bc_lb, bc_ub;
    // problem file's own `init` tuple: returns false when the entry is not provided
    bool (*init_state)(double t, double* x);
    bool (*init_control)(double t, double* u);
    bool (*init_variable)(double* v);
};

static const double kInf = std::numeric_limits<double>::infinity();
inline bool no_init_t(double, double*) { return false; }
inline bool no_init_v(double*) { return false; }

// -------------------------------------------------------------------------------------------------------
// Goddard rocket, abstract form: test/problems/goddard.jl:7-49
// -------------------------------------------------------------------------------------------------------
struct GoddardOCP {
    static constexpr int NX = 3, NU = 1, NV = 1, NPATH = 0, NBC = 4;
    static constexpr int IT0 = -1, ITF = 0;
    static constexpr bool HAS_LAGRANGE = false, HAS_MAYER = true;
    static constexpr bool DYN_T = false, DYN_V = false, PATH_T = false, PATH_V = false;
    static constexpr bool LAG_T = false, LAG_V = false;      // explicit dependence of the Lagrange cost on t / v
    static constexpr int DC = CTD_DC(4);                     // dual directions per pass
    static constexpr int MAXB = 1024;                       // largest workgroup the kernels are compiled for (register budget)
    static constexpr bool HAS_SYM = false, HAS_SYM_DYN = false, HAS_SYM_PATH = false, HAS_SYM_LAG = false;   // symbolic functions: generated specialisations (ctd_sym_registry.hpp)
    static constexpr double t0_fixed() { return 0.0; }
    static constexpr double tf_fixed() { return 0.0; }
    // xdot = F0(x) + u F1(x)   (goddard.jl:7-16, :44)
    template <class T> CTD_HD static void dynamics(T* dx, const T&, const T* x, const T* u, const T*) {
        const double Cd = 310.0, beta = 500.0, b = 2.0, Tmax = 3.5;
        const T drag = Cd * d_sqr(x[1]) * d_exp(-beta * (x[0] - 1.0));
        const T f0_v = -drag / x[2] - 1.0 / d_sqr(x[0]);
        const T f1_v = Tmax / x[2];
        dx[0] = x[1] + u[0] * 0.0;
        dx[1] = f0_v + u[0] * f1_v;
        dx[2] = 0.0 + u[0] * (-b * Tmax);
    }
    template <class T> CTD_HD static T lagrange(const T&, const T*, const T*, const T*) { return T(0.0); }
    template <class T> CTD_HD static T mayer(const T*, const T* xf, const T*) { return xf[0]; }          // r(tf) -> max :45
    template <class T> CTD_HD static void path(T*, const T&, const T*, const T*, const T*) {}
    template <class T> CTD_HD static void boundary(T* r, const T* x0, const T* xf, const T*) {           // :38-39
        r[0] = x0[0]; r[1] = x0[1]; r[2] = x0[2]; r[3] = xf[2];
    }
    static bool init_state(double, double* x) { x[0] = 1.01; x[1] = 0.05; x[2] = 0.8; return true; }    // :48
    static ProblemInfo info() {
        ProblemInfo p{"goddard", NX, NU, NV, NPATH, NBC, IT0, ITF, 0.0, 0.0, HAS_LAGRANGE, HAS_MAYER, true,
                      {{0, 1.0, 1.1}, {1, 0.0, 0.1}, {2, 0.6, 1.0}}, {{0, 0.0, 1.0}}, {{0, 0.01, kInf}},
                      {}, {}, {1.0, 0.0, 1.0, 0.6}, {1.0, 0.0, 1.0, 0.6}, init_state, no_init_t, no_init_v};
        return p;
    }
};

// -------------------------------------------------------------------------------------------------------
// goddard_all: functional form with three nonlinear path constraints, test/problems/goddard.jl:87-158
// -------------------------------------------------------------------------------------------------------
struct GoddardAllOCP {
    static constexpr int NX = 3, NU = 1, NV = 1, NPATH = 3, NBC = 4;
    static constexpr int IT0 = -1, ITF = 0;
    static constexpr bool HAS_LAGRANGE = false, HAS_MAYER = true;
    static constexpr bool DYN_T = false, DYN_V = false, PATH_T = false, PATH_V = true;
    static constexpr bool LAG_T = false, LAG_V = false;      // explicit dependence of the Lagrange cost on t / v
    static constexpr int DC = CTD_DC(4);
    static constexpr int MAXB = 1024;                       // largest workgroup the kernels are compiled for (register budget)
    static constexpr bool HAS_SYM = false, HAS_SYM_DYN = false, HAS_SYM_PATH = false, HAS_SYM_LAG = false;   // symbolic functions: generated specialisations (ctd_sym_registry.hpp)
    static constexpr double t0_fixed() { return 0.0; }
    static constexpr double tf_fixed() { return 0.0; }
    template <class T> CTD_HD static void dynamics(T* dx, const T&, const T* x, const T* u, const T*) {  // f! :127-132
        const double Cd = 310.0, beta = 500.0, b = 2.0, Tmax = 3.5;
        dx[0] = x[1];
        const T drag = Cd * d_sqr(x[1]) * d_exp(-beta * (x[0] - 1.0));
        dx[1] = -drag / x[2] - 1.0 / d_sqr(x[0]) + u[0] * Tmax / x[2];
        dx[2] = -b * Tmax * u[0];
    }
    template <class T> CTD_HD static T lagrange(const T&, const T*, const T*, const T*) { return T(0.0); }
    template <class T> CTD_HD static T mayer(const T*, const T* xf, const T*) { return xf[0]; }
    template <class T> CTD_HD static void path(T* r, const T&, const T* x, const T* u, const T* v) {     // path! :117-121
        r[0] = x[1];
        r[1] = u[0];
        r[2] = x[0] + x[1] + x[2] + u[0] + v[0];
    }
    template <class T> CTD_HD static void boundary(T* r, const T* x0, const T* xf, const T*) {           // bc! :134-139
        r[0] = x0[0]; r[1] = x0[1]; r[2] = x0[2]; r[3] = xf[2];
    }
    static bool init_state(double, double* x) { x[0] = 1.01; x[1] = 0.05; x[2] = 0.8; return true; }
    static ProblemInfo info() {
        ProblemInfo p{"goddard_all", NX, NU, NV, NPATH, NBC, IT0, ITF, 0.0, 0.0, HAS_LAGRANGE, HAS_MAYER, true,
                      {{0, 1.0, kInf}, {1, 0.0, kInf}, {2, 0.0, 1.0}}, {{0, 0.0, kInf}}, {{0, 0.01, kInf}},
                      {-kInf, -kInf, 0.0}, {0.1, 1.0, kInf}, {1.0, 0.0, 1.0, 0.6}, {1.0, 0.0, 1.0, 0.6},
                      init_state, no_init_t, no_init_v};
        return p;
    }
};

// -------------------------------------------------------------------------------------------------------
// double integrator (min energy, T = 2: test/problems/double_integrator.jl:42-58) with the build-defined
// nonlinear path constraint q + 0.1 w^2 <= 1.05 and control box |u| <= 5 (BASELINE config 3)
// -------------------------------------------------------------------------------------------------------
struct DoubleIntegratorPathOCP {
    static constexpr int NX = 2, NU = 1, NV = 0, NPATH = 1, NBC = 4;
    static constexpr int IT0 = -1, ITF = -1;
    static constexpr bool HAS_LAGRANGE = true, HAS_MAYER = false;
    static constexpr bool DYN_T = false, DYN_V = false, PATH_T = false, PATH_V = false;
    static constexpr bool LAG_T = false, LAG_V = false;      // explicit dependence of the Lagrange cost on t / v
    static constexpr int DC = CTD_DC(3);
    static constexpr int MAXB = 1024;                       // largest workgroup the kernels are compiled for (register budget)
    static constexpr bool HAS_SYM = false, HAS_SYM_DYN = false, HAS_SYM_PATH = false, HAS_SYM_LAG = false;   // symbolic functions: generated specialisations (ctd_sym_registry.hpp)
    static constexpr double t0_fixed() { return 0.0; }
    static constexpr double tf_fixed() { return 2.0; }
    template <class T> CTD_HD static void dynamics(T* dx, const T&, const T* x, const T* u, const T*) {
        dx[0] = x[1];
        dx[1] = u[0];
    }
    template <class T> CTD_HD static T lagrange(const T&, const T*, const T* u, const T*) { return d_sqr(u[0]); }
    template <class T> CTD_HD static T mayer(const T*, const T*, const T*) { return T(0.0); }
    template <class T> CTD_HD static void path(T* r, const T&, const T* x, const T*, const T*) {
        r[0] = x[0] + 0.1 * d_sqr(x[1]);
    }
    template <class T> CTD_HD static void boundary(T* r, const T* x0, const T* xf, const T*) {
        r[0] = x0[0]; r[1] = x0[1]; r[2] = xf[0]; r[3] = xf[1];
    }
    static ProblemInfo info() {
        ProblemInfo p{"double_integrator_path", NX, NU, NV, NPATH, NBC, IT0, ITF, 0.0, 2.0, HAS_LAGRANGE, HAS_MAYER, false,
                      {}, {{0, -5.0, 5.0}}, {}, {-kInf}, {1.05}, {0.0, 0.0, 1.0, 0.0}, {0.0, 0.0, 1.0, 0.0},
                      no_init_t, no_init_t, no_init_v};
        return p;
    }
};

// -------------------------------------------------------------------------------------------------------
// quadrotor, the reference's 8-state / 4-control model: test/problems/quadrotor.jl:7-105
// -------------------------------------------------------------------------------------------------------
struct QuadrotorOCP {
    static constexpr int NX = 8, NU = 4, NV = 1, NPATH = 1, NBC = 14;
    static constexpr int IT0 = -1, ITF = 0;
    static constexpr bool HAS_LAGRANGE = true, HAS_MAYER = true;
    static constexpr bool DYN_T = false, DYN_V = false, PATH_T = false, PATH_V = false;
    static constexpr bool LAG_T = false, LAG_V = false;      // explicit dependence of the Lagrange cost on t / v
    static constexpr int DC = CTD_DC(4);
    static constexpr int MAXB = 320;                       // largest workgroup the kernels are compiled for (register budget)
    static constexpr bool HAS_SYM = false, HAS_SYM_DYN = false, HAS_SYM_PATH = false, HAS_SYM_LAG = false;   // symbolic functions: generated specialisations (ctd_sym_registry.hpp)
    static constexpr double t0_fixed() { return 0.0; }
    static constexpr double tf_fixed() { return 0.0; }
    template <class T> CTD_HD static void dynamics(T* dx, const T&, const T* x, const T* u, const T*) {  // :20-40
        const double g = 9.81;
        const T cphi = d_cos(x[6]), sphi = d_sin(x[6]);
        const T cth = d_cos(x[7]), sth = d_sin(x[7]);
        const T cpsi = d_cos(u[3]), spsi = d_sin(u[3]);
        // R * [0; 0; at]: only the third column of R contributes
        const T ax = (cpsi * sth * cphi + spsi * sphi) * u[0];
        const T ay = (spsi * sth * cphi - cpsi * sphi) * u[0];
        const T az = (cth * cphi) * u[0];
        dx[0] = x[3]; dx[1] = x[4]; dx[2] = x[5];
        dx[3] = ax; dx[4] = ay; dx[5] = az - g;
        dx[6] = u[1]; dx[7] = u[2];
    }
    template <class T> CTD_HD static T lagrange(const T&, const T* x, const T* u, const T*) {            // :98
        return 1e-8 * (d_sqr(x[6]) + d_sqr(x[7]) + d_sqr(u[3]) + d_sqr(u[0])) + 1e2 * d_sqr(u[3]);
    }
    template <class T> CTD_HD static T mayer(const T*, const T*, const T* v) { return v[0]; }
    template <class T> CTD_HD static void path(T* r, const T&, const T* x, const T*, const T*) {         // :75
        r[0] = d_cos(x[7]) * d_cos(x[6]);
    }
    template <class T> CTD_HD static void boundary(T* r, const T* x0, const T* xf, const T*) {           // :77-92
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = x0[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) r[8 + i] = xf[i];
    }
    static bool init_state(double t, double* x) {                                                      // :101
        x[0] = 0.0 + (0.01 - 0.0) * t; x[1] = 0.0 + (5.0 - 0.0) * t; x[2] = 2.5 + (2.5 - 2.5) * t;
        for (int i = 3; i < 8; ++i) x[i] = 0.0;
        return true;
    }
    static bool init_control(double, double* u) { u[0] = 10.0; u[1] = 0.0; u[2] = 0.0; u[3] = 0.0; return true; }
    static bool init_variable(double* v) { v[0] = 1.0; return true; }
    static ProblemInfo info() {
        const double hp = 3.14159265358979323846 / 2;
        ProblemInfo p{"quadrotor", NX, NU, NV, NPATH, NBC, IT0, ITF, 0.0, 0.0, HAS_LAGRANGE, HAS_MAYER, false,
                      {{6, -hp, hp}, {7, -hp, hp}}, {{0, 0.0, 9.18 * 5}, {1, -3.0, 3.0}, {2, -3.0, 3.0}}, {{0, 0.1, kInf}},
                      {std::cos(1.1 / 2)}, {kInf},
                      {0.0, 0.0, 2.5, 0, 0, 0, 0, 0, 0.01, 5.0, 2.5, 0, 0, 0}, {0.0, 0.0, 2.5, 0, 0, 0, 0, 0, 0.01, 5.0, 2.5, 0, 0, 0},
                      init_state, init_control, init_variable};
        return p;
    }
};

// -------------------------------------------------------------------------------------------------------
// quadrotor-12: 12-state rigid body (BASELINE config 5).  Not in the reference; defined by the build
//   x = (p[3], v[3], phi, theta, psi, w[3]), u = (at, tau[3]), v = (tf)
// -------------------------------------------------------------------------------------------------------
struct Quadrotor12OCP {
    static constexpr int NX = 12, NU = 4, NV = 1, NPATH = 1, NBC = 23;
    static constexpr int IT0 = -1, ITF = 0;
    static constexpr bool HAS_LAGRANGE = true, HAS_MAYER = true;
    static constexpr bool DYN_T = false, DYN_V = false, PATH_T = false, PATH_V = false;
    static constexpr bool LAG_T = false, LAG_V = false;      // explicit dependence of the Lagrange cost on t / v
    static constexpr int DC = CTD_DC(4);
    static constexpr int MAXB = 320;                       // largest workgroup the kernels are compiled for (register budget)
    static constexpr bool HAS_SYM = false, HAS_SYM_DYN = false, HAS_SYM_PATH = false, HAS_SYM_LAG = false;   // symbolic functions: generated specialisations (ctd_sym_registry.hpp)
    static constexpr double t0_fixed() { return 0.0; }
    static constexpr double tf_fixed() { return 0.0; }
    template <class T> CTD_HD static void dynamics(T* dx, const T&, const T* x, const T* u, const T*) {
        const double g = 9.81, Jx = 0.03, Jy = 0.03, Jz = 0.06;
        const T cphi = d_cos(x[6]), sphi = d_sin(x[6]);
        const T cth = d_cos(x[7]), sth = d_sin(x[7]);
        const T cpsi = d_cos(x[8]), spsi = d_sin(x[8]);
        dx[0] = x[3]; dx[1] = x[4]; dx[2] = x[5];
        dx[3] = (cpsi * sth * cphi + spsi * sphi) * u[0];
        dx[4] = (spsi * sth * cphi - cpsi * sphi) * u[0];
        dx[5] = (cth * cphi) * u[0] - g;
        const T tth = sth / cth;
        dx[6] = x[9] + sphi * tth * x[10] + cphi * tth * x[11];
        dx[7] = cphi * x[10] - sphi * x[11];
        dx[8] = (sphi * x[10] + cphi * x[11]) / cth;
        dx[9] = ((Jy - Jz) * x[10] * x[11] + u[1]) / Jx;
        dx[10] = ((Jz - Jx) * x[11] * x[9] + u[2]) / Jy;
        dx[11] = ((Jx - Jy) * x[9] * x[10] + u[3]) / Jz;
    }
    template <class T> CTD_HD static T lagrange(const T&, const T* x, const T* u, const T*) {
        return 1e-8 * (d_sqr(x[6]) + d_sqr(x[7]) + d_sqr(u[0])) + 1e-2 * (d_sqr(u[1]) + d_sqr(u[2]) + d_sqr(u[3])) + 1e2 * d_sqr(x[8]);
    }
    template <class T> CTD_HD static T mayer(const T*, const T*, const T* v) { return v[0]; }
    template <class T> CTD_HD static void path(T* r, const T&, const T* x, const T*, const T*) {
        r[0] = d_cos(x[7]) * d_cos(x[6]);
    }
    template <class T> CTD_HD static void boundary(T* r, const T* x0, const T* xf, const T*) {
#pragma unroll
        for (int i = 0; i < 12; ++i) r[i] = x0[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) r[12 + i] = xf[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) r[20 + i] = xf[9 + i];
    }
    static bool init_state(double t, double* x) {
        x[0] = 0.01 * t; x[1] = 5.0 * t; x[2] = 2.5;
        for (int i = 3; i < 12; ++i) x[i] = 0.0;
        return true;
    }
    static bool init_control(double, double* u) { u[0] = 10.0; u[1] = 0.0; u[2] = 0.0; u[3] = 0.0; return true; }
    static bool init_variable(double* v) { v[0] = 1.0; return true; }
    static ProblemInfo info() {
        const double hp = 3.14159265358979323846 / 2;
        std::vector<double> bcv(23, 0.0);
        bcv[2] = 2.5; bcv[12] = 0.01; bcv[13] = 5.0; bcv[14] = 2.5;
        ProblemInfo p{"quadrotor12", NX, NU, NV, NPATH, NBC, IT0, ITF, 0.0, 0.0, HAS_LAGRANGE, HAS_MAYER, false,
                      {{6, -hp, hp}, {7, -hp, hp}}, {{0, 0.0, 45.9}, {1, -1.0, 1.0}, {2, -1.0, 1.0}, {3, -1.0, 1.0}}, {{0, 0.1, kInf}},
                      {std::cos(0.55)}, {kInf}, bcv, bcv, init_state, init_control, init_variable};
        return p;
    }
};

// -------------------------------------------------------------------------------------------------------
// scalar stagewise test problem: test/ci/test_discretization_stagewise.jl:1-14
// -------------------------------------------------------------------------------------------------------
struct StagewiseScalarOCP {
    static constexpr int NX = 1, NU = 1, NV = 0, NPATH = 0, NBC = 2;
    static constexpr int IT0 = -1, ITF = -1;
    static constexpr bool HAS_LAGRANGE = true, HAS_MAYER = false;
    static constexpr bool DYN_T = false, DYN_V = false, PATH_T = false, PATH_V = false;
    static constexpr bool LAG_T = false, LAG_V = false;      // explicit dependence of the Lagrange cost on t / v
    static constexpr int DC = CTD_DC(2);
    static constexpr int MAXB = 1024;                       // largest workgroup the kernels are compiled for (register budget)
    static constexpr bool HAS_SYM = false, HAS_SYM_DYN = false, HAS_SYM_PATH = false, HAS_SYM_LAG = false;   // symbolic functions: generated specialisations (ctd_sym_registry.hpp)
    static constexpr double t0_fixed() { return 0.0; }
    static constexpr double tf_fixed() { return 1.0; }
    template <class T> CTD_HD static void dynamics(T* dx, const T&, const T*, const T* u, const T*) { dx[0] = u[0]; }
    template <class T> CTD_HD static T lagrange(const T&, const T*, const T* u, const T*) { return d_sqr(u[0]); }
    template <class T> CTD_HD static T mayer(const T*, const T*, const T*) { return T(0.0); }
    template <class T> CTD_HD static void path(T*, const T&, const T*, const T*, const T*) {}
    template <class T> CTD_HD static void boundary(T* r, const T* x0, const T* xf, const T*) { r[0] = x0[0]; r[1] = xf[0]; }
    static ProblemInfo info() {
        ProblemInfo p{"stagewise_scalar", NX, NU, NV, NPATH, NBC, IT0, ITF, 0.0, 1.0, HAS_LAGRANGE, HAS_MAYER, false,
                      {}, {{0, 0.0, 2.0}}, {}, {}, {}, {0.0, 1.0}, {0.0, 1.0}, no_init_t, no_init_t, no_init_v};
        return p;
    }
};

// -------------------------------------------------------------------------------------------------------
// zero-control parameter estimation problems: test/problems/autonomous_system.jl
// -------------------------------------------------------------------------------------------------------
struct EstimateInitialConditionOCP {                                                      // :6-43
    static constexpr int NX = 2, NU = 0, NV = 2, NPATH = 0, NBC = 2;
    static constexpr int IT0 = -1, ITF = -1;
    static constexpr bool HAS_LAGRANGE = false, HAS_MAYER = true;
    static constexpr bool DYN_T = false, DYN_V = false, PATH_T = false, PATH_V = false;
    static constexpr bool LAG_T = false, LAG_V = false;      // explicit dependence of the Lagrange cost on t / v
    static constexpr int DC = CTD_DC(2);
    static constexpr int MAXB = 1024;                       // largest workgroup the kernels are compiled for (register budget)
    static constexpr bool HAS_SYM = false, HAS_SYM_DYN = false, HAS_SYM_PATH = false, HAS_SYM_LAG = false;   // symbolic functions: generated specialisations (ctd_sym_registry.hpp)
    static constexpr double t0_fixed() { return 0.0; }
    static constexpr double tf_fixed() { return 3.14159265358979323846 / 2; }
    template <class T> CTD_HD static void dynamics(T* dx, const T&, const T* x, const T*, const T*) { dx[0] = -x[1]; dx[1] = x[0]; }
    template <class T> CTD_HD static T lagrange(const T&, const T*, const T*, const T*) { return T(0.0); }
    template <class T> CTD_HD static T mayer(const T*, const T* xf, const T*) { return d_sqr(xf[0] - 0.0) + d_sqr(xf[1] - 1.0); }
    template <class T> CTD_HD static void path(T*, const T&, const T*, const T*, const T*) {}
    template <class T> CTD_HD static void boundary(T* r, const T* x0, const T*, const T* v) { r[0] = x0[0] - v[0]; r[1] = x0[1] - v[1]; }
    static ProblemInfo info() {
        ProblemInfo p{"estimate_initial_condition", NX, NU, NV, NPATH, NBC, IT0, ITF, 0.0, tf_fixed(), HAS_LAGRANGE, HAS_MAYER, false,
                      {}, {}, {}, {}, {}, {0.0, 0.0}, {0.0, 0.0}, no_init_t, no_init_t, no_init_v};
        return p;
    }
};

struct EstimateRotationRateOCP {                                                          // :46-87
    static constexpr int NX = 2, NU = 0, NV = 1, NPATH = 0, NBC = 2;
    static constexpr int IT0 = -1, ITF = -1;
    static constexpr bool HAS_LAGRANGE = false, HAS_MAYER = true;
    static constexpr bool DYN_T = false, DYN_V = true, PATH_T = false, PATH_V = false;
    static constexpr bool LAG_T = false, LAG_V = false;      // explicit dependence of the Lagrange cost on t / v
    static constexpr int DC = CTD_DC(3);
    static constexpr int MAXB = 1024;                       // largest workgroup the kernels are compiled for (register budget)
    static constexpr bool HAS_SYM = false, HAS_SYM_DYN = false, HAS_SYM_PATH = false, HAS_SYM_LAG = false;   // symbolic functions: generated specialisations (ctd_sym_registry.hpp)
    static constexpr double t0_fixed() { return 0.0; }
    static constexpr double tf_fixed() { return 1.0; }
    template <class T> CTD_HD static void dynamics(T* dx, const T&, const T* x, const T*, const T* v) {
        dx[0] = v[0] * (-x[1]);
        dx[1] = v[0] * x[0];
    }
    template <class T> CTD_HD static T lagrange(const T&, const T*, const T*, const T*) { return T(0.0); }
    template <class T> CTD_HD static T mayer(const T*, const T* xf, const T* v) {
        return d_sqr(xf[0] - 0.0) + d_sqr(xf[1] - 1.0) + 0.01 * d_sqr(v[0]);
    }
    template <class T> CTD_HD static void path(T*, const T&, const T*, const T*, const T*) {}
    template <class T> CTD_HD static void boundary(T* r, const T* x0, const T*, const T*) { r[0] = x0[0] - 1.0; r[1] = x0[1] - 0.0; }
    static ProblemInfo info() {
        ProblemInfo p{"estimate_rotation_rate", NX, NU, NV, NPATH, NBC, IT0, ITF, 0.0, 1.0, HAS_LAGRANGE, HAS_MAYER, false,
                      {}, {}, {{0, 0.0, 10.0}}, {}, {}, {0.0, 0.0}, {0.0, 0.0}, no_init_t, no_init_t, no_init_v};
        return p;
    }
};

struct LeastSquaresConstraintOCP {                                                        // :90-138
    static constexpr int NX = 2, NU = 0, NV = 2, NPATH = 1, NBC = 2;
    static constexpr int IT0 = -1, ITF = -1;
    static constexpr bool HAS_LAGRANGE = true, HAS_MAYER = true;
    static constexpr bool DYN_T = false, DYN_V = false, PATH_T = false, PATH_V = false;
    static constexpr bool LAG_T = true, LAG_V = false;       // explicit dependence of the Lagrange cost on t / v
    static constexpr int DC = CTD_DC(2);
    static constexpr int MAXB = 1024;                       // largest workgroup the kernels are compiled for (register budget)
    static constexpr bool HAS_SYM = false, HAS_SYM_DYN = false, HAS_SYM_PATH = false, HAS_SYM_LAG = false;   // symbolic functions: generated specialisations (ctd_sym_registry.hpp)
    static constexpr double t0_fixed() { return 0.0; }
    static constexpr double tf_fixed() { return 1.0; }
    template <class T> CTD_HD static void dynamics(T* dx, const T&, const T* x, const T*, const T*) { dx[0] = -x[1]; dx[1] = x[0]; }
    template <class T> CTD_HD static T lagrange(const T& t, const T* x, const T*, const T*) {
        return d_sqr(t - 0.5) * (d_sqr(x[0] - 0.7) + d_sqr(x[1] - 0.7));
    }
    template <class T> CTD_HD static T mayer(const T*, const T*, const T* v) { return 0.01 * (d_sqr(v[0]) + d_sqr(v[1])); }
    template <class T> CTD_HD static void path(T* r, const T&, const T* x, const T*, const T*) { r[0] = d_sqr(x[0]) + d_sqr(x[1]); }
    template <class T> CTD_HD static void boundary(T* r, const T* x0, const T*, const T* v) { r[0] = x0[0] - v[0]; r[1] = x0[1] - v[1]; }
    static ProblemInfo info() {
        ProblemInfo p{"least_squares_with_constraint", NX, NU, NV, NPATH, NBC, IT0, ITF, 0.0, 1.0, HAS_LAGRANGE, HAS_MAYER, false,
                      {}, {}, {}, {-kInf}, {2.0}, {0.0, 0.0}, {0.0, 0.0}, no_init_t, no_init_t, no_init_v};
        return p;
    }
};

// -------------------------------------------------------------------------------------------------------
// double integrator with free t0 and tf: test/problems/double_integrator.jl:79-99 (tf - t0 >= 0.01 carried
// as a fifth boundary row; build's choice)
// -------------------------------------------------------------------------------------------------------
struct DoubleIntegratorFreeT0TfOCP {
    static constexpr int NX = 2, NU = 1, NV = 2, NPATH = 0, NBC = 5;
    static constexpr int IT0 = 0, ITF = 1;
    static constexpr bool HAS_LAGRANGE = false, HAS_MAYER = true;
    static constexpr bool DYN_T = false, DYN_V = false, PATH_T = false, PATH_V = false;
    static constexpr bool LAG_T = false, LAG_V = false;      // explicit dependence of the Lagrange cost on t / v
    static constexpr int DC = CTD_DC(3);
    static constexpr int MAXB = 1024;                       // largest workgroup the kernels are compiled for (register budget)
    static constexpr bool HAS_SYM = false, HAS_SYM_DYN = false, HAS_SYM_PATH = false, HAS_SYM_LAG = false;   // symbolic functions: generated specialisations (ctd_sym_registry.hpp)
    static constexpr double t0_fixed() { return 0.0; }
    static constexpr double tf_fixed() { return 0.0; }
    template <class T> CTD_HD static void dynamics(T* dx, const T&, const T* x, const T* u, const T*) { dx[0] = x[1]; dx[1] = u[0]; }
    template <class T> CTD_HD static T lagrange(const T&, const T*, const T*, const T*) { return T(0.0); }
    template <class T> CTD_HD static T mayer(const T*, const T*, const T* v) { return v[0]; }
    template <class T> CTD_HD static void path(T*, const T&, const T*, const T*, const T*) {}
    template <class T> CTD_HD static void boundary(T* r, const T* x0, const T* xf, const T* v) {
        r[0] = x0[0]; r[1] = x0[1]; r[2] = xf[0]; r[3] = xf[1]; r[4] = v[1] - v[0];
    }
    static ProblemInfo info() {
        ProblemInfo p{"double_integrator_freet0tf", NX, NU, NV, NPATH, NBC, IT0, ITF, 0.0, 0.0, HAS_LAGRANGE, HAS_MAYER, true,
                      {}, {{0, -1.0, 1.0}}, {{0, 0.05, 10.0}, {1, 0.05, 10.0}}, {}, {},
                      {0.0, 0.0, 1.0, 0.0, 0.01}, {0.0, 0.0, 1.0, 0.0, kInf}, no_init_t, no_init_t, no_init_v};
        return p;
    }
};

constexpr int kNumProblems = 10;

// static dispatch over the registry: f(TypeTag<OCP>{})
template <class P> struct TypeTag { using type = P; };
template <class F> inline bool for_problem(int id, F&& f) {
    switch (id) {
        case 0: f(TypeTag<GoddardOCP>{}); return true;
        case 1: f(TypeTag<GoddardAllOCP>{}); return true;
        case 2: f(TypeTag<DoubleIntegratorPathOCP>{}); return true;
        case 3: f(TypeTag<QuadrotorOCP>{}); return true;
        case 4: f(TypeTag<Quadrotor12OCP>{}); return true;
        case 5: f(TypeTag<StagewiseScalarOCP>{}); return true;
        case 6: f(TypeTag<EstimateInitialConditionOCP>{}); return true;
        case 7: f(TypeTag<EstimateRotationRateOCP>{}); return true;
        case 8: f(TypeTag<LeastSquaresConstraintOCP>{}); return true;
        case 9: f(TypeTag<DoubleIntegratorFreeT0TfOCP>{}); return true;
        default: return false;
    }
}

}  // namespace ctd
