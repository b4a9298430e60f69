// ctd_engine.hip -- handle, device buffers, launches and the extern "C" entry points of include/ctdirect_hip.h.
//
// There is no CPU compute path in this library: every hot-path entry point launches the HIP kernels of
// ctd_kernels.hpp on the handle's device and fails with CTD_ENODEVICE when the handle has none.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <dlfcn.h>

#include <cstdio>
#include <functional>
#include <map>
#include <mutex>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "../../include/ctdirect_hip.h"
#include "ctd_host.hpp"
#include "ctd_kernels.hpp"
#include "ctd_hess_kernels.hpp"
#include "ctd_hess_step.hpp"
#include "ctd_iter_kernels.hpp"
#include "ctd_jit.hpp"

using namespace ctd;

namespace ctd {
// the instantiations live in the per-problem translation units
CTD_EXTERN_LAUNCHERS(GoddardOCP)
CTD_EXTERN_LAUNCHERS(GoddardAllOCP)
CTD_EXTERN_LAUNCHERS(DoubleIntegratorPathOCP)
CTD_EXTERN_LAUNCHERS(QuadrotorOCP)
CTD_EXTERN_LAUNCHERS(Quadrotor12OCP)
CTD_EXTERN_LAUNCHERS(StagewiseScalarOCP)
CTD_EXTERN_LAUNCHERS(EstimateInitialConditionOCP)
CTD_EXTERN_LAUNCHERS(EstimateRotationRateOCP)
CTD_EXTERN_LAUNCHERS(LeastSquaresConstraintOCP)
CTD_EXTERN_LAUNCHERS(DoubleIntegratorFreeT0TfOCP)
CTD_EXTERN_HESS(GoddardOCP)
CTD_EXTERN_HESS(GoddardAllOCP)
CTD_EXTERN_HESS(DoubleIntegratorPathOCP)
CTD_EXTERN_HESS(QuadrotorOCP)
CTD_EXTERN_HESS(Quadrotor12OCP)
CTD_EXTERN_HESS(StagewiseScalarOCP)
CTD_EXTERN_HESS(EstimateInitialConditionOCP)
CTD_EXTERN_HESS(EstimateRotationRateOCP)
CTD_EXTERN_HESS(LeastSquaresConstraintOCP)
CTD_EXTERN_HESS(DoubleIntegratorFreeT0TfOCP)
CTD_EXTERN_HESS_STEP(GoddardOCP)
CTD_EXTERN_HESS_STEP(GoddardAllOCP)
CTD_EXTERN_HESS_STEP(DoubleIntegratorPathOCP)
CTD_EXTERN_HESS_STEP(QuadrotorOCP)
CTD_EXTERN_HESS_STEP(Quadrotor12OCP)
CTD_EXTERN_HESS_STEP(StagewiseScalarOCP)
CTD_EXTERN_HESS_STEP(EstimateInitialConditionOCP)
CTD_EXTERN_HESS_STEP(EstimateRotationRateOCP)
CTD_EXTERN_HESS_STEP(LeastSquaresConstraintOCP)
CTD_EXTERN_HESS_STEP(DoubleIntegratorFreeT0TfOCP)
CTD_EXTERN_ITER(GoddardOCP)
CTD_EXTERN_ITER(GoddardAllOCP)
CTD_EXTERN_ITER(DoubleIntegratorPathOCP)
CTD_EXTERN_ITER(QuadrotorOCP)
CTD_EXTERN_ITER(Quadrotor12OCP)
CTD_EXTERN_ITER(StagewiseScalarOCP)
CTD_EXTERN_ITER(EstimateInitialConditionOCP)
CTD_EXTERN_ITER(EstimateRotationRateOCP)
CTD_EXTERN_ITER(LeastSquaresConstraintOCP)
CTD_EXTERN_ITER(DoubleIntegratorFreeT0TfOCP)
}  // namespace ctd

struct ctd_handle {
    Model model;
    int device = -1;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int64_t step_begin = 0, step_end = 0;
    int tile = 0, block = 256;
    KParams kp;                 // device pointers filled in, outputs set per call
    size_t lds_bytes = 0;
    int grid = 0;
    // static device data
    double* d_tau = nullptr;
    uint32_t* d_tmpl = nullptr;
    uint32_t* d_vtmpl = nullptr;
    uint16_t* d_pos = nullptr;
    int64_t* d_edge_idx = nullptr;
    uint32_t* d_edge_code = nullptr;
    // staging for the host-pointer entry points
    double* d_x = nullptr;
    double* d_c = nullptr;
    double* d_vals = nullptr;
    // objective
    double* d_partial = nullptr;
    double* d_obj = nullptr;
    double* d_g = nullptr;          // staging for ctd_grad (host pointers)
    double* d_gpartial = nullptr;   // per-workgroup partial sums of dg/dv
    int gblocks = 0;
    int obj_blocks = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // run-time defined OCP (ctd_register_ocp): kernels compiled with hiprtc and launched through the module API
    const RtOcp* rt = nullptr;
    hipModule_t jit_mod = nullptr, jit_hmod = nullptr;
    hipFunction_t f_cons_jac = nullptr, f_obj_partial = nullptr, f_obj_finish = nullptr, f_grad_units = nullptr,
                  f_grad_finish = nullptr, f_hess = nullptr, f_hess_finish = nullptr;
    // Hessian of the Lagrangian: tables are uploaded by the first Hessian call (hess_ready)
    bool hess_ready = false;
    HParams hp;
    int hess_tile = 0;
    size_t hess_lds_bytes = 0;
    uint32_t *d_htptr = nullptr, *d_hterms = nullptr, *d_hvptr = nullptr, *d_hvterms = nullptr, *d_heptr = nullptr,
             *d_hevptr = nullptr, *d_heterms = nullptr;
    int64_t* d_hedge_idx = nullptr;
    double* d_hpair_c = nullptr;
    uint32_t *d_hcpos = nullptr, *d_hzpos = nullptr;
    // lane-per-step Hessian kernel (ctd_hess_step.hpp): position tables, parameters, the tile kernel's parameters for its edge blocks
    int32_t *d_hssrc = nullptr, *d_hschunk = nullptr;
    double* d_hsck = nullptr;
    bool hess_step = false;
    SParams sp{};
    HParams hp_step{};
    size_t hess_step_lds = 0;
    uint32_t *d_htasks = nullptr, *d_hptasks = nullptr, *d_hbtasks = nullptr;
    double *d_hpartials = nullptr, *d_y = nullptr, *d_hvals = nullptr;
    // sharded iterate read in place (ctd_set_x_shards): device table of the other shards' buffers, host copy of what it holds
    XHalo* d_halo = nullptr;
    XHalo halo_host{};
    // ctd_stitch_c: padded send block and gathered blocks
    double *d_stitch_send = nullptr, *d_stitch_recv = nullptr;
    int64_t stitch_cap = 0;
    std::string err;
};

static thread_local std::string g_create_err;      // error of this thread's last failed handle-less call (ctd_last_error(NULL))

static int32_t fail(ctd_handle* h, int32_t code, const std::string& msg) {
    if (h) h->err = msg; else g_create_err = msg;
    return code;
}

#define HIP_TRY(h, expr)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(h, CTD_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

// Every entry point that touches the device runs on the handle's device and leaves the caller's current device as it found
// it (a process driving several GPUs keeps its own notion of "current"); hipGetDevice is a thread-local read, and the
// hipSetDevice pair is only paid when the two differ.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) { err = hipSetDevice(dev); switched = (err == hipSuccess); }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

template <class T> static hipError_t upload(T** dst, const std::vector<T>& src) {
    if (*dst) { (void)hipFree(*dst); }       // idempotent: a retry after a partial failure (ensure_hess) replaces, never leaks
    *dst = nullptr;
    if (src.empty()) return hipSuccess;
    hipError_t e = hipMalloc((void**)dst, src.size() * sizeof(T));
    if (e != hipSuccess) return e;
    return hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice);
}

static int env_int(const char* name, int dflt) {
    const char* s = std::getenv(name);
    return (s && *s) ? std::atoi(s) : dflt;
}

static void free_device(ctd_handle* h) {
    if (h->device < 0) return;
    DeviceGuard dg_(h->device);
    (void)hipStreamSynchronize(h->stream);      // enqueue-only calls may still be running on the tables freed below
    for (void* p : {(void*)h->d_pos, (void*)h->d_tau, (void*)h->d_tmpl, (void*)h->d_vtmpl, (void*)h->d_edge_idx, (void*)h->d_edge_code,
                    (void*)h->d_x, (void*)h->d_c, (void*)h->d_vals, (void*)h->d_partial, (void*)h->d_obj, (void*)h->d_g,
                    (void*)h->d_gpartial, (void*)h->d_htptr, (void*)h->d_hterms, (void*)h->d_hvptr, (void*)h->d_hvterms,
                    (void*)h->d_heptr, (void*)h->d_hevptr, (void*)h->d_heterms, (void*)h->d_hedge_idx, (void*)h->d_htasks,
                    (void*)h->d_hptasks, (void*)h->d_hbtasks, (void*)h->d_hpair_c, (void*)h->d_hcpos, (void*)h->d_hzpos, (void*)h->d_hssrc, (void*)h->d_hschunk, (void*)h->d_hsck, (void*)h->d_hpartials, (void*)h->d_y, (void*)h->d_hvals, (void*)h->d_halo, (void*)h->d_stitch_send, (void*)h->d_stitch_recv})
        if (p) (void)hipFree(p);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    if (h->jit_mod) (void)hipModuleUnload(h->jit_mod);
    if (h->jit_hmod) (void)hipModuleUnload(h->jit_hmod);
}

// ---- run-time compilation of the kernel templates for a registered OCP -------------------------------------------------
namespace {
std::string jit_include_dir() {
    const char* env = std::getenv("CTD_JIT_INCLUDE");
    if (env && *env) return env;
    Dl_info info;
    if (dladdr((const void*)&free_device, &info) && info.dli_fname) {
        std::string path(info.dli_fname);
        const size_t slash = path.find_last_of('/');
        return (slash == std::string::npos ? std::string(".") : path.substr(0, slash)) + "/csrc";
    }
    return "csrc";
}

std::mutex g_jit_mu;
std::map<std::string, std::pair<std::string, std::vector<std::string>>> g_jit_cache;   // key -> (code object, lowered names)

// compiles `header` + the OCP's functor for gfx950 and returns the code object and the lowered names of `exprs`
int32_t jit_compile(const RtOcp& ro, const char* header, const std::vector<std::string>& exprs, const char* fp_contract,
                    std::string& code, std::vector<std::string>& lowered, std::string& err) {
    std::string key = ro.name + "|" + std::to_string((size_t)&ro) + "|" + header;
    for (const std::string& e : exprs) key += "|" + e;
    // diagnostics: extra compiler options for the run-time kernels (e.g. CTD_JIT_EXTRA="-DCTD_NO_FOLD -DCTD_NO_SPLIT"), split at blanks
    std::vector<std::string> extra;
    if (const char* ex = std::getenv("CTD_JIT_EXTRA")) {
        std::string cur;
        for (const char* c = ex;; ++c) {
            if (*c == ' ' || *c == 0) { if (!cur.empty()) extra.push_back(cur); cur.clear(); if (*c == 0) break; }
            else cur += *c;
        }
        key += std::string("|") + ex;
    }
    {
        std::lock_guard<std::mutex> lk(g_jit_mu);
        auto it = g_jit_cache.find(key);
        if (it != g_jit_cache.end()) { code = it->second.first; lowered = it->second.second; return CTD_OK; }
    }
    const std::string src = std::string("#include \"") + header + "\"\n" + ro.functor_src;
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, src.c_str(), "ctd_user_ocp.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
        err = "hiprtcCreateProgram failed";
        return CTD_EHIP;
    }
    for (const std::string& e : exprs) (void)hiprtcAddNameExpression(prog, e.c_str());
    const std::string inc = "-I" + jit_include_dir();
    const std::string fpc = std::string("-ffp-contract=") + fp_contract;
    std::vector<const char*> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", fpc.c_str(), inc.c_str()};
    for (const std::string& e : extra) opts.push_back(e.c_str());
    const hiprtcResult rc = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
    if (rc != HIPRTC_SUCCESS) {
        size_t ls = 0;
        (void)hiprtcGetProgramLogSize(prog, &ls);
        std::string log(ls, '\0');
        if (ls) (void)hiprtcGetProgramLog(prog, &log[0]);
        err = "run-time compilation of OCP '" + ro.name + "' failed: " + log.substr(0, 4000);
        (void)hiprtcDestroyProgram(&prog);
        return CTD_EHIP;
    }
    lowered.clear();
    for (const std::string& e : exprs) {
        const char* ln = nullptr;
        if (hiprtcGetLoweredName(prog, e.c_str(), &ln) != HIPRTC_SUCCESS || !ln) { err = "no lowered name for " + e; (void)hiprtcDestroyProgram(&prog); return CTD_EHIP; }
        lowered.emplace_back(ln);
    }
    size_t cs = 0;
    (void)hiprtcGetCodeSize(prog, &cs);
    code.assign(cs, '\0');
    (void)hiprtcGetCode(prog, &code[0]);
    (void)hiprtcDestroyProgram(&prog);
    std::lock_guard<std::mutex> lk(g_jit_mu);
    g_jit_cache[key] = {code, lowered};
    return CTD_OK;
}

// s: stages of a Gauss-Legendre scheme; for the midpoint scheme the controls per step (control_steps)
std::vector<std::string> jit_first_exprs(int sc, int s) {
    const std::string P = "ctd::UserOCP", a = std::to_string(sc), b = std::to_string((sc == SC_IRK || sc == SC_MIDPOINT) && s > 0 ? s : 1);
    return {"ctd::cons_jac_kernel<" + P + ", " + a + ", " + b + ", false>", "ctd::obj_partial_kernel<" + P + ", " + a + ">",
            "ctd::obj_finish_kernel<" + P + ">", "ctd::grad_units_kernel<" + P + ", " + a + ", " + b + ">",
            "ctd::grad_finish_kernel<" + P + ">"};
}
std::vector<std::string> jit_hess_exprs(int sc, int s) {
    const std::string P = "ctd::UserOCP", a = std::to_string(sc), b = std::to_string((sc == SC_IRK || sc == SC_MIDPOINT) && s > 0 ? s : 1);
    return {"ctd::hess_kernel<" + P + ", " + a + ", " + b + ", false>", "ctd::hess_finish_kernel<" + P + ">"};
}

hipError_t jit_launch(hipFunction_t f, int grid, int block, size_t lds, hipStream_t st, void** args, hipEvent_t e0 = nullptr,
                      hipEvent_t e1 = nullptr) {
    if (e0 || e1)
        return hipExtModuleLaunchKernel(f, (uint32_t)grid * (uint32_t)block, 1, 1, (uint32_t)block, 1, 1, lds, st, args, nullptr, e0, e1, 0);
    return hipModuleLaunchKernel(f, (uint32_t)grid, 1, 1, (uint32_t)block, 1, 1, (uint32_t)lds, st, args, nullptr);
}
}  // namespace

static int32_t jit_load_first(ctd_handle* h) {
    std::string code, err;
    std::vector<std::string> names;
    const Layout& Lj = h->model.L;
    int32_t st = jit_compile(*h->rt, "ctd_kernels.hpp", jit_first_exprs(Lj.sc, Lj.sc == SC_MIDPOINT ? Lj.cs : Lj.s), "off", code, names, err);
    if (st) return fail(nullptr, st, err);
    HIP_TRY(nullptr, hipModuleLoadData(&h->jit_mod, code.data()));
    hipFunction_t* f[] = {&h->f_cons_jac, &h->f_obj_partial, &h->f_obj_finish, &h->f_grad_units, &h->f_grad_finish};
    for (int i = 0; i < 5; ++i) HIP_TRY(nullptr, hipModuleGetFunction(f[i], h->jit_mod, names[i].c_str()));
    return CTD_OK;
}
static int32_t jit_load_hess(ctd_handle* h) {
    std::string code, err;
    std::vector<std::string> names;
    int32_t st = jit_compile(*h->rt, "ctd_hess_kernels.hpp", jit_hess_exprs(h->model.L.sc, h->model.L.sc == SC_MIDPOINT ? h->model.L.cs : h->model.L.s), "fast", code, names, err);
    if (st) return fail(h, st, err);
    HIP_TRY(h, hipModuleLoadData(&h->jit_hmod, code.data()));
    HIP_TRY(h, hipModuleGetFunction(&h->f_hess, h->jit_hmod, names[0].c_str()));
    HIP_TRY(h, hipModuleGetFunction(&h->f_hess_finish, h->jit_hmod, names[1].c_str()));
    return CTD_OK;
}

extern "C" {

const char* ctd_strerror(int32_t st) {
    switch (st) {
        case CTD_OK: return "ok";
        case CTD_EINVAL: return "invalid argument";
        case CTD_EGRID: return "given time grid is not strictly increasing";
        case CTD_ESCHEME: return "unknown discretization method";
        case CTD_EPATTERN: return "sparsity pattern not available";
        case CTD_EPROBLEM: return "problem not in the compiled registry";
        case CTD_ENODEVICE: return "no HIP device bound to this handle (there is no CPU fallback)";
        case CTD_EHIP: return "HIP runtime error";
        case CTD_ENOMEM: return "out of memory";
        default: return "unknown status";
    }
}

const char* ctd_last_error(const ctd_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int32_t ctd_create(const ctd_desc* desc, ctd_handle** out) {
    if (!desc || !out) return fail(nullptr, CTD_EINVAL, "ctd_create: null argument");
    *out = nullptr;
    struct Del { void operator()(ctd_handle* p) const { if (p) { free_device(p); delete p; } } };
    std::unique_ptr<ctd_handle, Del> h(new (std::nothrow) ctd_handle());
    if (!h) return fail(nullptr, CTD_ENOMEM, "ctd_create: out of memory");
    HostDesc hd{desc->problem, desc->scheme, desc->pattern_mode, desc->grid_size, desc->time_grid, desc->time_grid_len,
                desc->control_steps > 1 ? desc->control_steps : 1, desc->value_order};
    if (desc->reserved0 != 0) return fail(nullptr, CTD_EINVAL, "ctd_create: ctd_desc.reserved0 must be 0 (zero-initialise the descriptor: memset / ctd_desc d = {0})");
    std::string err;
    int st;
    try {
        st = build_model(hd, h->model, err);
    } catch (const std::bad_alloc&) {
        return fail(nullptr, CTD_ENOMEM, "ctd_create: out of memory while building the model");
    }
    if (st) return fail(nullptr, st, err);
    const Model& mo = h->model;
    if (mo.L.cs > 3 && desc->device >= 0 && !runtime_ocp(mo.problem))      // (host-only handles -- sizes, bounds, patterns -- take any)
        return fail(nullptr, CTD_EINVAL, "ctd_create: control_steps > 3 needs an OCP registered at run time (ctd_register_ocp); the compiled registry holds the midpoint kernels for 1, 2 and 3 controls per step");
    h->step_begin = desc->step_begin;
    h->step_end = desc->step_end;
    if (h->step_begin == 0 && h->step_end == 0) h->step_end = mo.L.N;
    if (h->step_begin < 0 || h->step_end > mo.L.N || h->step_begin >= h->step_end)
        return fail(nullptr, CTD_EINVAL, "ctd_create: shard [step_begin, step_end) is not inside [0, N)");
    h->tile = env_int("CTD_TILE", 0);
    if (h->tile <= 0) h->tile = default_tile(mo, h->step_end - h->step_begin);
    int maxb = 256;
    for_problem(mo.problem, [&](auto tag) { maxb = decltype(tag)::type::MAXB; });
    h->rt = runtime_ocp(mo.problem);
    if (h->rt) maxb = h->rt->maxb;
    h->block = env_int("CTD_BLOCK", 0);
    if (h->block <= 0) {
        // 256 lanes, or 320 on small grids (everything resident at once) when a fifth wave lets the emit phase hold one more
        // replica of the CSC period: 10 000-step Goddard / GL2, period 102: 3 x 102 = 306 of 320 lanes, 7.5 us vs 7.8 us
        h->block = 256;
        const int64_t ntl = (h->step_end - h->step_begin + h->tile - 1) / h->tile;
        // (not for the wide OCPs: their kernels keep fewer workgroups resident with five waves each -- 8-state quadrotor, 2 controls per
        // step, optimized pattern, 40-step tiles: 501 workgroups no longer start together, 14.4 us against 8.8)
        if (ntl <= 512 && mo.nch_dyn <= 1 && mo.Lseg > 0 && 320 / mo.Lseg > 256 / mo.Lseg && mo.Lseg * (320 / mo.Lseg) * 10 >= 320 * 9) h->block = 320;
        // (a fifth wave for the symbolic path rows of the wide OCPs was measured slower -- quadrotor12 midpoint N = 20 000 30.1 us
        // against 24.4 -- and is gone: profiles/r03_experiments.md)
    }
    if (h->block < 64 || (h->block % 64)) h->block = 256;
    if (h->block > maxb) h->block = maxb;
    mo.fill_kparams(h->kp, h->step_begin, h->step_end, h->tile);
    h->lds_bytes = (size_t)lds_doubles(h->kp) * sizeof(double);
    const int debug_stop = env_int("CTD_DEBUG_STOP", 0);
    const size_t lds_cap = h->rt ? 64 * 1024 : 80 * 1024;   // module-API kernels stay inside the default 64 KiB of dynamic LDS
    while (h->lds_bytes > lds_cap && h->tile > 1) {        // a requested tile that does not fit (2 workgroups / CU) is shrunk, not rejected
        h->tile = (h->tile + 1) / 2;
        mo.fill_kparams(h->kp, h->step_begin, h->step_end, h->tile);
        h->lds_bytes = (size_t)lds_doubles(h->kp) * sizeof(double);
    }
    // Gauss-Legendre schemes of the narrow OCPs on long grids: the kernel keeps R workgroups per CU resident (R from the runtime:
    // registers and LDS of the instantiation), a round of 256 R tiles, and its time follows rounds x steps per tile -- Goddard GL3,
    // N = 80 000 at the 32 steps of default_tile: 2501 tiles = 2.44 rounds of 1024, the workgroups start in three waves
    // (profiles/r03_phase_stamps_final.log: start percentiles 0.1 / 10.1 / 19.3 us).  A tile of up to 48 steps with the same R fixes
    // the number of rounds, the steps are spread evenly over them: 40 steps = 2001 tiles, 28.3 -> 26.9 us, optimized pattern
    // 20.3 -> 17.9; N = 200 000: 79.9 -> 74.3 (profiles/r03_experiments.md).  Only ever a LARGER tile than the rule's.
    const bool long_mid = mo.L.sc == SC_MIDPOINT && mo.L.cs == 1;      // midpoint and both Euler schemes (one point per step): long-grid rule only
    if (desc->device >= 0 && !h->rt && env_int("CTD_TILE", 0) <= 0 && env_int("CTD_ROUND_TILES", 1) && mo.nch_dyn <= 1 && (mo.L.sc == SC_IRK || long_mid) && h->tile >= 16) {
        DeviceGuard dgq(desc->device);
        if (dgq.err == hipSuccess) {
            auto resident = [&](int tile, size_t& lds) {
                KParams kq;
                mo.fill_kparams(kq, h->step_begin, h->step_end, tile);
                lds = (size_t)lds_doubles(kq) * sizeof(double);
                int r = 0;
                for_problem(mo.problem, [&](auto tag) { r = occupancy_cons_jac<typename decltype(tag)::type>(mo.L.sc, kq, h->block, lds); });
                (void)hipGetLastError();
                return r;
            };
            size_t l0 = 0;
            const int r0 = resident(h->tile, l0);
            const int64_t ns = h->step_end - h->step_begin, cap = 256 * (int64_t)r0;
            // VERY long grids (8 rounds of resident workgroups and more: from ~half a million steps; crossover measured in
            // profiles/r04_long_grid_crossover.log: 2^18 steps +-5 % either way, 2^19 +3 .. +15 %, 2^20 +5 .. +18 %) are bandwidth-bound, the rounds no
            // longer quantise anything, and what is left per tile is its fixed cost -- launch slot, argument pinning, time table,
            // prologue latency.  Eight waves per workgroup and the largest tile whose records fit 64 KiB of LDS (two workgroups
            // per CU) amortise it: Goddard GL2, 4 194 304 steps 930 -> 795 us (0.59 -> 0.69 of 8 TB/s), GL3 1333 -> 1297,
            // double integrator + path GL2 508 -> 460, double integrator free t0 / tf GL3 915 -> 800, goddard_all GL2 (2 M steps,
            // staged driver) 598 -> 556; the midpoint scheme (records up to 74 KiB): double integrator + path, 8 M steps 506 -> 421, Goddard 4 M 350 -> 283, explicit / implicit Euler 335 -> 288 / 294, DI + path Euler 8 M 455 -> 367;
            // trapeze: flat, left alone (profiles/r04_tiles_long_grids.log; CTD_LONG_GRID=0: off; CTD_LONG_GRID_ROUNDS: the threshold, for tests)
            if (r0 > 0 && (ns + h->tile - 1) / h->tile >= env_int("CTD_LONG_GRID_ROUNDS", 8) * cap && maxb >= 512 && env_int("CTD_BLOCK", 0) <= 0 &&
                env_int("CTD_LONG_GRID", 1)) {
                int tbest = h->tile;
                for (int tt = h->tile + 1; tt <= 256; ++tt) {
                    KParams kq;
                    mo.fill_kparams(kq, h->step_begin, h->step_end, tt);
                    if ((size_t)lds_doubles(kq) * sizeof(double) > (long_mid ? 74u : 64u) * 1024) break;
                    tbest = tt;
                }
                if (tbest > h->tile) {
                    h->tile = tbest;
                    h->block = 512;
                    mo.fill_kparams(h->kp, h->step_begin, h->step_end, h->tile);
                    h->lds_bytes = (size_t)lds_doubles(h->kp) * sizeof(double);
                }
            } else if (mo.L.sc == SC_IRK && r0 > 0 && (ns + h->tile - 1) / h->tile > cap) {
                int tmax = 48;
                for (; tmax > h->tile; --tmax) { size_t l = 0; if (resident(tmax, l) >= r0 && l <= lds_cap) break; }
                const int64_t rounds = (ns + (int64_t)tmax * cap - 1) / ((int64_t)tmax * cap);
                const int64_t tb = (ns + rounds * cap - 1) / (rounds * cap);
                if (tb > h->tile && tb <= tmax) {
                    h->tile = (int)tb;
                    mo.fill_kparams(h->kp, h->step_begin, h->step_end, h->tile);
                    h->lds_bytes = (size_t)lds_doubles(h->kp) * sizeof(double);
                }
            }
        }
    }
    h->grid = h->kp.ntiles + h->kp.has_edge;
    h->kp.debug_stop = debug_stop;
    {   // write-through stores (emit_store) for launches whose outputs are small: what a kernel leaves dirty in the XCDs' L2s is written
        // back at its end, serial with the next launch.  CTD_WT_STORE: 0 never, 1 always, unset: outputs of this handle up to
        // CTD_WT_MB megabytes (default from the MI355X sweep in profiles/r03_experiments.md)
        const int wt = env_int("CTD_WT_STORE", -1);
        const double out_mb = 8.0 * ((double)(h->step_end - h->step_begin) * (mo.L.cb + mo.Lseg + (double)mo.L.nv * mo.vr)) / 1.0e6;
        h->kp.wt_store = wt >= 0 ? (wt ? 1 : 0) : (out_mb <= (double)env_int("CTD_WT_MB", 64) ? 1 : 0);
    }
    h->kp.xcd_remap = env_int("CTD_XCD", 0);          // 1: every XCD walks one contiguous run of tiles (measured neutral)
    h->device = desc->device;
    if (h->device >= 0) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || h->device >= ndev)
            return fail(nullptr, CTD_ENODEVICE, "ctd_create: HIP device not available");
        ctd_handle* hp = h.get();
        if (h->lds_bytes > 160 * 1024) return fail(nullptr, CTD_EINVAL, "ctd_create: tile does not fit the 160 KiB LDS");
        DeviceGuard dg_(hp->device); HIP_TRY(nullptr, dg_.err);
        if (desc->stream_mode == CTD_STREAM_GIVEN) { hp->stream = (hipStream_t)desc->stream; hp->own_stream = false; }
        else { HIP_TRY(nullptr, hipStreamCreateWithFlags(&hp->stream, hipStreamNonBlocking)); hp->own_stream = true; }
        // The normalized grid is always read from this table (N + 1 doubles, L2-resident): for a uniform grid the kernels
        // could compute tau_i = i / N themselves, but an FP64 division is ~35 dependent instructions on the evaluating lane's
        // critical path while the table load is issued together with the loads of x (the values are identical: the host
        // fills the table with the same division).
        HIP_TRY(nullptr, upload(&hp->d_tau, mo.tau));
        HIP_TRY(nullptr, upload(&hp->d_tmpl, mo.tmpl));
        HIP_TRY(nullptr, upload(&hp->d_vtmpl, mo.vtmpl));
        HIP_TRY(nullptr, upload(&hp->d_edge_idx, mo.edge_idx));
        HIP_TRY(nullptr, upload(&hp->d_edge_code, mo.edge_code));
        hp->kp.tau = hp->d_tau;
        hp->kp.tmpl = hp->d_tmpl;
        hp->kp.vtmpl = hp->d_vtmpl;
        // early emission (ctd_layout.hpp KParams::pos): direct tiles of the Gauss-Legendre schemes, when the lead wave can hold one
        // lane per early output of a step and the late positions fit the workgroup.  OFF by default (CTD_EARLY=1 switches it on):
        // measured on MI355X it LOSES -- Goddard GL2 N = 10 000: 8.4 us against 6.2, GL3 optimized pattern 28.8 against 20.1
        // (profiles/r03_experiments.md) -- the early and the late stores each write PARTS of the same 128-byte lines (the state
        // rows and the stage rows of one CSC column interleave), and two partial-line writes cost more than the overlap gains
        {
            const int S = mo.L.s, r_dyn = S * mo.nch_dyn, r_path = mo.nch_path;
            int lgT = 0;
            while ((1 << lgT) < hp->tile) ++lgT;
            const int leadbase = (((r_dyn + r_path) << lgT) + 63) & ~63;
            const int early_lanes = mo.n_early + mo.c_early + mo.L.nv * mo.vr_early;
            if (env_int("CTD_EARLY", 0) && mo.n_early > 0 && !hp->rt && mo.fused && hp->tile <= 32 && leadbase + 64 <= hp->block && early_lanes <= 64 &&
                mo.n_late > 0 && mo.n_late <= hp->block) {
                HIP_TRY(nullptr, upload(&hp->d_pos, mo.pos_order));
                hp->kp.pos = hp->d_pos;
                hp->kp.n_late = mo.n_late; hp->kp.n_early = mo.n_early; hp->kp.c_early = mo.c_early; hp->kp.vr_early = mo.vr_early;
                hp->kp.div_late = make_fastdiv((uint32_t)mo.n_late);
            }
        }
        hp->kp.edge_idx = hp->d_edge_idx;
        hp->kp.edge_code = hp->d_edge_code;
        hp->obj_blocks = 256;
        HIP_TRY(nullptr, hipMalloc((void**)&hp->d_partial, sizeof(double) * hp->obj_blocks));
        HIP_TRY(nullptr, hipMalloc((void**)&hp->d_obj, sizeof(double)));
        HIP_TRY(nullptr, hipEventCreate(&hp->ev0));
        HIP_TRY(nullptr, hipEventCreate(&hp->ev1));
        if (hp->rt) {
            if (hp->lds_bytes > 64 * 1024) return fail(nullptr, CTD_EINVAL, "ctd_create: one step of this run-time OCP does not fit 64 KiB of LDS");
            int32_t jst = jit_load_first(hp);
            if (jst) return jst;
        }
        // Multi-tile workgroups (staged driver only; KParams::wg_stride): when the tiles need several rounds of resident
        // workgroups, launch ONE round and let every workgroup walk its share of the tiles -- templates / v / codes fetched once
        // per workgroup, the next tile's x slice in flight during the emission.  The round is what the runtime says is resident
        // (registers and LDS of this very kernel).  EXPERIMENT: builds with -DCTD_MULTI_TILE_LOOP=1 and CTD_MULTI_TILE=1 (k > 1: k
        // workgroups per CU) only -- measured slower than the hardware's own dispatch of one tile per workgroup (ctd_kernels.hpp).
        {
            const bool direct = mo.fused && mo.L.sc != SC_TRAPEZE;
            const int mt = env_int("CTD_MULTI_TILE", 0);
            if (!direct && mt > 0 && kMultiTileLoop && !hp->rt) {
                int per_cu = mt > 1 ? mt : 0, cus = 256;
                if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, hp->device) != hipSuccess) cus = 256;
                if (per_cu == 0) {
                    if (hp->rt) {
                        if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, hp->f_cons_jac, hp->block, hp->lds_bytes) != hipSuccess) per_cu = 0;
                    } else {
                        for_problem(mo.problem, [&](auto tag) {
                            using P = typename decltype(tag)::type;
                            per_cu = occupancy_cons_jac<P>(mo.L.sc, hp->kp, hp->block, hp->lds_bytes);
                        });
                    }
                }
                const int64_t cap = (int64_t)per_cu * cus - hp->kp.has_edge;      // (the edge blocks hold slots of their own)
                if (cap > 0 && hp->kp.ntiles > cap) {
                    hp->kp.wg_stride = (int)cap;
                    hp->grid = hp->kp.wg_stride + hp->kp.has_edge;
                }
            }
        }
    }
    *out = h.release();
    return CTD_OK;
}

int32_t ctd_register_ocp(const ctd_ocp_def* def, int32_t* problem_id) {
    if (!def || !problem_id) return fail(nullptr, CTD_EINVAL, "ctd_register_ocp: null argument");
    std::string err;
    int id = -1;
    const int st = register_runtime_ocp(def, &id, err);
    if (st) return fail(nullptr, st, "ctd_register_ocp: " + err);
    *problem_id = id;
    return CTD_OK;
}

int32_t ctd_ocp_source(int32_t problem_id, char* buf, int64_t cap) {
    const RtOcp* ro = runtime_ocp(problem_id);
    if (!ro) return fail(nullptr, CTD_EINVAL, "ctd_ocp_source: not a run-time problem id");
    const int64_t need = (int64_t)ro->functor_src.size() + 1;
    if (!buf || cap < need)        // never a silently truncated text: the message names the capacity to come back with
        return fail(nullptr, CTD_EINVAL, "ctd_ocp_source: buffer too small, needs " + std::to_string(need) + " bytes");
    std::memcpy(buf, ro->functor_src.c_str(), (size_t)need);
    return CTD_OK;
}

// compile-only check (no device needed): the kernels of `scheme` for a registered OCP build for gfx950
int32_t ctd_jit_check(int32_t problem_id, int32_t scheme) {
    const RtOcp* ro = runtime_ocp(problem_id);
    if (!ro) return fail(nullptr, CTD_EPROBLEM, "ctd_jit_check: not a run-time problem id");
    if (scheme < 0 || scheme > 8) return fail(nullptr, CTD_ESCHEME, "Unknown discretization method");
    const int sc = scheme == 0 ? SC_TRAPEZE : ((scheme == 1 || scheme >= 7) ? SC_MIDPOINT : SC_IRK);
    int s = (scheme < 2 || scheme >= 7) ? 0 : (scheme <= 4 ? scheme - 1 : scheme - 3);
    if (scheme == 1) s = env_int("CTD_JIT_CHECK_CS", 0);       // midpoint: controls per step of the kernels to build (default 1)
    std::string code, err;
    std::vector<std::string> names;
    int32_t st = jit_compile(*ro, "ctd_kernels.hpp", jit_first_exprs(sc, s), "off", code, names, err);
    if (st == CTD_OK) st = jit_compile(*ro, "ctd_hess_kernels.hpp", jit_hess_exprs(sc, s), "fast", code, names, err);
    if (st) return fail(nullptr, st, err);
    return CTD_OK;
}

int32_t ctd_destroy(ctd_handle* h) {
    if (!h) return CTD_EINVAL;
    free_device(h);
    delete h;
    return CTD_OK;
}

int32_t ctd_sizes(const ctd_handle* h, int64_t* nvar, int64_t* ncon, int64_t* nnzj, int64_t* nnzh) {
    if (!h) return CTD_EINVAL;
    if (nvar) *nvar = h->model.L.nvar;
    if (ncon) *ncon = h->model.L.ncon;
    if (nnzj) *nnzj = h->model.nnzj;
    if (nnzh) *nnzh = h->model.H.nnzh;
    return CTD_OK;
}

int32_t ctd_dims(const ctd_handle* h, int64_t* o) {
    if (!h || !o) return CTD_EINVAL;
    const Layout& L = h->model.L;
    const ProblemInfo& pi = h->model.info;
    o[0] = L.n; o[1] = L.m; o[2] = L.nv; o[3] = L.p; o[4] = L.bc; o[5] = L.N;
    o[6] = L.blk; o[7] = L.eqs; o[8] = L.p; o[9] = L.s; o[10] = L.final_control;
    o[11] = pi.it0 >= 0; o[12] = pi.itf >= 0; o[13] = pi.lagrange; o[14] = pi.mayer; o[15] = pi.maximize;
    return CTD_OK;
}

int32_t ctd_time_grid(const ctd_handle* h, double* normalized, double* fixed) {
    if (!h) return CTD_EINVAL;
    const Model& mo = h->model;
    if (normalized) std::memcpy(normalized, mo.tau.data(), sizeof(double) * (mo.L.N + 1));
    if (fixed) for (int64_t i = 0; i <= mo.L.N; ++i) fixed[i] = mo.fixed_time(i);
    return CTD_OK;
}

// get_time_grid(xu, docp), src/DOCP_data.jl:437-458 (host arithmetic on the tail of x: post-processing, not the hot path)
int32_t ctd_time_grid_at(const ctd_handle* h, const double* x, double* grid) {
    if (!h || !x || !grid) return CTD_EINVAL;
    const Model& mo = h->model;
    const Layout& L = mo.L;
    const double t0 = L.it0 >= 0 ? x[L.v_off + L.it0] : L.t0;
    const double tf = L.itf >= 0 ? x[L.v_off + L.itf] : L.tf;
    for (int64_t i = 0; i <= L.N; ++i) grid[i] = L.free_time ? t0 + mo.tau[i] * (tf - t0) : mo.fixed_time(i);
    return CTD_OK;
}

int32_t ctd_butcher(const ctd_handle* h, double* a, double* b, double* c) {
    if (!h) return CTD_EINVAL;
    const Layout& L = h->model.L;
    for (int i = 0; i < L.s; ++i) {
        if (b) b[i] = L.b[i];
        if (c) c[i] = L.c[i];
        if (a) for (int j = 0; j < L.s; ++j) a[i * L.s + j] = L.a[3 * i + j];
    }
    return CTD_OK;
}

int32_t ctd_bounds(const ctd_handle* h, double* lvar, double* uvar, double* lcon, double* ucon) {
    if (!h) return CTD_EINVAL;
    const Model& mo = h->model;
    mo.fill_bounds(lvar, uvar, lcon, ucon);
    return CTD_OK;
}

int32_t ctd_initial_guess(const ctd_handle* h, double* x0, const ctd_init* init) {
    if (!h || !x0) return CTD_EINVAL;
    if (init) {
        InitSamples sm;
        if (init->n_samples > 0) {
            if (!init->t_samples) return fail(const_cast<ctd_handle*>(h), CTD_EINVAL, "ctd_initial_guess: t_samples is NULL");
            for (int64_t k = 1; k < init->n_samples; ++k)
                if (!(init->t_samples[k] > init->t_samples[k - 1]))
                    return fail(const_cast<ctd_handle*>(h), CTD_EGRID, "ctd_initial_guess: t_samples must be strictly increasing");
            sm.n = init->n_samples; sm.t = init->t_samples; sm.state = init->state_samples; sm.control = init->control_samples;
        }
        model_initial_guess(h->model, x0, init->use_problem_default != 0, init->state, init->control, init->variable, sm);
    }
    else model_initial_guess(h->model, x0, false, nullptr, nullptr, nullptr);
    return CTD_OK;
}

int32_t ctd_jac_structure(const ctd_handle* h, int64_t* rows, int64_t* cols) {
    if (!h || !rows || !cols) return CTD_EINVAL;
    const Model& mo = h->model;
    std::vector<int64_t> r;
    int64_t nz = 0;
    if (mo.order == 1) {                 // CTD_ORDER_CSR: the k-th value belongs to the k-th entry read by rows
        for (int64_t i = 0; i < mo.L.ncon; ++i) {
            mo.gen_row(i, r);
            for (int64_t col : r) { rows[nz] = i + 1; cols[nz] = col + 1; ++nz; }
        }
        return nz == mo.nnzj ? CTD_OK : CTD_EPATTERN;
    }
    for (int64_t j = 0; j < mo.L.nvar; ++j) {
        mo.gen_column(j, r);
        for (int64_t row : r) { rows[nz] = row + 1; cols[nz] = j + 1; ++nz; }
    }
    return nz == mo.nnzj ? CTD_OK : CTD_EPATTERN;
}

int32_t ctd_jac_csc(const ctd_handle* h, int64_t* colptr, int64_t* rowval) {
    if (!h || !colptr || !rowval) return CTD_EINVAL;
    const Model& mo = h->model;
    std::vector<int64_t> r;
    int64_t nz = 0;
    for (int64_t j = 0; j < mo.L.nvar; ++j) {
        colptr[j] = nz;
        mo.gen_column(j, r);
        for (int64_t row : r) rowval[nz++] = row;
    }
    colptr[mo.L.nvar] = nz;
    return nz == mo.nnzj ? CTD_OK : CTD_EPATTERN;
}

int32_t ctd_jac_csr(const ctd_handle* h, int64_t* rowptr, int64_t* colind) {
    if (!h || !rowptr || !colind) return CTD_EINVAL;
    const Model& mo = h->model;
    std::vector<int64_t> c;
    int64_t nz = 0;
    for (int64_t i = 0; i < mo.L.ncon; ++i) {
        rowptr[i] = nz;
        if (mo.order == 1 && mo.row_start(i) != nz) return fail(const_cast<ctd_handle*>(h), CTD_EPATTERN, "internal: row_start disagrees with the generated rows");
        mo.gen_row(i, c);
        for (int64_t col : c) colind[nz++] = col;
    }
    rowptr[mo.L.ncon] = nz;
    return nz == mo.nnzj ? CTD_OK : CTD_EPATTERN;
}

int32_t ctd_value_order(const ctd_handle* h, int32_t* order) {
    if (!h || !order) return CTD_EINVAL;
    *order = h->model.order;
    return CTD_OK;
}

int32_t ctd_dropped_nonzeros(const ctd_handle* h, int64_t* count) {
    if (!h || !count) return CTD_EINVAL;
    *count = h->model.dropped;
    return CTD_OK;
}

int32_t ctd_shard_info(const ctd_handle* h, int64_t* o) {
    if (!h || !o) return CTD_EINVAL;
    const Model& mo = h->model;
    const Layout& L = mo.L;
    const bool first = h->step_begin == 0, last = h->step_end == L.N;
    o[0] = h->step_begin; o[1] = h->step_end;
    o[2] = h->step_begin * L.cb;
    o[3] = h->step_end * L.cb;      // (+ the p + bc tail rows [N*cb, ncon), which every shard writes)
    o[4] = mo.shard_vals_begin(h->step_begin);
    o[5] = mo.shard_vals_end(h->step_end);
    o[6] = first; o[7] = last;
    return CTD_OK;
}

// Sharded iterate read in place: from now on the constraint / Jacobian kernels of this handle load the variables of OTHER
// shards -- the next shard's first node, the previous shard's last step block, X_1, X_{N+1} -- from x_bufs[k] (peer-mapped or
// IPC-mapped full-length buffers) instead of the x passed to the call.  The table is a few hundred bytes of device memory,
// rewritten only when the arguments change.
int32_t ctd_set_x_shards(ctd_handle* h, int32_t n_shards, const int64_t* step_begin, const double* const* x_bufs, int32_t self) {
    if (!h) return CTD_EINVAL;
    if (h->device < 0) return fail(h, CTD_ENODEVICE, "host-only handle");
    if (n_shards <= 0 || !step_begin || !x_bufs) {         // back to "x holds everything this handle reads"
        h->kp.halo = nullptr;
        h->halo_host.G = 0;
        return CTD_OK;
    }
    const Layout& L = h->model.L;
    if (n_shards > kMaxShards) return fail(h, CTD_EINVAL, "ctd_set_x_shards: more than 16 shards");
    if (self < 0 || self >= n_shards || step_begin[self] != h->step_begin || step_begin[self + 1] != h->step_end)
        return fail(h, CTD_EINVAL, "ctd_set_x_shards: step_begin[self], step_begin[self + 1] do not name this handle's shard");
    if (step_begin[0] != 0 || step_begin[n_shards] != L.N) return fail(h, CTD_EINVAL, "ctd_set_x_shards: the shards must cover [0, N)");
    XHalo t{};
    t.G = n_shards; t.self = self;
    for (int k = 0; k < n_shards; ++k) {
        if (step_begin[k + 1] <= step_begin[k]) return fail(h, CTD_EINVAL, "ctd_set_x_shards: empty shard");
        if (k != self && !x_bufs[k]) return fail(h, CTD_EINVAL, "ctd_set_x_shards: null buffer");
        t.vbegin[k] = step_begin[k] * (int64_t)L.blk;
        t.x[k] = k == self ? nullptr : x_bufs[k];
    }
    t.vbegin[n_shards] = L.v_off;       // the last shard also owns the final node; v is replicated
    if (h->kp.halo && std::memcmp(&t, &h->halo_host, sizeof(XHalo)) == 0) return CTD_OK;
    DeviceGuard dg_(h->device); HIP_TRY(h, dg_.err);
    if (!h->d_halo) HIP_TRY(h, hipMalloc((void**)&h->d_halo, sizeof(XHalo)));
    HIP_TRY(h, hipStreamSynchronize(h->stream));           // launches in flight still read the old table
    HIP_TRY(h, hipMemcpy(h->d_halo, &t, sizeof(XHalo), hipMemcpyHostToDevice));
    h->halo_host = t;
    h->kp.halo = h->d_halo;
    h->kp.near = make_xnear(t, L.blk, L.N, L.v_off);
    return CTD_OK;
}

// ---- the stitched constraint vector for one-process-per-GPU hosts: RCCL all-gather inside the library --------------------
// RCCL is reached through the copy of librccl the host process already has (a Julia / Python host created the communicator
// with it): resolved with dlopen at first use, no link-time dependency.  CTD_RCCL_LIB names the library explicitly.
namespace {
typedef int (*nccl_allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef const char* (*nccl_errstr_fn)(int);
nccl_allgather_fn g_nccl_allgather = nullptr;
nccl_errstr_fn g_nccl_errstr = nullptr;
std::once_flag g_nccl_once;
void nccl_resolve() {
    std::call_once(g_nccl_once, [] {
        void* lib = nullptr;
        if (const char* p = std::getenv("CTD_RCCL_LIB")) lib = dlopen(p, RTLD_NOW);
        for (const char* nm : {"librccl.so.1", "librccl.so"})            // the copy already in the process, whoever loaded it
            if (!lib) lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
        for (const char* nm : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if (!lib) lib = dlopen(nm, RTLD_NOW);
        if (!lib) return;
        g_nccl_allgather = (nccl_allgather_fn)dlsym(lib, "ncclAllGather");
        g_nccl_errstr = (nccl_errstr_fn)dlsym(lib, "ncclGetErrorString");
    });
}
__global__ void stitch_pack_kernel(const double* __restrict__ c, double* __restrict__ send, int64_t row0, int64_t own, int64_t tail0, int64_t tail) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < own) send[e] = c[row0 + e];
    else if (e < own + tail) send[e] = c[tail0 + (e - own)];
}
__global__ void stitch_unpack_kernel(const double* __restrict__ recv, double* __restrict__ c, int64_t ncon, int64_t N, int cb, int G, int64_t smax) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < ncon) c[r] = recv[stitch_src(r, N, cb, G, smax)];
}
}  // namespace

int32_t ctd_shard_steps(int64_t N, int32_t n_shards, int32_t k, int64_t* begin, int64_t* end) {
    if (N < 1 || n_shards < 1 || n_shards > N || k < 0 || k >= n_shards || !begin || !end) return CTD_EINVAL;
    *begin = shard_begin(N, n_shards, k);
    *end = k + 1 == n_shards ? N : shard_begin(N, n_shards, k + 1);
    return CTD_OK;
}

int32_t ctd_stitch_c(ctd_handle* h, void* nccl_comm, int32_t n_ranks, int32_t rank, double* c_dev) {
    if (!h) return CTD_EINVAL;
    if (h->device < 0) return fail(h, CTD_ENODEVICE, "host-only handle");
    if (!nccl_comm || !c_dev || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(h, CTD_EINVAL, "ctd_stitch_c: bad argument");
    const Layout& L = h->model.L;
    const int64_t N = L.N;
    const int G = n_ranks;
    if (G > N || h->step_begin != shard_begin(N, G, rank) || h->step_end != (rank + 1 == G ? N : shard_begin(N, G, rank + 1)))
        return fail(h, CTD_EINVAL, "ctd_stitch_c: the handle's shard is not block `rank` of the balanced split of the grid over n_ranks (ctd_shard_steps)");
    nccl_resolve();
    if (!g_nccl_allgather) return fail(h, CTD_ERCCL, "ctd_stitch_c: librccl (ncclAllGather) not found in this process; set CTD_RCCL_LIB");
    DeviceGuard dg_(h->device); HIP_TRY(h, dg_.err);
    const int64_t tail = L.ncon - N * L.cb;
    auto check = [&](int rc, const char* what) -> int32_t {
        if (rc == 0) return CTD_OK;
        return fail(h, CTD_ERCCL, std::string(what) + ": " + (g_nccl_errstr ? g_nccl_errstr(rc) : "RCCL error ") + " (" + std::to_string(rc) + ")");
    };
    constexpr int kNcclDouble = 8;       // ncclFloat64
    if (N % G == 0 && tail == 0) {       // equal blocks, nothing after them: gathered in place
        const int64_t S = (N / G) * L.cb;
        return check(g_nccl_allgather(c_dev + rank * S, c_dev, (size_t)S, kNcclDouble, nccl_comm, h->stream), "ncclAllGather");
    }
    const int64_t smax = ((N + G - 1) / G) * L.cb + tail;
    if (h->stitch_cap < (int64_t)G * smax) {
        if (h->d_stitch_send) (void)hipFree(h->d_stitch_send);
        if (h->d_stitch_recv) (void)hipFree(h->d_stitch_recv);
        h->d_stitch_send = h->d_stitch_recv = nullptr;
        h->stitch_cap = 0;
        HIP_TRY(h, hipMalloc((void**)&h->d_stitch_send, sizeof(double) * smax));
        HIP_TRY(h, hipMalloc((void**)&h->d_stitch_recv, sizeof(double) * G * smax));
        HIP_TRY(h, hipMemsetAsync(h->d_stitch_send, 0, sizeof(double) * smax, h->stream));
        h->stitch_cap = (int64_t)G * smax;
    }
    const int64_t own = (h->step_end - h->step_begin) * L.cb, mytail = rank + 1 == G ? tail : 0;
    const int64_t np = own + mytail;
    stitch_pack_kernel<<<(unsigned)((np + 255) / 256), 256, 0, h->stream>>>(c_dev, h->d_stitch_send, h->step_begin * L.cb, own, N * L.cb, mytail);
    HIP_TRY(h, hipGetLastError());
    const int32_t st = check(g_nccl_allgather(h->d_stitch_send, h->d_stitch_recv, (size_t)smax, kNcclDouble, nccl_comm, h->stream), "ncclAllGather");
    if (st) return st;
    stitch_unpack_kernel<<<(unsigned)((L.ncon + 255) / 256), 256, 0, h->stream>>>(h->d_stitch_recv, c_dev, L.ncon, N, L.cb, G, smax);
    HIP_TRY(h, hipGetLastError());
    return CTD_OK;
}

// ---- device buffers shared between the processes of one node (one process per GPU) ------------------------------------
// hipIpcGetMemHandle names a whole allocation: the handle of the allocation that holds dev_ptr plus dev_ptr's offset in it
// (a torch tensor is a slice of the caching allocator's block).
int32_t ctd_ipc_export(int32_t device, const void* dev_ptr, void* handle64, int64_t* offset) {
    if (!dev_ptr || !handle64 || !offset) return CTD_EINVAL;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "CTD_IPC_HANDLE_BYTES");
    DeviceGuard dg(device);
    hipError_t e = dg.err;
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (e == hipSuccess) e = hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)dev_ptr);
    hipIpcMemHandle_t hd;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&hd, (void*)base);
    if (e != hipSuccess) { g_create_err = std::string("ctd_ipc_export: ") + hipGetErrorString(e); (void)hipGetLastError(); return CTD_EHIP; }
    std::memcpy(handle64, &hd, sizeof(hd));
    *offset = (int64_t)((const char*)dev_ptr - (const char*)base);
    return CTD_OK;
}
int32_t ctd_ipc_open(int32_t device, const void* handle64, void** base) {
    if (!handle64 || !base) return CTD_EINVAL;
    *base = nullptr;
    DeviceGuard dg(device);
    hipError_t e = dg.err;
    hipIpcMemHandle_t hd;
    std::memcpy(&hd, handle64, sizeof(hd));
    if (e == hipSuccess) e = hipIpcOpenMemHandle(base, hd, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { g_create_err = std::string("ctd_ipc_open: ") + hipGetErrorString(e); (void)hipGetLastError(); return CTD_ERCCL; }
    return CTD_OK;
}
// reads `bytes` (<= 64) at a mapped pointer with a device-to-device copy on `device`: a mapping the device cannot reach comes back
// as an error code here instead of as a fault inside a kernel
int32_t ctd_ipc_probe(int32_t device, const void* ptr, size_t bytes) {
    if (!ptr || bytes == 0 || bytes > 64) return CTD_EINVAL;
    DeviceGuard dg(device);
    hipError_t e = dg.err;
    void* tmp = nullptr;
    if (e == hipSuccess) e = hipMalloc(&tmp, 64);
    if (e == hipSuccess) e = hipMemcpy(tmp, ptr, bytes, hipMemcpyDeviceToDevice);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (tmp) (void)hipFree(tmp);
    if (e != hipSuccess) { g_create_err = std::string("ctd_ipc_probe: ") + hipGetErrorString(e); (void)hipGetLastError(); return CTD_ERCCL; }
    return CTD_OK;
}
int32_t ctd_ipc_close(int32_t device, void* base) {
    if (!base) return CTD_OK;
    DeviceGuard dg(device);
    return (dg.err == hipSuccess && hipIpcCloseMemHandle(base) == hipSuccess) ? CTD_OK : CTD_EHIP;
}

int32_t ctd_launch_info(const ctd_handle* h, int64_t* o) {
    if (!h || !o) return CTD_EINVAL;
    o[0] = h->grid; o[1] = h->block; o[2] = (int64_t)h->lds_bytes; o[3] = h->tile; o[4] = h->model.Lseg;
    o[5] = (int64_t)h->model.edge_idx.size();
    o[6] = (h->model.fused && h->model.L.sc != SC_TRAPEZE) ? 1 : 0;      // DirectTile<P, SC>
    // resident workgroups per CU of the kernel this handle launches, as the runtime computes it from the kernel's registers and
    // LDS (0: host-only handle / query failed): the tiles' rounds = grid / (o[7] * CUs)
    o[7] = 0;
    if (h->device >= 0) {
        DeviceGuard dg(h->device);
        int per_cu = 0;
        if (dg.err == hipSuccess) {
            if (h->rt) { if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, h->f_cons_jac, h->block, h->lds_bytes) != hipSuccess) per_cu = 0; }
            else for_problem(h->model.problem, [&](auto tag) {
                using P = typename decltype(tag)::type;
                per_cu = occupancy_cons_jac<P>(h->model.L.sc, h->kp, h->block, h->lds_bytes);
            });
        }
        (void)hipGetLastError();
        o[7] = per_cu;
    }
    return CTD_OK;
}

// ---- hot path ------------------------------------------------------------------------------------------------

static int32_t enqueue_cons_jac(ctd_handle* h, const double* x_dev, double* c_dev, double* vals_dev, hipEvent_t te0 = nullptr,
                                hipEvent_t te1 = nullptr) {
    if (!h) return CTD_EINVAL;
    if (h->device < 0) return fail(h, CTD_ENODEVICE, "compute call on a host-only handle (device = -1); there is no CPU fallback");
    if (!x_dev) return fail(h, CTD_EINVAL, "x is null");
    DeviceGuard dg_(h->device); HIP_TRY(h, dg_.err);
    KParams kp = h->kp;
    kp.c = c_dev;
    kp.vals = vals_dev;
    hipError_t e = hipErrorInvalidValue;
    const int sc = h->model.L.sc;
    if (h->rt) {
        void* args[] = {&kp, &x_dev};
        e = jit_launch(h->f_cons_jac, h->grid, h->block, h->lds_bytes, h->stream, args, te0, te1);
    }
    for_problem(h->model.problem, [&](auto tag) {
        using P = typename decltype(tag)::type;
        e = launch_cons_jac<P>(sc, kp, x_dev, h->grid, h->block, h->lds_bytes, h->stream, te0, te1);
    });
    if (e != hipSuccess) return fail(h, CTD_EHIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return CTD_OK;
}

int32_t ctd_cons_jac_dev_async(ctd_handle* h, const double* x_dev, double* c_dev, double* vals_dev) {
    return enqueue_cons_jac(h, x_dev, c_dev, vals_dev);
}

// Launch on another stream from now on (e.g. the capturing stream while the caller records a HIP graph of a whole solver
// iteration).  The handle never owns a stream passed this way.
int32_t ctd_set_stream(ctd_handle* h, void* stream) {
    if (!h) return CTD_EINVAL;
    if (h->device < 0) return fail(h, CTD_ENODEVICE, "host-only handle");
    if (h->own_stream && h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    h->stream = (hipStream_t)stream;
    h->own_stream = false;
    return CTD_OK;
}

int32_t ctd_sync(ctd_handle* h) {
    if (!h) return CTD_EINVAL;
    if (h->device < 0) return fail(h, CTD_ENODEVICE, "host-only handle");
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return CTD_OK;
}

int32_t ctd_cons_jac_dev(ctd_handle* h, const double* x_dev, double* c_dev, double* vals_dev) {
    int32_t st = enqueue_cons_jac(h, x_dev, c_dev, vals_dev);
    if (st) return st;
    return ctd_sync(h);
}

static int32_t ensure_staging(ctd_handle* h, bool need_c, bool need_vals) {
    const Model& mo = h->model;
    DeviceGuard dg_(h->device); HIP_TRY(h, dg_.err);
    if (!h->d_x) HIP_TRY(h, hipMalloc((void**)&h->d_x, sizeof(double) * mo.L.nvar));
    if (need_c && !h->d_c) HIP_TRY(h, hipMalloc((void**)&h->d_c, sizeof(double) * mo.L.ncon));
    if (need_vals && !h->d_vals) HIP_TRY(h, hipMalloc((void**)&h->d_vals, sizeof(double) * (mo.nnzj > 0 ? mo.nnzj : 1)));
    return CTD_OK;
}

static int32_t host_cons_jac(ctd_handle* h, const double* x, double* c, double* vals) {
    if (!h) return CTD_EINVAL;
    if (h->device < 0) return fail(h, CTD_ENODEVICE, "compute call on a host-only handle (device = -1); there is no CPU fallback");
    if (!x) return fail(h, CTD_EINVAL, "x is null");
    const Model& mo = h->model;
    int32_t st = ensure_staging(h, c != nullptr, vals != nullptr);
    if (st) return st;
    HIP_TRY(h, hipMemcpyAsync(h->d_x, x, sizeof(double) * mo.L.nvar, hipMemcpyHostToDevice, h->stream));
    st = enqueue_cons_jac(h, h->d_x, c ? h->d_c : nullptr, vals ? h->d_vals : nullptr);
    if (st) return st;
    if (c) HIP_TRY(h, hipMemcpyAsync(c, h->d_c, sizeof(double) * mo.L.ncon, hipMemcpyDeviceToHost, h->stream));
    if (vals) HIP_TRY(h, hipMemcpyAsync(vals, h->d_vals, sizeof(double) * mo.nnzj, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return CTD_OK;
}

int32_t ctd_host_alloc(void** ptr, size_t bytes) {
    if (!ptr) return CTD_EINVAL;
    *ptr = nullptr;
    const hipError_t e = hipHostMalloc(ptr, bytes ? bytes : 8, hipHostMallocDefault);
    if (e != hipSuccess) { g_create_err = std::string("ctd_host_alloc: ") + hipGetErrorString(e); return CTD_EHIP; }
    return CTD_OK;
}
int32_t ctd_host_free(void* ptr) {
    if (!ptr) return CTD_OK;
    return hipHostFree(ptr) == hipSuccess ? CTD_OK : CTD_EHIP;
}

int32_t ctd_cons(ctd_handle* h, const double* x, double* c) {
    if (h && !c) return fail(h, CTD_EINVAL, "c is null");
    return host_cons_jac(h, x, c, nullptr);
}
int32_t ctd_jac_coord(ctd_handle* h, const double* x, double* vals) {
    if (h && !vals) return fail(h, CTD_EINVAL, "vals is null");
    return host_cons_jac(h, x, nullptr, vals);
}
int32_t ctd_cons_jac(ctd_handle* h, const double* x, double* c, double* vals) {
    if (h && (!c || !vals)) return fail(h, CTD_EINVAL, "c or vals is null");
    return host_cons_jac(h, x, c, vals);
}

// kernel parameters of the objective pass; returns the quadrature workgroups (0 for a Mayer-only cost)
static int fill_obj_params(ctd_handle* h, double* f_dev, ObjParams& op);

static int32_t enqueue_obj(ctd_handle* h, const double* x_dev, double* f_dev) {
    if (!h) return CTD_EINVAL;
    if (h->device < 0) return fail(h, CTD_ENODEVICE, "compute call on a host-only handle (device = -1); there is no CPU fallback");
    if (!x_dev || !f_dev) return fail(h, CTD_EINVAL, "null argument");
    DeviceGuard dg_(h->device); HIP_TRY(h, dg_.err);
    const Layout& L = h->model.L;
    ObjParams op;
    const int blocks = fill_obj_params(h, f_dev, op);
    const bool lagrange = h->model.info.lagrange;
    hipError_t e = hipErrorInvalidValue;
    if (h->rt) {
        void* args[] = {&op, &x_dev};
        e = lagrange ? jit_launch(h->f_obj_partial, blocks, 256, 0, h->stream, args) : hipSuccess;
        if (e == hipSuccess) e = jit_launch(h->f_obj_finish, 1, 64, 0, h->stream, args);
    }
    for_problem(h->model.problem, [&](auto tag) {
        using P = typename decltype(tag)::type;
        e = launch_obj<P>(L.sc, op, x_dev, lagrange ? blocks : 0, 256, h->stream);
    });
    if (e != hipSuccess) return fail(h, CTD_EHIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return CTD_OK;
}

static int fill_obj_params(ctd_handle* h, double* f_dev, ObjParams& op) {
    const Layout& L = h->model.L;
    std::memset(&op, 0, sizeof(op));
    op.L = L;
    op.tau = h->d_tau;
    const bool last = h->step_end == L.N;
    // quadrature units: trapeze sums over nodes (node N belongs to the last shard), the others over steps
    op.unit_begin = h->step_begin;
    op.unit_end = (L.sc == SC_TRAPEZE && last) ? L.N + 1 : h->step_end;
    op.add_mayer = last ? 1 : 0;
    op.partial = h->d_partial;
    op.out = f_dev;
    op.halo = h->kp.halo;
    op.near = h->kp.near;
    const int64_t units = op.unit_end - op.unit_begin;
    int blocks = (int)((units + 255) / 256);
    if (blocks > h->obj_blocks) blocks = h->obj_blocks;
    if (blocks < 1) blocks = 1;
    const bool lagrange = h->model.info.lagrange;      // Mayer-only cost: no quadrature pass, the finish kernel alone
    op.nblocks = lagrange ? blocks : 0;
    return lagrange ? blocks : 0;
}

int32_t ctd_obj_dev_async(ctd_handle* h, const double* x_dev, double* f_dev) { return enqueue_obj(h, x_dev, f_dev); }

int32_t ctd_obj_dev(ctd_handle* h, const double* x_dev, double* f_host) {
    if (h && !f_host) return fail(h, CTD_EINVAL, "null argument");
    int32_t st = enqueue_obj(h, x_dev, h ? h->d_obj : nullptr);
    if (st) return st;
    HIP_TRY(h, hipMemcpyAsync(f_host, h->d_obj, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return CTD_OK;
}

static int32_t enqueue_grad(ctd_handle* h, const double* x_dev, double* g_dev, GradParams* only_params = nullptr, bool shard = false);
int32_t ctd_grad_dev_async(ctd_handle* h, const double* x_dev, double* g_dev) { return enqueue_grad(h, x_dev, g_dev); }
// the shard's own entries of the gradient from a sharded iterate read in place (see include/ctdirect_hip.h)
int32_t ctd_grad_shard_dev_async(ctd_handle* h, const double* x_dev, double* g_dev) { return enqueue_grad(h, x_dev, g_dev, nullptr, true); }
int32_t ctd_grad_dev(ctd_handle* h, const double* x_dev, double* g_dev) {
    int32_t st = enqueue_grad(h, x_dev, g_dev);
    if (st) return st;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return CTD_OK;
}

static int32_t enqueue_grad(ctd_handle* h, const double* x_dev, double* g_dev, GradParams* only_params, bool shard) {
    if (!h) return CTD_EINVAL;
    if (h->device < 0) return fail(h, CTD_ENODEVICE, "compute call on a host-only handle (device = -1); there is no CPU fallback");
    if (!x_dev || !g_dev) return fail(h, CTD_EINVAL, "null argument");
    // ctd_grad*: the gradient of the WHOLE objective (O(nvar) work on every rank), from the x it is given -- a whole iterate.
    // ctd_grad_shard_dev_async (shard = true): the quadrature units of this handle's steps only (the last shard also the final node),
    // neighbours' blocks through the shard table like the other callbacks: the shard's own entries of g + its partial d/dv
    DeviceGuard dg_(h->device); HIP_TRY(h, dg_.err);
    const Layout& L = h->model.L;
    const bool last = h->step_end == L.N, first = h->step_begin == 0;
    const int64_t all_units = (L.sc == SC_IRK) ? L.N : L.N + 1;
    const int64_t ub = shard ? h->step_begin : 0, ue = shard ? ((L.sc != SC_IRK && last) ? L.N + 1 : h->step_end) : all_units;
    const int64_t units = ue - ub;
    const int blocks = (int)((units + 255) / 256);
    if (!h->d_gpartial || h->gblocks < blocks) {
        if (h->d_gpartial) (void)hipFree(h->d_gpartial);
        h->d_gpartial = nullptr;
        HIP_TRY(h, hipMalloc((void**)&h->d_gpartial, sizeof(double) * (size_t)blocks * kMaxNV));
        h->gblocks = blocks;
    }
    GradParams gp;
    std::memset(&gp, 0, sizeof(gp));
    gp.L = L;
    gp.tau = h->d_tau;
    gp.g = g_dev;
    gp.partial = h->d_gpartial;
    gp.nblocks = blocks;
    gp.unit_begin = ub; gp.unit_end = ue;
    gp.owns_first = (!shard || first) ? 1 : 0;
    gp.owns_last = (!shard || last) ? 1 : 0;
    if (shard) { gp.halo = h->kp.halo; gp.near = h->kp.near; }
    if (only_params) { gp.nblocks = h->model.info.lagrange ? blocks : 0; *only_params = gp; return CTD_OK; }
    // With a Lagrange cost the per-step kernel writes every entry of the step blocks (owner computes), only the tail
    // (final state, variables) needs zeroing before the finish kernel adds the Mayer part; a Mayer-only gradient is zero
    // except at x_0, x_f, v: one memset and the finish kernel
    const bool lagrange = h->model.info.lagrange;
    if (shard) {
        // a shard touches its own entries only: zero what the unit kernel does not write (everything, for a Mayer-only cost), the
        // final-state entries on the last shard, and the nv variable entries (the finish kernel assigns them)
        const int64_t lo = h->step_begin * (int64_t)L.blk, hi = h->step_end * (int64_t)L.blk;
        if (!lagrange) HIP_TRY(h, hipMemsetAsync(g_dev + lo, 0, sizeof(double) * (hi - lo), h->stream));
        if (last && (!lagrange || L.sc == SC_IRK))
            HIP_TRY(h, hipMemsetAsync(g_dev + L.N * (int64_t)L.blk, 0, sizeof(double) * (L.v_off - L.N * (int64_t)L.blk), h->stream));
    } else if (lagrange) HIP_TRY(h, hipMemsetAsync(g_dev + L.N * (int64_t)L.blk, 0, sizeof(double) * (L.nvar - L.N * (int64_t)L.blk), h->stream));
    else HIP_TRY(h, hipMemsetAsync(g_dev, 0, sizeof(double) * L.nvar, h->stream));
    gp.nblocks = lagrange ? blocks : 0;
    hipError_t e = hipErrorInvalidValue;
    if (h->rt) {
        void* args[] = {&gp, &x_dev};
        e = lagrange ? jit_launch(h->f_grad_units, blocks, 256, 0, h->stream, args) : hipSuccess;
        if (e == hipSuccess) e = jit_launch(h->f_grad_finish, 1, 64, 0, h->stream, args);
    }
    for_problem(h->model.problem, [&](auto tag) {
        using P = typename decltype(tag)::type;
        e = launch_grad<P>(L.sc, L.s, gp, x_dev, lagrange ? blocks : 0, h->stream);
    });
    if (e != hipSuccess) return fail(h, CTD_EHIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return CTD_OK;
}

int32_t ctd_grad(ctd_handle* h, const double* x, double* g) {
    if (!h) return CTD_EINVAL;
    if (h->device < 0) return fail(h, CTD_ENODEVICE, "compute call on a host-only handle (device = -1); there is no CPU fallback");
    if (!x || !g) return fail(h, CTD_EINVAL, "null argument");
    int32_t st = ensure_staging(h, false, false);
    if (st) return st;
    const int64_t nvar = h->model.L.nvar;
    if (!h->d_g) HIP_TRY(h, hipMalloc((void**)&h->d_g, sizeof(double) * nvar));
    HIP_TRY(h, hipMemcpyAsync(h->d_x, x, sizeof(double) * nvar, hipMemcpyHostToDevice, h->stream));
    st = ctd_grad_dev(h, h->d_x, h->d_g);
    if (st) return st;
    HIP_TRY(h, hipMemcpy(g, h->d_g, sizeof(double) * nvar, hipMemcpyDeviceToHost));
    return CTD_OK;
}

int32_t ctd_obj(ctd_handle* h, const double* x, double* f) {
    if (!h) return CTD_EINVAL;
    if (h->device < 0) return fail(h, CTD_ENODEVICE, "compute call on a host-only handle (device = -1); there is no CPU fallback");
    if (!x || !f) return fail(h, CTD_EINVAL, "null argument");
    int32_t st = ensure_staging(h, false, false);
    if (st) return st;
    HIP_TRY(h, hipMemcpyAsync(h->d_x, x, sizeof(double) * h->model.L.nvar, hipMemcpyHostToDevice, h->stream));
    return ctd_obj_dev(h, h->d_x, f);
}

int32_t ctd_debug_stamps(ctd_handle* h, const double* x_dev, double* c_dev, double* vals_dev, uint64_t* out, int64_t cap) {
    if (!h || !out) return CTD_EINVAL;
    if (h->device < 0) return fail(h, CTD_ENODEVICE, "compute call on a host-only handle (device = -1); there is no CPU fallback");
    int64_t words = (int64_t)h->grid * 12;
    if (cap < words) return fail(h, CTD_EINVAL, "stamp buffer too small");
    if (cap >= (int64_t)h->grid * 28) words = (int64_t)h->grid * 28;      // + sub-stamps of experiment builds (CTD_SUBSTAMPS)
    DeviceGuard dg_(h->device); HIP_TRY(h, dg_.err);
    unsigned long long* d_st = nullptr;
    HIP_TRY(h, hipMalloc((void**)&d_st, sizeof(unsigned long long) * words));
    HIP_TRY(h, hipMemsetAsync(d_st, 0, sizeof(unsigned long long) * words, h->stream));
    int32_t st = enqueue_cons_jac(h, x_dev, c_dev, vals_dev);     // warm, no stamps
    if (st == CTD_OK) {
        h->kp.stamps = d_st;
        st = enqueue_cons_jac(h, x_dev, c_dev, vals_dev);
        h->kp.stamps = nullptr;
    }
    if (st == CTD_OK) {
        hipError_t e = hipMemcpyAsync(out, d_st, sizeof(unsigned long long) * words, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) st = fail(h, CTD_EHIP, hipGetErrorString(e));
    }
    (void)hipFree(d_st);
    return st;
}

// Mean duration of one dispatch of a kernel, measured the way the timed region of bench.py runs it: the launches are
// enqueued back to back in batches (the GPU stays busy, clocks stay up), every launch carries its own pair of dispatch
// events (start / stop timestamps taken by the dispatch of THAT kernel on the handle's stream), one synchronisation per batch
static int32_t time_dispatches(ctd_handle* h, int32_t iters, double* mean_ms,
                               const std::function<int32_t(hipEvent_t, hipEvent_t)>& launch) {
    constexpr int kBatch = 32;
    DeviceGuard dg_(h->device); HIP_TRY(h, dg_.err);
    static thread_local std::map<int, std::vector<hipEvent_t>> ev_by_device;    // events belong to the device they were created on
    std::vector<hipEvent_t>& ev = ev_by_device[h->device];
    if (ev.empty()) ev.assign(2 * kBatch, nullptr);
    for (hipEvent_t& e : ev)
        if (!e) HIP_TRY(h, hipEventCreate(&e));
    int32_t st = launch(nullptr, nullptr);   // warm
    if (st) return st;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    double total = 0.0;
    for (int done = 0; done < iters;) {
        const int nb = iters - done < kBatch ? iters - done : kBatch;
        for (int i = 0; i < nb; ++i) {
            st = launch(ev[2 * i], ev[2 * i + 1]);
            if (st) return st;
        }
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        for (int i = 0; i < nb; ++i) {
            float ms = 0.f;
            HIP_TRY(h, hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
            total += (double)ms;
        }
        done += nb;
    }
    *mean_ms = total / iters;
    return CTD_OK;
}

int32_t ctd_time_cons_jac_dev(ctd_handle* h, const double* x_dev, double* c_dev, double* vals_dev, int32_t iters, double* mean_ms) {
    if (!h || !mean_ms || iters < 1) return CTD_EINVAL;
    if (h->device < 0) return fail(h, CTD_ENODEVICE, "compute call on a host-only handle (device = -1); there is no CPU fallback");
    return time_dispatches(h, iters, mean_ms, [&](hipEvent_t a, hipEvent_t b) { return enqueue_cons_jac(h, x_dev, c_dev, vals_dev, a, b); });
}

// ---- Hessian of the Lagrangian ----------------------------------------------------------------------------------------

int32_t ctd_hess_structure(const ctd_handle* h, int64_t* rows, int64_t* cols) {
    if (!h || !rows || !cols) return CTD_EINVAL;
    const Model& mo = h->model;
    std::vector<int64_t> r;
    int64_t k = 0;
    for (int64_t j = 0; j < mo.L.nvar; ++j) {
        mo.hess_gen_column(j, r);
        for (int64_t row : r) { rows[k] = row + 1; cols[k] = j + 1; ++k; }
    }
    return CTD_OK;
}

int32_t ctd_hess_csc(const ctd_handle* h, int64_t* colptr, int64_t* rowval);
// upper triangle by rows = lower triangle by columns (symmetric matrix): the same arrays, the same value order
int32_t ctd_hess_csr(const ctd_handle* h, int64_t* rowptr, int64_t* colind) { return ctd_hess_csc(h, rowptr, colind); }

int32_t ctd_hess_csc(const ctd_handle* h, int64_t* colptr, int64_t* rowval) {
    if (!h || !colptr || !rowval) return CTD_EINVAL;
    const Model& mo = h->model;
    std::vector<int64_t> r;
    int64_t k = 0;
    for (int64_t j = 0; j < mo.L.nvar; ++j) {
        colptr[j] = k;
        mo.hess_gen_column(j, r);
        for (int64_t row : r) rowval[k++] = row;
    }
    colptr[mo.L.nvar] = k;
    return k == mo.H.nnzh ? CTD_OK : CTD_EPATTERN;
}

static int32_t ensure_hess(ctd_handle* h) {
    if (h->hess_ready) return CTD_OK;
    const Model& mo = h->model;
    const HessModel& H = mo.H;
    DeviceGuard dg_(h->device); HIP_TRY(h, dg_.err);
    h->hess_tile = env_int("CTD_HESS_TILE", 0);
    if (h->hess_tile <= 0) h->hess_tile = default_hess_tile(mo);
    mo.fill_hparams(h->hp, h->hess_tile, h->step_begin, h->step_end);
    h->hess_lds_bytes = (size_t)hess_lds_doubles(h->hp) * sizeof(double);
    const size_t hess_cap = h->rt ? 64 * 1024 : 96 * 1024;
    while (h->hess_lds_bytes > hess_cap && h->hess_tile > 1) {
        h->hess_tile = (h->hess_tile + 1) / 2;
        mo.fill_hparams(h->hp, h->hess_tile, h->step_begin, h->step_end);
        h->hess_lds_bytes = (size_t)hess_lds_doubles(h->hp) * sizeof(double);
    }
    if (h->hess_lds_bytes > (h->rt ? 64u : 160u) * 1024) return fail(h, CTD_EINVAL, "Hessian records of one step do not fit the LDS");
    if (h->rt && !h->jit_hmod) {
        int32_t jst = jit_load_hess(h);
        if (jst) return jst;
    }
    HIP_TRY(h, upload(&h->d_htptr, H.ctptr));
    HIP_TRY(h, upload(&h->d_hcpos, H.cpos));
    HIP_TRY(h, upload(&h->d_hzpos, H.zpos));
    HIP_TRY(h, upload(&h->d_hterms, H.tcode));
    HIP_TRY(h, upload(&h->d_hpair_c, H.pair_c));
    HIP_TRY(h, upload(&h->d_hvptr, H.vptr));
    HIP_TRY(h, upload(&h->d_hvterms, H.vterms));
    HIP_TRY(h, upload(&h->d_hedge_idx, H.edge_idx));
    HIP_TRY(h, upload(&h->d_heptr, H.eptr));
    HIP_TRY(h, upload(&h->d_hevptr, H.evptr));
    HIP_TRY(h, upload(&h->d_heterms, H.eterms));
    HIP_TRY(h, upload(&h->d_htasks, H.tasks));
    HIP_TRY(h, upload(&h->d_hptasks, H.ptasks));
    HIP_TRY(h, upload(&h->d_hbtasks, H.btasks));
    // Gauss-Legendre schemes with 2 / 3 stages of registry OCPs: the lane-per-step kernel takes the regular steps (CTD_HESS_STEP=0:
    // the tile kernel everywhere)
    std::vector<int32_t> ssrc, schunk;
    int snout = -1;
    const int64_t sb = std::max<int64_t>(h->hp.step_begin, H.reg_first), se = std::min<int64_t>(h->hp.step_end, H.reg_last);
    // CTD_HESS_STEP: 0 never, 2 whenever the OCP has assembly functions, 1 (default) from the grid size on where it wins: a wave
    // of the step kernel needs ~17 us for its 64 steps whatever the grid, the tile kernel's time grows with it (MI355X, Goddard:
    // 3 stages 18.5 vs 19.9 us at 32 000 steps, 28.7 vs 37.7 at 80 000; 2 stages 10.4 vs 10.9 us at 10 000, 14.0 vs 17.8 at 50 000)
    const int step_mode = env_int("CTD_HESS_STEP", 1);
    const int64_t step_min = mo.L.s == 3 ? 28000 : 9000;
    if (!h->rt && mo.L.sc == SC_IRK && (mo.L.s == 2 || mo.L.s == 3) && se > sb && step_mode != 0 && (step_mode == 2 || se - sb >= step_min)) {
        const short* prs = nullptr;
        for_problem(mo.problem, [&](auto tag) { prs = hess_step_pairs<typename decltype(tag)::type>(mo.L.s, mo.L.stagewise != 0, &snout); });
        h->hess_step = prs && snout >= 0 && build_hess_step_tables(mo, prs, snout, kSymStepChunk, ssrc, schunk);
    }
    const int step_wgs = h->hess_step ? (int)((h->hp.step_end - h->hp.step_begin + kStepBlock - 1) / kStepBlock) : 0;
    const int edge_step = 32;      // (upper bound of the edge workgroups of the step launch)
    if (h->d_hpartials) { (void)hipFree(h->d_hpartials); h->d_hpartials = nullptr; }
    HIP_TRY(h, hipMalloc((void**)&h->d_hpartials, sizeof(double) * (size_t)(std::max(h->hp.ntiles, step_wgs) + std::max(h->hp.n_edge_blocks, edge_step)) *
                                                       (H.nvv > 0 ? H.nvv : 1)));
    HParams& hp = h->hp;
    hp.tau = h->d_tau;
    hp.tptr = h->d_htptr; hp.terms = h->d_hterms; hp.pair_c = h->d_hpair_c;
    hp.cpos = h->d_hcpos; hp.zpos = h->d_hzpos;
    hp.vptr = h->d_hvptr; hp.vterms = h->d_hvterms;
    hp.edge_idx = h->d_hedge_idx; hp.eptr = h->d_heptr; hp.evptr = h->d_hevptr; hp.eterms = h->d_heterms;
    hp.tasks = h->d_htasks; hp.ptasks = h->d_hptasks; hp.btasks = h->d_hbtasks;
    hp.partials = h->d_hpartials;
    // V x V partials: added in a fixed order by a second, one-workgroup kernel (a last-workgroup finish inside the main
    // kernel measured 2-7x slower on MI355X, profiles/r01_hessian_kernel.md: device-scope release per workgroup)
    hp.debug_stop = env_int("CTD_HESS_STOP", 0);
    {   // write-through value stores for small Hessians (as for the constraint / Jacobian kernel, ctd_create)
        const int wt = env_int("CTD_HESS_WT", -1);
        const double out_mb = 8.0 * (double)(h->hp.step_end - h->hp.step_begin) * (double)H.Lseg / 1.0e6;
        hp.wt_store = wt >= 0 ? (wt ? 1 : 0) : (out_mb <= (double)std::min(16, env_int("CTD_WT_MB", 64)) ? 1 : 0);       // (sweep: gains up to ~10 MB, even at 16, losses from ~80)
    }
    hp.xcd_remap = env_int("CTD_XCD", 0);
    if (h->hess_step) {
        HIP_TRY(h, upload(&h->d_hssrc, ssrc));
        HIP_TRY(h, upload(&h->d_hschunk, schunk));
        SParams& sp = h->sp;
        sp = SParams{};
        sp.L = mo.L;
        sp.tau = h->d_tau;
        sp.step_begin = h->hp.step_begin; sp.step_end = h->hp.step_end;
        sp.reg_lo = sb; sp.reg_hi = se;
        sp.seg_base = H.seg_base; sp.reg_first = H.reg_first;
        sp.Lseg = H.Lseg; sp.nout = snout; sp.nchunk = (int)schunk.size() - 1; sp.nvv = H.nvv;
        sp.src = h->d_hssrc; sp.chunk_pos = h->d_hschunk;
        {   // constant parts of the chain-rule coefficients (HC_*, ctd_hess.hpp) and of their pairs
            const Layout& L = mo.L;
            double kc[kHC];
            for (int ci = 0; ci < kHC; ++ci) {
                if (ci == HC_ONE) kc[ci] = 1.0;
                else if (ci == HC_HALF) kc[ci] = 0.5;
                else if (ci < HC_A) kc[ci] = L.a[ci - HC_HA];
                else if (ci < HC_B) kc[ci] = L.a[ci - HC_A];
                else if (ci < HC_NBH) kc[ci] = L.b[ci - HC_B];
                else kc[ci] = -L.b[(ci - HC_NBH) % 3];
            }
            std::vector<double> ck((size_t)kHC * kHC);
            for (int c1 = 0; c1 < kHC; ++c1)
                for (int c2 = 0; c2 < kHC; ++c2) ck[(size_t)c1 * kHC + c2] = kc[c1] * kc[c2];
            HIP_TRY(h, upload(&h->d_hsck, ck));
            sp.ck = h->d_hsck;
        }
        sp.partials = h->d_hpartials;
        // the tile kernel's edge path rides in the same launch on kStepBlock lanes: one edge workgroup per 64 edge entries
        h->hp_step = hp;
        const int ntot = (hp.edge_end - hp.edge_begin) + (hp.edge2_end - hp.edge2_begin);
        h->hp_step.n_edge_blocks = std::max(1, std::min(edge_step, (ntot + kStepBlock - 1) / kStepBlock));
        h->hp_step.ntiles = step_wgs;                 // (partials the finish kernel adds)
        sp.part_base = h->hp_step.n_edge_blocks;
        sp.wt_store = hp.wt_store;
        h->hess_step_lds = std::max(hess_step_lds_bytes(snout, mo.L.nv, H.Lseg), (size_t)hess_edge_lds_doubles(h->hp_step) * sizeof(double));
    }
    h->hess_ready = true;
    return CTD_OK;
}

static int32_t enqueue_hess(ctd_handle* h, const double* x_dev, const double* y_dev, double obj_weight, double* vals_dev,
                            hipEvent_t te0 = nullptr, hipEvent_t te1 = nullptr) {
    if (!h) return CTD_EINVAL;
    if (h->device < 0) return fail(h, CTD_ENODEVICE, "compute call on a host-only handle (device = -1); there is no CPU fallback");
    if (!x_dev || !y_dev || !vals_dev) return fail(h, CTD_EINVAL, "null argument");
    int32_t st = ensure_hess(h);
    if (st) return st;
    DeviceGuard dg_(h->device); HIP_TRY(h, dg_.err);
    HParams hp = h->hp;
    hp.obj_weight = obj_weight;
    hp.vals = vals_dev;
    hp.halo = h->kp.halo;
    hp.near = h->kp.near;
    if (hp.halo) { hp.own_lo = h->halo_host.vbegin[h->halo_host.self]; hp.own_hi = h->halo_host.vbegin[h->halo_host.self + 1]; }
    hipError_t e = hipErrorInvalidValue;
    if (h->rt) {
        void* args[] = {&hp, &x_dev, &y_dev};
        e = jit_launch(h->f_hess, hp.ntiles + hp.n_edge_blocks, kHessBlock, h->hess_lds_bytes, h->stream, args, te0, te1);
        if (e == hipSuccess && hp.nvv > 0) e = jit_launch(h->f_hess_finish, 1, kHessBlock, 0, h->stream, args);
    }
    for_problem(h->model.problem, [&](auto tag) {
        using P = typename decltype(tag)::type;
        if (h->hess_step && !hp.stamps && !hp.debug_stop) {
            HParams he = h->hp_step;
            he.obj_weight = obj_weight;
            he.vals = vals_dev;
            he.halo = hp.halo; he.near = hp.near; he.own_lo = hp.own_lo; he.own_hi = hp.own_hi;       // (its edge blocks; a step lane reads its own step only)
            SParams sp = h->sp;
            sp.obj_weight = obj_weight;
            sp.vals = vals_dev;
            e = launch_hess_step<P>(he, sp, x_dev, y_dev, h->hess_step_lds, h->stream, te0, te1);
        } else {
            e = launch_hess<P>(hp, x_dev, y_dev, h->hess_lds_bytes, h->stream, te0, te1);
        }
    });
    if (e != hipSuccess) return fail(h, CTD_EHIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return CTD_OK;
}

int32_t ctd_hess_coord_dev_async(ctd_handle* h, const double* x_dev, const double* y_dev, double obj_weight, double* vals_dev) {
    return enqueue_hess(h, x_dev, y_dev, obj_weight, vals_dev);
}

int32_t ctd_hess_coord_dev(ctd_handle* h, const double* x_dev, const double* y_dev, double obj_weight, double* vals_dev) {
    int32_t st = enqueue_hess(h, x_dev, y_dev, obj_weight, vals_dev);
    if (st) return st;
    return ctd_sync(h);
}

int32_t ctd_hess_coord(ctd_handle* h, const double* x, const double* y, double obj_weight, double* vals) {
    if (!h) return CTD_EINVAL;
    if (h->device < 0) return fail(h, CTD_ENODEVICE, "compute call on a host-only handle (device = -1); there is no CPU fallback");
    if (!x || !y || !vals) return fail(h, CTD_EINVAL, "null argument");
    const Model& mo = h->model;
    int32_t st = ensure_staging(h, false, false);
    if (st) return st;
    if (!h->d_y) HIP_TRY(h, hipMalloc((void**)&h->d_y, sizeof(double) * mo.L.ncon));
    if (!h->d_hvals) HIP_TRY(h, hipMalloc((void**)&h->d_hvals, sizeof(double) * (mo.H.nnzh > 0 ? mo.H.nnzh : 1)));
    HIP_TRY(h, hipMemcpyAsync(h->d_x, x, sizeof(double) * mo.L.nvar, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_y, y, sizeof(double) * mo.L.ncon, hipMemcpyHostToDevice, h->stream));
    st = enqueue_hess(h, h->d_x, h->d_y, obj_weight, h->d_hvals);
    if (st) return st;
    HIP_TRY(h, hipMemcpyAsync(vals, h->d_hvals, sizeof(double) * mo.H.nnzh, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return CTD_OK;
}

// ---- one solver iteration in one call --------------------------------------------------------------------------------
// obj(x), grad(x), cons(x) + jac_coord(x) and hess_coord(x, y; obj_weight) only depend on (x, y).  Registry problems: the
// horizontally fused kernel of ctd_iter_kernels.hpp (+ its finish kernel): two launches per iteration.  Run-time OCPs (their
// kernels are compiled by hiprtc per callback): the same callbacks one after the other on the handle's stream.
int32_t ctd_eval_all_dev_async(ctd_handle* h, const double* x_dev, const double* y_dev, double obj_weight, double* f_dev, double* g_dev,
                               double* c_dev, double* vals_dev, double* hvals_dev) {
    if (!h) return CTD_EINVAL;
    if (h->device < 0) return fail(h, CTD_ENODEVICE, "compute call on a host-only handle (device = -1); there is no CPU fallback");
    if (!x_dev) return fail(h, CTD_EINVAL, "x is null");
    if (hvals_dev && !y_dev) return fail(h, CTD_EINVAL, "y is null");
    DeviceGuard dg_(h->device); HIP_TRY(h, dg_.err);
    int32_t st = CTD_OK;
    if (hvals_dev) { st = ensure_hess(h); if (st) return st; }      // first use uploads the tables (not capturable)
    if (h->rt || h->model.L.cs > 1 || env_int("CTD_ITER_SERIAL", 0)) {
        if (f_dev) st = enqueue_obj(h, x_dev, f_dev);
        if (!st && g_dev) st = enqueue_grad(h, x_dev, g_dev);
        if (!st && (c_dev || vals_dev)) st = enqueue_cons_jac(h, x_dev, c_dev, vals_dev);
        if (!st && hvals_dev) st = enqueue_hess(h, x_dev, y_dev, obj_weight, hvals_dev);
        return st;
    }
    IterParams ip;
    std::memset(&ip, 0, sizeof(ip));
    size_t lds = 4 * kMaxNV * sizeof(double) + 64;
    ip.kp = h->kp;       // (Layout is read from kp / hp / gp / op by the respective bodies)
    ip.hp = h->hp;
    ip.hp.halo = h->kp.halo;
    ip.hp.near = h->kp.near;
    if (ip.hp.halo) { ip.hp.own_lo = h->halo_host.vbegin[h->halo_host.self]; ip.hp.own_hi = h->halo_host.vbegin[h->halo_host.self + 1]; }
    // The fused grid always carries the TILE body of the Hessian, also on handles whose stand-alone hess_coord runs the
    // lane-per-step kernel (Gauss-Legendre 2 from 9 000 steps, 3 from 28 000: ensure_hess).  Measured (CTD_ITER_HESS_APART=1: fused
    // first-order grid, then the step kernel and its finish as two more launches): Goddard GL2 N = 10 000 22.9 us against 14-16,
    // GL3 N = 80 000 64.0 against 62.6 -- the two extra launches cost what the faster kernel gains
    const bool hess_apart = hvals_dev && h->hess_step && env_int("CTD_ITER_HESS_APART", 0);
    if (hvals_dev && !hess_apart) {
        ip.hp.obj_weight = obj_weight;
        ip.hp.vals = hvals_dev;
        ip.nb_h = ip.hp.ntiles + ip.hp.n_edge_blocks;
        lds = std::max(lds, h->hess_lds_bytes);
    }
    if (c_dev || vals_dev) {
        ip.kp.c = c_dev;
        ip.kp.vals = vals_dev;
        ip.nb_cj = h->grid;
        lds = std::max(lds, h->lds_bytes);
    }
    if (g_dev) {
        st = enqueue_grad(h, x_dev, g_dev, &ip.gp);          // fills the parameters only (and sizes the partials buffer)
        if (st) return st;
        ip.zero_g = h->model.info.lagrange ? 0 : 1;
        const Layout& L = h->model.L;
        ip.nb_g = ip.zero_g ? (int)std::min<int64_t>(64, (L.nvar + 4095) / 4096) : ip.gp.nblocks;
        if (ip.nb_g < 1) ip.nb_g = 1;
    }
    if (f_dev) ip.nb_o = fill_obj_params(h, f_dev, ip.op);
    hipError_t e = hipErrorInvalidValue;
    for_problem(h->model.problem, [&](auto tag) {
        using P = typename decltype(tag)::type;
        e = launch_iter<P>(ip, x_dev, y_dev, lds, h->stream);
    });
    if (e != hipSuccess) return fail(h, CTD_EHIP, std::string("kernel launch: ") + hipGetErrorString(e));
    if (hess_apart) return enqueue_hess(h, x_dev, y_dev, obj_weight, hvals_dev);
    return CTD_OK;
}

// out[0..9]: grid (workgroups), block, LDS bytes, steps per tile, CSC period of the lower triangle, edge entries, eval lanes
// per stage point / path point / boundary point, terms of the periodic segment
int32_t ctd_hess_launch_info(ctd_handle* h, int64_t* o) {
    if (!h || !o) return CTD_EINVAL;
    if (h->device < 0) return fail(h, CTD_ENODEVICE, "host-only handle");
    int32_t st = ensure_hess(h);
    if (st) return st;
    o[0] = h->hp.ntiles + h->hp.n_edge_blocks; o[1] = kHessBlock; o[2] = (int64_t)h->hess_lds_bytes; o[3] = h->hess_tile; o[4] = h->hp.Lseg; o[5] = h->hp.n_edge;
    o[6] = h->hp.ntask; o[7] = h->hp.nptask; o[8] = h->hp.nbtask; o[9] = h->hp.nterms;
    return CTD_OK;
}

int32_t ctd_hess_kernel_info(ctd_handle* h, int64_t* o) {
    if (!h || !o) return CTD_EINVAL;
    if (h->device < 0) return fail(h, CTD_ENODEVICE, "host-only handle");
    int32_t st = ensure_hess(h);
    if (st) return st;
    o[0] = h->hess_step ? 1 : 0;
    o[1] = h->hess_step ? h->hp_step.n_edge_blocks + h->hp_step.ntiles : h->hp.ntiles + h->hp.n_edge_blocks;
    return CTD_OK;
}

int32_t ctd_hess_debug_stamps(ctd_handle* h, const double* x_dev, const double* y_dev, double obj_weight, double* vals_dev,
                              uint64_t* out, int64_t cap) {
    if (!h || !out) return CTD_EINVAL;
    int32_t st = enqueue_hess(h, x_dev, y_dev, obj_weight, vals_dev);     // warm, no stamps
    if (st) return st;
    const int64_t words = (int64_t)(h->hp.ntiles + h->hp.n_edge_blocks) * 10;
    if (cap < words) return fail(h, CTD_EINVAL, "stamp buffer too small");
    unsigned long long* d_st = nullptr;
    HIP_TRY(h, hipMalloc((void**)&d_st, sizeof(unsigned long long) * words));
    HIP_TRY(h, hipMemsetAsync(d_st, 0, sizeof(unsigned long long) * words, h->stream));
    h->hp.stamps = d_st;
    st = enqueue_hess(h, x_dev, y_dev, obj_weight, vals_dev);
    h->hp.stamps = nullptr;
    if (st == CTD_OK) {
        hipError_t e = hipMemcpyAsync(out, d_st, sizeof(unsigned long long) * words, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) st = fail(h, CTD_EHIP, hipGetErrorString(e));
    }
    (void)hipFree(d_st);
    return st;
}

// out[0..3 + nvv): vals_main_begin, vals_main_end (contiguous CSC range of the shard's step columns), nvv, then the nvv
// positions of the V x V entries -- after ctd_hess_coord_dev on a shard they hold the shard's PARTIAL sums and must be
// added over the shards (one all-reduce of nvv doubles)
int32_t ctd_hess_shard_info(const ctd_handle* h, int64_t* o) {
    if (!h || !o) return CTD_EINVAL;
    const Model& mo = h->model;
    const HessModel& H = mo.H;
    const int64_t a = h->step_begin > H.reg_first ? h->step_begin : H.reg_first;
    const int64_t b = h->step_end < H.reg_last ? h->step_end : H.reg_last;
    o[0] = H.seg_base + (a - H.reg_first) * (int64_t)H.Lseg;
    o[1] = b > a ? H.seg_base + (b - H.reg_first) * (int64_t)H.Lseg : o[0];
    int nvv = 0;                                     // (optimized pattern: entries no evaluation point feeds are not in the pattern)
    for (int e = 0; e < H.nvv; ++e)
        if (H.vv_idx[e] >= 0) o[3 + nvv++] = H.vv_idx[e];
    o[2] = nvv;
    return CTD_OK;
}

int32_t ctd_time_hess_dev(ctd_handle* h, const double* x_dev, const double* y_dev, double obj_weight, double* vals_dev, int32_t iters,
                          double* mean_ms) {
    if (!h || !mean_ms || iters < 1) return CTD_EINVAL;
    return time_dispatches(h, iters, mean_ms,
                           [&](hipEvent_t a, hipEvent_t b) { return enqueue_hess(h, x_dev, y_dev, obj_weight, vals_dev, a, b); });
}

// ---- one transcription on several GPUs of one process --------------------------------------------------------------------
}  // extern "C"

struct ctd_sharded {
    std::vector<ctd_handle*> h;
    std::vector<int> dev;
    std::vector<hipEvent_t> x_ready;     // recorded on shard k's stream when this call's reads of x_dev[k] by OTHER shards may start
    std::vector<hipEvent_t> done;        // recorded on shard k's stream after its kernel (its rows of c are final)
    std::vector<hipEvent_t> pulled;      // recorded on shard k's stream after it has pulled the others' row blocks (stitching)
    bool pulls_pending = false;          // the last call stitched: the next kernels must not overwrite rows still being pulled
    std::vector<char> peer_ok;           // [a * G + b]: device of shard a can load from device of shard b (same device, or peer access enabled)
    int64_t N = 0;
    std::string err;
    ~ctd_sharded() {
        for (size_t k = 0; k < h.size(); ++k) {
            if (h[k]) { DeviceGuard dg(dev[k]); if (x_ready[k]) (void)hipEventDestroy(x_ready[k]); if (done[k]) (void)hipEventDestroy(done[k]); if (pulled[k]) (void)hipEventDestroy(pulled[k]); }
            if (h[k]) (void)ctd_destroy(h[k]);
        }
    }
};

static int32_t sfail(ctd_sharded* s, int32_t code, const std::string& msg) {
    if (s) s->err = msg; else g_create_err = msg;
    return code;
}
// a copy between two shards' buffers on shard k's stream: peer-to-peer over xGMI (or a plain device copy when both shards
// sit on one device)
static hipError_t shard_copy(ctd_sharded* s, int k, double* dst, int from, const double* src, int64_t count) {
    if (count <= 0) return hipSuccess;
    if (s->dev[k] == s->dev[from]) return hipMemcpyAsync(dst, src, sizeof(double) * count, hipMemcpyDeviceToDevice, s->h[k]->stream);
    return hipMemcpyPeerAsync(dst, s->dev[k], src, s->dev[from], sizeof(double) * count, s->h[k]->stream);
}
#define SH_TRY(s, expr)                                                                                  \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return sfail(s, CTD_ERCCL, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

extern "C" {

int32_t ctd_create_sharded(const ctd_desc* desc, const int32_t* devices, int32_t n_devices, ctd_sharded** out) {
    if (!desc || !devices || !out || n_devices < 1) return sfail(nullptr, CTD_EINVAL, "ctd_create_sharded: bad argument");
    *out = nullptr;
    if (desc->step_begin != 0 || desc->step_end != 0) return sfail(nullptr, CTD_EINVAL, "ctd_create_sharded: the descriptor must name the whole grid");
    const int64_t N = desc->time_grid ? desc->time_grid_len - 1 : desc->grid_size;
    if (N < n_devices) return sfail(nullptr, CTD_EINVAL, "ctd_create_sharded: fewer time steps than devices");
    std::unique_ptr<ctd_sharded> s(new (std::nothrow) ctd_sharded());
    if (!s) return sfail(nullptr, CTD_ENOMEM, "ctd_create_sharded: out of memory");
    s->N = N;
    s->h.assign(n_devices, nullptr); s->dev.assign(devices, devices + n_devices);
    s->x_ready.assign(n_devices, nullptr); s->done.assign(n_devices, nullptr); s->pulled.assign(n_devices, nullptr);
    const int64_t base = N / n_devices, rem = N % n_devices;
    for (int k = 0; k < n_devices; ++k) {
        ctd_desc dk = *desc;
        dk.device = devices[k];
        dk.step_begin = k * base + (k < rem ? k : rem);
        dk.step_end = dk.step_begin + base + (k < rem ? 1 : 0);
        dk.stream = nullptr;
        dk.stream_mode = CTD_STREAM_OWN;
        const int32_t st = ctd_create(&dk, &s->h[k]);
        if (st) return st;                                   // message in ctd_last_error(NULL)
        DeviceGuard dg(devices[k]);
        if (dg.err != hipSuccess || hipEventCreateWithFlags(&s->x_ready[k], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s->done[k], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s->pulled[k], hipEventDisableTiming) != hipSuccess)
            return sfail(nullptr, CTD_EHIP, "ctd_create_sharded: event creation failed");
    }
    // peer access between every pair of distinct devices (xGMI); hipMemcpyPeerAsync works without it, through the host.  Which
    // pairs can load from each other is RECORDED: CTD_X_SHARDED_IN_PLACE dereferences the other shards' buffers inside the kernels
    // and falls back to the copying protocol when a pair it needs is not peer-capable (a fault inside a kernel otherwise)
    s->peer_ok.assign((size_t)n_devices * n_devices, 0);
    for (int a = 0; a < n_devices; ++a)
        for (int b = 0; b < n_devices; ++b) {
            if (devices[a] == devices[b]) { s->peer_ok[(size_t)a * n_devices + b] = (a == b || !env_int("CTD_TEST_NO_PEER", 0)) ? 1 : 0; continue; }      // (test knob: pretend distinct shards cannot reach each other)
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[a], devices[b]) == hipSuccess && can) {
                DeviceGuard dg(devices[a]);
                const hipError_t e = hipDeviceEnablePeerAccess(devices[b], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                    return sfail(nullptr, CTD_ERCCL, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
                (void)hipGetLastError();
                s->peer_ok[(size_t)a * n_devices + b] = 1;
            }
        }
    *out = s.release();
    return CTD_OK;
}

int32_t ctd_sharded_destroy(ctd_sharded* s) {
    if (!s) return CTD_EINVAL;
    delete s;
    return CTD_OK;
}

const char* ctd_sharded_last_error(const ctd_sharded* s) { return s ? s->err.c_str() : g_create_err.c_str(); }

int32_t ctd_sharded_handle(ctd_sharded* s, int32_t k, ctd_handle** h) {
    if (!s || !h || k < 0 || k >= (int32_t)s->h.size()) return CTD_EINVAL;
    *h = s->h[k];
    return CTD_OK;
}

int32_t ctd_sharded_shard_info(const ctd_sharded* s, int32_t k, int64_t* o) {
    if (!s || !o || k < 0 || k >= (int32_t)s->h.size()) return CTD_EINVAL;
    o[0] = (int64_t)s->h.size();
    o[1] = s->dev[k];
    return ctd_shard_info(s->h[k], o + 2);
}

int32_t ctd_sharded_sync(ctd_sharded* s) {
    if (!s) return CTD_EINVAL;
    for (size_t k = 0; k < s->h.size(); ++k) {
        const int32_t st = ctd_sync(s->h[k]);
        if (st) return sfail(s, st, ctd_last_error(s->h[k]));
    }
    return CTD_OK;
}

int32_t ctd_dev_alloc(int32_t device, size_t bytes, void** ptr) {
    if (!ptr) return CTD_EINVAL;
    *ptr = nullptr;
    DeviceGuard dg(device);
    hipError_t e = dg.err;
    if (e == hipSuccess) e = hipMalloc(ptr, bytes ? bytes : 8);
    if (e != hipSuccess) { g_create_err = std::string("ctd_dev_alloc: ") + hipGetErrorString(e); return CTD_EHIP; }
    return CTD_OK;
}
int32_t ctd_dev_free(int32_t device, void* ptr) {
    if (!ptr) return CTD_OK;
    DeviceGuard dg(device);
    return (dg.err == hipSuccess && hipFree(ptr) == hipSuccess) ? CTD_OK : CTD_EHIP;
}
int32_t ctd_dev_copy(int32_t device, void* dst, const void* src, size_t bytes, int32_t kind) {
    if (!dst || !src || (kind != 0 && kind != 1)) return CTD_EINVAL;
    DeviceGuard dg(device);
    hipError_t e = dg.err;
    if (e == hipSuccess) e = hipMemcpy(dst, src, bytes, kind == 0 ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost);
    if (e != hipSuccess) { g_create_err = std::string("ctd_dev_copy: ") + hipGetErrorString(e); return CTD_EHIP; }
    return CTD_OK;
}

int32_t ctd_cons_jac_sharded_dev_async(ctd_sharded* s, double* const* x_dev, double* const* c_dev, double* const* vals_dev,
                                       int32_t x_mode, int32_t stitch) {
    if (!s || !x_dev) return CTD_EINVAL;
    const int G = (int)s->h.size();
    for (int k = 0; k < G; ++k)
        if (!x_dev[k] || (stitch && (!c_dev || !c_dev[k]))) return sfail(s, CTD_EINVAL, "ctd_cons_jac_sharded_dev_async: null buffer");
    if (x_mode != CTD_X_IN_PLACE && x_mode != CTD_X_SHARDED && x_mode != CTD_X_FROM_DEVICE0 && x_mode != CTD_X_SHARDED_COPY &&
        x_mode != CTD_X_SHARDED_IN_PLACE)
        return sfail(s, CTD_EINVAL, "ctd_cons_jac_sharded_dev_async: unknown x_mode");
    const Layout& L = s->h[0]->model.L;
    const int64_t N = L.N;
    // shards whose buffers shard k reads besides its own: its neighbours, the first (X_1) and the last (X_{N+1})
    auto reads = [&](int k, int j) { return j != k && (j == k - 1 || j == k + 1 || j == 0 || j == G - 1); };
    // (0) CTD_X_SHARDED_IN_PLACE: the kernels read the neighbours' entries in place through the peer mappings -- nothing is copied;
    // the table of buffers is rewritten only when the caller passes other pointers than last time.  A pair of devices that cannot
    // load from each other (no peer access: hipDeviceCanAccessPeer said no at creation) would fault inside the kernel: the call then
    // takes the copying protocol of CTD_X_SHARDED instead (same results; ctd_sharded_last_error says so)
    if (x_mode == CTD_X_SHARDED_IN_PLACE && G > 1) {
        if (G > kMaxShards) return sfail(s, CTD_EINVAL, "ctd_cons_jac_sharded_dev_async: CTD_X_SHARDED_IN_PLACE supports at most 16 shards");
        for (int k = 0; k < G && x_mode == CTD_X_SHARDED_IN_PLACE; ++k)
            for (int j = 0; j < G; ++j)
                if (reads(k, j) && !s->peer_ok[(size_t)k * G + j]) {
                    s->err = "CTD_X_SHARDED_IN_PLACE: device " + std::to_string(s->dev[k]) + " has no peer access to device " + std::to_string(s->dev[j]) +
                             "; the halo entries are copied instead (CTD_X_SHARDED)";
                    x_mode = CTD_X_SHARDED;
                    break;
                }
    }
    {
        int64_t sb[kMaxShards + 1];
        const bool peer = x_mode == CTD_X_SHARDED_IN_PLACE && G > 1;
        if (peer) {
            for (int k = 0; k < G; ++k) sb[k] = s->h[k]->step_begin;
            sb[G] = N;
        }
        for (int k = 0; k < G; ++k) {
            const int32_t st = peer ? ctd_set_x_shards(s->h[k], G, sb, x_dev, k) : (s->h[k]->kp.halo ? ctd_set_x_shards(s->h[k], 0, nullptr, nullptr, 0) : CTD_OK);
            if (st) return sfail(s, st, ctd_last_error(s->h[k]));
        }
        // ... and the cheap half of the copying protocol stays: every shard marks the point of ITS stream behind which its buffer holds
        // this iterate (work the caller queued there -- e.g. its update of x on that device's stream -- included), and a shard's kernel
        // waits for the marks of the shards it reads.  No copy; two event operations per neighbour
        if (peer) {
            for (int k = 0; k < G; ++k) {
                DeviceGuard dg(s->dev[k]);
                SH_TRY(s, hipEventRecord(s->x_ready[k], s->h[k]->stream));
            }
            for (int k = 0; k < G; ++k) {
                DeviceGuard dg(s->dev[k]);
                for (int j = 0; j < G; ++j)
                    if (reads(k, j)) SH_TRY(s, hipStreamWaitEvent(s->h[k]->stream, s->x_ready[j], 0));
            }
        }
    }
    // (1) iterate distribution.  Every shard first marks the point of its stream behind which its x buffer may be read by
    // the others (its own previous kernel has finished with it; the caller's writes are complete by contract).
    if ((x_mode == CTD_X_SHARDED || x_mode == CTD_X_SHARDED_COPY || x_mode == CTD_X_FROM_DEVICE0) && G > 1) {
        for (int k = 0; k < G; ++k) {
            DeviceGuard dg(s->dev[k]);
            SH_TRY(s, hipEventRecord(s->x_ready[k], s->h[k]->stream));
        }
        const int64_t w = L.n + (L.final_control ? L.m : 0);
        // (whatever the Jacobian's value order needs: the gradient and the Hessian of a one-point scheme read the previous block too)
        const int64_t lo = (L.sc == SC_IRK) ? 0 : L.blk;
        for (int k = 0; k < G; ++k) {
            DeviceGuard dg(s->dev[k]);
            ctd_handle* hk = s->h[k];
            if (x_mode == CTD_X_FROM_DEVICE0) {
                if (k == 0) continue;
                SH_TRY(s, hipStreamWaitEvent(hk->stream, s->x_ready[0], 0));
                SH_TRY(s, shard_copy(s, k, x_dev[k], 0, x_dev[0], L.nvar));
                continue;
            }
            const int64_t b = hk->step_begin, e = hk->step_end;
            if (k + 1 < G) {      // next shard's first node; the final state (boundary rows, Mayer cost)
                SH_TRY(s, hipStreamWaitEvent(hk->stream, s->x_ready[k + 1], 0));
                SH_TRY(s, shard_copy(s, k, x_dev[k] + e * L.blk, k + 1, x_dev[k + 1] + e * L.blk, w));
                if (k + 1 != G - 1) SH_TRY(s, hipStreamWaitEvent(hk->stream, s->x_ready[G - 1], 0));
                SH_TRY(s, shard_copy(s, k, x_dev[k] + N * L.blk, G - 1, x_dev[G - 1] + N * L.blk, L.n));
            }
            if (k > 0) {          // previous shard's last block (one-point schemes); the first state (boundary rows)
                SH_TRY(s, hipStreamWaitEvent(hk->stream, s->x_ready[k - 1], 0));
                if (lo) SH_TRY(s, shard_copy(s, k, x_dev[k] + (b - 1) * L.blk, k - 1, x_dev[k - 1] + (b - 1) * L.blk, lo));
                if (k - 1 != 0) SH_TRY(s, hipStreamWaitEvent(hk->stream, s->x_ready[0], 0));
                SH_TRY(s, shard_copy(s, k, x_dev[k], 0, x_dev[0], L.n));
            }
        }
    }
    // (2) the shards' kernels (after every other device has finished pulling this shard's rows of the previous stitched call)
    if (s->pulls_pending) {
        for (int k = 0; k < G; ++k) {
            DeviceGuard dg(s->dev[k]);
            for (int j = 0; j < G; ++j)
                if (j != k) SH_TRY(s, hipStreamWaitEvent(s->h[k]->stream, s->pulled[j], 0));
        }
        s->pulls_pending = false;
    }
    for (int k = 0; k < G; ++k) {
        const int32_t st = enqueue_cons_jac(s->h[k], x_dev[k], c_dev ? c_dev[k] : nullptr, vals_dev ? vals_dev[k] : nullptr);
        if (st) return sfail(s, st, ctd_last_error(s->h[k]));
    }
    // (3) stitching: every device pulls the other shards' row blocks of c once those shards' kernels are done
    if (stitch && G > 1) {
        for (int k = 0; k < G; ++k) {
            DeviceGuard dg(s->dev[k]);
            SH_TRY(s, hipEventRecord(s->done[k], s->h[k]->stream));
        }
        for (int k = 0; k < G; ++k) {
            DeviceGuard dg(s->dev[k]);
            for (int j = 0; j < G; ++j) {
                if (j == k) continue;
                SH_TRY(s, hipStreamWaitEvent(s->h[k]->stream, s->done[j], 0));
                // (the p + bc tail rows come from the last shard: the only one that holds everything they read when x is sharded)
                const int64_t r0 = s->h[j]->step_begin * L.cb, r1 = (j == G - 1) ? L.ncon : s->h[j]->step_end * L.cb;
                SH_TRY(s, shard_copy(s, k, c_dev[k] + r0, j, c_dev[j] + r0, r1 - r0));
            }
            SH_TRY(s, hipEventRecord(s->pulled[k], s->h[k]->stream));
        }
        s->pulls_pending = true;
    }
    return CTD_OK;
}

}  // extern "C"
