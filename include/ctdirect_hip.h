/*
 * ctdirect_hip.h -- C ABI of the MI355X-native collocation engine (libctdirect_hip.so).
 *
 * This is the drop-in boundary for CTDirect.jl's NLP-callback hot path.  In the reference the boundary is the call
 *     ADNLPModels.ADNLPModel!(f, x0, lvar, uvar, c!, lcon, ucon; minimize, backends...)   src/collocation.jl:137-149
 * inside build_adnlp_model (src/collocation.jl:90-153), after which ADNLPModels serves the NLPModels API
 * (obj, cons!, jac_structure!, jac_coord!, ...) to Ipopt/MadNLP by calling the two Julia closures
 *     f  = x -> CTDirect.__objective(x, docp)          src/collocation.jl:97,  src/DOCP_functions.jl:23-54
 *     c! = (c, x) -> CTDirect.__constraints!(c, x, docp)  src/collocation.jl:98,  src/DOCP_functions.jl:80-115
 * and differentiating c! with coloured ForwardDiff passes over DOCP_Jacobian_pattern(docp) (src/collocation.jl:116-120).
 * A Julia shim subtypes NLPModels.AbstractNLPModel and forwards each NLPModels method to ONE entry
 * point below through `ccall`; every entry point states the reference interface it replaces.
 *
 * Conventions
 *   - all functions return an int32 status (CTD_OK == 0); no C++ exception crosses this boundary;
 *   - the caller owns every buffer passed in; the engine owns its device buffers and staging inside the handle;
 *   - host-pointer calls copy H2D / D2H and return when the result is in the caller's buffer;
 *     `_dev` calls take device pointers (hipMalloc'ed, on the handle's device) and are synchronous;
 *     `_dev_async` calls only enqueue on the handle's stream (ctd_sync waits);
 *   - external layouts are the reference's: variables step-major [X_i, U_i.., K_i..]..., X_{N+1}, [U_{N+1}], V
 *     (src/ode/trapeze.jl:1-4, midpoint.jl:1-7, irk.jl:1-9, irk_stagewise.jl:6-11); constraints
 *     [C_i^x, C_i^{k,1..s}, G_i]..., G_{N+1}, B (src/ode/irk_stagewise.jl:13-30, src/DOCP_functions.jl:92-111);
 *     Jacobian values in the CSC order of SparseArrays.sparse(Is, Js, ...) as DOCP_Jacobian_pattern returns it (default), or by rows
 *     (ctd_desc.value_order = CTD_ORDER_CSR);
 *   - all arithmetic is FP64; there is NO CPU fallback: compute entry points on a handle without a device fail
 *     with CTD_ENODEVICE.
 *   - a handle is not thread-safe; distinct handles are independent (one HIP stream each).
 */
#ifndef CTDIRECT_HIP_H
#define CTDIRECT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ctd_handle ctd_handle;

/* status codes; the Julia shim rethrows CTD_EGRID as ArgumentError (src/DOCP_data.jl:186-189), CTD_ESCHEME as
 * error("Unknown discretization method") (src/DOCP_data.jl:342-349), CTD_EPATTERN as the error of the
 * DOCP_Jacobian_pattern stubs (src/ode/common.jl:252-272) */
enum {
    CTD_OK = 0,
    CTD_EINVAL = 1,     /* null pointer / bad argument */
    CTD_EGRID = 2,      /* time grid not strictly increasing */
    CTD_ESCHEME = 3,    /* unknown discretization scheme */
    CTD_EPATTERN = 4,   /* sparsity pattern not available / not step-periodic */
    CTD_EPROBLEM = 5,   /* problem id not in the compiled registry */
    CTD_ENODEVICE = 6,  /* compute call on a host-only handle, or no usable HIP device */
    CTD_EHIP = 7,       /* a HIP runtime call failed (message in ctd_last_error) */
    CTD_ENOMEM = 8,
    CTD_ERCCL = 9       /* an exchange between the devices of a multi-device handle failed (peer access / peer copy) */
};

/* scheme symbols of the reference, src/DOCP_data.jl:307-349 */
enum {
    CTD_SCHEME_TRAPEZE = 0,                          /* :trapeze                              */
    CTD_SCHEME_MIDPOINT = 1,                         /* :midpoint                             */
    CTD_SCHEME_GAUSS_LEGENDRE_1 = 2,                 /* Gauss_Legendre_1 (test only, irk.jl:20) */
    CTD_SCHEME_GAUSS_LEGENDRE_2_CONSTANT_CONTROL = 3,/* :gauss_legendre_2_constant_control    */
    CTD_SCHEME_GAUSS_LEGENDRE_3_CONSTANT_CONTROL = 4,/* :gauss_legendre_3_constant_control    */
    CTD_SCHEME_GAUSS_LEGENDRE_2 = 5,                 /* :gauss_legendre_2 (stagewise controls) */
    CTD_SCHEME_GAUSS_LEGENDRE_3 = 6,                 /* :gauss_legendre_3 (stagewise controls) */
    CTD_SCHEME_EULER = 7,                            /* :euler | :euler_explicit | :euler_forward (src/DOCP_data.jl:315-317, src/ode/euler.jl) */
    CTD_SCHEME_EULER_IMPLICIT = 8                    /* :euler_implicit | :euler_backward (:318-320)                                           */
};

/* compiled OCP registry (the reference takes arbitrary Julia closures from CTModels; a GPU engine behind a C ABI
 * cannot -- see DESIGN.md "the unavoidable gap") */
enum {
    CTD_PROBLEM_GODDARD = 0,                    /* test/problems/goddard.jl:18-49                  */
    CTD_PROBLEM_GODDARD_ALL = 1,                /* test/problems/goddard.jl:87-158                 */
    CTD_PROBLEM_DOUBLE_INTEGRATOR_PATH = 2,     /* double_integrator.jl:42-58 + build-defined path */
    CTD_PROBLEM_QUADROTOR = 3,                  /* test/problems/quadrotor.jl:7-105 (8 states)     */
    CTD_PROBLEM_QUADROTOR12 = 4,                /* build-defined 12-state rigid body               */
    CTD_PROBLEM_STAGEWISE_SCALAR = 5,           /* test/ci/test_discretization_stagewise.jl:1-14   */
    CTD_PROBLEM_ESTIMATE_INITIAL_CONDITION = 6, /* test/problems/autonomous_system.jl:6-43         */
    CTD_PROBLEM_ESTIMATE_ROTATION_RATE = 7,     /* test/problems/autonomous_system.jl:46-87        */
    CTD_PROBLEM_LEAST_SQUARES_CONSTRAINT = 8,   /* test/problems/autonomous_system.jl:90-138       */
    CTD_PROBLEM_DOUBLE_INTEGRATOR_FREET0TF = 9  /* test/problems/double_integrator.jl:79-99        */
};

enum {
    CTD_PATTERN_REFERENCE_MANUAL = 0, /* DOCP_Jacobian_pattern exactly as written (bug-compatible, SURVEY hazards H1/H2) */
    CTD_PATTERN_STRUCTURAL = 1,       /* = manual + the dynamics-row x V block trapeze.jl:203 omits */
    CTD_PATTERN_OPTIMIZED = 2         /* the sparsity ADNLPModels' default backend detects itself (src/collocation.jl:131-134:
                                         SparseConnectivityTracer over c! and the Lagrangian): operator-level dependence of every
                                         row, e.g. nnzj 4504 / nnzh 5259 instead of 6028 / 6519 for Goddard, midpoint, N = 250
                                         (test/ci/test_modeler_solver.jl:32,37).  Fewer entries = fewer bytes per evaluation */
};

/* replaces the arguments of CTDirect.DOCP(ocp, grid_size, control_steps, scheme, time_grid), src/DOCP_data.jl:293-365,
 * as selected by Collocation(; grid_size, scheme, time_grid), src/collocation.jl:16-48,57-73 */
typedef struct ctd_desc {
    int32_t problem;          /* CTD_PROBLEM_*                                                        */
    int32_t scheme;           /* CTD_SCHEME_*                                                         */
    int32_t pattern_mode;     /* CTD_PATTERN_*                                                        */
    int32_t device;           /* HIP device ordinal; -1 = host-only handle (sizes, bounds, x0, patterns) */
    int64_t grid_size;        /* N; used when time_grid == NULL (uniform grid, src/DOCP_data.jl:179-183) */
    const double* time_grid;  /* optional strictly increasing grid of time_grid_len = N+1 points (:184-199) */
    int64_t time_grid_len;
    int64_t step_begin;       /* shard of the time grid this handle evaluates: steps [step_begin, step_end),  */
    int64_t step_end;         /*   0-based; 0,0 = all N steps (single GPU)                                    */
    void* stream;             /* hipStream_t to launch on when stream_mode == CTD_STREAM_GIVEN                */
    int32_t stream_mode;      /* CTD_STREAM_OWN: the handle creates a private non-blocking stream (default);  */
                              /* CTD_STREAM_GIVEN: launch on `stream` (NULL = the device's default stream),   */
                              /* so launches are ordered with the caller's other work on that stream          */
    int32_t control_steps;    /* controls per time step: DOCP(ocp, grid_size, control_steps, scheme, time_grid), src/DOCP_data.jl:293.
                               * 0 or 1: collocation (src/collocation.jl:65).  > 1: the direct-shooting layout of
                               * src/direct_shooting.jl:55-71 -- step block [X_i, U_i^1 .. U_i^cs] (midpoint.jl:20), dynamics summed over
                               * the control sub-steps (midpoint.jl:47-72,137-155), cost midpoint.jl:99-116, bounds / initial guess for
                               * every control (DOCP_variables.jl:44,138); path constraints see U_i^1 (common.jl:140-155).  CTD_SCHEME_MIDPOINT
                               * only.  KNOWN DEVIATION (decided, not an omission): the reference accepts control_steps > 1 with every scheme --
                               * it sizes the step block with it (trapeze.jl:20, euler.jl:22, irk.jl:141) and bounds / initialises the extra
                               * controls (DOCP_variables.jl:44-47,138-140) -- but only midpoint.jl integrates over them, and elsewhere its own
                               * layout is incoherent: the stage-variable getter of the IRK schemes keeps the offset n + m (common.jl:166-170),
                               * so K_i^j ALIASES the controls U_i^2.. it has just bounded; the trapeze `:manual` pattern addresses X_{i+1} at
                               * n + m behind X_i (trapeze.jl:180-186), i.e. inside the extra controls, and misses the real X_{i+1} columns;
                               * the stagewise schemes ignore control_steps altogether (irk_stagewise.jl:138-146); Euler is coherent (extra
                               * controls that nothing reads).  The engine returns CTD_ESCHEME for all of them instead of reproducing aliased or
                               * dead variables (direct shooting, the only caller that sets
                               * control_steps, advertises :midpoint and :trapeze, src/direct_shooting.jl:33-37; its default is :midpoint).
                               * Compiled problems: control_steps <= 3; problems registered at run time: any.  Constraints, Jacobian (all three
                               * patterns), objective, gradient, hess_structure and hess_coord (one second-order evaluation point per control). */
    int32_t value_order;      /* CTD_ORDER_CSC (0): Jacobian values in the order of SparseArrays.sparse(Is, Js, ...) as DOCP_Jacobian_pattern
                               * returns it (src/ode/midpoint.jl:229-232, irk_stagewise.jl:555-558) -- what ADNLPModels' jac_coord! fills.
                               * CTD_ORDER_CSR (1): the same entries by ROWS -- what GPU KKT consumers (rocSPARSE / hipSOLVER) take, and what
                               * BASELINE north_star names ("assembled ... in CSR on device").  Same kernel, same bytes; the rows of a time step,
                               * V entries inline, are one contiguous run of the value array, so a shard of the grid owns ONE range
                               * (ctd_shard_info).  ctd_jac_structure / ctd_jac_coord / ctd_cons_jac* follow the order chosen here. */
    int32_t reserved0;        /* must be 0.  ZERO-INITIALISE the whole struct (memset / `ctd_desc d = {0}` / Ref{ctd_desc}(zeros)): fields that were
                               * reserved in an earlier version of this header get a meaning later (control_steps was one), and a nonzero
                               * reserved field is refused with CTD_EINVAL */
} ctd_desc;

enum { CTD_ORDER_CSC = 0, CTD_ORDER_CSR = 1 };

enum { CTD_STREAM_OWN = 0, CTD_STREAM_GIVEN = 1 };

/* replaces the `init` tuple handed to __initial_guess(docp, CTModels.build_initial_guess(ocp, init)),
 * src/collocation.jl:101-102, src/DOCP_variables.jl:122-145, src/ode/irk_stagewise.jl:302-335.
 * NULL pointers mean "not provided" (entries stay at the 0.1 default, src/DOCP_variables.jl:126). */
typedef struct ctd_init {
    int32_t use_problem_default; /* nonzero: use the init tuple of the problem file (e.g. goddard.jl:48)    */
    const double* state;         /* constant state guess  [n]  or NULL                                      */
    const double* control;       /* constant control guess [m] or NULL                                      */
    const double* variable;      /* variable guess [nv] or NULL                                             */
    /* time-dependent guesses -- the functional / interpolated / warm-start forms of CTModels' init.state(t), init.control(t)
     * (test/ci/test_initial_guess.jl): n_samples > 0 gives trajectories sampled at the increasing times t_samples[k];
     * state_samples is row-major [n_samples][n], control_samples [n_samples][m] (either may be NULL).  They are evaluated
     * where the reference evaluates the functions (node times t_i; stage times t_ij for stagewise controls) by linear
     * interpolation, end values held outside the sampled span, and take precedence over the constant entries above. */
    int64_t n_samples;
    const double* t_samples;
    const double* state_samples;
    const double* control_samples;
} ctd_init;

/* ---- OCPs defined at run time ----------------------------------------------------------------------------------
 * The reference obtains the OCP functions as Julia closures from CTModels (CTModels.dynamics(ocp)(dx, t, x, u, v) called
 * at src/ode/trapeze.jl:66, midpoint.jl:64, irk.jl:291, irk_stagewise.jl:441; lagrange / mayer src/DOCP_functions.jl:35-48;
 * path / boundary constraints :108-110,136-138).  A closure cannot cross a C ABI, so an OCP that is not in the compiled
 * registry is handed over as TEXT: one arithmetic expression per output.  Grammar: + - * / parentheses, numbers, ^ with a CONSTANT
 * exponent (small non-negative integers multiply out; any other real exponent is pow), exp log sin cos tan atan tanh sqrt abs asin acos
 * sinh cosh floor, max(a, b) min(a, b) (derivative of the selected operand: ForwardDiff's rule, at a tie max follows b and min a;
 * floor has derivative 0) -- everything the reference's problem folder (test/problems, every .jl file) uses --, the names t, x1..xn, u1..um,
 * v1..vnv (dynamics, lagrange, path) or x0_1.., xf_1.., v1.. (mayer, boundary), and what `constants` declares: "Cd=310; beta=500"
 * (numbers) and "aux = 543 + 186*cos(x4); g13 = -(105 + 2*cos(2*x4)) / (2*aux)" (ALIASES: named sub-expressions, the `aux = ...`
 * lines of a CTParser @def block such as test/problems/swimmer.jl:39-53; substituted where they are used, may use each other).
 * No C++ is accepted.  ctd_register_ocp parses
 * the expressions, generates a functor of the registry's shape and returns a problem id (>= 1000) for ctd_desc.problem;
 * ctd_create then compiles the SAME kernel templates for it with hiprtc (gfx950) -- same code path as a built-in problem.
 * dims, flags and bounds restate DOCPdims / DOCPFlags (src/DOCP_data.jl:24-30,88-94) and the boxes of CTModels
 * (src/DOCP_variables.jl:88-98).  Bound arrays may be NULL (boxes: free; path / boundary: equality with 0). */
typedef struct ctd_ocp_def {
    const char* name;
    int32_t n, m, nv, npath, nbc;        /* state, control, variable, path-constraint, boundary-constraint dimensions */
    int32_t it0, itf;                    /* 0-based index of t0 / tf inside v, -1 = fixed                             */
    double t0, tf;                       /* fixed values when the index is -1                                          */
    int32_t maximize;                    /* DOCPFlags.max                                                              */
    int32_t reserved;
    const char* const* dynamics;         /* n expressions                                                              */
    const char* lagrange;                /* running cost or NULL                                                       */
    const char* mayer;                   /* terminal cost or NULL                                                      */
    const char* const* path;             /* npath expressions                                                          */
    const char* const* boundary;         /* nbc expressions                                                            */
    const char* constants;               /* "name=value; ..." or NULL                                                  */
    const double *state_lb, *state_ub, *control_lb, *control_ub, *variable_lb, *variable_ub;
    const double *path_lb, *path_ub, *boundary_lb, *boundary_ub;
} ctd_ocp_def;
int32_t ctd_register_ocp(const ctd_ocp_def* def, int32_t* problem_id);
/* the functor text generated for a registered OCP (diagnostics / tests).  cap too small: CTD_EINVAL and ctd_last_error(NULL)
 * = "... needs <n> bytes" (nothing is written, never a truncated text) */
int32_t ctd_ocp_source(int32_t problem_id, char* buf, int64_t cap);
/* compile-only check, no device needed: the kernels of `scheme` build for gfx950 (ctd_last_error(NULL) holds the log) */
int32_t ctd_jit_check(int32_t problem_id, int32_t scheme);

/* ---- lifecycle ------------------------------------------------------------------------------------------ */
/* get_docp: DOCP(...) + __variables_bounds! + __constraints_bounds!   (src/collocation.jl:57-73) */
int32_t ctd_create(const ctd_desc* desc, ctd_handle** out);
int32_t ctd_destroy(ctd_handle* h);
const char* ctd_last_error(const ctd_handle* h);   /* h may be NULL: error of the last failed ctd_create */
/* Page-locked host memory for the vectors a caller hands to the host-pointer entry points (x, c, vals, g, y ...): the
 * copies then run as direct DMA at PCIe rate instead of through the runtime's bounce buffers.  Optional -- any host pointer
 * is accepted everywhere -- and independent of handles.  The reference's vectors are plain Julia arrays owned by the solver
 * (src/collocation.jl:137-149); a shim would wrap these with unsafe_wrap. */
int32_t ctd_host_alloc(void** ptr, size_t bytes);
int32_t ctd_host_free(void* ptr);
const char* ctd_strerror(int32_t status);

/* ---- sizes and static data (host) ------------------------------------------------------------------------ */
/* docp.dim_NLP_variables, docp.dim_NLP_constraints (src/DOCP_data.jl:285-286), nlp.meta.nnzj, nlp.meta.nnzh
 * (nnzh = entries of the lower triangle of DOCP_Hessian_pattern, e.g. 6519 for Goddard / midpoint / 250 steps,
 * test/archives/AD_backend.md:86) */
int32_t ctd_sizes(const ctd_handle* h, int64_t* nvar, int64_t* ncon, int64_t* nnzj, int64_t* nnzh);
/* out[0..15]: NLP_x, NLP_u, NLP_v, path_cons, boundary_cons (DOCPdims, src/DOCP_data.jl:88-94), steps,
 * _step_variables_block, _state_stage_eqs_block, _step_pathcons_block, stage, _final_control, freet0, freetf,
 * lagrange, mayer, max (DOCPFlags, :24-30) */
int32_t ctd_dims(const ctd_handle* h, int64_t* out16);
/* docp.time.normalized_grid / fixed_grid (src/DOCP_data.jl:147-152), each N+1 */
int32_t ctd_time_grid(const ctd_handle* h, double* normalized, double* fixed);
/* get_time_grid(xu, docp), src/DOCP_data.jl:437-458: grid[i] = t0 + tau_i (tf - t0) with t0 / tf fixed or read from the
 * tail of x (host pointers; post-processing helper for the solution rebuild, src/DOCP_data.jl:514-633) */
int32_t ctd_time_grid_at(const ctd_handle* h, const double* x, double* grid);
/* Butcher tables of the scheme struct (row-major a[stage*stage], b[stage], c[stage]), src/ode/irk_stagewise.jl:61-64,103-109 */
int32_t ctd_butcher(const ctd_handle* h, double* a, double* b, double* c);
/* docp.bounds.var_l/var_u/con_l/con_u : __variables_bounds! (src/DOCP_variables.jl:21-63, irk_stagewise.jl:250-300)
 * and __constraints_bounds! (src/DOCP_functions.jl:163-191) */
int32_t ctd_bounds(const ctd_handle* h, double* lvar, double* uvar, double* lcon, double* ucon);
/* __initial_guess (src/DOCP_variables.jl:122-145, irk_stagewise.jl:302-335) */
int32_t ctd_initial_guess(const ctd_handle* h, double* x0, const ctd_init* init);
/* jac_structure!(nlp, rows, cols): 1-based (row, col) of every entry of DOCP_Jacobian_pattern(docp) in CSC order
 * (src/ode/trapeze.jl:149-233, midpoint.jl:163-233, irk.jl:315-416, irk_stagewise.jl:468-558) */
int32_t ctd_jac_structure(const ctd_handle* h, int64_t* rows, int64_t* cols);
/* same pattern as 0-based CSC (colptr[nvar+1], rowval[nnzj]) -- the order of the values of a CTD_ORDER_CSC handle */
int32_t ctd_jac_csc(const ctd_handle* h, int64_t* colptr, int64_t* rowval);
/* same pattern as 0-based CSR (rowptr[ncon+1], colind[nnzj]) -- the order of the values of a CTD_ORDER_CSR handle: pass rowptr / colind and the
 * device array ctd_cons_jac_dev fills straight to rocsparse_create_csr_descr.  Available on every handle (the pattern is the same set). */
int32_t ctd_jac_csr(const ctd_handle* h, int64_t* rowptr, int64_t* colind);
/* ctd_desc.value_order of the handle */
int32_t ctd_value_order(const ctd_handle* h, int32_t* order);
/* number of structurally nonzero Jacobian entries that the selected pattern does not hold (0 except
 * REFERENCE_MANUAL + trapeze + a free time / v-dependent dynamics: hazard H1) */
int32_t ctd_dropped_nonzeros(const ctd_handle* h, int64_t* count);

/* ---- the hot path: host pointers -------------------------------------------------------------------------- */
/* obj(nlp, x)       = __objective(x, docp)                       src/DOCP_functions.jl:23-54 */
int32_t ctd_obj(ctd_handle* h, const double* x, double* f);
/* grad!(nlp, x, g): gradient of __objective with respect to the nvar NLP variables (the reference obtains it with
 * ReverseDiff over the objective closure: gradient_backend = ReverseDiffADGradient, src/collocation.jl:127).
 * Always the gradient of the whole objective, also on a sharded handle (it is O(nvar) work). */
int32_t ctd_grad(ctd_handle* h, const double* x, double* g);
/* cons!(nlp, x, c)  = __constraints!(c, x, docp)                 src/DOCP_functions.jl:80-115 */
int32_t ctd_cons(ctd_handle* h, const double* x, double* c);
/* jac_coord!(nlp, x, vals): values of dc/dx in the pattern's CSC order (ADNLPModels.SparseADJacobian,
 * call site src/collocation.jl:116-120) */
int32_t ctd_jac_coord(ctd_handle* h, const double* x, double* vals);
/* fused cons! + jac_coord! at one x: the benchmarked call (BASELINE.json metric) */
int32_t ctd_cons_jac(ctd_handle* h, const double* x, double* c, double* vals);

/* ---- the hot path: device pointers (inputs and outputs stay in HBM) --------------------------------------- */
/* c_dev has ncon entries and vals_dev nnzj entries (global indexing also for sharded handles: a shard writes only
 * its rows / its CSC ranges, see ctd_shard_info).  Either of c_dev / vals_dev may be NULL to skip that output. */
int32_t ctd_cons_jac_dev(ctd_handle* h, const double* x_dev, double* c_dev, double* vals_dev);
int32_t ctd_cons_jac_dev_async(ctd_handle* h, const double* x_dev, double* c_dev, double* vals_dev);
int32_t ctd_obj_dev(ctd_handle* h, const double* x_dev, double* f_host);
int32_t ctd_grad_dev(ctd_handle* h, const double* x_dev, double* g_dev);
/* enqueue-only variants for device-resident solver loops: the objective goes to f_dev[0] (device memory), nothing is
 * copied to the host and nothing waits; ctd_sync (or any later work on the handle's stream) orders the results */
/* Launch on `stream` (a hipStream_t, NULL = the device's default stream) from now on, e.g. the capturing stream while the
 * caller records a HIP graph of a whole solver iteration: every *_dev_async entry point only enqueues kernels (and one
 * memset) and is capturable.  Handles of run-time OCPs and the first Hessian call compile / upload on first use: call
 * each entry point once before capturing. */
int32_t ctd_set_stream(ctd_handle* h, void* stream);
int32_t ctd_obj_dev_async(ctd_handle* h, const double* x_dev, double* f_dev);
int32_t ctd_grad_dev_async(ctd_handle* h, const double* x_dev, double* g_dev);
/* The gradient of a SHARDED transcription without an all-gathered iterate (round 4): on a handle restricted to the steps
 * [step_begin, step_end) this writes ONLY the gradient entries of the shard's own variables -- its step blocks; the last shard also
 * the final state (and final control) -- into the full-length g_dev, and leaves the shard's PARTIAL sums of d/dv in the nv variable
 * entries (the caller adds them over the shards: one all-reduce of nv doubles, as for the V x V entries of the Hessian).  The Mayer
 * term's d/dx0 goes to the shard that owns X_1, its d/dxf and d/dv to the one that owns X_{N+1}.  Neighbours' entries (the next
 * shard's first node and the previous shard's last block for the midpoint / Euler quadrature, X_1 / X_{N+1} for the Mayer term) are read
 * through the table of ctd_set_x_shards like the other callbacks do, or from x_dev itself when no table is set (halos copied).  The
 * quadrature being differentiated: src/DOCP_functions.jl:23-54 with trapeze.jl:78-110, midpoint.jl:79-116, irk_stagewise.jl:344-384.
 * On an unsharded handle it equals ctd_grad_dev_async. */
int32_t ctd_grad_shard_dev_async(ctd_handle* h, const double* x_dev, double* g_dev);
int32_t ctd_sync(ctd_handle* h);

/* ---- multi-GPU shards (time-step partition) ------------------------------------------ */
/* out[0..7]: step_begin, step_end, c_row_begin, c_row_end (step rows this shard writes; in addition EVERY shard writes
 * the p + bc tail rows [N*cb, ncon) -- final-time path and boundary constraints, x being replicated -- so stitching c
 * is a single all-gather), vals_main_begin, vals_main_end, owns_first, owns_last.
 * CTD_ORDER_CSC: [vals_main_begin, vals_main_end) is the contiguous CSC range of the shard's step columns; the shard ALSO writes its
 * slice of every V column, the first shard the irregular first-step columns and the last shard the final-state columns (SURVEY 8e
 * "CSC caveat").  CTD_ORDER_CSR: [vals_main_begin, vals_main_end) is EVERYTHING the shard writes: its step rows, V entries inline
 * (the last shard's range runs to nnzj: the p + bc tail rows follow the step rows) -- one range per rank, the ranges of the ranks
 * partition the value array. */
int32_t ctd_shard_info(const ctd_handle* h, int64_t* out8);

/* ---- measurement ------------------------------------------------------------------------------------------ */
/* Launches the fused kernel `iters` times back-to-back on the handle's stream between two hipEvents (recorded on
 * that same stream) and returns the mean duration of one launch in milliseconds.  Used by bench.py for the
 * roofline figure; the outputs are the normal outputs of ctd_cons_jac_dev. */
int32_t ctd_time_cons_jac_dev(ctd_handle* h, const double* x_dev, double* c_dev, double* vals_dev,
                              int32_t iters, double* mean_ms);
/* Diagnostics: one launch of the fused kernel with in-kernel phase stamps.  out receives, per workgroup, 6 pairs
 * {100 MHz realtime counter, shader cycle counter} taken by lane 0 at: start, after load, after eval, after fin, after
 * emit issue, after the workgroup's stores drained.  cap = capacity of out in uint64 words (needs grid * 12). */
int32_t ctd_debug_stamps(ctd_handle* h, const double* x_dev, double* c_dev, double* vals_dev, uint64_t* out, int64_t cap);
/* kernel launch geometry: out[0..7] = grid blocks, block threads, dynamic LDS bytes, steps per tile, interior
 * CSC period L (entries per regular step), number of edge entries, 1 when the tiles run the direct driver (x read
 * straight from global memory by the evaluating lanes, one workgroup barrier) and 0 for the staged one, resident workgroups
 * per CU of that kernel (the runtime's occupancy query for its registers and LDS; 0 on a host-only handle) */
int32_t ctd_launch_info(const ctd_handle* h, int64_t* out8);


/* ---- Hessian of the Lagrangian --------------------------------------------------------------------------------
 * Replaces hess_structure!(nlp, rows, cols) / hess_coord!(nlp, x, y, vals; obj_weight) of the ADNLPModel built at
 * src/collocation.jl:137-149 (sparse Hessian backend selected at :121-125).  The sparsity pattern is the lower triangle
 * (row >= col, NLPModels convention) of DOCP_Hessian_pattern(docp) -- src/ode/trapeze.jl:240-303, midpoint.jl:240-300,
 * irk.jl:423-496, irk_stagewise.jl:565-638 -- in CSC order; CTD_PATTERN_STRUCTURAL adds the final-state x variable
 * block whose add_nonzero_block! call has an empty range in the reference (irk.jl:483 / irk_stagewise.jl:625).
 * Values:  vals[k] = obj_weight * d2 f/dx_i dx_j + sum_r y[r] * d2 c_r/dx_i dx_j  at (i, j) = (rows[k], cols[k]),
 * f = __objective (src/DOCP_functions.jl:23-54), c = __constraints! (:80-115); entries of the pattern that are
 * structurally zero receive 0.0.  A sharded handle (step_begin, step_end) writes the entries of its step columns (one
 * contiguous CSC range), the first / last shard also the irregular first-step / final-state entries, and leaves its PARTIAL
 * sums in the nv (nv+1)/2 variable x variable entries (they sum over every step): ctd_hess_shard_info names them for the
 * one all-reduce the caller has to do. */
/* One solver iteration in one call: obj(nlp, x) -> f_dev[0], grad!(nlp, x, g), cons!(nlp, x, c) + jac_coord!(nlp, x, vals) and
 * hess_coord!(nlp, x, y, hvals; obj_weight) -- the NLPModels calls an interior-point iteration makes on the ADNLPModel of
 * src/collocation.jl:137-149 -- in TWO launches for registry problems: one grid whose workgroups take the bodies of the four
 * evaluation kernels by index range (all resident together: the launch costs about what the longest of them, the Hessian,
 * costs instead of their sum), then the fixed-order cross-workgroup sums.  Any output may be NULL to skip that callback.
 * Enqueue-only, like the *_dev_async entry points. */
int32_t ctd_eval_all_dev_async(ctd_handle* h, const double* x_dev, const double* y_dev, double obj_weight, double* f_dev,
                               double* g_dev, double* c_dev, double* vals_dev, double* hvals_dev);

/* 1-based (rows[k], cols[k]), k < nnzh, CSC order */
int32_t ctd_hess_structure(const ctd_handle* h, int64_t* rows, int64_t* cols);
/* same pattern as 0-based CSC (colptr[nvar + 1], rowval[nnzh]) */
int32_t ctd_hess_csc(const ctd_handle* h, int64_t* colptr, int64_t* rowval);
/* The Hessian values in CSR: the Hessian is symmetric, so the value array of ctd_hess_coord* (lower triangle by columns) IS the CSR value
 * array of the UPPER triangle (row j of the upper triangle = column j of the lower one).  rowptr[nvar + 1], colind[nnzh], 0-based,
 * colind >= row: hand them with the values to a CSR consumer as a symmetric matrix stored by its upper triangle
 * (rocsparse_fill_mode_upper).  No second value order is needed for the Hessian -- nothing is permuted, no extra bytes move. */
int32_t ctd_hess_csr(const ctd_handle* h, int64_t* rowptr, int64_t* colind);
/* host pointers: x[nvar], y[ncon] (constraint multipliers), vals[nnzh]; includes the PCIe copies */
int32_t ctd_hess_coord(ctd_handle* h, const double* x, const double* y, double obj_weight, double* vals);
/* device pointers on the handle's device; the first returns after the handle's stream has drained, the second only
 * enqueues (ctd_sync waits) */
int32_t ctd_hess_coord_dev(ctd_handle* h, const double* x_dev, const double* y_dev, double obj_weight, double* vals_dev);
int32_t ctd_hess_coord_dev_async(ctd_handle* h, const double* x_dev, const double* y_dev, double obj_weight, double* vals_dev);
/* measurement: mean duration (ms) of the Hessian kernel over `iters` launches, per-dispatch events on the handle's stream */
int32_t ctd_time_hess_dev(ctd_handle* h, const double* x_dev, const double* y_dev, double obj_weight, double* vals_dev,
                          int32_t iters, double* mean_ms);
/* launch geometry of the Hessian kernel: out[0..9] = grid blocks, block threads, dynamic LDS bytes, steps per tile,
 * CSC period of the lower triangle (entries per regular step), number of edge entries, second-order eval lanes per
 * stage point / path point / boundary point (after the structural-sparsity probe), terms of the periodic segment */
int32_t ctd_hess_launch_info(ctd_handle* h, int64_t* out10);
/* Which kernel ctd_hess_coord* launches for the regular steps of this handle: out[0] = 0 tile kernel (every scheme), 1 lane-per-step
 * kernel (Gauss-Legendre schemes with 2 / 3 stages of the light registry OCPs on large grids); out[1] = its grid. */
int32_t ctd_hess_kernel_info(ctd_handle* h, int64_t* out2);
/* out[0..3 + nvv): vals_main_begin, vals_main_end (0-based contiguous range of the shard's step columns in the CSC value
 * array), nvv = nv (nv+1)/2, then the 0-based positions of the variable x variable entries (partial sums on a shard) */
int32_t ctd_hess_shard_info(const ctd_handle* h, int64_t* out13);
/* Diagnostics: one launch of the Hessian kernel with in-kernel phase stamps: per workgroup 5 pairs {100 MHz realtime
 * counter, shader cycle counter} at start, after load, after eval, after emit issue, after the stores drained.
 * cap = capacity of out in uint64 words (needs grid * 10). */
int32_t ctd_hess_debug_stamps(ctd_handle* h, const double* x_dev, const double* y_dev, double obj_weight, double* vals_dev,
                              uint64_t* out, int64_t cap);


/* ---- one transcription on several GPUs of ONE process (SURVEY.md section 8b "device list", 8e) ------------------------------
 * The reference has no counterpart (it is single-threaded): this serves a Julia host that owns all GPUs of a node from one
 * process -- the same `ccall` shim, one multi-device handle instead of one handle.  The N time steps are split into
 * n_devices contiguous blocks (block k = steps [k N / n, (k+1) N / n), the ceil / floor rule of a balanced split; the loop
 * being partitioned is src/DOCP_functions.jl:92-98); shard k lives on devices[k] with a stream of its own.  Buffers are
 * FULL-LENGTH on every device (global indexing): x_dev[k] (nvar), c_dev[k] (ncon), vals_dev[k] (nnzj) are device pointers on
 * devices[k]; shard k writes its rows of c, its contiguous CSC range and its slices of the V columns
 * (ctd_sharded_shard_info).  The same device may appear several times (tests on a one-GPU box).  Exchanges between the
 * devices are peer-to-peer copies over xGMI (hipMemcpyPeerAsync on the shards' streams, ordered by events); a failure of one
 * of them is reported as CTD_ERCCL. */
typedef struct ctd_sharded ctd_sharded;

enum {
    CTD_X_IN_PLACE = 0,      /* every x_dev[k] already holds what shard k reads (e.g. the solver replicates x)               */
    CTD_X_SHARDED = 1,       /* x_dev[k] holds shard k's own variables (+ the replicated v).  The few entries a shard needs    */
                             /*   from its neighbours -- next shard's first node, previous shard's last block (midpoint /      */
                             /*   Euler), X_1, X_{N+1} -- are COPIED into x_dev[k] (hipMemcpyPeerAsync on the shards' streams, */
                             /*   ordered by events: a shard's buffer is read behind the work already queued on ITS stream);   */
                             /*   afterwards every callback of shard k's handle can run on x_dev[k].  Works on any topology    */
                             /*   (without peer access the copies go through the host).  The meaning this value has had since  */
                             /*   round 2; round 3 had given it to the in-place mode below                                      */
    CTD_X_FROM_DEVICE0 = 2,  /* x_dev[0] holds the whole iterate: the engine copies all of it to the other devices          */
    CTD_X_SHARDED_COPY = 3,  /* = CTD_X_SHARDED (kept for callers of round 3)                                                 */
    CTD_X_SHARDED_IN_PLACE = 4 /* as CTD_X_SHARDED, but nothing is copied: the kernels LOAD those entries in place from the   */
                             /*   owner's buffer through the peer mappings (xGMI; ctd_set_x_shards).  Ordering: every shard's  */
                             /*   stream records an event, the shards that read its buffer wait for it -- so work the caller   */
                             /*   queued on x_dev[k]'s device stream (ctd_set_stream) before this call is ordered before the   */
                             /*   neighbours' reads; writes from anywhere else must be complete.  The other callbacks of a     */
                             /*   shard handle then read the neighbours' entries through the same table (objective, Hessian)   */
                             /*   -- except ctd_grad*, which needs a whole iterate.  Needs peer access between the devices of  */
                             /*   neighbouring shards, of the first and of the last shard (recorded by ctd_create_sharded): a  */
                             /*   pair without it makes the call take the CTD_X_SHARDED protocol instead of faulting inside    */
                             /*   the kernel, and ctd_sharded_last_error names the pair                                         */
};

/* desc->device, step_begin / step_end and stream are ignored (must be 0 / NULL); devices[k] are HIP ordinals */
int32_t ctd_create_sharded(const ctd_desc* desc, const int32_t* devices, int32_t n_devices, ctd_sharded** out);
int32_t ctd_sharded_destroy(ctd_sharded* s);
const char* ctd_sharded_last_error(const ctd_sharded* s);
/* the k-th shard's single-device handle (owned by s): sizes, patterns, bounds, ctd_shard_info, ctd_set_stream, and every
 * single-device callback (objective, gradient, Hessian) on that shard */
int32_t ctd_sharded_handle(ctd_sharded* s, int32_t k, ctd_handle** h);
/* out[0..9] = n_devices, devices[k], then ctd_shard_info(shard k)[0..7] */
int32_t ctd_sharded_shard_info(const ctd_sharded* s, int32_t k, int64_t* out10);
/* Enqueue one fused evaluation on every device: [iterate distribution per x_mode] -> shard kernels -> [stitch != 0: every
 * device receives the other shards' row blocks of c (the p + bc tail rows from the last shard), so each c_dev[k] ends up
 * whole].  Nothing waits on the host.  The
 * caller's writes to x_dev must be complete (or ordered before the shards' streams) when this is called. */
int32_t ctd_cons_jac_sharded_dev_async(ctd_sharded* s, double* const* x_dev, double* const* c_dev, double* const* vals_dev,
                                       int32_t x_mode, int32_t stitch);
/* ---- sharded iterate read in place ------------------------------------------------------------------------------------------
 * One shard handle (ctd_desc.step_begin / step_end) usually finds everything it reads in the x passed to the call.  With the
 * iterate sharded like the steps (a shard holds its own steps' variables and the replicated v in a
 * full-length buffer) a few entries belong to other shards: the next shard's first node, the previous shard's last step block
 * (one-point schemes), X_1 and X_{N+1} (boundary rows).  After ctd_set_x_shards the constraint / Jacobian, objective and Hessian kernels of `h` load
 * those entries straight from x_bufs[k], the full-length buffer of shard k = steps [step_begin[k], step_begin[k+1]) -- a
 * peer-mapped pointer of another device of this process, or the mapping of another process' buffer (ctd_ipc_open below) --
 * so the evaluation of a shard contains no exchange step at all: the loop being split is src/DOCP_functions.jl:92-98, and
 * step i reads only X_i, U_i, K_i, X_{i+1}, v.  `self` = this handle's shard (x_bufs[self] is ignored: its x is the call's
 * argument); n_shards <= 16; n_shards = 0 switches back.  The writers' updates of x must be complete (or ordered before this
 * handle's stream) when an evaluation is enqueued -- the acceptance test of a solver iteration is a collective already.
 * The objective (src/DOCP_functions.jl:23-54: its quadrature over the shard's steps, the Mayer term on the last shard) and the
 * Hessian callbacks (ctd_hess_coord*, ctd_eval_all_dev_async; the multipliers y are replicated) follow the same table since
 * round 3.  ctd_grad* is the gradient of the WHOLE objective on every rank (see there) and ignores the table: it reads only the
 * x it is given, which must then hold every variable (an all-gathered copy); ctd_grad_shard_dev_async (round 4) is the sharded
 * form that follows the table. */
int32_t ctd_set_x_shards(ctd_handle* h, int32_t n_shards, const int64_t* step_begin, const double* const* x_bufs, int32_t self);
/* One process per GPU (e.g. a Julia host under MPI): the "RCCL all-gather over xGMI for the stitched constraint vector" of the
 * north star, inside the library.  comm is an ncclComm_t of n_ranks ranks created by the host with ITS copy of librccl (found
 * with dlopen at first use; no link-time dependency; CTD_RCCL_LIB overrides); rank r evaluates block r of the balanced split
 * (ctd_shard_steps gives the [begin, end) to put into ctd_desc).  c_dev is the full-length constraint vector in which this
 * rank's rows are already written (ctd_cons_jac_dev_async); on return of the stream every rank holds all N cb step rows, and the
 * p + bc tail rows as computed by the LAST rank.  Equal blocks without tail rows: ONE in-place ncclAllGather; otherwise padded
 * blocks + one index kernel.  Enqueue-only on the handle's stream.  A failing collective returns CTD_ERCCL. */
int32_t ctd_stitch_c(ctd_handle* h, void* nccl_comm, int32_t n_ranks, int32_t rank, double* c_dev);
/* block k of the balanced split of N steps over n_shards (the first N % n_shards blocks hold one step more) */
int32_t ctd_shard_steps(int64_t N, int32_t n_shards, int32_t k, int64_t* begin, int64_t* end);

/* One process per GPU: export a device buffer (any pointer inside a hipMalloc'ed allocation, e.g. a tensor of a caching
 * allocator: `handle64` names the allocation, `offset` the pointer's byte offset in it) and map it in another process of the
 * node (ctd_ipc_open returns the allocation's base in this process; add the offset).  hipIpcGetMemHandle / hipIpcOpenMemHandle. */
#define CTD_IPC_HANDLE_BYTES 64
int32_t ctd_ipc_export(int32_t device, const void* dev_ptr, void* handle64, int64_t* offset);
int32_t ctd_ipc_open(int32_t device, const void* handle64, void** base);
int32_t ctd_ipc_close(int32_t device, void* base);
/* reads `bytes` (<= 64) at a mapped pointer with a device-to-device copy: CTD_ERCCL if `device` cannot reach it (an error code
 * here instead of a fault inside a kernel); hosts call it once per mapping before ctd_set_x_shards */
int32_t ctd_ipc_probe(int32_t device, const void* ptr, size_t bytes);

/* waits for every shard's stream */
int32_t ctd_sharded_sync(ctd_sharded* s);
/* Device memory for hosts without a HIP binding of their own (examples/cabi_demo.c; a Julia host passes AMDGPU.jl arrays
 * instead): hipMalloc / hipFree / hipMemcpy on the given device.  kind: 0 = host -> device, 1 = device -> host. */
int32_t ctd_dev_alloc(int32_t device, size_t bytes, void** ptr);
int32_t ctd_dev_free(int32_t device, void* ptr);
int32_t ctd_dev_copy(int32_t device, void* dst, const void* src, size_t bytes, int32_t kind);

#ifdef __cplusplus
}
#endif
#endif /* CTDIRECT_HIP_H */
