/* examples/cabi_demo.c -- the C ABI of include/ctdirect_hip.h from plain C99, the way a foreign-function binding uses it.
 *
 *   gcc -std=c99 -Iinclude examples/cabi_demo.c -o cabi_demo -Lctdirect.jl_amd -lctdirect_hip -Wl,-rpath,$PWD/ctdirect.jl_amd
 *   ./cabi_demo            host-only part: sizes, bounds, default initial guess, Jacobian structure   (no GPU needed)
 *   ./cabi_demo gpu        + one fused evaluation of constraints and Jacobian values on device 0
 *
 * Goddard problem, Gauss-Legendre 2 (stagewise controls), 100 steps: CTDirect.DOCP(goddard().ocp, 100, 1,
 * :gauss_legendre_2, nothing) in the reference (src/DOCP_data.jl:293). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ctdirect_hip.h"

#define CHECK(call)                                                                        \
    do {                                                                                   \
        int32_t st_ = (call);                                                              \
        if (st_ != CTD_OK) {                                                               \
            fprintf(stderr, "%s -> %s: %s\n", #call, ctd_strerror(st_), ctd_last_error(h)); \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

int main(int argc, char** argv) {
    const int use_gpu = argc > 1 && strcmp(argv[1], "gpu") == 0;
    ctd_handle* h = NULL;
    ctd_desc d;
    memset(&d, 0, sizeof(d));
    d.problem = CTD_PROBLEM_GODDARD;
    d.scheme = CTD_SCHEME_GAUSS_LEGENDRE_2;
    d.pattern_mode = CTD_PATTERN_REFERENCE_MANUAL;
    d.device = use_gpu ? 0 : -1;
    d.grid_size = 100;
    CHECK(ctd_create(&d, &h));

    int64_t nvar = 0, ncon = 0, nnzj = 0, nnzh = 0;
    CHECK(ctd_sizes(h, &nvar, &ncon, &nnzj, &nnzh));
    printf("nvar %lld ncon %lld nnzj %lld nnzh %lld\n", (long long)nvar, (long long)ncon, (long long)nnzj, (long long)nnzh);

    double* x = (double*)malloc(sizeof(double) * (size_t)nvar);
    double* lvar = (double*)malloc(sizeof(double) * (size_t)nvar);
    double* uvar = (double*)malloc(sizeof(double) * (size_t)nvar);
    double* lcon = (double*)malloc(sizeof(double) * (size_t)ncon);
    double* ucon = (double*)malloc(sizeof(double) * (size_t)ncon);
    int64_t* rows = (int64_t*)malloc(sizeof(int64_t) * (size_t)nnzj);
    int64_t* cols = (int64_t*)malloc(sizeof(int64_t) * (size_t)nnzj);
    CHECK(ctd_bounds(h, lvar, uvar, lcon, ucon));
    CHECK(ctd_initial_guess(h, x, NULL));                 /* everything 0.1, src/DOCP_variables.jl:126 */
    CHECK(ctd_jac_structure(h, rows, cols));              /* 1-based, CSC order */
    printf("x0[0] %.3f  tf bounds [%g, %g]  first entry (%lld, %lld)  last entry (%lld, %lld)\n", x[0], lvar[nvar - 1],
           uvar[nvar - 1], (long long)rows[0], (long long)cols[0], (long long)rows[nnzj - 1], (long long)cols[nnzj - 1]);

    if (use_gpu) {
        double* c = (double*)malloc(sizeof(double) * (size_t)ncon);
        double* vals = (double*)malloc(sizeof(double) * (size_t)nnzj);
        double f = 0.0;
        CHECK(ctd_cons_jac(h, x, c, vals));               /* cons!(nlp, x, c) + jac_coord!(nlp, x, vals) in one launch */
        CHECK(ctd_obj(h, x, &f));
        printf("objective %.6f  c[0] %.6e  vals[0] %.6e\n", f, c[0], vals[0]);
        free(c); free(vals);
    } else {
        double f;
        int32_t st = ctd_obj(h, x, &f);                   /* no CPU fallback: a host-only handle refuses compute calls */
        printf("compute call on a host-only handle: %s\n", ctd_strerror(st));
        if (st != CTD_ENODEVICE) return 2;
    }
    free(x); free(lvar); free(uvar); free(lcon); free(ucon); free(rows); free(cols);
    ctd_destroy(h);
    return 0;
}
