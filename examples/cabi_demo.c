/* examples/cabi_demo.c -- the C ABI of include/ctdirect_hip.h from plain C99, the way a foreign-function binding uses it.
 *
 *   gcc -std=c99 -Iinclude examples/cabi_demo.c -o cabi_demo -Lctdirect.jl_amd -lctdirect_hip -Wl,-rpath,$PWD/ctdirect.jl_amd
 *   ./cabi_demo            host-only part: sizes, bounds, default initial guess, Jacobian structure   (no GPU needed)
 *   ./cabi_demo gpu        + one fused evaluation of constraints and Jacobian values on device 0, and the same evaluation
 *                          through the multi-device entry points (two shards; both on device 0 here, devices {0, 1, ..} on a
 *                          multi-GPU node): iterate sent from shard 0, residual stitched on every shard; and the Jacobian in CSR order
 *                          (ctd_desc.value_order = CTD_ORDER_CSR: rowptr / colind for a GPU KKT consumer), checked entry by entry
 *                          against the CSC values
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
        {   /* one transcription on several devices of this process (ctd_create_sharded) */
            ctd_sharded* s = NULL;
            const int32_t devices[2] = {0, 0};
            void *xd[2], *cd[2], *vd[2];
            double* c2 = (double*)malloc(sizeof(double) * (size_t)ncon);
            int64_t info[10];
            int k, bad = 0;
            d.device = -1;
            if (ctd_create_sharded(&d, devices, 2, &s) != CTD_OK) { fprintf(stderr, "%s\n", ctd_sharded_last_error(NULL)); return 3; }
            for (k = 0; k < 2; ++k) {
                if (ctd_dev_alloc(devices[k], sizeof(double) * (size_t)nvar, &xd[k]) || ctd_dev_alloc(devices[k], sizeof(double) * (size_t)ncon, &cd[k]) ||
                    ctd_dev_alloc(devices[k], sizeof(double) * (size_t)nnzj, &vd[k])) return 3;
                ctd_sharded_shard_info(s, k, info);
                printf("shard %d on device %lld: steps [%lld, %lld), c rows [%lld, %lld), CSC range [%lld, %lld)\n", k, (long long)info[1],
                       (long long)info[2], (long long)info[3], (long long)info[4], (long long)info[5], (long long)info[6], (long long)info[7]);
            }
            if (ctd_dev_copy(devices[0], xd[0], x, sizeof(double) * (size_t)nvar, 0)) return 3;       /* the iterate lives on shard 0 */
            if (ctd_cons_jac_sharded_dev_async(s, (double* const*)xd, (double* const*)cd, (double* const*)vd, CTD_X_FROM_DEVICE0, 1) ||
                ctd_sharded_sync(s)) { fprintf(stderr, "%s\n", ctd_sharded_last_error(s)); return 3; }
            if (ctd_dev_copy(devices[1], c2, cd[1], sizeof(double) * (size_t)ncon, 1)) return 3;      /* stitched: whole c on shard 1 too */
            for (k = 0; k < (int)ncon; ++k) bad += (c2[k] != c[k]);
            printf("multi-device: stitched c on shard 1 differs from the single-device c in %d of %lld rows\n", bad, (long long)ncon);
            for (k = 0; k < 2; ++k) { ctd_dev_free(devices[k], xd[k]); ctd_dev_free(devices[k], cd[k]); ctd_dev_free(devices[k], vd[k]); }
            ctd_sharded_destroy(s);
            free(c2);
            if (bad) return 4;
        }
        {   /* the same Jacobian assembled in CSR order: rowptr / colind (0-based) + values by rows */
            ctd_handle* hr = NULL;
            int64_t* rowptr = (int64_t*)malloc(sizeof(int64_t) * (size_t)(ncon + 1));
            int64_t* colind = (int64_t*)malloc(sizeof(int64_t) * (size_t)nnzj);
            double* cr = (double*)malloc(sizeof(double) * (size_t)ncon);
            double* vr = (double*)malloc(sizeof(double) * (size_t)nnzj);
            int64_t k, q, bad = 0;
            d.device = 0;
            d.value_order = CTD_ORDER_CSR;
            if (ctd_create(&d, &hr) != CTD_OK) { fprintf(stderr, "%s\n", ctd_last_error(NULL)); return 5; }
            if (ctd_jac_csr(hr, rowptr, colind) || ctd_cons_jac(hr, x, cr, vr)) { fprintf(stderr, "%s\n", ctd_last_error(hr)); return 5; }
            for (k = 0; k < nnzj; ++k) {           /* CSC entry k = (rows[k], cols[k]) 1-based: find it in its CSR row */
                const int64_t r = rows[k] - 1, cc = cols[k] - 1;
                for (q = rowptr[r]; q < rowptr[r + 1] && colind[q] != cc; ++q) {}
                bad += (q == rowptr[r + 1] || vr[q] != vals[k]);
            }
            printf("CSR order: rowptr[ncon] %lld, values differ from the CSC values in %lld of %lld entries\n", (long long)rowptr[ncon], (long long)bad,
                   (long long)nnzj);
            ctd_destroy(hr);
            free(rowptr); free(colind); free(cr); free(vr);
            d.value_order = CTD_ORDER_CSC;
            if (bad) return 6;
        }
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
