/* Plain-C use of the drop-in boundary (include/pcs_hip.h): template chain on a toy problem.
 *
 *   gcc -std=c99 -Iinclude examples/c_api_demo.c -o /tmp/c_api_demo -Lpycamset_amd -lpcs_hip \
 *       -Wl,-rpath,$PWD/pycamset_amd -Wl,-rpath,/opt/rocm/lib
 *
 * Needs an MI355X at run time; without one pcs_create reports PCS_ERR_NODEVICE (no CPU fallback).
 */
#include <stdio.h>
#include <stdlib.h>

#include "pcs_hip.h"

#define CHECK(call)                                                      \
    do {                                                                 \
        int rc_ = (call);                                                \
        if (rc_ != PCS_OK) {                                             \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, pcs_last_error()); \
            return 1;                                                    \
        }                                                                \
    } while (0)

int main(void) {
    enum { C = 2, I = 2, K = 4, N = 8 };
    /* detection table [cam, im, key, u, v] (pyCamSet target_detections.py:51-55) */
    double det[N * 5];
    for (int n = 0; n < N; ++n) {
        det[5 * n + 0] = n / 4;
        det[5 * n + 1] = (n / 2) % 2;
        det[5 * n + 2] = n % 4;
        det[5 * n + 3] = 500.0 + 3.0 * n;
        det[5 * n + 4] = 480.0 - 2.0 * n;
    }
    const double tmpl[K * 3] = {-0.01, -0.01, 0, 0.01, -0.01, 0, 0.01, 0.01, 0, -0.01, 0.01, 0.002};
    /* parameter string: intr 9 per camera | extr 6 per camera | pose 6 per image (afb:793-818) */
    double prm[15 * C + 6 * I];
    for (int c = 0; c < C; ++c) {
        const double intr[9] = {1000, 500, 1000, 500, 0.01, 0.001, 1e-4, -1e-4, 1e-5};
        for (int j = 0; j < 9; ++j) prm[9 * c + j] = intr[j];
        const double ext[6] = {0.0, 0.3 * c, 0.0, 0.0, 0.0, 0.2};
        for (int j = 0; j < 6; ++j) prm[9 * C + 6 * c + j] = ext[j];
    }
    for (int i = 0; i < I; ++i)
        for (int j = 0; j < 6; ++j) prm[15 * C + 6 * i + j] = i ? 0.01 * (j + 1) : 0.0; /* pose 0 exactly zero */

    pcs_engine *h = NULL;
    CHECK(pcs_create(&h, PCS_CHAIN_TEMPLATE, PCS_F64, C, I, K, 0));
    CHECK(pcs_set_detections_table(h, det, N));
    CHECK(pcs_set_template(h, tmpl));
    const int P = pcs_row_len(h);
    double *resid = (double *)malloc(sizeof(double) * 2 * N);
    double *jac = (double *)malloc(sizeof(double) * 2 * N * P);
    CHECK(pcs_eval(h, prm, resid, jac));
    printf("n_params %lld, row length %d\n", (long long)pcs_n_params(h), P);
    for (int n = 0; n < 2; ++n) printf("detection %d: residual (%.6f, %.6f), du/dfx %.6f, dv/dfy %.6f\n", n, resid[2 * n], resid[2 * n + 1],
                                       jac[(2 * n) * P + 0], jac[(2 * n + 1) * P + 2]);
    int64_t nnz = 0;
    CHECK(pcs_csr_structure(h, NULL, NULL, NULL, &nnz));
    printf("CSR nnz (all parameters free) %lld\n", (long long)nnz);
    free(resid);
    free(jac);
    CHECK(pcs_destroy(h));
    return 0;
}
