/* A plain C host calling the C ABI of libjstsp_mi355x.so exactly as the MEX gateway does (host memory in,
 * host memory out, one context) - no Python, no torch in the process.  Built and run by
 * tests/test_gpu_capi_c_host.py, which regenerates the same inputs and checks the outputs against the oracle.
 *
 *   gcc -O2 -I include tests/capi/host_example.c -o host_example -L jstsp19_amd/csrc -ljstsp_mi355x -lm
 *   ./host_example out.bin
 *
 * Inputs: a deterministic 64-bit LCG (same recurrence in the Python test) fills subY, Omega, A, B.
 * Output file: S (Gr*G2 complex64), Y (N*M complex64), ce (Imax*3 doubles), then x_hat / indexSet of OMP. */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include "jstsp.h"

static uint64_t lcg_state = 0x2545F4914F6CDD1DULL;
static double lcg_uniform(void)          /* (0,1) */
{
    lcg_state = lcg_state * 6364136223846793005ULL + 1442695040888963407ULL;
    return ((double)(lcg_state >> 11) + 0.5) / 9007199254740992.0;
}
static void fill_c32(jstsp_c32 *x, size_t n, double scale)
{
    for (size_t i = 0; i < n; ++i) {
        x[i].re = (float)(scale * (2.0 * lcg_uniform() - 1.0));
        x[i].im = (float)(scale * (2.0 * lcg_uniform() - 1.0));
    }
}
#define CHECK(call)                                                                          \
    do {                                                                                     \
        int rc_ = (call);                                                                    \
        if (rc_ != 0) {                                                                      \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, jstsp_last_error());         \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s out.bin\n", argv[0]); return 2; }
    enum { N = 12, M = 40, Gr = 10, G2 = 18, Imax = 25, OMP_M = 5 };
    jstsp_c32 *subY = malloc(sizeof(jstsp_c32) * N * M), *A = malloc(sizeof(jstsp_c32) * N * Gr),
              *B = malloc(sizeof(jstsp_c32) * G2 * M), *S = malloc(sizeof(jstsp_c32) * Gr * G2),
              *Y = malloc(sizeof(jstsp_c32) * N * M), *xh = malloc(sizeof(jstsp_c32) * Gr * G2);
    float *Omega = malloc(sizeof(float) * N * M);
    double *ce = malloc(sizeof(double) * Imax * 3);
    int32_t idx[OMP_M];
    if (!subY || !A || !B || !S || !Y || !xh || !Omega || !ce) return 3;
    fill_c32(A, (size_t)N * Gr, 1.0 / sqrt((double)N));
    fill_c32(B, (size_t)G2 * M, 1.0 / sqrt((double)G2));
    fill_c32(subY, (size_t)N * M, 1.0);
    for (size_t i = 0; i < (size_t)N * M; ++i) {          /* random sampling mask, subY supported on it */
        Omega[i] = lcg_uniform() < 0.4 ? 1.f : 0.f;
        subY[i].re *= Omega[i];
        subY[i].im *= Omega[i];
    }
    const double tau_Y = 0.02, tau_S = 0.01, rho = 0.35;

    jstsp_ctx *ctx = NULL;
    CHECK(jstsp_create(0, &ctx));
    CHECK(jstsp_proposed_algorithm_c32(ctx, N, M, Gr, G2, 1, subY, Omega, A, 0, B, 0, Imax, &tau_Y, &tau_S, &rho,
                                       JSTSP_TYPE_APPROXIMATE, NULL, S, Y, ce, JSTSP_HOST));
    /* OMP on the Kronecker dictionary kron(B.', A) with v = vec(subY) */
    CHECK(jstsp_omp_kron_c32(ctx, N, M, Gr, G2, 1, A, 0, B, 0, subY, OMP_M, xh, idx, JSTSP_HOST));
    /* argument checking: a NULL array must be refused with an error code, not crash */
    if (jstsp_svt_c32(ctx, 4, 4, 1, NULL, &tau_Y, Y, JSTSP_HOST) == 0) { fprintf(stderr, "NULL accepted\n"); return 4; }
    CHECK(jstsp_destroy(ctx));

    FILE *f = fopen(argv[1], "wb");
    if (!f) return 5;
    fwrite(S, sizeof(jstsp_c32), (size_t)Gr * G2, f);
    fwrite(Y, sizeof(jstsp_c32), (size_t)N * M, f);
    fwrite(ce, sizeof(double), (size_t)Imax * 3, f);
    fwrite(xh, sizeof(jstsp_c32), (size_t)Gr * G2, f);
    fwrite(idx, sizeof(int32_t), OMP_M, f);
    fclose(f);
    printf("ok %s\n", jstsp_version());
    return 0;
}
