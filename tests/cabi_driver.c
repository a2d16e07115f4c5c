/* Plain-C driver of the C ABI (include/pioran_hip.h): proves the boundary without Python marshalling.
 *   gcc -std=c11 -Iinclude tests/cabi_driver.c -o tests/cabi_driver -Lpioran.jl_amd -lpioran_hip -Wl,-rpath,'$ORIGIN/../pioran.jl_amd' -lm
 *   tests/cabi_driver case.txt
 * case.txt (written by tests/test_gpu_cabi.py from the reference's literal inputs and the oracle's values):
 *   J N B
 *   c[J]  d[J]  t[N]  y[N]  sigma2[N]
 *   B rows:  a[J] b[J] mu nu  expected_logl
 *   expected_dense_nll_of_row_0   (mu subtracted, nu applied)
 * Exit code 0 and "CABI OK" when every entry point agrees with the expected values to 1e-10 (relative). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "pioran_hip.h"

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        int rc_ = (call);                                                                             \
        if (rc_ != PIORAN_OK) {                                                                       \
            fprintf(stderr, "%s -> %d (%s; %s)\n", #call, rc_, pioran_strerror(rc_), ctx ? pioran_last_hip_error(ctx) : ""); \
            return 2;                                                                                 \
        }                                                                                             \
    } while (0)

static double* read_vec(FILE* f, long n)
{
    double* v = (double*)malloc(sizeof(double) * (size_t)n);
    for (long i = 0; i < n; ++i)
        if (fscanf(f, "%lf", &v[i]) != 1) { fprintf(stderr, "short input\n"); exit(3); }
    return v;
}

static int close_enough(double got, double want) { return fabs(got - want) <= 1e-10 * fabs(want); }

int main(int argc, char** argv)
{
    pioran_ctx* ctx = NULL;
    if (argc < 2) { fprintf(stderr, "usage: cabi_driver case.txt\n"); return 3; }
    FILE* f = fopen(argv[1], "r");
    if (!f) { perror(argv[1]); return 3; }
    long J, N, B;
    if (fscanf(f, "%ld %ld %ld", &J, &N, &B) != 3) return 3;
    double *c = read_vec(f, J), *d = read_vec(f, J), *t = read_vec(f, N), *y = read_vec(f, N), *s2 = read_vec(f, N);
    double *A = malloc(sizeof(double) * B * J), *Bc = malloc(sizeof(double) * B * J), *mu = malloc(sizeof(double) * B),
           *nu = malloc(sizeof(double) * B), *want = malloc(sizeof(double) * B), *out = malloc(sizeof(double) * B);
    int32_t* status = malloc(sizeof(int32_t) * B);
    for (long b = 0; b < B; ++b) {
        for (long j = 0; j < J; ++j) if (fscanf(f, "%lf", &A[b * J + j]) != 1) return 3;
        for (long j = 0; j < J; ++j) if (fscanf(f, "%lf", &Bc[b * J + j]) != 1) return 3;
        if (fscanf(f, "%lf %lf %lf", &mu[b], &nu[b], &want[b]) != 3) return 3;
    }
    double want_dense;
    if (fscanf(f, "%lf", &want_dense) != 1) return 3;
    fclose(f);

    if (pioran_abi_version() != 7) { fprintf(stderr, "ABI version %d, this driver was written for 7\n", pioran_abi_version()); return 2; }
    CHECK(pioran_ctx_create(0, &ctx));
    pioran_ds* ds = NULL;
    CHECK(pioran_dataset_create(ctx, N, t, y, s2, &ds));
    /* batch entry, shared (c, d), per-draw mu / nu */
    CHECK(pioran_celerite_logl_batch(ds, B, J, A, Bc, c, d, 1, mu, nu, NULL, NULL, out, status));
    int bad = 0;
    for (long b = 0; b < B; ++b)
        if (status[b] != 0 || !close_enough(out[b], want[b])) { fprintf(stderr, "batch[%ld]: %.17g vs %.17g (status %d)\n", b, out[b], want[b], status[b]); ++bad; }
    /* scalar drop-in for logl: y with the mean subtracted, sigma2 scaled, as the reference passes them */
    double* y0 = malloc(sizeof(double) * N); double* s0 = malloc(sizeof(double) * N);
    for (long n = 0; n < N; ++n) { y0[n] = y[n] - mu[0]; s0[n] = nu[0] * s2[n]; }
    double one = 0.0; int32_t st1 = -1;
    CHECK(pioran_celerite_logl(ctx, N, J, A, Bc, c, d, t, y0, s0, &one, &st1));
    if (st1 != 0 || !close_enough(one, want[0])) { fprintf(stderr, "scalar: %.17g vs %.17g\n", one, want[0]); ++bad; }
    /* dense solver: +NLL, the reference's sign (src/direct_solver.jl:19) */
    double nll = 0.0; int32_t info = -1;
    CHECK(pioran_dense_nll(ctx, N, J, A, Bc, c, d, t, y0, s0, &nll, &info));
    if (info != 0 || !close_enough(nll, want_dense) || fabs(nll + one) > 1e-8 * fabs(nll)) { fprintf(stderr, "dense: %.17g vs %.17g\n", nll, want_dense); ++bad; }
    /* argument errors come back as codes, nothing unwinds */
    if (pioran_celerite_logl_batch(ds, 0, J, A, Bc, c, d, 1, NULL, NULL, NULL, NULL, out, NULL) != PIORAN_ERR_ARG) ++bad;
    if (pioran_celerite_logl_batch_dev(ds, B, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL) != PIORAN_ERR_ARG) ++bad;
    CHECK(pioran_ctx_trim(ctx));
    CHECK(pioran_dataset_destroy(ds));
    CHECK(pioran_ctx_destroy(ctx));
    if (bad) { fprintf(stderr, "%d mismatches\n", bad); return 1; }
    printf("CABI OK  logl[0] = %.15g  dense nll = %.15g  config = %s\n", one, nll, pioran_celerite_config_name(2 * J));
    return 0;
}
