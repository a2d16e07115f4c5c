/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product.
 *
 * The reference's ScalableGP log-likelihood (mlefkir/Pioran.jl v1.2.0, logl, src/celerite_solver.jl:312-334) evaluated in
 * __float128 (libquadmath, 113-bit significand): the HIGHER-PRECISION TRUTH against which the fp64 evaluations — the fp64 oracle
 * (celerite_oracle.c), the step-by-step GPU kernels and the windowed GPU kernels — are compared on ill-conditioned draws
 * (tests/test_oracle.py, tests/test_gpu_parity.py::test_ill_conditioned_draws_vs_quad_truth, tools/quad_truth.py).
 *
 * Same recurrence as init_semi_separable! (:12-100) and the forward half of solve_prec! (:115-142), same loop order; every input is an
 * fp64 number taken exactly, every product, transcendental and sum is carried out in quad precision (d_j * t_n is NOT rounded to fp64
 * before the sine and cosine).  The quadratic form is taken as sum z_n^2 / D_n with z from the forward substitution, which equals the
 * reference's y' K^-1 y (its backward sweep :145-155 and the dot product :333) exactly in exact arithmetic and to ~1e-30 here.
 * A relative perturbation eps of K's entries moves y' K^-1 y by ~ eps cond(K): with eps = 2^-113 this evaluation is exact to fp64
 * output precision for every conditioning an fp64 input can express.
 */
#include <math.h>
#include <quadmath.h>
#include <stdint.h>
#include <stdlib.h>

typedef __float128 q128;

/* returns log L rounded to fp64; D_min (optional): the smallest D_n (fp64), for the callers' positive-definiteness report */
static double logl_quad_one(int64_t N, int64_t J, const double *a, const double *b, const double *c, const double *d,
                            const double *tau, const double *y, const double *sigma2, double *D_min)
{
    const int64_t R = 2 * J;
    q128 *S = calloc((size_t)(R * R), sizeof(q128));
    q128 *u = malloc(sizeof(q128) * (size_t)R), *v = malloc(sizeof(q128) * (size_t)R), *phi = malloc(sizeof(q128) * (size_t)R);
    q128 *w = malloc(sizeof(q128) * (size_t)R), *wn = malloc(sizeof(q128) * (size_t)R), *f = calloc((size_t)R, sizeof(q128));
    double res = NAN;
    if (!S || !u || !v || !phi || !w || !wn || !f) goto done;
    q128 suma = 0; /* :21 */
    for (int64_t j = 0; j < J; ++j) suma += (q128)a[j];
    /* :27-42 first row */
    q128 Dp = suma + (q128)sigma2[0];
    for (int64_t j = 0; j < J; ++j) {
        q128 arg = (q128)d[j] * (q128)tau[0];
        w[2 * j] = cosq(arg) / Dp;
        w[2 * j + 1] = sinq(arg) / Dp;
    }
    q128 logdet = logq(Dp); /* :126 (a negative D_1 gives NaN, like the reference's DomainError) */
    q128 zp = (q128)y[0];
    q128 quad = zp * zp / Dp;
    double dmin = (double)Dp;
    for (int64_t n = 1; n < N; ++n) {
        q128 dtau = (q128)tau[n] - (q128)tau[n - 1];
        for (int64_t j = 0; j < J; ++j) { /* :51-64 */
            q128 arg = (q128)d[j] * (q128)tau[n];
            q128 co = cosq(arg), si = sinq(arg), ec = expq(-(q128)c[j] * dtau);
            phi[2 * j] = ec; phi[2 * j + 1] = ec;
            u[2 * j] = (q128)a[j] * co + (q128)b[j] * si;
            u[2 * j + 1] = (q128)a[j] * si - (q128)b[j] * co;
            v[2 * j] = co; v[2 * j + 1] = si;
        }
        /* :69-90: S <- (phi phi') o (S + D_{n-1} w w'), q = S u, s = u'Su; W numerator v - q */
        q128 s = 0;
        for (int64_t j = 0; j < R; ++j) wn[j] = v[j];
        for (int64_t j = 0; j < R; ++j) {
            q128 dnw = Dp * w[j];
            for (int64_t k = 0; k < j; ++k) {
                q128 r = phi[j] * phi[k] * (S[k * R + j] + dnw * w[k]);
                S[k * R + j] = r;
                q128 vv = u[j] * r;
                wn[k] -= vv;
                wn[j] -= u[k] * r;
                s += 2 * vv * u[k];
            }
            S[j * R + j] = phi[j] * phi[j] * (S[j * R + j] + dnw * w[j]);
            q128 r = S[j * R + j] * u[j];
            s += r * u[j];
            wn[j] -= r;
        }
        q128 Dn = suma + (q128)sigma2[n] - s; /* :92 */
        /* forward substitution (:132-142): f <- (f + W_{n-1} z_{n-1}) o phi; z_n = y_n - u'f */
        q128 uf = 0;
        for (int64_t j = 0; j < R; ++j) {
            f[j] = (f[j] + w[j] * zp) * phi[j];
            uf += u[j] * f[j];
        }
        q128 zn = (q128)y[n] - uf;
        for (int64_t j = 0; j < R; ++j) w[j] = wn[j] / Dn; /* :95-97 */
        logdet += logq(fabsq(Dn)); /* :140 */
        quad += zn * zn / Dn;
        if ((double)Dn < dmin) dmin = (double)Dn;
        Dp = Dn;
        zp = zn;
    }
    {
        q128 r = -logdet / 2 - (q128)N * logq(2 * M_PIq) / 2 - quad / 2;
        res = (double)r;
    }
    if (D_min) *D_min = dmin;
done:
    free(S); free(u); free(v); free(phi); free(w); free(wn); free(f);
    return res;
}

double oracle_logl_quad(int64_t N, int64_t J, const double *a, const double *b, const double *c, const double *d, const double *tau,
                        const double *y, const double *sigma2, double *D_min)
{
    return logl_quad_one(N, J, a, b, c, d, tau, y, sigma2, D_min);
}

/* B independent evaluations, OpenMP over the batch; A, Bc: draw b at A + b J; shared (c, d); mu, nu as in oracle_logl_batch */
void oracle_logl_quad_batch(int64_t N, int64_t J, int64_t B, const double *A, const double *Bc, const double *C, const double *Dd,
                            const double *mu, const double *nu, const double *tau, const double *y, const double *sigma2, double *out,
                            double *D_min, int nthreads)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int64_t bi = 0; bi < B; ++bi) {
        double *yb = malloc(sizeof(double) * (size_t)N), *sb = malloc(sizeof(double) * (size_t)N);
        if (!yb || !sb) { out[bi] = NAN; free(yb); free(sb); continue; }
        /* y - mu and nu sigma2 are formed in fp64 exactly as the fp64 paths form them (src/scalable_GP.jl:164; the models' nu rescaling):
         * the truth is the value of the SAME fp64 inputs */
        for (int64_t n = 0; n < N; ++n) {
            yb[n] = mu ? y[n] - mu[bi] : y[n];
            sb[n] = nu ? nu[bi] * sigma2[n] : sigma2[n];
        }
        double dm = 0.0;
        out[bi] = logl_quad_one(N, J, A + bi * J, Bc + bi * J, C, Dd, tau, yb, sb, &dm);
        if (D_min) D_min[bi] = dm;
        free(yb); free(sb);
    }
}
