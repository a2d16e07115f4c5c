/* TEST INFRASTRUCTURE ONLY (see celerite_oracle.c): the same restatement of init_semi_separable! / solve_prec! / logl
 * (src/celerite_solver.jl:12-100, 115-158, 312-334) carried out in COMPLEX arithmetic, so that the derivative of log L
 * with respect to any of (a_j, b_j, c_j, d_j, y_n, sigma2_n) is Im(logl(x + i h)) / h to rounding accuracy (complex
 * step, h = 1e-30) — what ForwardDiff's Duals give the reference through the generic `logl` (:316,
 * test/test_likelihood.jl:55-60).  t stays real.  log|D_n| (:140) is taken as the analytic log D_n, valid where the
 * likelihood is (D_n > 0). */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef double complex cplx;
#define IDX(M, R, j, n) ((M)[(size_t)(n) * (size_t)(R) + (size_t)(j)])

static void init_semi_separable_c(int64_t J, int64_t N, const cplx *a, const cplx *b, const cplx *c, const cplx *d,
                                  const double *tau, const cplx *sigma2, cplx *V, cplx *D, cplx *U, cplx *phi, cplx *S)
{
    const int64_t R = 2 * J;
    cplx suma = 0.0;
    for (int64_t j = 0; j < J; ++j) suma += a[j];
    D[0] = suma + sigma2[0];
    cplx dn = D[0];
    cplx buff = 1.0 / dn;
    for (int64_t j = 0; j < J; ++j) {
        const cplx co = ccos(d[j] * tau[0]), si = csin(d[j] * tau[0]);
        IDX(V, R, 2 * j + 1, 0) = si * buff;
        IDX(V, R, 2 * j, 0) = co * buff;
        IDX(U, R, 2 * j + 1, 0) = a[j] * si - b[j] * co;
        IDX(U, R, 2 * j, 0) = a[j] * co + b[j] * si;
    }
    for (int64_t n = 1; n < N; ++n) {
        cplx s = 0.0;
        const double taun = tau[n], dtau = taun - tau[n - 1];
        for (int64_t j = 0; j < J; ++j) {
            const cplx co = ccos(d[j] * taun), si = csin(d[j] * taun), ec = cexp(-c[j] * dtau);
            IDX(phi, R, 2 * j + 1, n - 1) = ec;
            IDX(phi, R, 2 * j, n - 1) = ec;
            IDX(U, R, 2 * j + 1, n) = a[j] * si - b[j] * co;
            IDX(U, R, 2 * j, n) = a[j] * co + b[j] * si;
            IDX(V, R, 2 * j + 1, n) = si;
            IDX(V, R, 2 * j, n) = co;
        }
        for (int64_t j = 0; j < R; ++j) {
            const cplx uj = IDX(U, R, j, n);
            const cplx phinj = IDX(phi, R, j, n - 1);
            const cplx vn = IDX(V, R, j, n - 1);
            dn = D[n - 1] * vn;
            cplx vnj = IDX(V, R, j, n);
            for (int64_t k = 0; k < j; ++k) {
                const cplx uk = IDX(U, R, k, n);
                const cplx r = phinj * IDX(phi, R, k, n - 1) * (S[k * R + j] + dn * IDX(V, R, k, n - 1));
                S[k * R + j] = r;
                const cplx v = uj * r;
                IDX(V, R, k, n) -= v;
                vnj -= uk * r;
                s += 2 * v * uk;
            }
            S[j * R + j] = (phinj * phinj) * (S[j * R + j] + dn * vn);
            const cplx r = S[j * R + j] * uj;
            s += r * uj;
            IDX(V, R, j, n) = vnj - r;
        }
        dn = suma + sigma2[n] - s;
        D[n] = dn;
        for (int64_t j = 0; j < R; ++j) IDX(V, R, j, n) /= dn;
    }
}

static cplx solve_prec_c(int64_t N, int64_t R, cplx *z, const cplx *y, const cplx *U, const cplx *W, const cplx *D,
                         const cplx *phi, cplx *f, cplx *g)
{
    for (int64_t j = 0; j < R; ++j) f[j] = 0.0, g[j] = 0.0;
    cplx logdetD = clog(D[0]);
    z[0] = y[0];
    for (int64_t n = 1; n < N; ++n) {
        cplx s = 0.0;
        const cplx z_p = z[n - 1];
        for (int64_t j = 0; j < R; ++j) {
            f[j] = (f[j] + IDX(W, R, j, n - 1) * z_p) * IDX(phi, R, j, n - 1);
            s += IDX(U, R, j, n) * f[j];
        }
        logdetD += clog(D[n]);
        z[n] = y[n] - s;
    }
    z[N - 1] /= D[N - 1];
    for (int64_t n = N - 2; n >= 0; --n) {
        cplx s = 0.0;
        const cplx zn = z[n + 1];
        for (int64_t j = 0; j < R; ++j) {
            g[j] = (g[j] + IDX(U, R, j, n + 1) * zn) * IDX(phi, R, j, n);
            s += IDX(W, R, j, n) * g[j];
        }
        z[n] = z[n] / D[n] - s;
    }
    return logdetD;
}

/* logl of complex (a, b, c, d, y, sigma2): inputs as separate real / imaginary arrays (im may be NULL = 0); returns Re, writes Im */
double oracle_logl_complex_cd(int64_t N, int64_t J, const double *a_re, const double *a_im, const double *b_re,
                              const double *b_im, const double *c_re, const double *c_im, const double *d_re, const double *d_im,
                              const double *tau, const double *y_re, const double *y_im, const double *s2_re, const double *s2_im,
                              double *im_out)
{
    const int64_t R = 2 * J;
    cplx *a = malloc(sizeof(cplx) * (size_t)J), *b = malloc(sizeof(cplx) * (size_t)J);
    cplx *c = malloc(sizeof(cplx) * (size_t)J), *d = malloc(sizeof(cplx) * (size_t)J);
    cplx *y = malloc(sizeof(cplx) * (size_t)N), *s2 = malloc(sizeof(cplx) * (size_t)N);
    for (int64_t j = 0; j < J; ++j) {
        a[j] = a_re[j] + I * (a_im ? a_im[j] : 0.0);
        b[j] = b_re[j] + I * (b_im ? b_im[j] : 0.0);
        c[j] = c_re[j] + I * (c_im ? c_im[j] : 0.0);
        d[j] = d_re[j] + I * (d_im ? d_im[j] : 0.0);
    }
    for (int64_t n = 0; n < N; ++n) {
        y[n] = y_re[n] + I * (y_im ? y_im[n] : 0.0);
        s2[n] = s2_re[n] + I * (s2_im ? s2_im[n] : 0.0);
    }
    cplx *S = calloc((size_t)(R * R), sizeof(cplx));
    cplx *phi = malloc(sizeof(cplx) * (size_t)(R * (N > 1 ? N - 1 : 1)));
    cplx *U = malloc(sizeof(cplx) * (size_t)(R * N)), *V = malloc(sizeof(cplx) * (size_t)(R * N));
    cplx *D = malloc(sizeof(cplx) * (size_t)N), *z = malloc(sizeof(cplx) * (size_t)N), *fg = malloc(sizeof(cplx) * (size_t)(2 * R));
    init_semi_separable_c(J, N, a, b, c, d, tau, s2, V, D, U, phi, S);
    const cplx logdetD = solve_prec_c(N, R, z, y, U, V, D, phi, fg, fg + R);
    cplx ytz = 0.0;
    for (int64_t n = 0; n < N; ++n) ytz += y[n] * z[n];
    const cplx res = -logdetD / 2 - (double)N * log(2 * M_PI) / 2 - ytz / 2;
    free(a); free(b); free(c); free(d); free(y); free(s2); free(S); free(phi); free(U); free(V); free(D); free(z); free(fg);
    if (im_out) *im_out = cimag(res);
    return creal(res);
}

double oracle_logl_complex(int64_t N, int64_t J, const double *a_re, const double *a_im, const double *b_re,
                           const double *b_im, const double *c, const double *d, const double *tau, const double *y_re,
                           const double *y_im, const double *s2_re, const double *s2_im, double *im_out)
{
    return oracle_logl_complex_cd(N, J, a_re, a_im, b_re, b_im, c, NULL, d, NULL, tau, y_re, y_im, s2_re, s2_im, im_out);
}
