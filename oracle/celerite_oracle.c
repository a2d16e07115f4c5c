/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product.
 *
 * A plain-C, fp64, CPU restatement of the reference's ScalableGP log-likelihood
 * hot path (mlefkir/Pioran.jl v1.2.0, Julia).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this; the product (pioran.jl_amd/)
 * never does.
 *
 * Pinning status: PINNED.  tests/test_oracle.py checks this file against
 *   - 6 000+ log-likelihood values that the reference itself produced (ultranest run
 *     stored under docs/src/data/inference of the reference; fixture
 *     tests/golden/ultranest_points.npz, generator oracle/make_golden.py);
 *   - the reference's own known-answer relation celerite == -dense
 *     (test/test_likelihood.jl:58-59, test/test_scalablegp.jl:128) on its literal inputs.
 *
 * Each function cites the reference lines it follows (paths relative to the reference root).
 * The loop order, the in-place updates and the abs() in the log-determinant are kept as in
 * the reference so that rounding behaviour is the same up to libm differences.
 *
 * Layout: U, V(W), phi are R x N column-major with R fastest, exactly as the Julia
 * matrices (src/celerite_solver.jl:322-326).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#define IDX(M, R, j, n) ((M)[(size_t)(n) * (size_t)(R) + (size_t)(j)])

/* src/celerite_solver.jl:12-100  init_semi_separable! */
static void init_semi_separable(int64_t J, int64_t N, const double *a, const double *b,
                                const double *c, const double *d, const double *tau,
                                const double *sigma2, double *V, double *D, double *U,
                                double *phi, double *S /* R x R col-major, zeroed */)
{
    const int64_t R = 2 * J;
    double suma = 0.0; /* :21 sum(a) */
    for (int64_t j = 0; j < J; ++j) suma += a[j];

    /* :27-42 first row */
    D[0] = suma + sigma2[0];
    double dn = D[0];
    double buff = 1.0 / dn;
    double tau1 = tau[0];
    for (int64_t j = 0; j < J; ++j) {
        double co = cos(d[j] * tau1);
        double si = sin(d[j] * tau1);
        IDX(V, R, 2 * j + 1, 0) = si * buff;
        IDX(V, R, 2 * j, 0) = co * buff;
        IDX(U, R, 2 * j + 1, 0) = a[j] * si - b[j] * co;
        IDX(U, R, 2 * j, 0) = a[j] * co + b[j] * si;
    }

    /* :44-99 */
    for (int64_t n = 1; n < N; ++n) {
        double s = 0.0;
        double taun = tau[n];
        double dtau = taun - tau[n - 1];
        /* :51-64 */
        for (int64_t j = 0; j < J; ++j) {
            double co = cos(d[j] * taun);
            double si = sin(d[j] * taun);
            double ec = exp(-c[j] * dtau);
            IDX(phi, R, 2 * j + 1, n - 1) = ec;
            IDX(phi, R, 2 * j, n - 1) = ec;
            IDX(U, R, 2 * j + 1, n) = a[j] * si - b[j] * co;
            IDX(U, R, 2 * j, n) = a[j] * co + b[j] * si;
            IDX(V, R, 2 * j + 1, n) = si;
            IDX(V, R, 2 * j, n) = co;
        }
        /* :69-90 lower triangle of S, u'Su and the W numerator at the same time */
        for (int64_t j = 0; j < R; ++j) {
            double uj = IDX(U, R, j, n);
            double phinj = IDX(phi, R, j, n - 1);
            double vn = IDX(V, R, j, n - 1);
            dn = D[n - 1] * vn;
            double vnj = IDX(V, R, j, n);
            for (int64_t k = 0; k < j; ++k) {
                double uk = IDX(U, R, k, n);
                double r = phinj * IDX(phi, R, k, n - 1) * (S[k * R + j] + dn * IDX(V, R, k, n - 1));
                S[k * R + j] = r; /* S_n[j,k], col-major */
                double v = uj * r;
                IDX(V, R, k, n) -= v;
                vnj -= uk * r;
                s += 2 * v * uk;
            }
            S[j * R + j] = (phinj * phinj) * (S[j * R + j] + dn * vn);
            double r = S[j * R + j] * uj;
            s += r * uj;
            IDX(V, R, j, n) = vnj - r;
        }
        /* :92-97 */
        dn = suma + sigma2[n] - s;
        D[n] = dn;
        for (int64_t j = 0; j < R; ++j) IDX(V, R, j, n) /= dn;
    }
}

/* src/celerite_solver.jl:115-158  solve_prec!  (returns logdetD, z <- K^-1 y) */
static double solve_prec(int64_t N, int64_t R, double *z, const double *y, const double *U,
                         const double *W, const double *D, const double *phi, double *f, double *g)
{
    /* f and g alias the "previous" buffers after the first step (:139,:153): in-place */
    for (int64_t j = 0; j < R; ++j) f[j] = 0.0, g[j] = 0.0;
    double logdetD = log(D[0]); /* :126, unguarded: NaN for D1 < 0 like Julia's DomainError */
    z[0] = y[0];
    for (int64_t n = 1; n < N; ++n) { /* :132-142 */
        double s = 0.0;
        double z_p = z[n - 1];
        for (int64_t j = 0; j < R; ++j) {
            f[j] = (f[j] + IDX(W, R, j, n - 1) * z_p) * IDX(phi, R, j, n - 1);
            s += IDX(U, R, j, n) * f[j];
        }
        logdetD += log(fabs(D[n])); /* :140 */
        z[n] = y[n] - s;
    }
    z[N - 1] /= D[N - 1]; /* :145 */
    for (int64_t n = N - 2; n >= 0; --n) { /* :146-155 */
        double s = 0.0;
        double zn = z[n + 1];
        for (int64_t j = 0; j < R; ++j) {
            g[j] = (g[j] + IDX(U, R, j, n + 1) * zn) * IDX(phi, R, j, n);
            s += IDX(W, R, j, n) * g[j];
        }
        z[n] = z[n] / D[n] - s;
    }
    return logdetD;
}

/* workspace of one logl call: what the reference allocates per call (:322-330) */
typedef struct {
    double *S, *phi, *U, *V, *D, *z, *fg;
} logl_ws;

static int ws_alloc(logl_ws *w, int64_t N, int64_t R)
{
    w->S = malloc(sizeof(double) * (size_t)(R * R));
    w->phi = malloc(sizeof(double) * (size_t)(R * (N > 1 ? N - 1 : 1)));
    w->U = malloc(sizeof(double) * (size_t)(R * N));
    w->V = malloc(sizeof(double) * (size_t)(R * N));
    w->D = malloc(sizeof(double) * (size_t)N);
    w->z = malloc(sizeof(double) * (size_t)N);
    w->fg = malloc(sizeof(double) * (size_t)(2 * R));
    return w->S && w->phi && w->U && w->V && w->D && w->z && w->fg;
}

static void ws_free(logl_ws *w)
{
    free(w->S); free(w->phi); free(w->U); free(w->V); free(w->D); free(w->z); free(w->fg);
}

static double logl_ws_run(logl_ws *w, int64_t N, int64_t J, const double *a, const double *b, const double *c,
                          const double *d, const double *tau, const double *y, const double *sigma2,
                          int32_t *status)
{
    const int64_t R = 2 * J;
    memset(w->S, 0, sizeof(double) * (size_t)(R * R)); /* S_n = zeros(T, R, R)  :322 */
    init_semi_separable(J, N, a, b, c, d, tau, sigma2, w->V, w->D, w->U, w->phi, w->S);
    double logdetD = solve_prec(N, R, w->z, y, w->U, w->V, w->D, w->phi, w->fg, w->fg + R);
    double ytz = 0.0; /* :333 y'z */
    for (int64_t n = 0; n < N; ++n) ytz += y[n] * w->z[n];
    double res = -logdetD / 2 - (double)N * log(2 * M_PI) / 2 - ytz / 2;
    if (status) {
        int32_t st = 0;
        for (int64_t n = 0; n < N; ++n)
            if (!(w->D[n] > 0.0)) st = 1;
        if (!isfinite(res)) st = 2;
        *status = st;
    }
    return res;
}

/* src/celerite_solver.jl:312-334  logl.
 * status (optional): 0 ok, 1 some D_n <= 0, 2 non-finite result. */
double oracle_logl(int64_t N, int64_t J, const double *a, const double *b, const double *c,
                   const double *d, const double *tau, const double *y, const double *sigma2,
                   int32_t *status)
{
    logl_ws w;
    double res = NAN;
    if (ws_alloc(&w, N, 2 * J)) res = logl_ws_run(&w, N, J, a, b, c, d, tau, y, sigma2, status);
    ws_free(&w);
    return res;
}

/* Same as oracle_logl, also returning D and z (for kernel-level diagnostics in tests). */
double oracle_logl_detail(int64_t N, int64_t J, const double *a, const double *b, const double *c,
                          const double *d, const double *tau, const double *y,
                          const double *sigma2, double *D_out, double *z_out)
{
    const int64_t R = 2 * J;
    double *S = calloc((size_t)(R * R), sizeof(double));
    double *phi = malloc(sizeof(double) * (size_t)(R * (N > 1 ? N - 1 : 1)));
    double *U = malloc(sizeof(double) * (size_t)(R * N));
    double *V = malloc(sizeof(double) * (size_t)(R * N));
    double *fg = malloc(sizeof(double) * (size_t)(2 * R));
    init_semi_separable(J, N, a, b, c, d, tau, sigma2, V, D_out, U, phi, S);
    double logdetD = solve_prec(N, R, z_out, y, U, V, D_out, phi, fg, fg + R);
    double ytz = 0.0;
    for (int64_t n = 0; n < N; ++n) ytz += y[n] * z_out[n];
    free(S); free(phi); free(U); free(V); free(fg);
    return -logdetD / 2 - (double)N * log(2 * M_PI) / 2 - ytz / 2;
}

/*
 * Batched driver used for parity sweeps and for bench.py's cpu_baseline leg:
 * the reference has no batch dimension; this is B independent calls of logl, as a
 * process farm would make them (docs/src/ultranest.md:143-149), OpenMP over the batch.
 *   A, Bc : J x B column-major (draw b at A + b*J).  C, Dd: J x B, or length J when cd_shared.
 *   mu (B or NULL): y <- y - mu_b  (src/scalable_GP.jl:164, ConstMean)
 *   nu (B or NULL): sigma2 <- nu_b * sigma2  (the models' "nu" rescaling, benchmark/benchmarks.jl:58)
 */
void oracle_logl_batch(int64_t N, int64_t J, int64_t B, const double *A, const double *Bc,
                       const double *C, const double *Dd, int cd_shared, const double *mu,
                       const double *nu, const double *tau, const double *y, const double *sigma2,
                       double *out, int32_t *status, int nthreads)
{
    /* each thread keeps ONE workspace for all its draws (a process-farm worker would let its allocator
     * reuse the same pages; per-call malloc of ~10 MB from 100+ threads measures the kernel's mmap lock) */
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
#endif
    {
        logl_ws w;
        double *yb = malloc(sizeof(double) * (size_t)N);
        double *sb = malloc(sizeof(double) * (size_t)N);
        const int okws = ws_alloc(&w, N, 2 * J) && yb && sb;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int64_t bi = 0; bi < B; ++bi) {
            if (!okws) { out[bi] = NAN; if (status) status[bi] = 2; continue; }
            double m = mu ? mu[bi] : 0.0, v = nu ? nu[bi] : 1.0;
            for (int64_t n = 0; n < N; ++n) {
                yb[n] = mu ? y[n] - m : y[n];
                sb[n] = nu ? v * sigma2[n] : sigma2[n];
            }
            const double *cb = cd_shared ? C : C + bi * J;
            const double *db = cd_shared ? Dd : Dd + bi * J;
            int32_t st = 0;
            out[bi] = logl_ws_run(&w, N, J, A + bi * J, Bc + bi * J, cb, db, tau, yb, sb, &st);
            if (status) status[bi] = st;
        }
        ws_free(&w);
        free(yb); free(sb);
    }
}

/* src/acvf.jl:138-140 kappa(::SumOfTerms) = sum of term kappas;
 * src/Celerite.jl:42-44 Celerite_covariance: exp(-c tau) (a cos(d tau) + b sin(d tau)). */
double oracle_kappa(int64_t J, const double *a, const double *b, const double *c, const double *d,
                    double tau)
{
    double k = 0.0;
    for (int64_t j = 0; j < J; ++j)
        k += exp(-c[j] * tau) * (a[j] * cos(d[j] * tau) + b[j] * sin(d[j] * tau));
    return k;
}

/* src/direct_solver.jl:6-21 log_likelihood_direct: dense K, Cholesky, returns +NLL
 *   = sum log U_ii + z'z/2 + N log(2 pi)/2 with z = U' \ y (K = U'U).
 * LAPACK dpotrf is replaced by a row-oriented (dot-product form) Cholesky, K = L L'
 * (L = U'), and dtrtrs by the matching forward substitution.
 * Returns NaN when K is not positive definite (Julia throws PosDefException). */
double oracle_dense_nll(int64_t N, int64_t J, const double *a, const double *b, const double *c,
                        const double *d, const double *t, const double *y, const double *sigma2)
{
    double *L = malloc(sizeof(double) * (size_t)N * (size_t)N); /* row-major lower */
    double *z = malloc(sizeof(double) * (size_t)N);
    if (!L || !z) { free(L); free(z); return NAN; }
    /* :9-15 K[i,j] = cov(t[i], t[j]) (Euclidean metric -> |ti - tj|), + Diagonal(sigma2) */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16)
#endif
    for (int64_t i = 0; i < N; ++i)
        for (int64_t j = 0; j <= i; ++j) {
            double k = oracle_kappa(J, a, b, c, d, fabs(t[i] - t[j]));
            if (i == j) k += sigma2[i];
            L[i * N + j] = k;
        }
    /* :16 cholesky */
    double logdet = 0.0;
    int bad = 0;
    for (int64_t j = 0; j < N && !bad; ++j) {
        double *Lj = L + j * N;
        double s = Lj[j];
        double acc = 0.0;
#pragma omp simd reduction(+ : acc)
        for (int64_t k = 0; k < j; ++k) acc += Lj[k] * Lj[k];
        s -= acc;
        if (!(s > 0.0)) { bad = 1; break; }
        double ljj = sqrt(s);
        Lj[j] = ljj;
        logdet += log(ljj);
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (N - j > 256)
#endif
        for (int64_t i = j + 1; i < N; ++i) {
            double *Li = L + i * N;
            double acc2 = 0.0;
#pragma omp simd reduction(+ : acc2)
            for (int64_t k = 0; k < j; ++k) acc2 += Li[k] * Lj[k];
            Li[j] = (Li[j] - acc2) / ljj;
        }
    }
    double res = NAN;
    if (!bad) {
        /* :17 z = U' \ y  (forward substitution with L) */
        double zz = 0.0;
        for (int64_t i = 0; i < N; ++i) {
            double *Li = L + i * N;
            double acc = 0.0;
#pragma omp simd reduction(+ : acc)
            for (int64_t k = 0; k < i; ++k) acc += Li[k] * z[k];
            z[i] = (y[i] - acc) / Li[i];
            zz += z[i] * z[i];
        }
        res = logdet + 0.5 * zz + 0.5 * (double)N * log(2 * M_PI); /* :19 */
    }
    free(L); free(z);
    return res;
}

/* src/celerite_solver.jl:515-549  sim: exact GP draw given q ~ N(0,1)^N.
 * (Used only to generate the synthetic benchmark series) */
void oracle_sim(int64_t N, int64_t J, const double *a, const double *b, const double *c,
                const double *d, const double *tau, const double *sigma2, const double *q,
                double *y_sim)
{
    const int64_t R = 2 * J;
    double *S = calloc((size_t)(R * R), sizeof(double));
    double *phi = calloc((size_t)(R * (N > 1 ? N - 1 : 1)), sizeof(double));
    double *U = calloc((size_t)(R * N), sizeof(double));
    double *V = calloc((size_t)(R * N), sizeof(double));
    double *D = calloc((size_t)N, sizeof(double));
    double *f = calloc((size_t)R, sizeof(double));
    init_semi_separable(J, N, a, b, c, d, tau, sigma2, V, D, U, phi, S);
    for (int64_t n = 0; n < N; ++n) y_sim[n] = 0.0;
    y_sim[0] = sqrt(D[0]) * q[0];
    for (int64_t n = 1; n < N; ++n) {
        for (int64_t j = 0; j < R; ++j) {
            /* g aliases f after the first step (:545) */
            f[j] = IDX(phi, R, j, n - 1) * (f[j] + IDX(V, R, j, n - 1) * sqrt(D[n - 1]) * q[n - 1]);
            y_sim[n] += IDX(U, R, j, n) * f[j];
        }
        y_sim[n] += sqrt(D[n]) * q[n];
    }
    free(S); free(phi); free(U); free(V); free(D); free(f);
}

/* src/celerite_solver.jl:363-483  pred: posterior mean at the (ascending) times tq[M] of the zero-mean GP given (t, y).
 * z = K^-1 y by init_semi_separable! + solve_prec! (:375-384), then the two walks over the merged sequence:
 *   forward pass (:392-428): Q <- (Q + z_n V_n) e^{-c (t_{n+1} - t_n)} for the data points left of tau (:397-404),
 *                            mu_m = sum_rows (Q + z_{n0} V_{n0}) e^{-c (tau - t_{n0})} U~(tau)       (:412-413,427)
 *   backward pass (:433-479): Q <- (Q + z_n U_n) e^{-c (t_n - t_{n-1})} from the right (:445-446),
 *                            mu_m += sum_rows (Q + z_{n0+1} U_{n0+1}) e^{-c (t_{n0+1} - tau)} V(tau)  (:457-458,475)
 * with n0 = searchsortedfirst(t, tau) - 1 = the number of t_n < tau (:388).  The reference's bookkeeping of which data
 * points Q has absorbed (`start`, `stop`, :395-424,:440-472) is restated as "Q has absorbed exactly the points it must
 * for this tau", which is what that bookkeeping maintains for ascending tau. */
void oracle_predict(int64_t N, int64_t J, const double *a, const double *b, const double *c, const double *d,
                    const double *t, const double *y, const double *sigma2, int64_t M, const double *tq, double *mu_out)
{
    const int64_t R = 2 * J;
    double *S = calloc((size_t)(R * R), sizeof(double));
    double *phi = malloc(sizeof(double) * (size_t)(R * (N > 1 ? N - 1 : 1)));
    double *U = malloc(sizeof(double) * (size_t)(R * N));
    double *V = malloc(sizeof(double) * (size_t)(R * N));
    double *D = malloc(sizeof(double) * (size_t)N);
    double *z = malloc(sizeof(double) * (size_t)N);
    double *fg = malloc(sizeof(double) * (size_t)(2 * R));
    double *Q = calloc((size_t)R, sizeof(double));
    int64_t *n0 = malloc(sizeof(int64_t) * (size_t)(M > 0 ? M : 1));
    init_semi_separable(J, N, a, b, c, d, t, sigma2, V, D, U, phi, S);
    (void)solve_prec(N, R, z, y, U, V, D, phi, fg, fg + R);
    for (int64_t m = 0; m < M; ++m) {   /* :388 */
        int64_t k = 0;
        while (k < N && t[k] < tq[m]) ++k;
        n0[m] = k;                      /* 1-based index of the last t_n < tau, 0 if none */
        mu_out[m] = 0.0;
    }
    /* forward pass: `done` = data points (1-based 1..done) already folded into Q, Q referred to t_{done+1} */
    int64_t done = 0;
    for (int64_t m = 0; m < M; ++m) {
        const int64_t k0 = n0[m];
        if (k0 == 0) continue;          /* nothing to the left: S stays zero (:407) */
        while (done < k0 - 1) {         /* :397-404 */
            const int64_t n = done;     /* 0-based data index */
            for (int64_t j = 0; j < J; ++j) {
                const double e = exp(-c[j] * (t[n + 1] - t[n]));
                Q[2 * j] = (Q[2 * j] + z[n] * cos(d[j] * t[n])) * e;
                Q[2 * j + 1] = (Q[2 * j + 1] + z[n] * sin(d[j] * t[n])) * e;
            }
            ++done;
        }
        const int64_t n = k0 - 1;
        double acc = 0.0;
        for (int64_t j = 0; j < J; ++j) {   /* :412-413, summed in row order like sum(S) (:427) */
            const double e = exp(-c[j] * (tq[m] - t[n]));
            const double ct = cos(d[j] * tq[m]), st = sin(d[j] * tq[m]);
            acc += (Q[2 * j] + z[n] * cos(d[j] * t[n])) * e * (a[j] * ct + b[j] * st);
            acc += (Q[2 * j + 1] + z[n] * sin(d[j] * t[n])) * e * (a[j] * st - b[j] * ct);
        }
        mu_out[m] = acc;
    }
    /* backward pass: `left` = data points (0-based left..N-1) already folded into Q, Q referred to t_{left-1} */
    memset(Q, 0, sizeof(double) * (size_t)R);   /* :430 */
    int64_t left = N;
    for (int64_t m = M - 1; m >= 0; --m) {
        const int64_t k0 = n0[m];
        if (k0 == N) continue;          /* nothing to the right (:436) */
        while (left > k0 + 1) {         /* :440-448: fold n = left-1 (0-based), decay to t_{n-1} */
            const int64_t n = left - 1;
            for (int64_t j = 0; j < J; ++j) {
                const double e = exp(-c[j] * (t[n] - t[n - 1]));
                const double ct = cos(d[j] * t[n]), st = sin(d[j] * t[n]);
                Q[2 * j] = (Q[2 * j] + z[n] * (a[j] * ct + b[j] * st)) * e;
                Q[2 * j + 1] = (Q[2 * j + 1] + z[n] * (a[j] * st - b[j] * ct)) * e;
            }
            --left;
        }
        const int64_t n = k0;           /* 0-based index of the first t_n >= tau */
        double acc = 0.0;
        for (int64_t j = 0; j < J; ++j) {   /* :457-458 */
            const double e = exp(-c[j] * (t[n] - tq[m]));
            const double ct = cos(d[j] * t[n]), st = sin(d[j] * t[n]);
            acc += (Q[2 * j] + z[n] * (a[j] * ct + b[j] * st)) * e * cos(d[j] * tq[m]);
            acc += (Q[2 * j + 1] + z[n] * (a[j] * st - b[j] * ct)) * e * sin(d[j] * tq[m]);
        }
        mu_out[m] += acc;               /* :475 */
    }
    free(S); free(phi); free(U); free(V); free(D); free(z); free(fg); free(Q); free(n0);
}
