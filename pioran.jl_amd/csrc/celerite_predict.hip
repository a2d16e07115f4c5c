// Posterior mean of the GP at new times (gfx950): `pred` of the reference (src/celerite_solver.jl:363-483) behind
// `predict` (:348-361) and `mean(::PosteriorGP, tau)` (src/scalable_GP.jl:64-72, 90-91).
//
//   z = K^-1 (y - mu)                      init_semi_separable! + solve_prec! (forward AND backward sweep)   :375-384
//   mu(tau) = sum_n z_n k(|tau - t_n|)     evaluated through the semi-separable structure                    :386-479
//
// The reference walks the merged (t, tau) sequence twice with running vectors Q.  Here the two recurrences are
// decoupled from the evaluation points so that the evaluation is embarrassingly parallel over tau:
//   1. celerite_wide_kernel<MODE 1> (celerite_wide.hip) factors and forward-solves, leaving W_n, D_n, z'_n in HBM;
//   2. predict_sweep_kernel, one wavefront per draw, lane = row: the backward sweep of solve_prec! (:145-155)
//        g <- phi_{n+1} o (g + U_{n+1} z_{n+1}) ;  z_n = z'_n / D_n - W_n' g          (one wave reduction per step)
//      which also gives  Qb_n = g + U_n z_n = sum_{k >= n} z_k U_k e^{-c (t_k - t_n)}  (the backward-pass Q of :437-470),
//      followed by the forward recurrence  Qf_n = phi_n o Qf_{n-1} + z_n V_n = sum_{k <= n} z_k V_k e^{-c (t_n - t_k)}
//      (the forward-pass Q of :397-424); both [N][R] per draw in HBM;
//   3. predict_eval_kernel, one thread per (draw, tau_m): n0 = #{t_n < tau} (searchsortedfirst - 1, :388),
//        mu_m = sum_r Qf[n0-1][r] e^{-c_r (tau - t_{n0-1})} U~_r(tau)   +   sum_r Qb[n0][r] e^{-c_r (t_{n0} - tau)} V_r(tau)
//      with U~_r(tau) = (a cos + b sin | a sin - b cos)(d tau), V_r(tau) = (cos | sin)(d tau)             (:412-413, :457-458).
// tau must be ascending for the reference; here any order works (each tau is independent).
#include "common.h"

#include <cmath>

namespace {

// sum over the 64 lanes of a wavefront, result in every lane
__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// The same sum, valid in lane 0 only, with the four stages inside a 16-lane row as DPP rotations (row_ror 8, 4, 2, 1: register moves,
// no LDS crossbar) and the three other rows fetched at once: 6 dependent exchange rounds become 4 short ones + 1.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_lane0(double x)
{
    x += dpp_move<0x128>(x);   // row_ror:8
    x += dpp_move<0x124>(x);   // row_ror:4
    x += dpp_move<0x122>(x);   // row_ror:2
    x += dpp_move<0x121>(x);   // row_ror:1   -> every lane holds its row's sum
    const double r1 = __shfl(x, 16), r2 = __shfl(x, 32), r3 = __shfl(x, 48);
    return (x + r1) + (r2 + r3);
}

struct RowDesc {
    double al, be;   // u = al v + be x
    int row;         // table row, < 0: lane idle
};

// Lane j of the wavefront owns rows j, j + 64 (and j + 128 with NR = 3: 129 .. 143 rows, round 4).
template <int NR>
__global__ void __launch_bounds__(64) predict_sweep_kernel(const ScanParams p, double* __restrict__ z, double* __restrict__ Qf,
                                                           double* __restrict__ Qb)
{
    const int64_t b = blockIdx.x, N = p.N;
    const int lane = threadIdx.x, R = p.R, Rp = R + 2, J = p.J;
    const int64_t rec = p.rec_stride;
    RowDesc rd[NR];
#pragma unroll
    for (int h = 0; h < NR; ++h) {
        const int r = lane + 64 * h;
        rd[h].row = r < R ? r : -1;
        rd[h].al = rd[h].be = 0.0;
        if (r < R) {
            const int rm = p.rowmap[r];
            const int term = rm & 0xfffff;
            const bool ks = (rm >> 30) & 1;
            const double a = p.A[b * J + term], bb = p.Bc[b * J + term];
            rd[h].al = a;
            rd[h].be = ks ? -bb : bb;
        }
    }
    const double* W = p.st_w + b * N * R;
    const double* Dd = p.st_d + b * N;
    const double* zf = p.st_z + b * N;
    double* zb = z + b * N;
    double* qf = Qf + b * N * R;
    double* qb = Qb + b * N * R;

    // What step n reads that does not depend on the recurrence (table record n, W_n, z'_n, D_n), fetched one step ahead
    // of its use: a lone wavefront per draw has nothing else to hide the latency behind.
    struct StepData {
        double v[NR], x[NR], ph[NR], w[NR], zf, Dn;
    };
    auto fetch = [&](int64_t n, StepData& sd) __attribute__((always_inline)) {
        const int64_t nn = n < 0 ? 0 : (n >= N ? N - 1 : n);
        const double* recn = p.tab + nn * rec;
#pragma unroll
        for (int h = 0; h < NR; ++h) {
            const int r = rd[h].row >= 0 ? rd[h].row : 0;
            sd.v[h] = recn[r];
            sd.x[h] = recn[Rp + r];
            sd.ph[h] = recn[2 * Rp + r];
            sd.w[h] = W[nn * R + r];
        }
        sd.zf = zf[nn];
        sd.Dn = Dd[nn];
    };

    // ---- backward sweep, :145-155 ----
    double g[NR] = {};
    double znext = 0.0;
    double unext[NR] = {}, phnext[NR];
#pragma unroll
    for (int h = 0; h < NR; ++h) phnext[h] = 1.0;   // U_{n+1}, phi between t_n and t_{n+1} (record n+1)
    StepData cur, nxt;
    fetch(N - 1, cur);
    for (int64_t n = N - 1; n >= 0; --n) {
        fetch(n - 1, nxt);
        double u[NR];
        double dot = 0.0;
#pragma unroll
        for (int h = 0; h < NR; ++h) {
            u[h] = 0.0;
            if (rd[h].row >= 0) {
                u[h] = rd[h].al * cur.v[h] + rd[h].be * cur.x[h];
                if (n < N - 1) g[h] = phnext[h] * (g[h] + unext[h] * znext);   // :149-150
                dot = fma(cur.w[h], g[h], dot);
            }
        }
        dot = wave_sum(dot);
        const double zn = cur.zf / cur.Dn - dot;                                // :146,151
#pragma unroll
        for (int h = 0; h < NR; ++h)
            if (rd[h].row >= 0) {
                qb[n * R + rd[h].row] = g[h] + u[h] * zn;
                unext[h] = u[h];
                phnext[h] = cur.ph[h];
            }
        if (lane == 0) zb[n] = zn;
        znext = zn;
        cur = nxt;
    }
    // ---- forward recurrence of the prediction, :397-404 (lane 0 wrote z: make it visible to the wavefront) ----
    __threadfence_block();
    double q[NR] = {};
    constexpr int FD = 4;   // z and the table record four steps ahead (no dependence on q)
    double zr[FD], vr[FD][NR], pr[FD][NR];
    auto fetch_f = [&](int64_t n, int slot) __attribute__((always_inline)) {
        const int64_t nn = n >= N ? N - 1 : n;
        const double* recn = p.tab + nn * rec;
        zr[slot] = zb[nn];
#pragma unroll
        for (int h = 0; h < NR; ++h) {
            const int r = rd[h].row >= 0 ? rd[h].row : 0;
            vr[slot][h] = recn[r];
            pr[slot][h] = recn[2 * Rp + r];
        }
    };
#pragma unroll
    for (int k = 0; k < FD; ++k) fetch_f(k, k);
    for (int64_t n0 = 0; n0 < N; n0 += FD) {
#pragma unroll
        for (int k = 0; k < FD; ++k) {
            const int64_t n = n0 + k;
            if (n < N) {
#pragma unroll
                for (int h = 0; h < NR; ++h)
                    if (rd[h].row >= 0) {
                        q[h] = fma(zr[k], vr[k][h], n > 0 ? pr[k][h] * q[h] : 0.0);
                        qf[n * R + rd[h].row] = q[h];
                    }
            }
            fetch_f(n + FD, k);
        }
    }
}

// one thread per (draw, tau_m)
__global__ void __launch_bounds__(256) predict_eval_kernel(const ScanParams p, const double* __restrict__ t,
                                                           const double* __restrict__ Qf, const double* __restrict__ Qb,
                                                           int64_t M, const double* __restrict__ tau, double* __restrict__ out)
{
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t b = blockIdx.y, N = p.N;
    if (m >= M) return;
    const int R = p.R, J = p.J;
    const double tm = tau[m];
    // n0 = number of t_n < tau  (searchsortedfirst(t, tau) - 1 in 1-based terms, :388)
    int64_t lo = 0, hi = N;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (t[mid] < tm) lo = mid + 1; else hi = mid;
    }
    const int64_t n0 = lo;
    const double* qf = n0 > 0 ? Qf + (b * N + (n0 - 1)) * R : nullptr;
    const double* qb = n0 < N ? Qb + (b * N + n0) * R : nullptr;
    const double dtf = n0 > 0 ? tm - t[n0 - 1] : 0.0;
    const double dtb = n0 < N ? t[n0] - tm : 0.0;
    double acc = 0.0;
    for (int r = 0; r < R; ++r) {
        const int rm = p.rowmap[r];
        const int term = rm & 0xfffff;
        const bool ks = (rm >> 30) & 1;
        const double a = p.A[b * J + term], bb = p.Bc[b * J + term], c = p.C[term], d = p.D[term];
        double s_, c_;
        sincos(d * tm, &s_, &c_);
        const double v = ks ? s_ : c_;                               // V_r(tau)
        const double u = ks ? a * s_ - bb * c_ : a * c_ + bb * s_;   // U~_r(tau)
        if (qf) acc = fma(qf[r] * exp(-c * dtf), u, acc);            // :412-413
        if (qb) acc = fma(qb[r] * exp(-c * dtb), v, acc);            // :457-458
    }
    out[b * M + m] = acc + (p.mu ? p.mu[b] : 0.0);
}

// ---- evaluation in two kernels (windowed path): what depends on tau alone is computed once, not once per draw ----------------
// predict_tau_kernel, one thread per tau_m: n0 (as above) and, per row, ef cos(d tau), ef sin(d tau), eb V_r(tau) with
// ef = e^{-c (tau - t_{n0-1})}, eb = e^{-c (t_{n0} - tau)} (zero where that side has no data point)  -> W [M][3][RP], RP = R rounded up to 16.
// blockIdx.y: draw of a batch with per-draw (c, d) (cd_stride = J, W at W + y W_stride); one shared table: gridDim.y = 1, strides 0.
__global__ void __launch_bounds__(256) predict_tau_kernel(const ScanParams p, const double* __restrict__ t, int64_t M,
                                                          const double* __restrict__ tau, int32_t* __restrict__ n0s, double* __restrict__ W,
                                                          int64_t cd_stride, int64_t W_stride)
{
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const double* Cc = p.C + (int64_t)blockIdx.y * cd_stride;
    const double* Dc = p.D + (int64_t)blockIdx.y * cd_stride;
    W += (int64_t)blockIdx.y * W_stride;
    const int64_t N = p.N;
    const int R = p.R, RP = (R + 15) & ~15;
    const double tm = tau[m];
    int64_t lo = 0, hi = N;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (t[mid] < tm) lo = mid + 1; else hi = mid;
    }
    n0s[m] = (int32_t)lo;
    const double dtf = lo > 0 ? tm - t[lo - 1] : 0.0, dtb = lo < N ? t[lo] - tm : 0.0;
    double* w = W + m * 3 * RP;
    for (int r = 0; r < RP; ++r) {
        double wc = 0.0, wsn = 0.0, wv = 0.0;
        if (r < R) {
            const int rm = p.rowmap[r];
            const int term = rm & 0xfffff;
            const bool ks = (rm >> 30) & 1;
            const double c = Cc[term], d = Dc[term];
            double s_, c_;
            sincos(d * tm, &s_, &c_);
            const double ef = lo > 0 ? exp(-c * dtf) : 0.0, eb = lo < N ? exp(-c * dtb) : 0.0;
            wc = ef * c_; wsn = ef * s_; wv = eb * (ks ? s_ : c_);
        }
        w[r] = wc; w[RP + r] = wsn; w[2 * RP + r] = wv;
    }
}

// predict_eval16_kernel: sixteen lanes per (draw, tau), lane = row (mod 16): the two Q rows and the three W rows of a tau are read
// as contiguous runs (the one-thread-per-tau kernel above reads them with a stride of R doubles between lanes).
//   mu_m = sum_r Qf[n0-1][r] (al_r Wc + be_r Ws) + Qb[n0][r] Wv ,   (al, be) = (a, b) on a cos row, (-b, a) on a sin row
constexpr int EVT = 8;   // taus per 16-lane group
__global__ void __launch_bounds__(256) predict_eval16_kernel(const ScanParams p, const double* __restrict__ Qf, const double* __restrict__ Qb,
                                                             int64_t M, const int32_t* __restrict__ n0s, const double* __restrict__ W,
                                                             double* __restrict__ out, int64_t W_stride)
{
    const int64_t b = blockIdx.y, N = p.N;
    W += b * W_stride;
    const int R = p.R, RP = (R + 15) & ~15, J = p.J, l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
    constexpr int MAXI = 8;   // R <= 128
    double al[MAXI], be[MAXI];
    const int ni = RP >> 4;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        al[i] = be[i] = 0.0;
        const int r = l16 + 16 * i;
        if (i < ni && r < R) {
            const int rm = p.rowmap[r];
            const int term = rm & 0xfffff;
            const bool ks = (rm >> 30) & 1;
            const double a = p.A[b * J + term], bb = p.Bc[b * J + term];
            al[i] = ks ? -bb : a;
            be[i] = ks ? a : bb;
        }
    }
    const double mu = p.mu ? p.mu[b] : 0.0;
    const double* qfb = Qf + b * N * R;
    const double* qbb = Qb + b * N * R;
    const int64_t m0 = ((int64_t)blockIdx.x * 16 + grp) * EVT;
#pragma unroll 2
    for (int e = 0; e < EVT; ++e) {
        const int64_t m = m0 + e;
        if (m >= M) break;            // (uniform within the 16-lane group)
        const int64_t n0 = n0s[m];
        const double* w = W + m * 3 * RP;
        const double* qf = qfb + (n0 > 0 ? n0 - 1 : 0) * R;
        const double* qb = qbb + (n0 < N ? n0 : N - 1) * R;
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int r = l16 + 16 * i;
            if (i < ni && r < R) {
                acc = fma(qf[r], fma(al[i], w[r], be[i] * w[RP + r]), acc);   // (W is zero on a side without data: the clamped row is harmless)
                acc = fma(qb[r], w[2 * RP + r], acc);
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (l16 == 0) out[b * M + m] = acc + mu;
    }
}

// ---- the two Q recurrences from a KNOWN z (windowed path below): no reduction, so the series is cut into segments --------------
// z = K^-1 (y - mu) comes out of the windowed reverse mode as -dL/dy (celerite_block.hip).  Then, per draw and row,
//   Qf_n = phi_n Qf_{n-1} + z_n V_n        Qb_n = U_n z_n + phi_{n+1} Qb_{n+1}
// are first-order linear recurrences: segment s (QSEG steps) first computes its own sum with a zero carry (E_s) and the product of
// its phis (P_s) [pass 0], one wavefront per (draw, direction) chains the carries over the segments [q_carry_kernel], and pass 1
// repeats the segment from its carry and stores every step.  One workgroup of 64 lanes = one (segment, draw, direction); lane = row.
constexpr int QSEG = 128;

// PD: (c, d) per draw (no shared table: cos / sin / exp of a step are computed where they are used; t: the time stamps)
template <int PASS, bool PD>
__global__ void __launch_bounds__(64) q_segment_kernel(const ScanParams p, const double* __restrict__ gy, double* __restrict__ Qf,
                                                       double* __restrict__ Qb, double* __restrict__ EP, const double* __restrict__ Cin,
                                                       const double* __restrict__ t)
{
    const int64_t N = p.N, s = blockIdx.x, b = blockIdx.y, nseg = gridDim.x;
    const int dir = blockIdx.z, lane = threadIdx.x, R = p.R, Rp = R + 2, J = p.J;
    if (lane >= R) return;
    const int64_t rec = PD ? 0 : p.rec_stride;
    const int rm0 = p.rowmap[lane];
    [[maybe_unused]] const bool sinrow = (rm0 >> 30) & 1;
    [[maybe_unused]] const double cpd = PD ? p.C[b * J + (rm0 & 0xfffff)] : 0.0, dpd = PD ? p.D[b * J + (rm0 & 0xfffff)] : 0.0;
    // (v, x, phi) of step n for this lane's row
    auto step = [&](int64_t n, double& v, double& x, double& ph) __attribute__((always_inline)) {
        if constexpr (PD) {
            double sn, cs;
            sincos(dpd * t[n], &sn, &cs);
            v = sinrow ? sn : cs; x = sinrow ? cs : sn;
            ph = n > 0 ? exp(-cpd * (t[n] - t[n - 1])) : 0.0;
        } else {
            const double* r = p.tab + lane + n * rec;
            v = r[0]; x = r[Rp]; ph = r[2 * Rp];
        }
    };
    const int64_t n_lo = s * QSEG, n_hi = n_lo + QSEG < N ? n_lo + QSEG : N;
    const double* zz = gy + b * N;
    const int64_t slot = ((b * 2 + dir) * nseg + s) * R + lane;
    if (dir == 0) {
        double q = PASS ? Cin[slot] : 0.0, prod = 1.0;
        double* qf = Qf + b * N * R + lane;
#pragma unroll 4
        for (int64_t n = n_lo; n < n_hi; ++n) {
            double v, x, ph;
            step(n, v, x, ph);
            if (n == 0) ph = 0.0;
            q = fma(-zz[n], v, ph * q);                              // z_n = -dL/dy_n
            if (PASS) qf[n * R] = q; else prod *= ph;
        }
        if (!PASS) { EP[2 * slot] = q; EP[2 * slot + 1] = prod; }
    } else {
        const int term = rm0 & 0xfffff;
        const double al = p.A[b * J + term], be = sinrow ? -p.Bc[b * J + term] : p.Bc[b * J + term];
        double q = PASS ? Cin[slot] : 0.0, prod = 1.0;
        double* qb = Qb + b * N * R + lane;
        double vn, xn, phn;                                          // the step after n: its phi links n to n + 1
        if (n_hi < N) step(n_hi, vn, xn, phn); else phn = 0.0;
#pragma unroll 4
        for (int64_t n = n_hi - 1; n >= n_lo; --n) {
            double v, x, phc;
            step(n, v, x, phc);
            const double u = al * v + be * x;
            const double ph = phn;                                   // phi between t_n and t_{n+1} (record n+1)
            phn = phc;
            q = fma(-zz[n], u, ph * q);
            if (PASS) qb[n * R] = q; else prod *= ph;
        }
        if (!PASS) { EP[2 * slot] = q; EP[2 * slot + 1] = prod; }
    }
}

// ---- ascending tau: the second pass and the evaluation in one kernel, Qf / Qb never written ---------------------------------------
// With tau sorted (the reference requires it) n0 is non-decreasing, so the evaluation times that need Qf of step n (n0 - 1 = n) or Qb of
// step n (n0 = n) are a contiguous run of indices: the wavefront that walks a segment (as pass 1 would, from the segment's carry) forms
//   sum_r Qf_n[r] (al_r Wc + be_r Ws)   resp.   sum_r Qb_n[r] Wv      (one wave reduction per evaluation time)
// right where q is in its registers and writes it to part[dir][b][m] (zero where a side has no data point: predict_part_init_kernel);
// predict_part_sum_kernel adds the two parts and mu_b (no atomics: the result does not depend on the order the blocks finish in).
// Saves the 2 N R doubles per draw that pass 1 writes and the evaluation reads back.
__global__ void __launch_bounds__(256) predict_part_init_kernel(int64_t n, double* __restrict__ part)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) part[i] = 0.0;
}
__global__ void __launch_bounds__(256) predict_part_sum_kernel(const ScanParams p, int64_t M, const double* __restrict__ part, double* __restrict__ out)
{
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t b = blockIdx.y, BM = (int64_t)gridDim.y * M;
    if (m < M) out[b * M + m] = (part[b * M + m] + part[BM + b * M + m]) + (p.mu ? p.mu[b] : 0.0);
}

template <bool PD>
__global__ void __launch_bounds__(64) q_eval_fused_kernel(const ScanParams p, const double* __restrict__ gy, const double* __restrict__ Cin,
                                                          const double* __restrict__ t, int64_t M, const int32_t* __restrict__ n0s,
                                                          const double* __restrict__ W, int64_t W_stride, double* __restrict__ out)
{
    const int64_t N = p.N, s = blockIdx.x, b = blockIdx.y, nseg = gridDim.x;
    const int dir = blockIdx.z, lane = threadIdx.x, R = p.R, Rp = R + 2, J = p.J, RP = (R + 15) & ~15;
    const bool live = lane < R;                        // (idle lanes stay in: the wave reductions below want all 64)
    const int rl = live ? lane : 0;
    const int64_t rec = PD ? 0 : p.rec_stride;
    const int rm0 = p.rowmap[rl];
    const bool sinrow = (rm0 >> 30) & 1;
    const int term = rm0 & 0xfffff;
    [[maybe_unused]] const double cpd = PD ? p.C[b * J + term] : 0.0, dpd = PD ? p.D[b * J + term] : 0.0;
    auto step = [&](int64_t n, double& v, double& x, double& ph) __attribute__((always_inline)) {
        if constexpr (PD) {
            double sn, cs;
            sincos(dpd * t[n], &sn, &cs);
            v = sinrow ? sn : cs; x = sinrow ? cs : sn;
            ph = n > 0 ? exp(-cpd * (t[n] - t[n - 1])) : 0.0;
        } else {
            const double* r = p.tab + rl + n * rec;
            v = r[0]; x = r[Rp]; ph = r[2 * Rp];
        }
    };
    auto lower = [&](int64_t key) {                    // first m with n0s[m] >= key
        int64_t lo = 0, hi = M;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (n0s[mid] < key) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    const int64_t n_lo = s * QSEG, n_hi = n_lo + QSEG < N ? n_lo + QSEG : N;
    const double* zz = gy + b * N;
    const int64_t slot = ((b * 2 + dir) * nseg + s) * R + rl;
    const double a_ = p.A[b * J + term], b_ = p.Bc[b * J + term];
    const double* Wb = W + b * W_stride + rl;
    double* ob = out + ((int64_t)dir * gridDim.y + b) * M;      // out: part[2][B][M]
    double q = live ? Cin[slot] : 0.0;
    if (dir == 0) {
        const double ae = live ? (sinrow ? -b_ : a_) : 0.0, bee = live ? (sinrow ? a_ : b_) : 0.0;   // U~_r(tau) = ae cos + bee sin
        int64_t mp = lower(n_lo + 1);                  // evaluation times with n0 - 1 in [n_lo, n_hi)
        const int64_t m_hi = lower(n_hi + 1);
        int64_t nn = mp < m_hi ? n0s[mp] - 1 : N;      // the step the next evaluation time waits for
        for (int64_t n = n_lo; n < n_hi; ++n) {
            double v, x, ph;
            step(n, v, x, ph);
            if (n == 0) ph = 0.0;
            q = fma(-zz[n], v, ph * q);
            while (nn == n) {
                const double* w = Wb + mp * 3 * RP;
                const double sum = wave_sum_lane0(q * fma(ae, w[0], bee * w[RP]));
                if (lane == 0) ob[mp] = sum;
                ++mp;
                nn = mp < m_hi ? n0s[mp] - 1 : N;
            }
        }
    } else {
        const double al = live ? a_ : 0.0, be = live ? (sinrow ? -b_ : b_) : 0.0;
        const int64_t m_lo = lower(n_lo);              // evaluation times with n0 in [n_lo, n_hi)
        int64_t mp = lower(n_hi) - 1;
        int64_t nn = mp >= m_lo ? n0s[mp] : -1;
        double vn, xn, phn;
        if (n_hi < N) step(n_hi, vn, xn, phn); else phn = 0.0;
        for (int64_t n = n_hi - 1; n >= n_lo; --n) {
            double v, x, phc;
            step(n, v, x, phc);
            const double u = al * v + be * x;
            const double ph = phn;
            phn = phc;
            q = fma(-zz[n], u, ph * q);
            while (nn == n) {
                const double* w = Wb + mp * 3 * RP;
                const double sum = wave_sum_lane0(live ? q * w[2 * RP] : 0.0);
                if (lane == 0) ob[mp] = sum;
                --mp;
                nn = mp >= m_lo ? n0s[mp] : -1;
            }
        }
    }
}

// carries into the segments: forward C_0 = 0, C_{s+1} = E_s + P_s C_s; backward the same from the last segment down
__global__ void __launch_bounds__(64) q_carry_kernel(int R, int64_t nseg, const double* __restrict__ EP, double* __restrict__ Cin)
{
    const int64_t b = blockIdx.x;
    const int dir = blockIdx.y, lane = threadIdx.x;
    if (lane >= R) return;
    const int64_t base = (b * 2 + dir) * nseg;
    double c = 0.0;
    for (int64_t k = 0; k < nseg; ++k) {
        const int64_t s = dir == 0 ? k : nseg - 1 - k;
        const int64_t slot = (base + s) * R + lane;
        Cin[slot] = c;
        c = fma(EP[2 * slot + 1], c, EP[2 * slot]);
    }
}

}  // namespace

size_t pioran_predict_q_workspace_doubles(int64_t B, int64_t N, int32_t R)
{
    // Qf, Qb: [B][N][R]; -z: [B][N]; E, P, C per (draw, direction, segment, row)
    const size_t nseg = (size_t)((N + QSEG - 1) / QSEG);
    return (size_t)B * (size_t)N * (2 * (size_t)R + 1) + (size_t)B * 2 * nseg * (size_t)R * 3;
}
// tau-only factors of the evaluation: W [M][3][RP] + n0 [M] (as int32, rounded up to whole doubles)
// (ntab: 1 for a shared (c, d), the number of draws of a launch with per-draw (c, d))
size_t pioran_predict_tau_workspace_doubles(int64_t M, int32_t R, int64_t ntab)
{
    return (size_t)ntab * (size_t)M * 3 * (size_t)((R + 15) & ~15) + (size_t)(M + 1) / 2 + 1;
}

// Windowed path: gy [B][N] = dL/dy of the windowed reverse mode (pioran_launch_block_grad with p.g_y) already on the stream.
// work: pioran_predict_q_workspace_doubles doubles whose FIRST B N hold gy; tau_work: pioran_predict_tau_workspace_doubles.
// R <= 64 (the windowed kernels stop at 63 rows).
int pioran_launch_predict_from_gy(ScanParams p, double* work, double* tau_work, const double* t, int64_t M, const double* tau, double* mean_out,
                                  hipStream_t stream, int cd_per_draw, int tau_sorted)
{
    // cd_per_draw: p.C, p.D are [B][J], no shared table (p.tab unused), tau_work holds B sets of factors (n0 after the last one)
    if ((!p.tab && !cd_per_draw) || p.npd_rows != 0 || p.R > 64 || M < 0 || p.N > 0x7fffffff) return PIORAN_ERR_UNSUPPORTED;
    const size_t BN = (size_t)p.B * (size_t)p.N;
    const int64_t nseg = (p.N + QSEG - 1) / QSEG;
    const double* gy = work;
    double* Qf = work + BN;
    double* Qb = Qf + BN * p.R;
    double* EP = Qb + BN * p.R;
    double* Cin = EP + (size_t)p.B * 2 * (size_t)nseg * (size_t)p.R * 2;
    const dim3 grid((unsigned)nseg, (unsigned)p.B, 2);
    if (cd_per_draw) hipLaunchKernelGGL((q_segment_kernel<0, true>), grid, dim3(64), 0, stream, p, gy, Qf, Qb, EP, Cin, t);
    else hipLaunchKernelGGL((q_segment_kernel<0, false>), grid, dim3(64), 0, stream, p, gy, Qf, Qb, EP, Cin, t);
    hipLaunchKernelGGL(q_carry_kernel, dim3((unsigned)p.B, 2), dim3(64), 0, stream, (int)p.R, nseg, EP, Cin);
    // tau_work: pioran_predict_tau_workspace_doubles; the tau-only factors are computed here (once per chunk of draws: 10 us)
    double* W = tau_work;
    const int64_t W_stride = cd_per_draw ? (int64_t)M * 3 * ((p.R + 15) & ~15) : 0;
    const int64_t ntab = cd_per_draw ? p.B : 1;
    int32_t* n0s = (int32_t*)(tau_work + (size_t)ntab * (size_t)M * 3 * (size_t)((p.R + 15) & ~15));
    if (M > 0)
        hipLaunchKernelGGL(predict_tau_kernel, dim3((unsigned)((M + 255) / 256), (unsigned)ntab), dim3(256), 0, stream, p, t, M, tau, n0s, W,
                           cd_per_draw ? (int64_t)p.J : (int64_t)0, W_stride);
    if (M > 0 && tau_sorted && M <= p.N * (int64_t)p.R) {
        // ascending tau: second pass and evaluation fused, Qf / Qb never written (their storage holds the two parts of the result)
        double* part = Qf;
        const int64_t np = 2 * (int64_t)p.B * M;
        hipLaunchKernelGGL(predict_part_init_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, stream, np, part);
        if (cd_per_draw) hipLaunchKernelGGL((q_eval_fused_kernel<true>), grid, dim3(64), 0, stream, p, gy, Cin, t, M, n0s, W, W_stride, part);
        else hipLaunchKernelGGL((q_eval_fused_kernel<false>), grid, dim3(64), 0, stream, p, gy, Cin, t, M, n0s, W, W_stride, part);
        hipLaunchKernelGGL(predict_part_sum_kernel, dim3((unsigned)((M + 255) / 256), (unsigned)p.B), dim3(256), 0, stream, p, M, part, mean_out);
        return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
    }
    if (cd_per_draw) hipLaunchKernelGGL((q_segment_kernel<1, true>), grid, dim3(64), 0, stream, p, gy, Qf, Qb, EP, Cin, t);
    else hipLaunchKernelGGL((q_segment_kernel<1, false>), grid, dim3(64), 0, stream, p, gy, Qf, Qb, EP, Cin, t);
    if (M > 0)
        hipLaunchKernelGGL(predict_eval16_kernel, dim3((unsigned)((M + 16 * EVT - 1) / (16 * EVT)), (unsigned)p.B), dim3(256), 0, stream, p, Qf,
                           Qb, M, n0s, W, mean_out, W_stride);
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}

size_t pioran_predict_workspace_doubles(int64_t B, int64_t N, int32_t R)
{
    // W, Qf, Qb: [B][N][R]; D, z', z: [B][N]
    return (size_t)B * (size_t)N * (3 * (size_t)R + 3);
}

// p: a shared-table launch description (tab, rowmap, A, Bc, C, D (shared, [J]), mu, nu, y/s2 or Y/S2, out, status).
// work: pioran_predict_workspace_doubles(B, N, R) doubles.  t: device [N].  tau: device [M].  mean_out: device [B][M].
int pioran_launch_predict(ScanParams p, double* work, const double* t, int64_t M, const double* tau, double* mean_out,
                          hipStream_t stream)
{
    if (!p.tab || p.npd_rows != 0 || p.R > 192 || M < 0) return PIORAN_ERR_UNSUPPORTED;
    const size_t BN = (size_t)p.B * (size_t)p.N;
    double* W = work;
    double* Qf = W + BN * p.R;
    double* Qb = Qf + BN * p.R;
    double* Dd = Qb + BN * p.R;
    double* zf = Dd + BN;
    double* z = zf + BN;
    p.st_w = W; p.st_d = Dd; p.st_z = zf;
    int rc = pioran_launch_scan_wide_store(p, stream);
    if (rc) return rc;
    if (p.R > 128) hipLaunchKernelGGL(predict_sweep_kernel<3>, dim3((unsigned)p.B), dim3(64), 0, stream, p, z, Qf, Qb);
    else hipLaunchKernelGGL(predict_sweep_kernel<2>, dim3((unsigned)p.B), dim3(64), 0, stream, p, z, Qf, Qb);
    if (M > 0)
        hipLaunchKernelGGL(predict_eval_kernel, dim3((unsigned)((M + 255) / 256), (unsigned)p.B), dim3(256), 0, stream, p, t, Qf,
                           Qb, M, tau, mean_out);
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}
