// Latency layout of the celerite scan for SMALL batches (gfx950): one draw per workgroup of four wavefronts.
//
// Same recurrence as celerite_scan.hip (init_semi_separable! + the forward half of solve_prec!,
// src/celerite_solver.jl:12-100,115-142, logl :312-334):
//   S <- (phi phi') o (S + D w w') ;  q = S u ;  D_n = sum(a) + sigma2_n - u'q ;  w = (v - q) / D_n
// with y carried as one more row (u = 0, v = y_n - mu, phi = 1), so z_n = v_y - q_y needs no recurrence of its own.
//
// celerite_scan.hip is built for throughput: 16..64 lanes per draw, 36..75 entries of S per lane, thousands of
// draws in flight to hide every latency.  With a handful of draws (the scalar `logl` drop-in, MCMC walkers) a time
// step is one long dependent chain on a single wavefront.  Here a draw gets 256 lanes arranged as 16 x 16:
//   DPP row g (16 of them over 4 waves) = ROW block g of S;  lane l inside the row = COLUMN block l;
//   RPL x RPL entries of S per lane (RPL <= 6: up to 95 rows + the y row; the throughput layouts stop at 79).
// Consequences:
//   * q = S u is a sum over the 16 lanes of a DPP row: four DPP butterfly stages, no LDS, no ds_bpermute;
//   * u and phi of the lane's column block are table data: the lane loads them itself (no broadcast);
//   * the only exchange per step is through LDS, ONE barrier: each DPP row publishes (v - q) of its RPL rows and its
//     share of u'q; every lane then reads the 16 shares (fixed summation order: all waves get the same bits),
//     forms D_n and its reciprocal, and the RPL values w_k = (v - q)_k / D_n of its column block.
//     Buffers alternate with the step parity, which makes the single barrier sufficient.
//   * the table records go HBM -> registers -> LDS: each thread fetches ONE double of the step record (two when it is
//     longer than 256 doubles), four records ahead, and the lanes read their 6 RPL + 2 values from LDS.  A lone
//     workgroup keeps nothing else in flight and the 10 MB table is not L2 resident at small B, so every record is
//     an HBM-latency miss that only depth can hide; and 6 RPL + 2 broadcast-heavy buffer loads per lane and step
//     (the first version) saturated the texture path instead.
// Shared-table launches only ((c, d) common to the batch, possibly with a few per-draw rows: mixed mode); launches
// with fully per-draw (c, d) stay on celerite_scan.hip.
//
// The same kernel carries three more modes (template MODE) for the callers either side of the likelihood
// (SURVEY.md 8(f)): 1 = also store the factor (W_n, D_n, forward-solved z_n) for the posterior mean
// (celerite_predict.hip); 2 = simulate (the extra row applies L instead of L^-1); 3 = also store S_n, v - q and D_n for
// celerite_adjoint_kernel below, which walks the recurrence backwards and returns the gradient of log L.
#include "common.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>

// Diagnostic hooks: compiled out in the product; tools/wide_probe.hip defines them to s_memtime accumulators.
#ifndef PIORAN_WSTAMP
#define PIORAN_WSTAMP(i)
#define PIORAN_WSTAMP_DECL
#define PIORAN_WSTAMP_FLUSH
#endif

namespace {

template <int I>
using ic = std::integral_constant<int, I>;
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (B < E) {
        f(ic<B>{});
        static_for<B + 1, E>(f);
    }
}

template <int CTRL>
__device__ __forceinline__ double dpp_perm(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// sum over the 16 lanes of a DPP row; every lane gets the bit-identical total
__device__ __forceinline__ double row16_sum(double x)
{
    x += dpp_perm<0xB1>(x);   // quad_perm [1,0,3,2]
    x += dpp_perm<0x4E>(x);   // quad_perm [2,3,0,1]
    x += dpp_perm<0x141>(x);  // row_half_mirror
    x += dpp_perm<0x140>(x);  // row_mirror
    return x;
}

typedef double d2 __attribute__((ext_vector_type(2)));

// 1/x to fp64 accuracy: v_rcp_f64 seed + two Newton steps (same sequence as celerite_scan.hip)
__device__ __forceinline__ double recip_f64(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

template <int RPL>
struct WideIn {   // what one time step reads: (v, x, phi) of the lane's row block and of its column block, y_n, sigma2_n
    double rv[RPL], rx[RPL], rp[RPL], cv[RPL], cx[RPL], cp[RPL], y, s2;
};

template <int RPL>
struct Slots {    // per-lane description of RPL consecutive row slots
    int ov[RPL], ox[RPL], op[RPL];   // positions (in doubles) of v, x, phi inside the staged step record
    double al[RPL], be[RPL];         // u = al v + be x
};

// Staged record of one time step (LDS), L = RS + 3 npd_rows doubles: the shared part of the table record
// [v x Rp | x x Rp | phi x Rp | y_n, sigma2_n] (RS = 3 Rp + 2) followed by THIS draw's per-draw rows
// [row][v, x, phi] (mixed mode), which sit at RS + 3 (b npd_rows + row) in the table record.
template <int RPL>
__device__ __forceinline__ void describe_slots(const ScanParams& p, int64_t b, int block, Slots<RPL>& s)
{
    const int R = p.R, Rp = R + 2, J = p.J, RS = 3 * Rp + 2;
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        const int slot = block * RPL + i;
        int trow;
        s.al[i] = 0.0;
        s.be[i] = 0.0;
        int pdrow = -1;
        if (slot < R) {
            const int rm = p.rowmap[slot];
            const int term = rm & 0xfffff;
            const bool ks = (rm >> 30) & 1;
            if ((rm >> 29) & 1) pdrow = (rm >> 20) & 0x1ff;
            const double a = p.A[b * J + term], bb = p.Bc[b * J + term];
            // cos row: v = co, x = si, u = a co + b si ; sin row: v = si, x = co, u = a si - b co   (:59-63)
            trow = slot;
            s.al[i] = a;
            s.be[i] = ks ? -bb : bb;
        } else if (slot == 16 * RPL - 1) {
            trow = R + 1;   // the y row: u = 0, v = y_n - mu, phi = 1
        } else {
            trow = R;       // inert padding row: u = 0, v = 1, phi = 1
        }
        if (pdrow >= 0) {
            s.ov[i] = RS + pdrow * 3;
            s.ox[i] = s.ov[i] + 1;
            s.op[i] = s.ov[i] + 2;
        } else {
            s.ov[i] = trow;
            s.ox[i] = trow + Rp;
            s.op[i] = trow + 2 * Rp;
        }
    }
}

constexpr int kWideMaxRecord = 512;   // staged doubles per step: two per thread

// MODE 0: log-likelihood.  MODE 1: the same, and the factor goes to HBM (W_n, D_n, forward-solved z_n per step) for the
// backward sweep of the prediction path (celerite_predict.hip).  MODE 2: simulation — the extra row carries
// f <- phi o (f + W_{n-1} x_{n-1}) instead of the forward solve and emits y_n = x_n + u_n'f, x_n = sqrt(D_n) q_n
// (sim, src/celerite_solver.jl:515-549); the noise q takes the place of y in the staged record.  MODE 3: log-likelihood,
// and v - q of all 16 RPL slots and D_n of every step plus S_n (lane layout) at the checkpoints n = k * ckpt_every go to HBM
// for the reverse pass below.
template <int RPL, int MODE = 0>
__global__ void __launch_bounds__(256, 1) celerite_wide_kernel(const ScanParams p)
{
    constexpr int YS = RPL - 1;                       // slot of the y row inside row / column block 15
    constexpr int DG = 4;                             // records in flight from HBM per thread
    const int tid = threadIdx.x;
    const int g = tid >> 4;                           // row block (DPP row of the draw)
    const int l = tid & 15;                           // column block
    const int64_t b = blockIdx.x;                     // grid = batch
    const int64_t N = p.N;
    const int J = p.J, Rp = p.R + 2, RS = 3 * Rp + 2;
    const int L = RS + 3 * p.npd_rows;                // staged doubles per step (<= kWideMaxRecord, checked on the host)

    __shared__ double sh_rec[2][kWideMaxRecord];      // staged records, by step parity
    __shared__ double sh_num[2][16 * RPL];            // (v - q) of every row = D_n w
    __shared__ double sh_uq[2][16];                   // u'q share of every row block

    Slots<RPL> rs_, cs_;
    describe_slots<RPL>(p, b, g, rs_);
    describe_slots<RPL>(p, b, l, cs_);
    double suma = 0.0;  // :21
    for (int j = 0; j < J; ++j) suma += p.A[b * J + j];
    const double mu = p.mu ? p.mu[b] : 0.0;
    const double nu = p.nu ? p.nu[b] : 1.0;
    const bool has_nu = p.nu != nullptr;
    const bool yrow = g == 15;                        // this lane's row slot YS is the y row
    const bool ycol = l == 15;                        // ... column slot YS

    // ---- HBM -> registers -> LDS staging: thread `tid` owns elements tid and tid + 256 of every staged record -------
    // element e < RS comes from table element e, e >= RS from this draw's per-draw rows; with a per-draw series
    // (Y, S2) the two elements y_n, sigma2_n come from there instead.  One 8-byte load per thread and step (two when
    // L > 256) instead of 6 RPL + 2 broadcast-heavy loads per lane: the texture path, not HBM, was the bottleneck.
    const bool two = L > 256;                         // uniform
    const double* src[2];
    int64_t stride[2], last[2];                       // address = src + min(n, last) * stride
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int e = tid + 256 * h;
        src[h] = p.tab + (e < RS ? e : e + (int64_t)b * p.npd_rows * 3);
        stride[h] = p.rec_stride;
        last[h] = N;                                  // the table holds N + 1 records
        if (p.Y && (e == 3 * Rp || e == 3 * Rp + 1)) {
            src[h] = (e == 3 * Rp ? p.Y : p.S2) + b * N;
            stride[h] = 1;
            last[h] = N - 1;
        }
        if (MODE == 2 && e == 3 * Rp) {   // the noise rides where y would
            src[h] = p.noise + b * N;
            stride[h] = 1;
            last[h] = N - 1;
        }
        if (e >= L) { src[h] = p.tab; stride[h] = 0; }   // idle thread: harmless re-read of element 0
    }
    double gv[DG][2];                                 // record m is (or will be) in gv[m % DG]
    auto fetch = [&](int64_t m, double (&dst)[2]) __attribute__((always_inline)) {
        dst[0] = src[0][(m < last[0] ? m : last[0]) * stride[0]];
        if (two) dst[1] = src[1][(m < last[1] ? m : last[1]) * stride[1]];
    };
    auto stage = [&](int par, const double (&v)[2]) __attribute__((always_inline)) {
        sh_rec[par][tid] = v[0];
        if (two) sh_rec[par][tid + 256] = v[1];
    };
    auto unstage = [&](int par, WideIn<RPL>& in) __attribute__((always_inline)) {
        const double* r = sh_rec[par];
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            in.rv[i] = r[rs_.ov[i]];
            in.rx[i] = r[rs_.ox[i]];
            in.rp[i] = r[rs_.op[i]];
            in.cv[i] = r[cs_.ov[i]];
            in.cx[i] = r[cs_.ox[i]];
            in.cp[i] = r[cs_.op[i]];
        }
        in.y = r[3 * Rp];
        in.s2 = r[3 * Rp + 1];
    };

#pragma unroll
    for (int m = 0; m < DG; ++m) fetch(m, gv[m]);
    WideIn<RPL> cur[2];                               // record m is consumed from cur[m & 1]
    stage(0, gv[0]);
    fetch(DG, gv[0]);
    __syncthreads();
    unstage(0, cur[0]);

    // ---- first row, :27-42 and :126-128 ----
    double S[RPL][RPL];   // [own row][own column]
#pragma unroll
    for (int i = 0; i < RPL; ++i)
#pragma unroll
        for (int c = 0; c < RPL; ++c) S[i][c] = 0.0;
    double num[RPL];      // (v - q) of this lane's rows at the last step = D_n W_n, the `dn` of :73
    double wc[RPL];       // W_n of this lane's columns
    double Dn = suma + (has_nu ? nu * cur[0].s2 : cur[0].s2);
    double rD = recip_f64(Dn);
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        const double v0r = (yrow && i == YS) ? cur[0].y - mu : cur[0].rv[i];   // z_1 = y_1      :128
        const double v0c = (ycol && i == YS) ? cur[0].y - mu : cur[0].cv[i];
        num[i] = v0r;
        wc[i] = v0c * rD;
    }
    double Pm = Dn;       // running product of |D| (sign of D_1 kept: log of a negative D_1 is NaN, :126)
    int Pe = 0;
    {
        int ex;
        Pm = frexp(Pm, &ex);
        Pe += ex;
    }
    double quad = num[YS] * num[YS] * rD;             // meaningful in the y-row lanes only
    bool nonpd = !(Dn > 0.0);
    // per-step outputs of MODE 1 / MODE 2 (one lane per DPP row stores)
    auto emit = [&](int64_t n, double yn) __attribute__((always_inline)) {
        if constexpr (MODE == 1) {
            if (l == 0) {
#pragma unroll
                for (int i = 0; i < RPL; ++i)
                    if (g * RPL + i < p.R) p.st_w[(b * N + n) * p.R + g * RPL + i] = num[i] * rD;
                if (yrow) {
                    p.st_d[b * N + n] = Dn;
                    p.st_z[b * N + n] = num[YS];
                }
            }
        }
        if constexpr (MODE == 3) {
            if (l == 0) {
#pragma unroll
                for (int i = 0; i < RPL; ++i) p.st_w[(b * N + n) * (16 * RPL) + g * RPL + i] = num[i];
                if (yrow) p.st_d[b * N + n] = Dn;
            }
        }
        if constexpr (MODE == 2) {
            const double x = sqrt(Dn) * yn;            // x_n = sqrt(D_n) q_n      :539,546
            if (yrow) {
                if (l == 0) p.ysim[b * N + n] = x + ((yn - mu) - num[YS]);   // x_n + u_n'f  (this row's v is q_n - mu, num = v - u'f)
                num[YS] = x;                           // what the extra row adds next step: W_{n} x_n   :543
            }
        }
    };
    emit(0, cur[0].y);
    // record 1 for the first step of the loop
    stage(1, gv[1 % DG]);
    fetch(DG + 1, gv[1 % DG]);
    __syncthreads();
    unstage(1, cur[1]);

    PIORAN_WSTAMP_DECL
    // one time step n: consumes `in` (record n); stages record n + 1 (fetched DG - 1 steps ago) for the next step
    // and refills its register slot with record n + 1 + DG
    auto do_step = [&](int64_t n, WideIn<RPL>& in, WideIn<RPL>& nxt, double (&gslot)[2]) __attribute__((always_inline)) {
        PIORAN_WSTAMP(0);
        double ur[RPL], uc[RPL], qt[RPL];
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            ur[i] = rs_.al[i] * in.rv[i] + rs_.be[i] * in.rx[i];
            uc[i] = cs_.al[i] * in.cv[i] + cs_.be[i] * in.cx[i];
            qt[i] = 0.0;
        }
        if (yrow) in.rv[YS] = in.y - mu;
        // ---- S update + this lane's share of q = S u ----
#pragma unroll
        for (int c = 0; c < RPL; ++c)
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const double m = fma(num[i], wc[c], S[i][c]);     // S + dn * V[k,n-1]          :78
                const double sn = (in.rp[i] * in.cp[c]) * m;      // phi_j phi_k ( ... )        :78,85
                S[i][c] = sn;
                qt[i] = fma(sn, uc[c], qt[i]);                    // (S u)_j                    :80-82,86-89
            }
        if constexpr (MODE == 3) {   // checkpoint of S_n, lane layout: each lane's block contiguous, padded to an even
                                     // number of doubles so that it moves as 16-byte pairs
            if (n % p.ckpt_every == 0) {   // uniform
                constexpr int SP = (RPL * RPL + 1) & ~1;
                const size_t nck = (size_t)((N - 1) / p.ckpt_every + 1);
                d2* dst = reinterpret_cast<d2*>(p.st_ck + ((size_t)b * nck + (size_t)(n / p.ckpt_every)) * 256 * SP) + tid;
#pragma unroll
                for (int e = 0; e < SP / 2; ++e) {
                    const int e0 = 2 * e, e1 = 2 * e + 1;
                    d2 v;
                    v.x = S[e0 / RPL][e0 % RPL];
                    v.y = e1 < RPL * RPL ? S[e1 / RPL][e1 % RPL] : 0.0;
                    dst[e * 256] = v;   // [pair][lane]: a wavefront's store is one contiguous KB
                }
            }
        }
        PIORAN_WSTAMP(1);
        double sp = 0.0;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            qt[i] = row16_sum(qt[i]);                             // over the 16 column blocks
            num[i] = in.rv[i] - qt[i];                            // :89
            sp = fma(ur[i], qt[i], sp);                           // this row block's share of u'Su   :83,88
        }
        PIORAN_WSTAMP(2);
        // ---- the one exchange of the step (+ the next record on its way through LDS) ----
        const int par = (int)(n & 1);
        if (l == 0) {
            sh_uq[par][g] = sp;
#pragma unroll
            for (int i = 0; i < RPL; ++i) sh_num[par][g * RPL + i] = num[i];
        }
        stage(par ^ 1, gslot);
        fetch(n + 1 + DG, gslot);
        __syncthreads();
        PIORAN_WSTAMP(3);
        double sh[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) sh[k] = sh_uq[par][k];
        double nc[RPL];
#pragma unroll
        for (int i = 0; i < RPL; ++i) nc[i] = sh_num[par][l * RPL + i];
        unstage(par ^ 1, nxt);
        const double s = (((sh[0] + sh[1]) + (sh[2] + sh[3])) + ((sh[4] + sh[5]) + (sh[6] + sh[7]))) +
                         (((sh[8] + sh[9]) + (sh[10] + sh[11])) + ((sh[12] + sh[13]) + (sh[14] + sh[15])));
        PIORAN_WSTAMP(4);
        Dn = suma + (has_nu ? nu * in.s2 : in.s2) - s;           // :92
        rD = recip_f64(Dn);
#pragma unroll
        for (int i = 0; i < RPL; ++i) wc[i] = nc[i] * rD;        // :96
        PIORAN_WSTAMP(5);
        const double z = num[YS];                                // y row: z_n = y_n - u'f      :141
        nonpd |= !(Dn > 0.0);
        Pm *= fabs(Dn);                                          // log(abs(D[n]))  :140
        {                                                        // mantissa/exponent split
            int ex;
            Pm = frexp(Pm, &ex);
            Pe += ex;
        }
        quad = fma(z * z, rD, quad);                             // z_n^2 / D_n  (== y'K^-1 y, :333)
        emit(n, in.y);
        PIORAN_WSTAMP(6);
    };

    int64_t n = 1;
    // unrolled by DG (even) so that the register slots are compile-time constants: at the loop top n = 1 (mod DG);
    // step n + k consumes cur[(1 + k) & 1] and stages record n + k + 1 out of gv[(2 + k) % DG]
    for (; n + DG - 1 < N; n += DG)
        static_for<0, DG>([&](auto Kc) __attribute__((always_inline)) {
            constexpr int k = decltype(Kc)::value;
            do_step(n + k, cur[(1 + k) & 1], cur[k & 1], gv[(2 + k) % DG]);
        });
    static_for<0, DG - 1>([&](auto Kc) __attribute__((always_inline)) {   // the last N - n < DG steps
        constexpr int k = decltype(Kc)::value;
        if (n + k < N) do_step(n + k, cur[(1 + k) & 1], cur[k & 1], gv[(2 + k) % DG]);
    });

    PIORAN_WSTAMP_FLUSH
    if (yrow && l == 0) {
        const double logdet = log(Pm) + (double)Pe * 0.6931471805599453094;
        const double res = -0.5 * logdet - 0.5 * (double)N * 1.8378770664093454836 - 0.5 * quad;
        p.out[b] = res;
        if (p.status) p.status[b] = !isfinite(res) ? 2 : (nonpd ? 1 : 0);
    }
}

// ---- lean form of the latency layout (round 3): 64 .. 143 rows ----------------------------------------------------------
// Same 16 x 16 arrangement, same single LDS exchange per step and the same arithmetic as celerite_wide_kernel<RPL, 0>, but the
// per-step inputs are prepared ONCE PER SLOT instead of once per lane: thread s < 16 RPL is the "slot thread" of row slot s — it
// fetches (v, x, phi) of its table row (three coalesced loads per step, DG records ahead), forms u_s = al_s v_s + be_s x_s and
// publishes (u, v, phi) of the slot; the y slot's thread publishes v = y_n - mu and sigma2_n.  The 256 lanes then read what
// they need from LDS at the point of use: u, v, phi of their RPL rows and u, phi of their RPL columns (5 RPL values, no x) —
// where celerite_wide_kernel keeps two register copies of 6 RPL + 2 inputs per lane (152 registers at RPL = 6) and recomputes
// every u 16 times.  With RPL >= 5 that kernel's loop moves 140 .. 200 values per step between VGPRs and AGPRs; here the
// registers hold S (RPL x RPL), the row / column vectors of the step and nothing else, which is what makes 7 .. 9 rows per
// lane (the reference benchmark's j = 64: 128 rows + y, benchmark/benchmarks.jl:16-18) fit at all.
// YC: y is not a row slot (all 16 RPL slots can be rows: R = 16 RPL exactly — the reference grid's j = 32 and 64 are R = 64 and 128 —
// runs with one row per lane less than it would with the y slot).  The forward solve is then the reference's own recurrence
// f <- phi o (f + W_{n-1} z_{n-1}), z_n = y_n - u_n'f (src/celerite_solver.jl:136-141) on the lane's COLUMN block (replicated in
// the 16 DPP rows), its dot product one more 16-lane sum per step; thread 255 stages y_n - mu and sigma2_n.
// SM (round 4): the MODEs of celerite_wide_kernel on this kernel — 1: the factor (W_n, D_n, forward-solved z_n) goes to HBM for the
// prediction; 2: simulation (the noise rides where y would); 3: the forward pass of the step-by-step reverse mode (v - q of all 16 RPL
// slots and D_n of every step, S_n at the checkpoints n = k * ckpt_every).  These carry 96 .. 143 rows, and 64 .. 95 faster than before.
template <int RPL, bool YC = false, int SM = 0>
__global__ void __launch_bounds__(256, 1) celerite_wide2_kernel(const ScanParams p)
{
    static_assert(!(YC && SM), "the store / simulate / gradient modes expect y as the last row slot");
    constexpr bool GM = SM == 3;
    constexpr int NS = 16 * RPL;                      // row slots; the last one is the y row
    // LDS pitch of a lane's block of RPL slots.  A pitch of 8 doubles = 16 dwords puts four of the 16 lanes of a DPP row on each
    // bank pair when they read element c of their blocks (RPL = 8 ran slower than RPL = 9: 13.0 vs 11.0 ms at N = 8192 once padded);
    // the other row counts are conflict-free or two-way at worst, and keeping their blocks 16-byte aligned (merged 16-byte
    // reads) is worth more than the padding (RPL = 4, 6 lost 5 - 8 % with an odd pitch)
    constexpr int PITCH = RPL == 8 ? 9 : RPL;
    constexpr int NSP = 16 * PITCH;
    constexpr int YS = RPL - 1;
    constexpr bool LAZY = RPL >= 7;                   // column operands one column ahead from LDS (do_step)
    constexpr int DG = LAZY ? 2 : 4;                  // records in flight from HBM per slot thread (a step of the big shapes takes
                                                      // ~2 us: two ahead cover an HBM miss, and the registers go to S)
    const int tid = threadIdx.x;
    const int g = tid >> 4;                           // row block (DPP row of the draw)
    const int l = tid & 15;                           // column block
    const int64_t b = blockIdx.x;                     // grid = batch
    const int64_t N = p.N;
    const int J = p.J, R = p.R, Rp = R + 2, RS = 3 * Rp + 2;
    const double* const tabb = p.tab + b * p.tab_draw_stride;   // per-draw tables (launches with per-draw (c, d)) or the shared one

    __shared__ double sh_rec[2][3 * NSP + 2];         // [u x NSP | v x NSP | phi x NSP | sigma2_n], by step parity; slot s at (s / RPL) PITCH + s % RPL
    __shared__ double sh_num[2][NSP];                 // (v - q) of every row = D_n w (same addressing)
    __shared__ double sh_uq[2][16];                   // u'q share of every row block

    double suma = 0.0;  // :21
    for (int j = 0; j < J; ++j) suma += p.A[b * J + j];
    const double mu = p.mu ? p.mu[b] : 0.0;
    const double nu = p.nu ? p.nu[b] : 1.0;
    const bool has_nu = p.nu != nullptr;
    const bool yrow = g == 15;                        // this lane's row slot YS is the y row

    // ---- slot threads: sources of (v, x, phi) of slot `tid`; address = src + min(n, last) * stride ----------------------
    const bool slot_thread = tid < NS || (YC && tid == 255);
    const bool yslot = YC ? tid == 255 : tid == NS - 1;
    const int sa_ = (tid / RPL) * PITCH + tid % RPL;  // LDS address of this thread's slot
    const double* src[3] = {tabb, tabb, tabb};
    int64_t stride[3] = {0, 0, 0}, last[3] = {N, N, N};
    double al = 0.0, be = 0.0;
    if (slot_thread) {
        if (yslot) {                                  // v <- y_n, x <- sigma2_n, phi <- the table's y row (1)
            src[0] = p.Y ? p.Y + b * N : tabb + 3 * Rp;
            src[1] = p.S2 ? p.S2 + b * N : tabb + 3 * Rp + 1;
            stride[0] = stride[1] = p.Y ? 1 : p.rec_stride;
            last[0] = last[1] = p.Y ? N - 1 : N;
            if constexpr (SM == 2) { src[0] = p.noise + b * N; stride[0] = 1; last[0] = N - 1; }   // the noise rides where y would
            src[2] = tabb + 2 * Rp + (R + 1);
            stride[2] = p.rec_stride;
        } else if (tid < R) {
            const int rm = p.rowmap[tid];
            const int term = rm & 0xfffff;
            const bool ks = (rm >> 30) & 1;
            const double a = p.A[b * J + term], bb = p.Bc[b * J + term];
            al = a;                                   // cos row: u = a co + b si ; sin row: u = a si - b co   (:59-63)
            be = ks ? -bb : bb;
            if ((rm >> 29) & 1) {                     // per-draw row (mixed mode): [draw][row][v, x, phi] inside the step record
                const double* base = tabb + RS + ((int64_t)b * p.npd_rows + ((rm >> 20) & 0x1ff)) * 3;
                src[0] = base; src[1] = base + 1; src[2] = base + 2;
            } else {
                src[0] = tabb + tid; src[1] = tabb + tid + Rp; src[2] = tabb + tid + 2 * Rp;
            }
            stride[0] = stride[1] = stride[2] = p.rec_stride;
        } else {                                      // inert padding slot: the table's padding row (v, x, phi) = (1, 0, 1), u = 0
            src[0] = tabb + R; src[1] = tabb + R + Rp; src[2] = tabb + R + 2 * Rp;
            stride[0] = stride[1] = stride[2] = p.rec_stride;
        }
    }
    double gv[DG][3];                                 // record m is (or will be) in gv[m % DG]
    auto fetch = [&](int64_t m, double (&dst)[3]) __attribute__((always_inline)) {
        if (slot_thread) {
#pragma unroll
            for (int h = 0; h < 3; ++h) dst[h] = src[h][(m < last[h] ? m : last[h]) * stride[h]];
        }
    };
    auto stage = [&](int par, const double (&v)[3]) __attribute__((always_inline)) {
        if (slot_thread) {
            double* r = sh_rec[par];
            if (YC && yslot) {
                r[3 * NSP + 1] = v[0] - mu;
            } else {
                r[sa_] = yslot ? 0.0 : al * v[0] + be * v[1];
                r[NSP + sa_] = yslot ? v[0] - mu : v[0];
                r[2 * NSP + sa_] = v[2];
            }
            if (yslot) r[3 * NSP] = has_nu ? nu * v[1] : v[1];
        }
    };

#pragma unroll
    for (int m = 0; m < DG; ++m) fetch(m, gv[m]);
    stage(0, gv[0]);
    fetch(DG, gv[0]);
    __syncthreads();

    // ---- first row, :27-42 and :126-128 ----
    double S[RPL][RPL];   // [own row][own column]
#pragma unroll
    for (int i = 0; i < RPL; ++i)
#pragma unroll
        for (int c = 0; c < RPL; ++c) S[i][c] = 0.0;
    double num[RPL];      // (v - q) of this lane's rows at the last step = D_n W_n, the `dn` of :73
    double wc[RPL];       // W_n of this lane's columns
    double Dn = suma + sh_rec[0][3 * NSP];
    double rD = recip_f64(Dn);
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        num[i] = sh_rec[0][NSP + g * PITCH + i];      // z_1 = y_1 in the y row      :128
        wc[i] = sh_rec[0][NSP + l * PITCH + i] * rD;
    }
    double Pm = Dn;       // running product of |D| (sign of D_1 kept: log of a negative D_1 is NaN, :126)
    int Pe = 0;
    {
        int ex;
        Pm = frexp(Pm, &ex);
        Pe += ex;
    }
    [[maybe_unused]] double fy[RPL];                  // YC: f of this lane's column block
    [[maybe_unused]] double zprev = YC ? sh_rec[0][3 * NSP + 1] : 0.0;   // z_1 = y_1 - mu      :128
#pragma unroll
    for (int i = 0; i < RPL; ++i) fy[i] = 0.0;
    double quad = YC ? zprev * zprev * rD : num[YS] * num[YS] * rD;   // (without YC: meaningful in the y-row lanes only)
    bool nonpd = !(Dn > 0.0);
    if constexpr (LAZY) {   // step 0's (v - q) = v_0 of every slot, where step 1 looks for the previous step's exchange values
        if (tid < NS) sh_num[0][sa_] = sh_rec[0][NSP + sa_];
    }
    [[maybe_unused]] auto emit = [&](int64_t n) __attribute__((always_inline)) {   // (as celerite_wide_kernel's)
        if constexpr (SM == 1) {
            if (l == 0) {
#pragma unroll
                for (int i = 0; i < RPL; ++i)
                    if (g * RPL + i < R) p.st_w[(b * N + n) * R + g * RPL + i] = num[i] * rD;
                if (yrow) {
                    p.st_d[b * N + n] = Dn;
                    p.st_z[b * N + n] = num[YS];
                }
            }
        }
        if constexpr (SM == 2) {
            if (yrow) {
                const double yn = sh_rec[n & 1][NSP + 15 * PITCH + YS];   // q_n - mu
                const double x = sqrt(Dn) * (yn + mu);                    // x_n = sqrt(D_n) q_n      :539,546
                if (l == 0) p.ysim[b * N + n] = x + (yn - num[YS]);       // x_n + u_n'f  (num = v - u'f)
                num[YS] = x;                                              // what the extra row adds next step: W_n x_n   :543
            }
        }
        if constexpr (SM == 3) {
            if (l == 0) {
#pragma unroll
                for (int i = 0; i < RPL; ++i) p.st_w[(b * N + n) * NS + g * RPL + i] = num[i];
                if (yrow) p.st_d[b * N + n] = Dn;
            }
        }
    };
    if constexpr (SM != 0) emit(0);
    stage(1, gv[1 % DG]);
    fetch(DG + 1, gv[1 % DG]);
    __syncthreads();

    // one time step n: consumes the staged record n (parity n & 1); stages record n + 1 and refills its register slot
    auto do_step = [&](int64_t n, double (&gslot)[3]) __attribute__((always_inline)) {
        const int par = (int)(n & 1);
        const double* r = sh_rec[par];
        double rp[RPL], qt[RPL];
        [[maybe_unused]] double zpart = 0.0;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            rp[i] = r[2 * NSP + g * PITCH + i];
            qt[i] = 0.0;
        }
        // ---- S update + this lane's share of q = S u ----
        if constexpr (LAZY) {
            // 7 .. 9 rows per lane: S alone is 98 .. 162 registers.  The column operands (u_c, phi_c and w_c = (v - q)_c / D of the
            // previous step, still in its exchange buffer) are read from LDS one column ahead instead of all RPL at the top of the
            // step: 6 live registers instead of 6 RPL, and the loop stops moving S through the AGPRs.
            const double* ncp = sh_num[par ^ 1] + l * PITCH;
            double ucn = r[l * PITCH], cpn = r[2 * NSP + l * PITCH], wcn = ncp[0] * rD;
#pragma unroll
            for (int c = 0; c < RPL; ++c) {
                const double uc = ucn, cp = cpn, wcc = wcn;
                if (c + 1 < RPL) { ucn = r[l * PITCH + c + 1]; cpn = r[2 * NSP + l * PITCH + c + 1]; wcn = ncp[c + 1] * rD; }
                asm volatile("" ::: "memory");   // compiler only: keep the next column's loads here, not at the top of the step
                if constexpr (YC) {
                    fy[c] = cp * fma(wcc, zprev, fy[c]);             // f <- phi o (f + W_{n-1} z_{n-1})    :136
                    zpart = fma(uc, fy[c], zpart);                   // u_n'f                               :137
                }
#pragma unroll
                for (int i = 0; i < RPL; ++i) {
                    const double m = fma(num[i], wcc, S[i][c]);       // S + dn * V[k,n-1]          :78
                    const double sn = (rp[i] * cp) * m;              // phi_j phi_k ( ... )        :78,85
                    S[i][c] = sn;
                    qt[i] = fma(sn, uc, qt[i]);                      // (S u)_j                    :80-82,86-89
                }
            }
        } else {
            double uc[RPL], cp[RPL];
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                uc[i] = r[l * PITCH + i];
                cp[i] = r[2 * NSP + l * PITCH + i];
            }
#pragma unroll
            for (int c = 0; c < RPL; ++c) {
                if constexpr (YC) {
                    fy[c] = cp[c] * fma(wc[c], zprev, fy[c]);         // f <- phi o (f + W_{n-1} z_{n-1})    :136
                    zpart = fma(uc[c], fy[c], zpart);                 // u_n'f                               :137
                }
#pragma unroll
                for (int i = 0; i < RPL; ++i) {
                    const double m = fma(num[i], wc[c], S[i][c]);     // S + dn * V[k,n-1]          :78
                    const double sn = (rp[i] * cp[c]) * m;            // phi_j phi_k ( ... )        :78,85
                    S[i][c] = sn;
                    qt[i] = fma(sn, uc[c], qt[i]);                    // (S u)_j                    :80-82,86-89
                }
            }
        }
        if constexpr (GM) {   // checkpoint of S_n: same layout as celerite_wide_kernel<RPL, 3>
            if (n % p.ckpt_every == 0) {   // uniform
                constexpr int SP = (RPL * RPL + 1) & ~1;
                const size_t nck = (size_t)((N - 1) / p.ckpt_every + 1);
                d2* dst = reinterpret_cast<d2*>(p.st_ck + ((size_t)b * nck + (size_t)(n / p.ckpt_every)) * 256 * SP) + tid;
#pragma unroll
                for (int e = 0; e < SP / 2; ++e) {
                    const int e0 = 2 * e, e1 = 2 * e + 1;
                    d2 v;
                    v.x = S[e0 / RPL][e0 % RPL];
                    v.y = e1 < RPL * RPL ? S[e1 / RPL][e1 % RPL] : 0.0;
                    dst[e * 256] = v;   // [pair][lane]: a wavefront's store is one contiguous KB
                }
            }
        }
        [[maybe_unused]] double zn = 0.0;
        if constexpr (YC) zn = r[3 * NSP + 1] - row16_sum(zpart);    // z_n = y_n - mu - u_n'f     :141
        double sp = 0.0;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            qt[i] = row16_sum(qt[i]);                             // over the 16 column blocks
            num[i] = r[NSP + g * PITCH + i] - qt[i];              // :89
            sp = fma(r[g * PITCH + i], qt[i], sp);                // this row block's share of u'Su   :83,88
        }
        const double s2n = r[3 * NSP];
        // ---- the one exchange of the step (+ the next record on its way through LDS) ----
        if (l == 0) {
            sh_uq[par][g] = sp;
#pragma unroll
            for (int i = 0; i < RPL; ++i) sh_num[par][g * PITCH + i] = num[i];
        }
        stage(par ^ 1, gslot);
        fetch(n + 1 + DG, gslot);
        __syncthreads();
        double sh[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) sh[k] = sh_uq[par][k];
        const double s = (((sh[0] + sh[1]) + (sh[2] + sh[3])) + ((sh[4] + sh[5]) + (sh[6] + sh[7]))) +
                         (((sh[8] + sh[9]) + (sh[10] + sh[11])) + ((sh[12] + sh[13]) + (sh[14] + sh[15])));
        Dn = suma + s2n - s;                                     // :92
        rD = recip_f64(Dn);
        if constexpr (!LAZY) {
#pragma unroll
            for (int i = 0; i < RPL; ++i) wc[i] = sh_num[par][l * PITCH + i] * rD;   // :96
        }
        const double z = YC ? zn : num[YS];                      // (y row: z_n = y_n - u'f      :141)
        if constexpr (YC) zprev = zn;
        nonpd |= !(Dn > 0.0);
        Pm *= fabs(Dn);                                          // log(abs(D[n]))  :140
        {                                                        // mantissa/exponent split
            int ex;
            Pm = frexp(Pm, &ex);
            Pe += ex;
        }
        quad = fma(z * z, rD, quad);                             // z_n^2 / D_n  (== y'K^-1 y, :333)
        if constexpr (SM != 0) emit(n);
    };

    int64_t n = 1;
    // unrolled by DG so that the register slots of the slot threads are compile-time constants: at the loop top n = 1 (mod DG);
    // step n + k stages record n + k + 1 out of gv[(2 + k) % DG]
    for (; n + DG - 1 < N; n += DG)
        static_for<0, DG>([&](auto Kc) __attribute__((always_inline)) {
            constexpr int k = decltype(Kc)::value;
            do_step(n + k, gv[(2 + k) % DG]);
        });
    static_for<0, DG - 1>([&](auto Kc) __attribute__((always_inline)) {   // the last N - n < DG steps
        constexpr int k = decltype(Kc)::value;
        if (n + k < N) do_step(n + k, gv[(2 + k) % DG]);
    });

    if (YC ? tid == 0 : (yrow && l == 0)) {
        const double logdet = log(Pm) + (double)Pe * 0.6931471805599453094;
        const double res = -0.5 * logdet - 0.5 * (double)N * 1.8378770664093454836 - 0.5 * quad;
        p.out[b] = res;
        if (p.status) p.status[b] = !isfinite(res) ? 2 : (nonpd ? 1 : 0);
    }
}

// ---- reverse mode through the recurrence ----------------------------------------------------------------------------
// log L = -1/2 sum_n (log D_n + z_n^2 / D_n) - N/2 log(2 pi), with per step (rows incl. the y row, u_y = 0, v_y = y_n - mu)
//   T = S_{n-1} + m_{n-1} m_{n-1}' / D_{n-1}   (m = v - q = D w) ;  S_n = (phi phi') o T ;  q = S_n u_n
//   D_n = sum(a) + nu sigma2_n - u_n'q ;  m_n = v_n - q ;  z_n = (m_n)_y
// Walking n = N-1 .. 0 with the adjoints (Sb, mb, Db) of (S_n, m_n, D_n) that the later steps left:
//   Db += -1/(2 D_n) + z_n^2 / (2 D_n^2) ;  (mb)_y -= z_n / D_n
//   qb = -mb - Db u ;  ub = -Db q + S_n qb ;  d/d(al_r) += ub_r v_r ;  d/d(be_r) += ub_r x_r        (u_r = al_r v_r + be_r x_r)
//   d/dsum(a) += Db ;  d/dnu += Db sigma2_n ;  d/dmu -= (mb)_y ;  d/dy_n = (mb)_y ;  d/dsigma2_n = nu Db
//   vb_r = ub_r al_r + mb_r ;  xb_r = ub_r be_r ;  with (v, x) = (cos, sin)(d t_n) for a cos row, (sin, cos) for a sin row:
//   d/dd_j += t_n s_r (vb_r x_r - xb_r v_r),  s_r = -1 (cos row), +1 (sin row)                       (src/celerite_solver.jl:52-53)
//   Sb <- Sb + (qb u' + u qb') / 2   (total adjoint of S_n; S is symmetric: only the symmetric part matters)
//   phi enters only through S_n = (phi phi') o T:  phi_i (dL/dphi_i) = 2 sum_k Sb_ik S_n,ik, and dphi_i/dc_j = -(t_n - t_{n-1}) phi_i, so
//   d/dc_j -= 2 (t_n - t_{n-1}) sum_{i in rows(j)} sum_k Sb_ik S_n,ik     — no division by phi               (:54)
//   Sb <- (phi phi') o Sb ;  mb <- 2 Sb m_{n-1} / D_{n-1} ;  Db <- -(m_{n-1}' mb) / (2 D_{n-1})
// Same 16 x 16 lane layout and the same single LDS exchange per step as the forward kernel (here: mb of the rows and
// the shares of m'mb); the two mat-vecs (S_n qb, Sb m) are DPP butterflies.  m_{n-1} and D_{n-1} ride through the staged
// record.
//
// Memory: the forward pass keeps S_n only at checkpoints (every K steps).  The reverse pass handles one segment between
// checkpoints per launch pair: celerite_replay_kernel rebuilds S_n of the segment — S_n = (phi phi') o (S_{n-1} + m m'/D) from
// the STORED (m, D): purely lane-local, no reduction, no barrier — and celerite_adjoint_kernel consumes it backwards.
// N = 1e4, R = 41: 20 KB per step and draw -> 180 MB per draw when every S_n was kept; now K x 20 KB for the segment plus
// N/K x 20 KB of checkpoints (K = 128: 4 MB) next to the 3.9 MB of (m, D).

// S_n, n = seg_n0 + 1 .. seg_hi, into st_s slots 0 .. seg_hi - seg_n0 - 1, from the checkpoint S_{seg_n0} (zero for seg_n0 = 0).
// Bit-identical to what the forward kernel held: same operations on the same stored operands.  The operands of a step —
// phi_n of every row (table), m_{n-1} of every slot and D_{n-1} (stored by the forward pass) — are contiguous runs: the 256
// threads fetch them coalesced for SB steps at a time into LDS (double-buffered, one barrier per SB steps) and every lane
// picks its 4 RPL + 1 values from there; per-lane global loads of those broadcast-heavy values ran at the texture path's
// pace (1 us per step for a lone workgroup).
template <int RPL>
__global__ void __launch_bounds__(256) celerite_replay_kernel(const ScanParams p)
{
    constexpr int NS = 16 * RPL, SP = (RPL * RPL + 1) & ~1;
    constexpr int SB = 8;                              // steps per staged block
    constexpr int EMAX = (RPL <= 6 ? 96 : 144) + 2 + NS + 1;   // phi of <= 95 (143) rows + padding + y | m | D
    constexpr int PT = (SB * EMAX + 255) / 256;        // staged elements per thread and block
    const int tid = threadIdx.x, g = tid >> 4, l = tid & 15;
    const int64_t b = blockIdx.x, N = p.N;
    const int Rp = p.R + 2;
    const int E = Rp + NS + 1;                         // staged doubles per step
    __shared__ double sh[2][SB * EMAX];
    int opr[RPL], opc[RPL];                            // positions of phi of this lane's row / column slots inside the staged step
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        auto pos = [&](int slot) { return slot < p.R ? slot : (slot == NS - 1 ? p.R + 1 : p.R); };
        opr[i] = pos(g * RPL + i);
        opc[i] = pos(l * RPL + i);
    }
    const double* numst = p.st_w + (size_t)b * (size_t)N * NS;
    const double* dst_ = p.st_d + (size_t)b * (size_t)N;
    const int64_t n_first = p.seg_n0 + 1, n_end = p.seg_hi;
    if (n_end < n_first) return;
    double S[RPL][RPL];
    if (p.seg_n0 == 0) {
#pragma unroll
        for (int i = 0; i < RPL; ++i)
#pragma unroll
            for (int c = 0; c < RPL; ++c) S[i][c] = 0.0;
    } else {
        const size_t nck = (size_t)((N - 1) / p.ckpt_every + 1);
        const d2* src = reinterpret_cast<const d2*>(p.st_ck + ((size_t)b * nck + (size_t)(p.seg_n0 / p.ckpt_every)) * 256 * SP) + tid;
#pragma unroll
        for (int e = 0; e < SP / 2; ++e) {
            const d2 v = src[e * 256];
            S[(2 * e) / RPL][(2 * e) % RPL] = v.x;
            if (2 * e + 1 < RPL * RPL) S[(2 * e + 1) / RPL][(2 * e + 1) % RPL] = v.y;
        }
    }
    d2* out = reinterpret_cast<d2*>(p.st_s + (size_t)b * (size_t)p.ckpt_every * 256 * SP) + tid;   // [step][pair][lane]
    double regs[PT];
    auto fetch_block = [&](int64_t blk) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < PT; ++k) {
            const int idx = tid + 256 * k;
            const int j = idx / E, e = idx - j * E;
            double v = 0.0;
            if (j < SB) {
                int64_t n = n_first + blk * SB + j;
                n = n > n_end ? n_end : n;             // past the segment: re-read its last step (never used)
                v = e < Rp ? p.tab[n * p.rec_stride + 2 * Rp + e] : (e < Rp + NS ? numst[(size_t)(n - 1) * NS + (e - Rp)] : dst_[n - 1]);
            }
            regs[k] = v;
        }
    };
    auto put_block = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < PT; ++k) {
            const int idx = tid + 256 * k;
            if (idx < SB * E) sh[buf][idx] = regs[k];
        }
    };
    const int64_t nblk = (n_end - n_first + SB) / SB;
    fetch_block(0);
    put_block(0);
    __syncthreads();
    for (int64_t blk = 0; blk < nblk; ++blk) {
        if (blk + 1 < nblk) fetch_block(blk + 1);      // in flight while this block is computed
        const double* base = sh[blk & 1];
#pragma unroll 2
        for (int j = 0; j < SB; ++j) {
            const int64_t n = n_first + blk * SB + j;
            if (n > n_end) break;                      // uniform
            const double* r = base + j * E;
            const double rD = recip_f64(r[Rp + NS]);
            double rp[RPL], mr[RPL];
#pragma unroll
            for (int i = 0; i < RPL; ++i) { rp[i] = r[opr[i]]; mr[i] = r[Rp + g * RPL + i]; }
#pragma unroll
            for (int c = 0; c < RPL; ++c) {
                const double cp = r[opc[c]], wc = r[Rp + l * RPL + c] * rD;
#pragma unroll
                for (int i = 0; i < RPL; ++i) S[i][c] = (rp[i] * cp) * fma(mr[i], wc, S[i][c]);
            }
            d2* dst = out + (size_t)(n - n_first) * 256 * (SP / 2);
#pragma unroll
            for (int e = 0; e < SP / 2; ++e) {
                const int e0 = 2 * e, e1 = 2 * e + 1;
                d2 v;
                v.x = S[e0 / RPL][e0 % RPL];
                v.y = e1 < RPL * RPL ? S[e1 / RPL][e1 % RPL] : 0.0;
                dst[e * 256] = v;
            }
        }
        if (blk + 1 < nblk) put_block((int)((blk + 1) & 1));
        __syncthreads();
    }
}

template <int RPL>
__global__ void __launch_bounds__(256, 1) celerite_adjoint_kernel(const ScanParams p)
{
    constexpr int YS = RPL - 1;
    constexpr int DG = 4;
    constexpr int NS = 16 * RPL;                      // row slots
    constexpr int NSTATE = RPL * RPL + 3 * RPL + 4;   // per lane: Sb | mbr mbc gphr | Db gA gnu gmu
    const int tid = threadIdx.x;
    const int g = tid >> 4, l = tid & 15;
    const int64_t b = blockIdx.x, N = p.N;
    const int64_t n_hi = p.seg_hi, n_lo = p.seg_lo, NP = n_hi - n_lo + 1;   // steps of this launch
    const int Rp = p.R + 2, RS = 3 * Rp + 2;
    const int L = RS + 3 * p.npd_rows;                // table part of the staged record
    const int LT = L + NS + 1;                        // + m_{n-1} of every slot + D_{n-1}

    constexpr int NE = RPL >= 7 ? 3 : 2;              // staged elements per thread and step (96 .. 143 rows: up to 582 doubles)
    __shared__ double sh_rec[2][NE * 256];
    __shared__ double sh_num[2][NS];
    __shared__ double sh_uq[2][16];
    // row accumulators of d/d(al), d/d(be), d/dd: identical in the 16 lanes of a DPP row, so they live here (updated by
    // the lane l = 0 of each row, off the critical path) instead of in 3 RPL registers of every lane
    __shared__ double sh_acc[3][NS];

    Slots<RPL> rs_, cs_;
    describe_slots<RPL>(p, b, g, rs_);
    describe_slots<RPL>(p, b, l, cs_);
    const double mu = p.mu ? p.mu[b] : 0.0;
    const double nu = p.nu ? p.nu[b] : 1.0;
    const bool yrow = g == 15, ycol = l == 15;
    int sinmask = 0, realmask = 0;                    // bit i: row slot i of this lane is a sin row / a real row at all
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        const int slot = g * RPL + i;
        if (slot < p.R) {
            realmask |= 1 << i;
            if ((p.rowmap[slot] >> 30) & 1) sinmask |= 1 << i;
        }
    }
    const double* numst = p.st_w + (size_t)b * (size_t)N * NS;
    const double* dst_ = p.st_d + (size_t)b * (size_t)N;

    // staged element e of step n: e < L table record n (as in the forward kernel); L <= e < L + NS: m_{n-1}[e - L];
    // e == L + NS: D_{n-1}.  address = src + clamp(n + shift, 0, last) * stride
    const double* src[NE];
    int64_t stride[NE], last[NE], shift[NE];
#pragma unroll
    for (int h = 0; h < NE; ++h) {
        const int e = tid + 256 * h;
        src[h] = p.tab + (e < RS ? e : e + (int64_t)b * p.npd_rows * 3);
        stride[h] = p.rec_stride; last[h] = N; shift[h] = 0;
        if (p.Y && (e == 3 * Rp || e == 3 * Rp + 1)) {
            src[h] = (e == 3 * Rp ? p.Y : p.S2) + b * N;
            stride[h] = 1; last[h] = N - 1;
        }
        if (e >= L && e < L + NS) { src[h] = numst + (e - L); stride[h] = NS; last[h] = N - 1; shift[h] = -1; }
        if (e == L + NS) { src[h] = dst_; stride[h] = 1; last[h] = N - 1; shift[h] = -1; }
        if (e >= LT) { src[h] = p.tab; stride[h] = 0; }
    }
    const int ne = (LT + 255) / 256;                  // uniform
    auto fetch = [&](int64_t n, double (&dstv)[NE]) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < NE; ++h) {
            if (h >= ne) break;
            int64_t k = n + shift[h];
            k = k < 0 ? 0 : (k > last[h] ? last[h] : k);
            dstv[h] = src[h][k * stride[h]];
        }
    };
    auto stage = [&](int par, const double (&v)[NE]) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < NE; ++h)
            if (h < ne) sh_rec[par][tid + 256 * h] = v[h];
    };
    // S_n of this lane from the replayed segment (slot n - seg_n0 - 1), SD steps ahead in registers
    constexpr int SP = (RPL * RPL + 1) & ~1;          // padded block (16-byte pairs)
    const double* sbase = p.st_s + (size_t)b * (size_t)p.ckpt_every * 256 * SP;
    const int64_t s_first = p.seg_n0 + 1;             // lowest step present in the segment buffer
    auto fetch_s = [&](int64_t n, double (&dsts)[RPL * RPL]) __attribute__((always_inline)) {
        int64_t k = n < s_first ? s_first : n;        // below the segment (or S_0): never used, any readable slot will do
        k = k > n_hi ? n_hi : k;
        const bool have = n_hi >= s_first;            // N = 1: nothing was replayed (the buffer exists; what is read then is never used)
        const d2* q_ = reinterpret_cast<const d2*>(sbase + (size_t)(have ? k - s_first : 0) * 256 * SP) + tid;
        // (unconditional: `v = 0; if (have) v = load` is a default value on a loaded register — the compiler waits for the loads in flight
        //  at the join, i.e. the prefetch would be consumed at once)
#pragma unroll
        for (int e = 0; e < SP / 2; ++e) {
            const d2 v = q_[e * 256];
            dsts[2 * e] = v.x;
            if (2 * e + 1 < RPL * RPL) dsts[2 * e + 1] = v.y;
        }
    };

    // steps are visited in DEscending n; "position" s = n_hi - n plays the role n plays in the forward kernel
    double gv[DG][NE];
#pragma unroll
    for (int m = 0; m < DG; ++m) fetch(n_hi - m, gv[m]);
    constexpr int SD = RPL <= 3 ? 2 : 1;              // S_n buffers (steps ahead); one where registers are short
    double sv[SD][RPL * RPL];
    fetch_s(n_hi, sv[0]);
    if constexpr (SD == 2) fetch_s(n_hi - 1, sv[1]);
    stage((int)(n_hi & 1), gv[0]);                    // record n is staged in sh_rec[n & 1] and READ FROM THERE at its
    fetch(n_hi - DG, gv[0]);                          // points of use (no register copy: the register budget goes to S)

    // m_n, D_n, z_n of the first step of this launch: straight from HBM, once
    double mr[RPL], mc[RPL];
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        mr[i] = numst[(size_t)n_hi * NS + g * RPL + i];
        mc[i] = numst[(size_t)n_hi * NS + l * RPL + i];
    }
    double Dn = dst_[n_hi];
    double zn = numst[(size_t)n_hi * NS + NS - 1];

    // adjoint state: zero at the first launch of the reverse pass, else what the previous launch (the later segment) parked
    double Sb[RPL][RPL];
    double mbr[RPL], mbc[RPL], gphr[RPL];
    double Db = 0.0, gA = 0.0, gnu = 0.0, gmu = 0.0;
    double* state = p.st_state + (size_t)b * (NSTATE * 256 + 3 * NS) + tid;      // [e][tid]: coalesced; then the 3 NS row accumulators
    double* state_acc = p.st_state + (size_t)b * (NSTATE * 256 + 3 * NS) + NSTATE * 256;
    if (p.seg_first) {
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
#pragma unroll
            for (int c = 0; c < RPL; ++c) Sb[i][c] = 0.0;
            mbr[i] = 0.0; mbc[i] = 0.0; gphr[i] = 0.0;
        }
        for (int e = tid; e < 3 * NS; e += 256) (&sh_acc[0][0])[e] = 0.0;
    } else {
        int e = 0;
#pragma unroll
        for (int i = 0; i < RPL; ++i)
#pragma unroll
            for (int c = 0; c < RPL; ++c) Sb[i][c] = state[256 * e++];
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            mbr[i] = state[256 * (e + 0 * RPL + i)]; mbc[i] = state[256 * (e + 1 * RPL + i)]; gphr[i] = state[256 * (e + 2 * RPL + i)];
        }
        e += 3 * RPL;
        Db = state[256 * e]; gA = state[256 * (e + 1)]; gnu = state[256 * (e + 2)]; gmu = state[256 * (e + 3)];
        for (int k = tid; k < 3 * NS; k += 256) (&sh_acc[0][0])[k] = state_acc[k];
    }
    __syncthreads();

    auto do_step = [&](int64_t n, double (&gslot)[NE], double (&sn)[RPL * RPL]) __attribute__((always_inline)) {
        const double* r = sh_rec[n & 1];               // record n: table part, then m_{n-1} of every slot, D_{n-1}
        double ur[RPL], uc[RPL], qbr[RPL], qbc[RPL], ub[RPL];
        const double rDn = recip_f64(Dn);
        const double tn = p.t[n];                      // uniform (scalar load)
        const double dtn = n > 0 ? tn - p.t[n - 1] : 0.0;
        Db += -0.5 * rDn + 0.5 * zn * zn * rDn * rDn;
        if (yrow) mbr[YS] -= zn * rDn;
        if (ycol) mbc[YS] -= zn * rDn;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            ur[i] = rs_.al[i] * r[rs_.ov[i]] + rs_.be[i] * r[rs_.ox[i]];
            uc[i] = cs_.al[i] * r[cs_.ov[i]] + cs_.be[i] * r[cs_.ox[i]];
            qbr[i] = -mbr[i] - Db * ur[i];
            qbc[i] = -mbc[i] - Db * uc[i];
            ub[i] = 0.0;
        }
        if (n > 0) {
#pragma unroll
            for (int i = 0; i < RPL; ++i)
#pragma unroll
                for (int c = 0; c < RPL; ++c) ub[i] = fma(sn[i * RPL + c], qbc[c], ub[i]);
#pragma unroll
            for (int i = 0; i < RPL; ++i) ub[i] = row16_sum(ub[i]);
        }
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const double rv = r[rs_.ov[i]], rx = r[rs_.ox[i]];
            const double vr = (yrow && i == YS) ? r[3 * Rp] - mu : rv;
            ub[i] = fma(-Db, vr - mr[i], ub[i]);       // q = v - m
            if (l == 0) {
                // d/dd: through v (in u and in m = v - q) and x (in u); s = -1 for a cos row, +1 for a sin row
                const double vb = fma(ub[i], rs_.al[i], mbr[i]), xb = ub[i] * rs_.be[i];
                const double sg = ((realmask >> i) & 1) ? (((sinmask >> i) & 1) ? tn : -tn) : 0.0;
                sh_acc[0][g * RPL + i] = fma(ub[i], rv, sh_acc[0][g * RPL + i]);
                sh_acc[1][g * RPL + i] = fma(ub[i], rx, sh_acc[1][g * RPL + i]);
                sh_acc[2][g * RPL + i] = fma(sg, fma(vb, rx, -xb * rv), sh_acc[2][g * RPL + i]);
            }
        }
        gA += Db;
        gnu = fma(Db, r[3 * Rp + 1], gnu);
        if (yrow) {
            gmu -= mbr[YS];
            if (l == 0) {
                if (p.g_y) p.g_y[b * N + n] = mbr[YS];
                if (p.g_s2) p.g_s2[b * N + n] = nu * Db;
            }
        }
        if (n == 0) return;
        // ---- adjoints of (S_{n-1}, m_{n-1}, D_{n-1}); d/dc through phi_n ----
        double nb[RPL], rp[RPL];
#pragma unroll
        for (int i = 0; i < RPL; ++i) { nb[i] = 0.0; rp[i] = r[rs_.op[i]]; }
#pragma unroll
        for (int c = 0; c < RPL; ++c) {
            const double cp = r[cs_.op[c]], pc = r[L + l * RPL + c];
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const double t_ = fma(0.5 * qbr[i], uc[c], fma(0.5 * ur[i], qbc[c], Sb[i][c]));   // total adjoint of S_n[i][c]
                gphr[i] = fma(dtn * t_, sn[i * RPL + c], gphr[i]);                                 // this lane's share of sum_k Sb_ik S_ik
                const double sbn = (rp[i] * cp) * t_;
                Sb[i][c] = sbn;
                nb[i] = fma(sbn, pc, nb[i]);
            }
        }
        fetch_s(n - SD, sn);                           // this register buffer is free again
        const double Dp = r[L + NS];
        const double rDp = recip_f64(Dp);
        double share = 0.0;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            nb[i] = 2.0 * row16_sum(nb[i]) * rDp;
            mr[i] = r[L + g * RPL + i];                // m_{n-1}: "previous" becomes "current"
            share = fma(mr[i], nb[i], share);
            mbr[i] = nb[i];
        }
        const int par = (int)(n & 1);
        if (l == 0) {
            sh_uq[par][g] = share;
#pragma unroll
            for (int i = 0; i < RPL; ++i) sh_num[par][g * RPL + i] = nb[i];
        }
#pragma unroll
        for (int i = 0; i < RPL; ++i) mc[i] = r[L + l * RPL + i];
        Dn = Dp;
        zn = r[L + NS - 1];                            // the y slot is the last one
        stage(par ^ 1, gslot);                         // record n - 1 -> sh_rec[(n-1) & 1]
        fetch(n - 1 - DG, gslot);
        __syncthreads();
        double sh[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) sh[k] = sh_uq[par][k];
#pragma unroll
        for (int i = 0; i < RPL; ++i) mbc[i] = sh_num[par][l * RPL + i];
        const double tot = (((sh[0] + sh[1]) + (sh[2] + sh[3])) + ((sh[4] + sh[5]) + (sh[6] + sh[7]))) +
                           (((sh[8] + sh[9]) + (sh[10] + sh[11])) + ((sh[12] + sh[13]) + (sh[14] + sh[15])));
        Db = -0.5 * tot * rDp;
    };

    // positions: at position s = n_hi - n the global ring slot (s + 1) % DG holds record n - 1,
    // the S buffer s % SD holds S_n.  Unrolled by DG so that all of these are compile-time constants.
    int64_t s0 = 0;
    for (; s0 + DG <= NP; s0 += DG)
        static_for<0, DG>([&](auto Kc) __attribute__((always_inline)) {
            constexpr int k = decltype(Kc)::value;
            do_step(n_hi - (s0 + k), gv[(k + 1) % DG], sv[k % SD]);
        });
    static_for<0, DG - 1>([&](auto Kc) __attribute__((always_inline)) {
        constexpr int k = decltype(Kc)::value;
        if (s0 + k < NP) do_step(n_hi - (s0 + k), gv[(k + 1) % DG], sv[k % SD]);
    });

    __syncthreads();
    if (n_lo > 0) {   // park the adjoint state for the launch that continues with the earlier segment
        int e = 0;
#pragma unroll
        for (int i = 0; i < RPL; ++i)
#pragma unroll
            for (int c = 0; c < RPL; ++c) state[256 * e++] = Sb[i][c];
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            state[256 * (e + 0 * RPL + i)] = mbr[i]; state[256 * (e + 1 * RPL + i)] = mbc[i]; state[256 * (e + 2 * RPL + i)] = gphr[i];
        }
        e += 3 * RPL;
        state[256 * e] = Db; state[256 * (e + 1)] = gA; state[256 * (e + 2)] = gnu; state[256 * (e + 3)] = gmu;
        for (int k = tid; k < 3 * NS; k += 256) state_acc[k] = (&sh_acc[0][0])[k];
        return;
    }
#pragma unroll
    for (int i = 0; i < RPL; ++i) gphr[i] = row16_sum(gphr[i]);   // over the 16 column blocks
    if (l == 0) {
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            p.g_al[b * NS + g * RPL + i] = sh_acc[0][g * RPL + i];
            p.g_be[b * NS + g * RPL + i] = sh_acc[1][g * RPL + i];
            p.g_d[b * NS + g * RPL + i] = sh_acc[2][g * RPL + i];
            p.g_c[b * NS + g * RPL + i] = -2.0 * gphr[i];
        }
        if (yrow) {
            p.g_scal[b * 4 + 0] = gA;
            p.g_scal[b * 4 + 1] = gnu;
            p.g_scal[b * 4 + 2] = gmu;
            p.g_scal[b * 4 + 3] = 0.0;
        }
    }
}

// ---- lean form of the reverse pass (round 4) ------------------------------------------------------------------------------------
// Same mathematics, segments, workspace and parked state as celerite_adjoint_kernel; what changes is where the per-step inputs live.
// As in celerite_wide2_kernel a SLOT THREAD s < 16 RPL fetches (v, x, phi) of its table row and m_{n-1}[s] (DG steps ahead) and
// publishes per slot  u | v | x | phi | m_{n-1} | q = v - m_n  (+ sigma2_n, D_{n-1}); the 256 lanes read row and column operands from
// LDS at the point of use (column operands one column ahead).  The registers then hold the adjoint state Sb, the replayed S_n and
// 5 RPL row vectors — celerite_adjoint_kernel kept 126 registers of slot descriptors and 11 RPL row / column vectors (147 spilled
// registers at RPL = 6, 3000 at RPL = 9).  The mat-vec S_n qb (only the accumulators of d/d(al, be, d) need it) is fused into the
// loop that updates Sb, so S_n is read once, in the middle of the step: its loads (one buffer ahead, two where the registers allow)
// have most of a step to land.
template <int RPL>
__global__ void __launch_bounds__(256, 1) celerite_adjoint2_kernel(const ScanParams p)
{
    constexpr int YS = RPL - 1;
    constexpr int DG = RPL >= 6 ? 2 : 4;             // records in flight per slot thread
    constexpr int NS = 16 * RPL;
    constexpr int PITCH = RPL == 8 ? 9 : RPL;         // (as celerite_wide2_kernel)
    constexpr int NSP = 16 * PITCH;
    constexpr int NSTATE = RPL * RPL + 3 * RPL + 4;   // layout of celerite_adjoint_kernel's parked state (the mbc slots stay unused)
    constexpr int SP = (RPL * RPL + 1) & ~1;
    constexpr int SD = RPL <= 5 ? 2 : 1;              // S_n buffers (a second one where the registers allow)
    // PL pairs of S_n do not pass through registers: every wavefront copies that part of its own lanes' block of the replayed segment
    // ([pair][lane], verbatim) into LDS with the LDS DMA one step ahead and reads the elements at their use; the other pairs are fetched into
    // registers as before.  7 and 8 rows per lane: all of S_n (Sb + S_n were 196 .. 256 of the 256 registers the vector ALU addresses: scratch);
    // 9 rows per lane: 33 of the 41 pairs (all of them would take 166 KB of LDS); up to 6: none (measured at 6: 47.8 against 45.3 ms).
    constexpr int PL = (RPL == 7 || RPL == 8) ? ((RPL * RPL + 1) & ~1) / 2 : (RPL == 9 ? 33 : 0);
    constexpr bool SL = PL > 0;
    constexpr int NREG = ((RPL * RPL + 1) & ~1) - 2 * PL;   // doubles of S_n that live in registers
    constexpr int oU = 0, oV = NSP, oX = 2 * NSP, oP = 3 * NSP, oM = 4 * NSP, oQ = 5 * NSP, oS = 6 * NSP;   // oS: sigma2_n, D_{n-1}
    const int tid = threadIdx.x;
    const int g = tid >> 4, l = tid & 15;
    const int64_t b = blockIdx.x, N = p.N;
    const int64_t n_hi = p.seg_hi, n_lo = p.seg_lo, NP = n_hi - n_lo + 1;
    const int R = p.R, Rp = R + 2;

    __shared__ double sh_rec[2][6 * NSP + 2];
    __shared__ double sh_num[2][NSP];                 // mb of every row slot (the exchange of the step)
    __shared__ double sh_uq[2][16];
    __shared__ double sh_acc[3][NSP];                 // row accumulators of d/d(al), d/d(be), d/dd
    __shared__ double sh_c[3][NSP];                   // al | be | sign of the d-derivative (-1 cos row, +1 sin row, 0 otherwise)
    __shared__ double sh_s[SL ? 256 * 2 * PL : 2];    // the first PL pairs of S_n of the step, [pair][lane][2]

    const double mu = p.mu ? p.mu[b] : 0.0;
    const double nu = p.nu ? p.nu[b] : 1.0;
    const bool yrow = g == 15, ycol = l == 15;
    const double* numst = p.st_w + (size_t)b * (size_t)N * NS;
    const double* dst_ = p.st_d + (size_t)b * (size_t)N;

    // ---- slot threads -------------------------------------------------------------------------------------------------------
    const bool slot_thread = tid < NS;
    const bool yslot = tid == NS - 1;
    const int sa_ = (tid / RPL) * PITCH + tid % RPL;
    const double* src[5] = {p.tab, p.tab, p.tab, p.tab, p.tab};    // v | x | phi | m_{n-1} | D_{n-1} (y slot)
    int64_t stride[5] = {0, 0, 0, 0, 0}, last[5] = {N, N, N, N - 1, N - 1}, shift[5] = {0, 0, 0, -1, -1};
    double al = 0.0, be = 0.0, sgn = 0.0;
    if (slot_thread) {
        if (yslot) {
            src[0] = p.Y ? p.Y + b * N : p.tab + 3 * Rp;
            src[1] = p.S2 ? p.S2 + b * N : p.tab + 3 * Rp + 1;
            stride[0] = stride[1] = p.Y ? 1 : p.rec_stride;
            last[0] = last[1] = p.Y ? N - 1 : N;
            src[2] = p.tab + 2 * Rp + (R + 1);
            stride[2] = p.rec_stride;
            src[4] = dst_; stride[4] = 1;
        } else {
            const int trow = tid < R ? tid : R;       // inert padding slots: the table's padding row (v, x, phi) = (1, 0, 1), u = 0
            if (tid < R) {
                const int rm = p.rowmap[tid];
                const int term = rm & 0xfffff;
                const bool ks = (rm >> 30) & 1;
                al = p.A[b * p.J + term];
                be = ks ? -p.Bc[b * p.J + term] : p.Bc[b * p.J + term];
                sgn = ks ? 1.0 : -1.0;
            }
            src[0] = p.tab + trow; src[1] = p.tab + trow + Rp; src[2] = p.tab + trow + 2 * Rp;
            stride[0] = stride[1] = stride[2] = p.rec_stride;
        }
        src[3] = numst + tid; stride[3] = NS;
        sh_c[0][sa_] = al; sh_c[1][sa_] = be; sh_c[2][sa_] = sgn;
    }
    double gv[DG][5];
    auto fetch = [&](int64_t n, double (&d)[5]) __attribute__((always_inline)) {
        if (slot_thread) {
#pragma unroll
            for (int h = 0; h < 5; ++h) {
                if (h == 4 && !yslot) break;
                int64_t k = n + shift[h];
                k = k < 0 ? 0 : (k > last[h] ? last[h] : k);
                d[h] = src[h][k * stride[h]];
            }
        }
    };
    double pm = slot_thread ? numst[(size_t)n_hi * NS + tid] : 0.0;   // m_n[s] of the record about to be staged
    auto stage = [&](int par, const double (&v)[5]) __attribute__((always_inline)) {
        if (slot_thread) {
            double* r = sh_rec[par];
            const double vv = yslot ? v[0] - mu : v[0];
            r[oU + sa_] = yslot ? 0.0 : al * v[0] + be * v[1];
            r[oV + sa_] = vv;
            r[oX + sa_] = yslot ? 0.0 : v[1];
            r[oP + sa_] = v[2];
            r[oM + sa_] = v[3];
            r[oQ + sa_] = vv - pm;
            pm = v[3];
            if (yslot) { r[oS] = v[1]; r[oS + 1] = v[4]; }
        }
    };
    const double* sbase = p.st_s + (size_t)b * (size_t)p.ckpt_every * 256 * SP;   // [step][pair][lane]
    const int64_t s_first = p.seg_n0 + 1;
    auto fetch_s = [&](int64_t n, double (&dsts)[NREG > 0 ? NREG : 1]) __attribute__((always_inline)) {
        int64_t k = n < s_first ? s_first : n;        // below the segment (or S_0): never used, any readable slot will do
        k = k > n_hi ? n_hi : k;
        const bool have = n_hi >= s_first;            // N = 1: nothing was replayed (the buffer exists; what is read then is never used)
        const d2* q_ = reinterpret_cast<const d2*>(sbase + (size_t)(have ? k - s_first : 0) * 256 * SP) + tid;
        // (unconditional: `v = 0; if (have) v = load` is a default value on a loaded register — the compiler waits for the loads in flight
        //  at the join, i.e. the prefetch would be consumed at once)
#pragma unroll
        for (int e = PL; e < SP / 2; ++e) {
            const d2 v = q_[e * 256];
            dsts[2 * (e - PL)] = v.x;
            if (2 * e + 1 < RPL * RPL) dsts[2 * (e - PL) + 1] = v.y;
        }
    };

    // (SL) this wavefront's 64 lanes x SP doubles of S_n -> sh_s: SP / 2 pieces of 1 KB.  Inline assembly: the compiler's wait-count pass
    // must not see an LDS write (it would wait for it in front of every LDS read); ordering is explicit — s_waitcnt vmcnt(0) before the
    // element loop reads, lgkmcnt(0) before the next copy is issued; only the issuing wavefront's lanes read what it copied.
    [[maybe_unused]] auto dma_s = [&](int64_t n) __attribute__((always_inline)) {
        int64_t k = n < s_first ? s_first : n;
        k = k > n_hi ? n_hi : k;
        const bool have = n_hi >= s_first;
        const double* g0 = sbase + (size_t)(have ? k - s_first : 0) * 256 * SP + (size_t)tid * 2;
        const unsigned l0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)sh_s + (unsigned)(tid & ~63) * 16u;
#pragma unroll
        for (int e = 0; e < PL; ++e) {
            const double* g_ = g0 + e * 512;
            const unsigned l = __builtin_amdgcn_readfirstlane(l0 + (unsigned)e * 4096u);
            unsigned m0_save;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(m0_save)
                         : "v"(g_), "s"(l)
                         : "memory");
        }
    };
#pragma unroll
    for (int m = 0; m < DG; ++m) fetch(n_hi - m, gv[m]);
    double sv[SD][NREG > 0 ? NREG : 1];
    if constexpr (SL) dma_s(n_hi);
    if constexpr (NREG > 0) {
        fetch_s(n_hi, sv[0]);
        if constexpr (SD == 2) fetch_s(n_hi - 1, sv[1]);
    }
    stage((int)(n_hi & 1), gv[0]);
    fetch(n_hi - DG, gv[0]);

    double Dn = dst_[n_hi];
    double zn = numst[(size_t)n_hi * NS + NS - 1];

    double Sb[RPL][RPL];
    double mbr[RPL], gphr[RPL];
    double Db = 0.0, gA = 0.0, gnu = 0.0, gmu = 0.0;
    double* state = p.st_state + (size_t)b * (NSTATE * 256 + 3 * NS) + tid;
    double* state_acc = p.st_state + (size_t)b * (NSTATE * 256 + 3 * NS) + NSTATE * 256;
    if (p.seg_first) {
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
#pragma unroll
            for (int c = 0; c < RPL; ++c) Sb[i][c] = 0.0;
            mbr[i] = 0.0; gphr[i] = 0.0;
        }
        for (int e = tid; e < 3 * NSP; e += 256) (&sh_acc[0][0])[e] = 0.0;
    } else {
        int e = 0;
#pragma unroll
        for (int i = 0; i < RPL; ++i)
#pragma unroll
            for (int c = 0; c < RPL; ++c) Sb[i][c] = state[256 * e++];
#pragma unroll
        for (int i = 0; i < RPL; ++i) { mbr[i] = state[256 * (e + 0 * RPL + i)]; gphr[i] = state[256 * (e + 2 * RPL + i)]; }
        e += 3 * RPL;
        Db = state[256 * e]; gA = state[256 * (e + 1)]; gnu = state[256 * (e + 2)]; gmu = state[256 * (e + 3)];
        if (slot_thread) {
#pragma unroll
            for (int k = 0; k < 3; ++k) sh_acc[k][sa_] = state_acc[k * NS + tid];
        }
    }
    // mb of the step above this launch's first one, where the column loop looks for it
    if (l == 0) {
#pragma unroll
        for (int i = 0; i < RPL; ++i) sh_num[(n_hi + 1) & 1][g * PITCH + i] = mbr[i];
    }
    __syncthreads();

    auto do_step = [&](int64_t n, double (&gslot)[5], double (&sn)[NREG > 0 ? NREG : 1]) __attribute__((always_inline)) {
        const int par = (int)(n & 1);
        const double* r = sh_rec[par];
        const double rDn = recip_f64(Dn);
        const double tn = p.t[n];                      // uniform (scalar load)
        const double dtn = n > 0 ? tn - p.t[n - 1] : 0.0;
        Db += -0.5 * rDn + 0.5 * zn * zn * rDn * rDn;
        const double yadj = zn * rDn;
        if (yrow) mbr[YS] -= yadj;
        double hq[RPL], hu[RPL], rp[RPL], ub[RPL], nb[RPL], gp[RPL];
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const double ur = r[oU + g * PITCH + i];
            hq[i] = 0.5 * (-mbr[i] - Db * ur);         // qb / 2
            hu[i] = 0.5 * ur;
            rp[i] = r[oP + g * PITCH + i];
            ub[i] = 0.0; nb[i] = 0.0; gp[i] = 0.0;
        }
        if (n > 0) {
            // ---- adjoints of (S_{n-1}, m_{n-1}); d/dc through phi_n; S_n qb -------------------------------------------------
            if constexpr (SL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // S_n has landed in sh_s
            const double* mbp = sh_num[par ^ 1] + l * PITCH;   // mb of the columns (the exchange of step n + 1)
            double ucn = r[oU + l * PITCH], cpn = r[oP + l * PITCH], pcn = r[oM + l * PITCH], mcn = mbp[0];
#pragma unroll
            for (int c = 0; c < RPL; ++c) {
                const double uc = ucn, cp = cpn, pc = pcn;
                double mbc = mcn;
                if (c + 1 < RPL) { ucn = r[oU + l * PITCH + c + 1]; cpn = r[oP + l * PITCH + c + 1]; pcn = r[oM + l * PITCH + c + 1]; mcn = mbp[c + 1]; }
                asm volatile("" ::: "memory");   // compiler only: the next column's loads stay here
                if (c == YS && ycol) mbc -= yadj;
                const double qbc = -mbc - Db * uc;
#pragma unroll
                for (int i = 0; i < RPL; ++i) {
                    double s_;
                    if ((i * RPL + c) < 2 * PL) s_ = sh_s[(((i * RPL + c) >> 1) * 256 + tid) * 2 + ((i * RPL + c) & 1)];   // (compile-time choice)
                    else s_ = sn[i * RPL + c - 2 * PL];
                    const double t_ = fma(hq[i], uc, fma(hu[i], qbc, Sb[i][c]));   // total adjoint of S_n[i][c]
                    ub[i] = fma(s_, qbc, ub[i]);
                    gp[i] = fma(t_, s_, gp[i]);
                    const double sbn = (rp[i] * cp) * t_;
                    Sb[i][c] = sbn;
                    nb[i] = fma(sbn, pc, nb[i]);
                }
            }
            if constexpr (SL) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every read of S_n has returned
                dma_s(n - 1);
            }
            if constexpr (NREG > 0) {
                if (!(p.exp & 1)) fetch_s(n - SD, sn);
            }
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                ub[i] = row16_sum(ub[i]);
                gphr[i] = fma(dtn, gp[i], gphr[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            ub[i] = fma(-Db, r[oQ + g * PITCH + i], ub[i]);   // q = v - m
            if (l == 0) {
                const int sa = g * PITCH + i;
                const double rv = r[oV + sa], rx = r[oX + sa];
                const double vb = fma(ub[i], sh_c[0][sa], mbr[i]), xb = ub[i] * sh_c[1][sa];
                sh_acc[0][sa] = fma(ub[i], rv, sh_acc[0][sa]);
                sh_acc[1][sa] = fma(ub[i], rx, sh_acc[1][sa]);
                sh_acc[2][sa] = fma(sh_c[2][sa] * tn, fma(vb, rx, -xb * rv), sh_acc[2][sa]);
            }
        }
        gA += Db;
        gnu = fma(Db, r[oS], gnu);
        if (yrow) {
            gmu -= mbr[YS];
            if (l == 0) {
                if (p.g_y) p.g_y[b * N + n] = mbr[YS];
                if (p.g_s2) p.g_s2[b * N + n] = nu * Db;
            }
        }
        if (n == 0) return;
        const double Dp = r[oS + 1];
        const double rDp = recip_f64(Dp);
        double share = 0.0;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            nb[i] = 2.0 * row16_sum(nb[i]) * rDp;
            share = fma(r[oM + g * PITCH + i], nb[i], share);
            mbr[i] = nb[i];
        }
        if (l == 0) {
            sh_uq[par][g] = share;
#pragma unroll
            for (int i = 0; i < RPL; ++i) sh_num[par][g * PITCH + i] = nb[i];
        }
        Dn = Dp;
        zn = r[oM + 15 * PITCH + YS];                  // the y slot is the last one
        stage(par ^ 1, gslot);
        fetch(n - 1 - DG, gslot);
        __syncthreads();
        double sh[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) sh[k] = sh_uq[par][k];
        const double tot = (((sh[0] + sh[1]) + (sh[2] + sh[3])) + ((sh[4] + sh[5]) + (sh[6] + sh[7]))) +
                           (((sh[8] + sh[9]) + (sh[10] + sh[11])) + ((sh[12] + sh[13]) + (sh[14] + sh[15])));
        Db = -0.5 * tot * rDp;
    };

    int64_t s0 = 0;
    for (; s0 + DG <= NP; s0 += DG)
        static_for<0, DG>([&](auto Kc) __attribute__((always_inline)) {
            constexpr int k = decltype(Kc)::value;
            do_step(n_hi - (s0 + k), gv[(k + 1) % DG], sv[k % SD]);
        });
    static_for<0, DG - 1>([&](auto Kc) __attribute__((always_inline)) {
        constexpr int k = decltype(Kc)::value;
        if (s0 + k < NP) do_step(n_hi - (s0 + k), gv[(k + 1) % DG], sv[k % SD]);
    });

    if constexpr (SL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last copy (never read) must not outlive the workgroup's LDS
    __syncthreads();
    if (n_lo > 0) {
        int e = 0;
#pragma unroll
        for (int i = 0; i < RPL; ++i)
#pragma unroll
            for (int c = 0; c < RPL; ++c) state[256 * e++] = Sb[i][c];
#pragma unroll
        for (int i = 0; i < RPL; ++i) { state[256 * (e + 0 * RPL + i)] = mbr[i]; state[256 * (e + 2 * RPL + i)] = gphr[i]; }
        e += 3 * RPL;
        state[256 * e] = Db; state[256 * (e + 1)] = gA; state[256 * (e + 2)] = gnu; state[256 * (e + 3)] = gmu;
        if (slot_thread) {
#pragma unroll
            for (int k = 0; k < 3; ++k) state_acc[k * NS + tid] = sh_acc[k][sa_];
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < RPL; ++i) gphr[i] = row16_sum(gphr[i]);
    if (l == 0) {
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int sa = g * PITCH + i;
            p.g_al[b * NS + g * RPL + i] = sh_acc[0][sa];
            p.g_be[b * NS + g * RPL + i] = sh_acc[1][sa];
            p.g_d[b * NS + g * RPL + i] = sh_acc[2][sa];
            p.g_c[b * NS + g * RPL + i] = -2.0 * gphr[i];
        }
        if (yrow) {
            p.g_scal[b * 4 + 0] = gA;
            p.g_scal[b * 4 + 1] = gnu;
            p.g_scal[b * 4 + 2] = gmu;
            p.g_scal[b * 4 + 3] = 0.0;
        }
    }
}

// row adjoints -> term gradients: a_j enters al of both rows and sum(a); b_j enters be of the cos row and -be of the sin row;
// c_j and d_j collect the accumulators of their rows
__global__ void __launch_bounds__(256) grad_finish_kernel(const ScanParams p, int NS, double* __restrict__ grad_a,
                                                          double* __restrict__ grad_b, double* __restrict__ grad_c,
                                                          double* __restrict__ grad_d, double* __restrict__ grad_nu,
                                                          double* __restrict__ grad_mu)
{
    const int64_t b = blockIdx.x;
    const int J = p.J;
    for (int j = threadIdx.x; j < J; j += 256) {
        double ga = p.g_scal[b * 4 + 0], gb = 0.0, gc = 0.0, gd = 0.0;
        for (int r = 0; r < p.R; ++r) {
            const int rm = p.rowmap[r];
            if ((rm & 0xfffff) != j) continue;
            const bool ks = (rm >> 30) & 1;
            ga += p.g_al[b * NS + r];
            gb += ks ? -p.g_be[b * NS + r] : p.g_be[b * NS + r];
            gc += p.g_c[b * NS + r];
            gd += p.g_d[b * NS + r];
        }
        grad_a[b * J + j] = ga;
        grad_b[b * J + j] = gb;
        if (grad_c) grad_c[b * J + j] = gc;
        if (grad_d) grad_d[b * J + j] = gd;
    }
    if (threadIdx.x == 0) {
        if (grad_nu) grad_nu[b] = p.g_scal[b * 4 + 1];
        if (grad_mu) grad_mu[b] = p.g_scal[b * 4 + 2];
    }
}

}  // namespace

// RPL <= 6 (95 rows): the shapes of rounds 1-2.  RPL 7 .. 9 (up to 143 rows, round 3): 49 .. 81 entries of S per lane, one
// wavefront per SIMD with the 512-register budget — the reference's own benchmark grid goes up to j = 64 terms = 128 rows
// (benchmark/benchmarks.jl:16-18), which used to fall to the HBM-resident any-rank kernel.
int pioran_wide_supported_rows() { return 143; }
// ... the store (prediction) and simulate modes: on the lean kernel since round 4
int pioran_wide_supported_rows_modes() { return 143; }
int pioran_predict_supported_rows() { return 143; }
// the step-by-step reverse mode: RPL 7 .. 9 since round 4 (forward pass: the lean kernel with GM)
int pioran_wide_supported_rows_grad() { return 143; }

// Batches up to this size take the latency layout (at most one workgroup per CU on the chip's 256 CUs).
int64_t pioran_wide_max_batch() { return 256; }

template <int MODE>
static int launch_wide_mode(const ScanParams& p, hipStream_t stream)
{
    if (!p.tab || p.R > 143 || p.B <= 0 || p.B > 0x7fffffffLL) return PIORAN_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)p.B), block(256);
    if constexpr (MODE == 0) {
        // the lean form: from 48 rows on (tools/sweep_wide.py, N = 8192: R = 48 6.0 -> 5.7 ms, 80 9.1 -> 7.5, 94 9.4 -> 7.6; below
        // that the register copies of celerite_wide_kernel are still cheaper: R = 40 4.8 vs 5.1 ms), or on request
        const bool lean = (p.opt && p.opt->wide2) || (p.R >= 48 && !(p.opt && p.opt->no_wide2));
        if (lean || p.R > 95 || p.tab_draw_stride != 0) {
            if (p.R % 16 == 0 && p.R >= 48 && p.npd_rows == 0) {   // exactly 16 RPL rows: y as a vector, one row per lane less
                switch (p.R / 16) {
                case 3: hipLaunchKernelGGL((celerite_wide2_kernel<3, true>), grid, block, 0, stream, p); break;
                case 4: hipLaunchKernelGGL((celerite_wide2_kernel<4, true>), grid, block, 0, stream, p); break;
                case 5: hipLaunchKernelGGL((celerite_wide2_kernel<5, true>), grid, block, 0, stream, p); break;
                case 6: hipLaunchKernelGGL((celerite_wide2_kernel<6, true>), grid, block, 0, stream, p); break;
                case 7: hipLaunchKernelGGL((celerite_wide2_kernel<7, true>), grid, block, 0, stream, p); break;
                default: hipLaunchKernelGGL((celerite_wide2_kernel<8, true>), grid, block, 0, stream, p); break;
                }
                return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
            }
            switch ((p.R + 1 + 15) / 16) {
            case 1: hipLaunchKernelGGL(celerite_wide2_kernel<1>, grid, block, 0, stream, p); break;
            case 2: hipLaunchKernelGGL(celerite_wide2_kernel<2>, grid, block, 0, stream, p); break;
            case 3: hipLaunchKernelGGL(celerite_wide2_kernel<3>, grid, block, 0, stream, p); break;
            case 4: hipLaunchKernelGGL(celerite_wide2_kernel<4>, grid, block, 0, stream, p); break;
            case 5: hipLaunchKernelGGL(celerite_wide2_kernel<5>, grid, block, 0, stream, p); break;
            case 6: hipLaunchKernelGGL(celerite_wide2_kernel<6>, grid, block, 0, stream, p); break;
            case 7: hipLaunchKernelGGL(celerite_wide2_kernel<7>, grid, block, 0, stream, p); break;
            case 8: hipLaunchKernelGGL(celerite_wide2_kernel<8>, grid, block, 0, stream, p); break;
            default: hipLaunchKernelGGL(celerite_wide2_kernel<9>, grid, block, 0, stream, p); break;
            }
            return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
        }
    }
    if constexpr (MODE == 1 || MODE == 2) {
        // store / simulate on the lean kernel: 96 .. 143 rows (round 4), and from 64 rows on (where the windowed kernels end) unless `no_wide2`
        if (p.tab_draw_stride == 0 && p.npd_rows == 0 && (p.R > 95 || (p.R >= 64 && !(p.opt && p.opt->no_wide2)))) {
            switch ((p.R + 1 + 15) / 16) {
            case 5: hipLaunchKernelGGL((celerite_wide2_kernel<5, false, MODE>), grid, block, 0, stream, p); break;
            case 6: hipLaunchKernelGGL((celerite_wide2_kernel<6, false, MODE>), grid, block, 0, stream, p); break;
            case 7: hipLaunchKernelGGL((celerite_wide2_kernel<7, false, MODE>), grid, block, 0, stream, p); break;
            case 8: hipLaunchKernelGGL((celerite_wide2_kernel<8, false, MODE>), grid, block, 0, stream, p); break;
            default: hipLaunchKernelGGL((celerite_wide2_kernel<9, false, MODE>), grid, block, 0, stream, p); break;
            }
            return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
        }
    }
    if (p.tab_draw_stride != 0) return PIORAN_ERR_UNSUPPORTED;   // per-draw tables: celerite_wide2_kernel (log-likelihood) only
    if (3 * (p.R + 2) + 2 + 3 * p.npd_rows > kWideMaxRecord) return PIORAN_ERR_UNSUPPORTED;   // staged record too long
    if (p.R <= 15) hipLaunchKernelGGL((celerite_wide_kernel<1, MODE>), grid, block, 0, stream, p);
    else if (p.R <= 31) hipLaunchKernelGGL((celerite_wide_kernel<2, MODE>), grid, block, 0, stream, p);
    else if (p.R <= 47) hipLaunchKernelGGL((celerite_wide_kernel<3, MODE>), grid, block, 0, stream, p);
    else if (p.R <= 63) hipLaunchKernelGGL((celerite_wide_kernel<4, MODE>), grid, block, 0, stream, p);
    else if (p.R <= 79) hipLaunchKernelGGL((celerite_wide_kernel<5, MODE>), grid, block, 0, stream, p);
    else if (p.R <= 95) {
        hipLaunchKernelGGL((celerite_wide_kernel<6, MODE>), grid, block, 0, stream, p);
    } else {
        return PIORAN_ERR_UNSUPPORTED;   // 96 .. 143 rows: celerite_wide2_kernel above (log-likelihood only)
    }
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}

int pioran_launch_scan_wide(const ScanParams& p, hipStream_t stream) { return launch_wide_mode<0>(p, stream); }

int pioran_launch_scan_wide_store(const ScanParams& p, hipStream_t stream)
{
    if (!p.st_w || !p.st_d || !p.st_z) return PIORAN_ERR_ARG;
    return launch_wide_mode<1>(p, stream);
}

int pioran_launch_scan_wide_sim(const ScanParams& p, hipStream_t stream)
{
    if (!p.noise || !p.ysim || p.Y) return PIORAN_ERR_ARG;
    return launch_wide_mode<2>(p, stream);
}

static int rpl_of(int R) { return R <= 15 ? 1 : R <= 31 ? 2 : R <= 47 ? 3 : R <= 63 ? 4 : R <= 79 ? 5 : R <= 95 ? 6 : R <= 111 ? 7 : R <= 127 ? 8 : 9; }

// checkpoint interval of the gradient's forward pass: ~2 sqrt(N), between 16 and 256 (memory K + N/K blocks of S per draw)
static int grad_ckpt_every(int64_t N)
{
    int k = 16;
    while (k < 256 && (int64_t)k * k < 4 * N) k *= 2;
    return k;
}

size_t pioran_grad_workspace_doubles(int64_t B, int64_t N, int32_t R)
{
    const size_t rpl = (size_t)rpl_of(R), ns = 16 * rpl;
    const size_t sp = (rpl * rpl + 1) & ~(size_t)1;
    const size_t K = (size_t)grad_ckpt_every(N), nck = (size_t)((N - 1) / (int64_t)K + 1);
    const size_t nstate = rpl * rpl + 3 * rpl + 4;   // per lane; + 3 ns row accumulators per draw
    // two S segment buffers 2 x [B][K][256][sp] | S checkpoints [B][nck][256][sp] | m [B][N][ns] | D [B][N] |
    // state [B][nstate][256] + [B][3 ns] | row accumulators 4 x [B][ns] | scalars [B][4]
    return (size_t)B * ((2 * K + nck) * 256 * sp + (size_t)N * (ns + 1) + nstate * 256 + 3 * ns + 4 * ns + 4) + 2;
}

// p: shared-table launch description with out / status set; work: pioran_grad_workspace_doubles doubles;
// grad_a, grad_b: device [B][J]; grad_c, grad_d: device [B][J] or nullptr; grad_nu, grad_mu: device [B] or nullptr;
// p.g_y / p.g_s2: device [B][N] or nullptr.  aux / ev (5 events): a second stream on which the replay of segment k - 1 runs
// while the adjoint kernel works through segment k (two segment buffers); aux == nullptr: everything on `stream`.
int pioran_launch_scan_wide_grad(ScanParams p, double* work, double* grad_a, double* grad_b, double* grad_c, double* grad_d,
                                 double* grad_nu, double* grad_mu, hipStream_t stream, hipStream_t aux, hipEvent_t* ev)
{
    if (!p.tab || p.R > 143 || p.npd_rows || p.B <= 0 || p.B > 0x7fffffffLL || !grad_a || !grad_b) return PIORAN_ERR_UNSUPPORTED;
    const int rpl = rpl_of(p.R), ns = 16 * rpl;
    p.exp = p.opt ? p.opt->exp : 0;
    if (3 * (p.R + 2) + 2 + 3 * p.npd_rows + ns + 1 > (rpl >= 7 ? 768 : kWideMaxRecord)) return PIORAN_ERR_UNSUPPORTED;
    const size_t sp = (size_t)((rpl * rpl + 1) & ~1), nstate = (size_t)(rpl * rpl + 3 * rpl + 4);
    const int K = grad_ckpt_every(p.N);
    const size_t nck = (size_t)((p.N - 1) / K + 1), B = (size_t)p.B, BN = B * (size_t)p.N;
    p.ckpt_every = K;
    p.st_s = work;                                     // hipMalloc'ed: 256-byte aligned, blocks of an even number of doubles
    const size_t seg_doubles = B * (size_t)K * 256 * sp;
    p.st_ck = p.st_s + 2 * seg_doubles;
    p.st_w = p.st_ck + B * nck * 256 * sp;
    p.st_d = p.st_w + BN * ns;
    p.st_state = p.st_d + BN;
    p.g_al = p.st_state + B * (nstate * 256 + 3 * (size_t)ns);
    p.g_be = p.g_al + B * ns;
    p.g_d = p.g_be + B * ns;
    p.g_c = p.g_d + B * ns;
    p.g_scal = p.g_c + B * ns;
    const dim3 grid((unsigned)p.B), block(256);
    auto run = [&](auto Rc) {
        constexpr int RPL = decltype(Rc)::value;
        // the lean reverse pass (celerite_adjoint2_kernel): always from 7 rows per lane on; below, by option (wide2 = 1: on, no_wide2: off)
        const bool lean = RPL >= 7 || (RPL >= 4 && !(p.opt && p.opt->no_wide2)) || (p.opt && p.opt->wide2);
        if constexpr (RPL >= 7) {
            hipLaunchKernelGGL((celerite_wide2_kernel<RPL, false, 3>), grid, block, 0, stream, p);
        } else if constexpr (RPL >= 4) {   // (64 .. 95 rows: the lean forward kernel is the faster one, 9.0 against 13.1 ms at 80 rows)
            if (lean) hipLaunchKernelGGL((celerite_wide2_kernel<RPL, false, 3>), grid, block, 0, stream, p);
            else hipLaunchKernelGGL((celerite_wide_kernel<RPL, 3>), grid, block, 0, stream, p);
        } else {
            hipLaunchKernelGGL((celerite_wide_kernel<RPL, 3>), grid, block, 0, stream, p);
        }
        // reverse pass, last segment first; segment k holds steps k K + 1 .. min((k + 1) K, N - 1); the first segment's
        // adjoint launch also takes step 0
        const int64_t nseg = (p.N - 1 + K - 1) / K;
        ScanParams q = p;
        q.seg_first = 1;
        if (nseg == 0) {                               // N = 1
            q.seg_n0 = 0; q.seg_hi = 0; q.seg_lo = 0;
            if (lean) hipLaunchKernelGGL(celerite_adjoint2_kernel<RPL>, grid, block, 0, stream, q);
            else hipLaunchKernelGGL(celerite_adjoint_kernel<RPL>, grid, block, 0, stream, q);
        }
        // events: ev[0] forward done; ev[1 + (k & 1)] replay of segment k done; ev[3 + (k & 1)] adjoint of segment k done
        hipStream_t rs = aux ? aux : stream;
        if (aux) {
            (void)hipEventRecord(ev[0], stream);
            (void)hipStreamWaitEvent(aux, ev[0], 0);
        }
        for (int64_t k = nseg - 1; k >= 0; --k) {
            q.seg_n0 = k * K;
            q.seg_hi = (k + 1) * K < p.N - 1 ? (k + 1) * K : p.N - 1;
            q.seg_lo = k == 0 ? 0 : q.seg_n0 + 1;
            q.st_s = p.st_s + (size_t)(k & 1) * seg_doubles;
            if (aux && k + 2 <= nseg - 1) (void)hipStreamWaitEvent(aux, ev[3 + (k & 1)], 0);   // its buffer was read by segment k + 2
            hipLaunchKernelGGL(celerite_replay_kernel<RPL>, grid, block, 0, rs, q);
            if (aux) {
                (void)hipEventRecord(ev[1 + (k & 1)], aux);
                (void)hipStreamWaitEvent(stream, ev[1 + (k & 1)], 0);
            }
            if (lean) hipLaunchKernelGGL(celerite_adjoint2_kernel<RPL>, grid, block, 0, stream, q);
            else hipLaunchKernelGGL(celerite_adjoint_kernel<RPL>, grid, block, 0, stream, q);
            if (aux) (void)hipEventRecord(ev[3 + (k & 1)], stream);
            q.seg_first = 0;
        }
    };
    switch (rpl) {
    case 1: run(ic<1>{}); break;
    case 2: run(ic<2>{}); break;
    case 3: run(ic<3>{}); break;
    case 4: run(ic<4>{}); break;
    case 5: run(ic<5>{}); break;
    case 6: run(ic<6>{}); break;    // 80 .. 95 rows (SHO-40, the dense configuration's model): round 3
    case 7: run(ic<7>{}); break;    // 96 .. 143 rows (the reference grid's j = 64 is 128): round 4
    case 8: run(ic<8>{}); break;
    default: run(ic<9>{}); break;
    }
    hipLaunchKernelGGL(grad_finish_kernel, grid, block, 0, stream, p, ns, grad_a, grad_b, grad_c, grad_d, grad_nu, grad_mu);
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}
