// Dense solver path on gfx950: log_likelihood_direct of the reference (src/direct_solver.jl:6-21).
//
//   K_ik = sum_j exp(-c_j tau) (a_j cos(d_j tau) + b_j sin(d_j tau)),  tau = |t_i - t_k|    :9-14
//          (kappa of src/acvf.jl:138-140 over src/Celerite.jl:42-44)      + Diagonal(sigma2)  :15
//   L = cholesky(K) ; z = L \ y ; return sum(log L_ii) + z'z/2 + N log(2 pi)/2              :16-19
//
// Data layout in HBM: one column-major slab A[ld x Mp], Mp = N rounded up to 64, ld = Mp + 64.
// Rows 0..Mp-1 hold the LOWER triangle of K (padding rows/columns: identity); row Mp holds y'.
// Factorising the augmented matrix [K y; y' .] in place leaves z' = (L^-1 y)' in row Mp, so the
// triangular solve (dtrtrs, :17) is part of the factorisation and costs no extra pass.
//
// Kernels (all fp64):
//   dense_build      one thread per lower-triangle entry: J x (exp + sincos); transcendental/VALU bound;
//                    coalesced 8-byte stores along i (algorithmic bytes: 4 N^2 written once).
//   dense_panel      per 64-column step: the rows below the (already factored) diagonal block, solved
//                    X L^T = A entirely on the matrix cores (blocked, with the 16 x 16 inverses the factor
//                    left in the workspace), results staying in accumulator registers between sub-steps.
//   dense_syrk       trailing update A22 -= P P^T on the matrix cores: v_mfma_f64_16x16x4_f64, one
//                    64 x 64 output tile per wavefront (16 accumulators), operands straight from L2
//                    in the MFMA A/B fragment layout; the product is oriented so that the C/D
//                    fragment's lane index runs along the memory-contiguous row index.
//                    Workgroup 0 is the critical path: four waves share the tile that is the NEXT diagonal
//                    block and factor it in LDS right away (factor_block64: DPP column sweeps for the
//                    16-column sub-panels, MFMA for the in-block update; reports the first non-positive pivot).
//                    MFMA bound: N^3/3 flop at N = 4096 => 2.3e10 flop (the only dense contraction here).
//   dense_finish     logdet + z'z reduction -> +NLL.
#include "../../include/pioran_hip.h"
#include "common.h"

#include <cmath>
#include <type_traits>
#include <utility>

namespace {

constexpr int NB = 64;

// Several matrices of the same size in ONE launch of every kernel below (blockIdx.z = matrix): slab z at A + z slab (its workspace
// behind it, as for one matrix), coefficients at a + z ab_stride, c + z cd_stride, mu[z] / nu[z] when the arrays are given, info[z],
// out[z].  All zero / nullptr with gridDim.z = 1: one matrix, the scalar mu / nu arguments.
struct DenseBatch {
    int64_t slab, ab_stride, cd_stride;
    const double* mu;
    const double* nu;
};
constexpr int kPairThreshold = 24;   // trailing tiles per side above which steps are taken in pairs (128-deep updates)
using f64x4 = __attribute__((ext_vector_type(4))) double;

__global__ void __launch_bounds__(256) dense_build_kernel(int64_t N, int64_t Mp, int64_t ld, int32_t J,
                                                          const double* __restrict__ a, const double* __restrict__ b,
                                                          const double* __restrict__ c, const double* __restrict__ d,
                                                          const double* __restrict__ t, const double* __restrict__ s2,
                                                          const double* __restrict__ y, double* __restrict__ A, int diag_only,
                                                          double mu, double nu, DenseBatch bt)
{
    {
        const int64_t mz = blockIdx.z;
        a += mz * bt.ab_stride; b += mz * bt.ab_stride; c += mz * bt.cd_stride; d += mz * bt.cd_stride; A += mz * bt.slab;
        if (bt.mu) mu = bt.mu[mz];
        if (bt.nu) nu = bt.nu[mz];
    }
    if (diag_only && (blockIdx.x >> 2) != (blockIdx.y >> 2)) return;  // keep only the diagonal 64 x 64 tiles
    // 16 x 16 tile of (i, k); i is the fast index (threadIdx.x) = memory-contiguous
    const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x & 15);
    const int64_t k = (int64_t)blockIdx.y * 16 + (threadIdx.x >> 4);
    if ((int64_t)blockIdx.x < (int64_t)blockIdx.y) return;  // tile strictly above the diagonal
    if (i >= Mp || k >= Mp) {
        return;
    }
    double v;
    if (i < N && k < N) {
        if (i < k) return;
        const double tau = fabs(t[i] - t[k]);
        v = 0.0;
        for (int j = 0; j < J; ++j) {
            double sn, cs;
            sincos(d[j] * tau, &sn, &cs);
            v += exp(-c[j] * tau) * (a[j] * cs + b[j] * sn);   // src/Celerite.jl:42-44, summed as acvf.jl:138-140
        }
        if (i == k) v = fma(nu, s2[i], v);                      // src/direct_solver.jl:15 (nu = 1: the data set's variances)
    } else {
        if (i < k) return;
        v = (i == k) ? 1.0 : 0.0;                               // identity padding
    }
    A[i + k * ld] = v;
    if (i == k) A[Mp + k * ld] = k < N ? y[k] - mu : 0.0;       // the y row (mean subtracted: src/scalable_GP.jl:164)
}

// Covariance of an arbitrary list of times, a NaN time marking an identity (padding) row/column; no y row.
// Used for the augmented matrix [[K(t,t) + diag(s2), K(t,tau)], [K(tau,t), K(tau,tau)]] of predict_cov.
__global__ void __launch_bounds__(256) dense_build_aug_kernel(int64_t Mtot, int64_t ld, int32_t J, const double* __restrict__ a,
                                                              const double* __restrict__ b, const double* __restrict__ c,
                                                              const double* __restrict__ d, const double* __restrict__ te,
                                                              const double* __restrict__ s2e, double* __restrict__ A)
{
    if ((int64_t)blockIdx.x < (int64_t)blockIdx.y) return;  // tile strictly above the diagonal
    const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x & 15);
    const int64_t k = (int64_t)blockIdx.y * 16 + (threadIdx.x >> 4);
    if (i >= Mtot || k >= Mtot || i < k) return;
    const double ti = te[i], tk = te[k];
    double v;
    if (isnan(ti) || isnan(tk)) {
        v = (i == k) ? 1.0 : 0.0;
    } else {
        const double tau = fabs(ti - tk);
        v = 0.0;
        for (int j = 0; j < J; ++j) {
            double sn, cs;
            sincos(d[j] * tau, &sn, &cs);
            v += exp(-c[j] * tau) * (a[j] * cs + b[j] * sn);   // src/Celerite.jl:42-44, summed as acvf.jl:138-140
        }
        if (i == k) v += s2e[i];
    }
    A[i + k * ld] = v;
}

// ---- fast covariance build for sorted time stamps -------------------------------------------------------------
// Off-diagonal 64 x 64 tile (rows i0.., columns k0.., i0 >= k0 + 64, t ascending => tau = t_i - t_k >= 0).
// With t_ref = t[i0] both factors of exp(-c tau) = exp(-c (t_i - t_ref)) exp(-c (t_ref - t_k)) are <= 1 (no
// overflow; underflow only where the true value underflows), and with the angle-addition formulas
//   k_j(tau) = E_i F_k [ (a Ck - b Sk) Ci + (a Sk + b Ck) Si ] = P_i G_k + Q_i H_k
// where P_i = E_i cos(d t_i), Q_i = E_i sin(d t_i), G_k = F_k (a Ck - b Sk), H_k = F_k (a Sk + b Ck).
// Per tile: 128 J (exp + sincos) instead of 4096 J, then 2 FMAs per (i, k, term).  The absolute angles d t are
// reduced by the device library's full-range sincos; their rounding (|d t| 2^-53) is the same size as the
// rounding of d * tau in the direct evaluation.
constexpr int BT = 64;    // build tile
constexpr int BJ = 16;    // terms per LDS chunk
__global__ void __launch_bounds__(256) dense_build_fast_kernel(int64_t N, int64_t ld, int32_t J,
                                                               const double* __restrict__ a, const double* __restrict__ b,
                                                               const double* __restrict__ c, const double* __restrict__ d,
                                                               const double* __restrict__ t, double* __restrict__ A, DenseBatch bt)
{
    {
        const int64_t mz = blockIdx.z;
        a += mz * bt.ab_stride; b += mz * bt.ab_stride; c += mz * bt.cd_stride; d += mz * bt.cd_stride; A += mz * bt.slab;
    }
    __shared__ double Ps[BJ][BT], Qs[BJ][BT], Gs[BJ][BT], Hs[BJ][BT];
    // linear block id -> strictly-lower tile pair (ti > tj)
    const int bid = blockIdx.x;
    int ti = (int)((sqrt(8.0 * bid + 1.0) + 1.0) * 0.5);
    while ((int64_t)ti * (ti - 1) / 2 > bid) --ti;
    while ((int64_t)(ti + 1) * ti / 2 <= bid) ++ti;
    const int tj = bid - (int)((int64_t)ti * (ti - 1) / 2);
    const int64_t i0 = (int64_t)ti * BT, k0 = (int64_t)tj * BT;
    const int tid = threadIdx.x;
    const double tref = t[i0 < N ? i0 : N - 1];
    // each thread owns rows (tid & 63) and columns 4 * (tid >> 6) + {0..3} + 16 m, m = 0..3  -> 16 entries
    const int li = tid & 63, lc = tid >> 6;
    double acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0;
    for (int j0 = 0; j0 < J; j0 += BJ) {
        __syncthreads();
        // 2 * 64 * BJ precomputations by 256 threads
        for (int e = tid; e < 2 * BT * BJ; e += 256) {
            const int jj = e / (2 * BT), rr = e % (2 * BT);
            const int j = j0 + jj;
            if (j >= J) {
                if (rr < BT) { Ps[jj][rr] = 0.0; Qs[jj][rr] = 0.0; } else { Gs[jj][rr - BT] = 0.0; Hs[jj][rr - BT] = 0.0; }
                continue;
            }
            if (rr < BT) {
                const int64_t i = i0 + rr;
                const double ti_ = t[i < N ? i : N - 1];
                double sn, cs;
                sincos(d[j] * ti_, &sn, &cs);
                const double E = exp(-c[j] * (ti_ - tref));
                Ps[jj][rr] = E * cs;
                Qs[jj][rr] = E * sn;
            } else {
                const int64_t k = k0 + rr - BT;
                const double tk = t[k < N ? k : N - 1];
                double sn, cs;
                sincos(d[j] * tk, &sn, &cs);
                const double F = exp(-c[j] * (tref - tk));
                Gs[jj][rr - BT] = F * (a[j] * cs - b[j] * sn);
                Hs[jj][rr - BT] = F * (a[j] * sn + b[j] * cs);
            }
        }
        __syncthreads();
#pragma unroll 4
        for (int jj = 0; jj < BJ; ++jj) {
            const double pi = Ps[jj][li], qi = Qs[jj][li];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kk = 16 * m + 4 * lc + q;
                    acc[4 * m + q] = fma(pi, Gs[jj][kk], fma(qi, Hs[jj][kk], acc[4 * m + q]));
                }
        }
    }
    const int64_t i = i0 + li;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t k = k0 + 16 * m + 4 * lc + q;
            A[i + k * ld] = (i < N && k < N) ? acc[4 * m + q] : 0.0;   // identity padding: off-diagonal zeros
        }
}

// ---- covariance build of a BATCH with shared (c, d) (late round 3) ----------------------------------------------------------------
// What depends on (c, d, t) alone — every exp and sincos — is computed once per tile for all matrices of the launch; the matrices differ
// in (a_j, b_j), mu and nu only.
// Off-diagonal tiles: P, Q (row side) and FC = F cos(d t_k), FS = F sin(d t_k) (column side) of ALL terms stay in LDS (4 x 64 J doubles:
// J <= 64); per matrix z the tile is the product over the 2 J-long axis  K'[k][i] = sum_j G_z[j][k] P[j][i] + H_z[j][k] Q[j][i],
// G_z = a_zj FC - b_zj FS, H_z = a_zj FS + b_zj FC, on the matrix cores: the A operand (column side) is formed in registers from two LDS
// reads, the product is oriented so that the result's lane index runs along i, the memory-contiguous index of the slab.
__global__ void __launch_bounds__(256) dense_build_fast_batch_kernel(int64_t N, int64_t ld, int32_t J, int32_t nbatch,
                                                                     const double* __restrict__ a, const double* __restrict__ b,
                                                                     const double* __restrict__ c, const double* __restrict__ d,
                                                                     const double* __restrict__ t, double* __restrict__ A, DenseBatch bt)
{
    extern __shared__ double bsh[];
    const int JP = (J + 3) & ~3;                      // terms padded to whole k-steps of the 16x16x4 product
    double* Ps = bsh;                                 // [JP][64]
    double* Qs = Ps + JP * BT;
    double* FCs = Qs + JP * BT;
    double* FSs = FCs + JP * BT;
    const int bid = blockIdx.x;
    int ti = (int)((sqrt(8.0 * bid + 1.0) + 1.0) * 0.5);
    while ((int64_t)ti * (ti - 1) / 2 > bid) --ti;
    while ((int64_t)(ti + 1) * ti / 2 <= bid) ++ti;
    const int tj = bid - (int)((int64_t)ti * (ti - 1) / 2);
    const int64_t i0 = (int64_t)ti * BT, k0 = (int64_t)tj * BT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
    const double tref = t[i0 < N ? i0 : N - 1];
    for (int e = tid; e < 2 * BT * JP; e += 256) {
        const int j = e / (2 * BT), rr = e % (2 * BT);
        double v0 = 0.0, v1 = 0.0;
        if (j < J) {
            const int64_t n = rr < BT ? i0 + rr : k0 + rr - BT;
            const double tn = t[n < N ? n : N - 1];
            double sn, cs;
            sincos(d[j] * tn, &sn, &cs);
            const double E = exp(rr < BT ? -c[j] * (tn - tref) : -c[j] * (tref - tn));
            v0 = n < N ? E * cs : 0.0;                // (identity padding: rows / columns past N contribute zeros)
            v1 = n < N ? E * sn : 0.0;
        }
        if (rr < BT) { Ps[j * BT + rr] = v0; Qs[j * BT + rr] = v1; } else { FCs[j * BT + rr - BT] = v0; FSs[j * BT + rr - BT] = v1; }
    }
    __syncthreads();
    const int nks = JP >> 2;
    for (int z = 0; z < nbatch; ++z) {
        const double* az = a + (int64_t)z * bt.ab_stride;
        const double* bz = b + (int64_t)z * bt.ab_stride;
        f64x4 acc[4];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) acc[ib] = f64x4{0.0, 0.0, 0.0, 0.0};
        for (int ks = 0; ks < nks; ++ks) {
            const int j = 4 * ks + lk;
            const double aj = j < J ? az[j] : 0.0, bj = j < J ? bz[j] : 0.0;
            const double fc = FCs[j * BT + 16 * wave + lr], fs = FSs[j * BT + 16 * wave + lr];
            const double g = fma(aj, fc, -bj * fs), h = fma(aj, fs, bj * fc);     // A operand: [m = column k][k-step = term]
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                acc[ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(g, Ps[j * BT + 16 * ib + lr], acc[ib], 0, 0, 0);
                acc[ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(h, Qs[j * BT + 16 * ib + lr], acc[ib], 0, 0, 0);
            }
        }
        // result register g of lane (lk, lr): column k0 + 16 wave + lk + 4 g, row i0 + 16 ib + lr
        double* Az = A + (int64_t)z * bt.slab;
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int g = 0; g < 4; ++g) Az[(i0 + 16 * ib + lr) + (k0 + 16 * wave + lk + 4 * g) * ld] = acc[ib][g];
    }
}

// Diagonal 64 x 64 tiles (direct evaluation, as dense_build_kernel's) for up to ZC matrices per pass: one thread per entry, the
// exp / sincos of a term once, then two FMAs per matrix; also the y row (y - mu_z), nu_z sigma2 on the diagonal and the identity padding.
constexpr int ZC = 16;
__global__ void __launch_bounds__(256) dense_build_diag_batch_kernel(int64_t N, int64_t Mp, int64_t ld, int32_t J, int32_t nbatch,
                                                                     const double* __restrict__ a, const double* __restrict__ b,
                                                                     const double* __restrict__ c, const double* __restrict__ d,
                                                                     const double* __restrict__ t, const double* __restrict__ s2,
                                                                     const double* __restrict__ y, double* __restrict__ A, DenseBatch bt,
                                                                     double mu0, double nu0)
{
    // blockIdx.x: 16-row block, blockIdx.y: 16-column block within the same 64-tile (4 x 4 per diagonal tile), blockIdx.z: pass of ZC matrices
    const int64_t tile = blockIdx.x >> 2;
    const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x & 15);
    const int64_t k = tile * 64 + (int64_t)blockIdx.y * 16 + (threadIdx.x >> 4);
    const int z0 = blockIdx.z * ZC;
    const int nz = nbatch - z0 < ZC ? nbatch - z0 : ZC;
    if (i >= Mp || k >= Mp || i < k) return;
    double v[ZC];
#pragma unroll
    for (int z = 0; z < ZC; ++z) v[z] = 0.0;
    const bool data = i < N && k < N;
    if (data) {
        const double tau = fabs(t[i] - t[k]);
        for (int j = 0; j < J; ++j) {
            double sn, cs;
            sincos(d[j] * tau, &sn, &cs);
            const double e = exp(-c[j] * tau);
            const double e1 = e * cs, e2 = e * sn;
#pragma unroll
            for (int z = 0; z < ZC; ++z)
                if (z < nz) v[z] = fma(a[(int64_t)(z0 + z) * bt.ab_stride + j], e1, fma(b[(int64_t)(z0 + z) * bt.ab_stride + j], e2, v[z]));
        }
    }
#pragma unroll
    for (int z = 0; z < ZC; ++z) {
        if (z >= nz) break;
        double* Az = A + (int64_t)(z0 + z) * bt.slab;
        const double mu = bt.mu ? bt.mu[z0 + z] : mu0, nu = bt.nu ? bt.nu[z0 + z] : nu0;
        double val = data ? v[z] : (i == k ? 1.0 : 0.0);
        if (data && i == k) val = fma(nu, s2[i], val);
        Az[i + k * ld] = val;
        if (i == k) Az[Mp + k * ld] = k < N ? y[k] - mu : 0.0;
    }
}

template <int I>
using icd = std::integral_constant<int, I>;
template <class F, int... Is>
__device__ __forceinline__ void static_for16_impl(F&& f, std::integer_sequence<int, Is...>)
{
    (f(icd<Is>{}), ...);
}
template <class F>
__device__ __forceinline__ void static_for16(F&& f)
{
    static_for16_impl(f, std::make_integer_sequence<int, 16>{});
}

// 1/sqrt(x) to fp64 accuracy: v_rsq_f64 + ONE third-order step, y (1 + e/2 + 3e^2/8) with e = 1 - x y^2.
// Four dependent operations instead of the six of two Newton steps: this sits on the per-column critical path of
// the diagonal-tile factorisation.  (|e| <= 2^-22 after v_rsq_f64 leaves a relative error ~ e^3 < 2^-66.)
__device__ __forceinline__ double rsqrt_f64(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y, y, 1.0);
    return fma(y * e, fma(0.375, e, 0.5), y);
}

__device__ __forceinline__ f64x4 mfma4(double a, double b, f64x4 c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

constexpr int LP = NB + 1;
// workspace written by whoever factors a diagonal block, read by the following panel kernel:
//   ws[0 .. 1023]   inv(L_ss), s = 0..3, row-major 16 x 16 each
[[maybe_unused]] constexpr int WS_DOUBLES = 4 * 16 * 16;

// Diagnostic hooks: compiled out in the product; tools/factor_probe.hip defines PIORAN_STAMP to s_memtime stamps.
#ifndef PIORAN_STAMP
#define PIORAN_STAMP(i)
#endif

// lane N of each 16-lane DPP row -> all lanes of that row
template <int N>
__device__ __forceinline__ double bcast16(double x)
{
    return __builtin_amdgcn_mov_dpp(x, 0x150 + N, 0xf, 0xf, true);
}

// ---- the 64 x 64 diagonal factor (round 6) -----------------------------------------------------------------------------------------
// 1 / x to fp64 accuracy: v_rcp_f64 + two Newton steps.
__device__ __forceinline__ double recip_f64d(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}

// The 16 x 16 LDL' + L^-1 of one diagonal tile as an in-place Gauss-Jordan sweep on DPP broadcasts — the form of the windowed celerite
// kernels (window_common.h), restated here for a tile that lives in the dense slab.  Lane (q, n) holds column n of the (symmetric) tile
// in m[0..15]; the four DPP rows q hold copies.  Step P: d_P = m[P] of lane P; mult_n = -m[P]_n / d_P (lane P: -2);
// m[j]_n += bcast_P(m[j]) mult_n for j > P — the rank-1 update of the reduced tile for n > P, the row operation on the identity for
// n < P, and lane P's own column becomes -L_jP d_P: column P of L^-1 scaled by d_P.  ONE v_fmac_f64_dpp per (P, j): 120 for the
// factorisation AND the inverse (the column sweep of rounds 1-5 spent 240 plus sixteen rsqrt chains on the same two results).
// Afterwards lane n holds   m[j]_n = L_nj d_j (j < n: row n of L D, frozen when step j passed it),  m[n]_n = d_n,
// m[j]_n = d_n (L^-1)_jn (j > n).  The next pivot's multiplier chain (broadcast, v_rcp_f64, third-order correction: four dependent DP
// operations) is spread between the updates of the current step, which do not depend on it.
// DPP hazard (two wait states between a VALU write of a register and a DPP read of it; not interlocked): the DPP source of every update
// is the row's own register m[j], last written by the PREVIOUS step's update of that row — at least the seven instructions of the
// multiplier chain earlier — so the updates need no s_nop (window_common.h's form pays one per group of four: 168 s_nop per sweep in
// the ISA); only the broadcast of the next pivot, which reads the register the instruction before it wrote, keeps its s_nop 1.
// multiplier -m / d without a finished reciprocal: r0 = v_rcp_f64(d), e = 1 - d r0, t0 = -m r0, mult = t0 (1 + e + e^2) (relative error e^3 < 1e-22)
__device__ __forceinline__ double gj_mult_of(double dn, double mrow)
{
    double r0, e, t0, pq, mn;
    asm volatile("v_rcp_f64 %0, %1" : "=v"(r0) : "v"(dn));
    asm volatile("s_nop 0\n\tv_fma_f64 %0, -%2, %3, 1.0\n\tv_mul_f64 %1, -%4, %2" : "=&v"(e), "=&v"(t0) : "v"(r0), "v"(dn), "v"(mrow));
    asm volatile("v_fma_f64 %0, %1, %1, %1" : "=v"(pq) : "v"(e));
    asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(mn) : "v"(t0), "v"(pq));
    return mn;
}
// one elimination step = ONE inline-assembly statement (ldl_steps.inc, generated by tools/gen_ldl_steps.py; shared with the windowed celerite kernels):
// as separate statements per piece the compiler put an `s_nop 0` at most of their boundaries (111 s_nop per sweep, on a chain that waits for every one)
#include "ldl_steps.inc"
template <int P>
__device__ __forceinline__ void gj_sweep(double (&m)[16], double& mult, int c16)
{
    if constexpr (P < 16) {
        ldl_step_fused<P>(m, mult, c16);
        gj_sweep<P + 1>(m, mult, c16);
    }
}

// Factor the 64 x 64 block held row-major (padded) in LDS `Ls`, in place, by a FOUR-wave workgroup (tid 0..255): on return the lower
// triangle holds the Cholesky factor C, the four 16 x 16 inverses inv(C_ss) (exact zeros above their diagonals) are parked in the
// block's strictly-upper part (s = 0, 1, 2: rows 0 .. 15, columns 16 (s + 1) ..; s = 3: rows 16 .. 31, columns 32 ..) and copied to
// `ws` (global, row-major 16 x 16 each).  Only the lower triangle of the input is read.  Returns the 1-based index of the first
// non-positive pivot (0 = none) to every thread.  `flag` is one int of LDS.
//
// Round s (16-column sub-panel s), ONE workgroup barrier per round:
//   wave 0      the dependent chain and nothing else: Gauss-Jordan sweep of the tile (s, s) (gj_sweep: d, L D and d L^-1 in registers),
//               raw result to LDS (G[s]), barrier; then — its inputs are its own or a round old — Y = L^-1 A~(s+1, s)' (4 matrix
//               instructions), tile (s+1, s+1) -= Y' D^-1 Y (4 more), and on to the next sweep.  No square root and no scaling on the chain.
//   waves 1..3  after the barrier, beside the chain's next sweep: the outputs of sub-panel s — C_ss = (L D)(n, j) d_j^-1/2 and
//               inv(C_ss) = d_i^-1/2 L^-1 straight from G[s]; X_t = A~(t, s) inv(C_ss)' (t > s) on the matrix cores, kept in registers and
//               stored in place after the NEXT barrier (other waves read the raw tile meanwhile) — and the right-looking update of the tiles
//               (rt, ct), ct > s, except (s+1, s+1), every wave forming the X tiles it needs itself (no second barrier).
// Measured (tools/factor_probe.hip, profiles/r06_factor_probe.txt): see there; rounds 1-5: 20 750 cycles with a 64 x 16 column sweep per
// sub-panel (3100 cycles each) and three barriers per round.
constexpr int GS = 18;       // row pitch of a raw Gauss-Jordan result in LDS
__device__ __forceinline__ int factor_block64(double* __restrict__ Ls, double* __restrict__ ws, int* __restrict__ flag,
                                              int tid)
{
    __shared__ double Gsh[4][16 * GS];
    const int lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    int bad = 0;
    f64x4 xown = f64x4{0.0, 0.0, 0.0, 0.0};     // waves 1..3: X(wave, s-1), stored in place after the barrier of round s
    // A operand of the solves: L^-1 [row lr][k = 4 ks + lk] (unit lower triangular), 1 / d_k and d_k^-1/2 (k = 4 ks + lk), d_lr^-1/2, from the
    // raw result of a sweep.  Nine independent Newton chains (four reciprocals, five reciprocal square roots), written LEVEL BY LEVEL
    // with scheduling fences in between: one wavefront per SIMD issues in order, and the compiler's own order finished one chain
    // before it started the next (five dependent DP operations each) and issued the LDS reads one by one between them.
    auto load_linv = [&](auto want_rs, const double* G, double (&li)[4], double (&idv)[4], double (&rs)[4], double& rsl, double& dl) {
        constexpr int NS = decltype(want_rs)::value ? 5 : 0;     // the chain wavefront needs no square roots
        double lv[4], dk[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kk = 4 * ks + lk;
            lv[ks] = G[kk * GS + lr];
            dk[ks] = G[kk * GS + kk];
        }
        dl = G[lr * GS + lr];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        double x[5], y[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, e[5], u[5];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) x[ks] = dk[ks];
        x[4] = dl;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) idv[ks] = __builtin_amdgcn_rcp(dk[ks]);
#pragma unroll
        for (int i = 0; i < NS; ++i) y[i] = __builtin_amdgcn_rsq(x[i]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) e[ks] = fma(-dk[ks], idv[ks], 1.0);
#pragma unroll
        for (int i = 0; i < NS; ++i) u[i] = x[i] * y[i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) idv[ks] = fma(e[ks], idv[ks], idv[ks]);
#pragma unroll
        for (int i = 0; i < NS; ++i) u[i] = fma(-u[i], y[i], 1.0);          // e = 1 - x y^2
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) e[ks] = fma(-dk[ks], idv[ks], 1.0);
#pragma unroll
        for (int i = 0; i < NS; ++i) { x[i] = y[i] * u[i]; u[i] = fma(0.375, u[i], 0.5); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) idv[ks] = fma(e[ks], idv[ks], idv[ks]);
#pragma unroll
        for (int i = 0; i < NS; ++i) y[i] = fma(x[i], u[i], y[i]);          // y (1 + e/2 + 3 e^2/8): rsqrt_f64's third-order step
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kk = 4 * ks + lk;
            li[ks] = kk < lr ? lv[ks] * idv[ks] : (kk == lr ? 1.0 : 0.0);     // the column arrives scaled by d_k
            rs[ks] = NS ? y[ks] : 0.0;
        }
        rsl = NS ? y[4] : 0.0;
    };
    // op' (16 x 16, rows = result rows) times the raw tile (t, s): register g of lane (lk, lr) = result[lr][lk + 4 g]
    auto solve_tile = [&](int t, int c0, const double (&aop)[4]) -> f64x4 {
        double bq[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bq[ks] = Ls[(16 * t + lr) * LP + c0 + 4 * ks + lk];
        f64x4 x = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) x = mfma4(aop[ks], bq[ks], x);
        return x;
    };
    // tile (rt, ct) -= X_rt X_ct'
    auto update_tile = [&](int rt, int ct, const f64x4& xr, const f64x4& xc) {
        f64x4 c;
#pragma unroll
        for (int g = 0; g < 4; ++g) c[g] = Ls[(16 * rt + lr) * LP + 16 * ct + lk + 4 * g];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) c = mfma4(-xc[ks], xr[ks], c);
#pragma unroll
        for (int g = 0; g < 4; ++g) Ls[(16 * rt + lr) * LP + 16 * ct + lk + 4 * g] = c[g];
    };
    static_for16([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s < 4) {
            constexpr int c0 = 16 * s;
            constexpr int ir0 = s == 3 ? 16 : 0, ic0 = s == 3 ? 32 : 16 * (s + 1);   // where inv(C_ss) is parked
            double* G = Gsh[s];
            PIORAN_STAMP(8 * s + 0);
            if (wave == 0) {
                double m[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) m[j] = Ls[(c0 + (j >= lr ? j : lr)) * LP + c0 + (j >= lr ? lr : j)];   // lower triangle only
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                double d0;
                asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "=v"(d0) : "v"(m[0]));
                asm volatile("s_nop 0");
                double mult = gj_mult_of(d0, m[0]);
                mult = lr == 0 ? -2.0 : mult;
                gj_sweep<0>(m, mult, lr);
                PIORAN_STAMP(8 * s + 1);
                if (lk == 0) {
                    typedef double d2 __attribute__((ext_vector_type(2)));
                    d2* dst = reinterpret_cast<d2*>(G + lr * GS);
#pragma unroll
                    for (int j = 0; j < 16; j += 2) dst[j / 2] = d2{m[j], m[j + 1]};
                }
            }
            __syncthreads();      // round s: G[s] visible; every wave's work of round s-1 finished and visible
            PIORAN_STAMP(8 * s + 2);
            if (wave == 0) {
                if constexpr (s < 3) {
                    constexpr int d0_ = 16 * (s + 1);
                    // (both tiles were last written by the other wavefronts in round s-1: readable only after this round's barrier)
                    f64x4 c;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int cc = lk + 4 * g;                           // symmetric tile: the lower triangle is the valid one
                        c[g] = Ls[(d0_ + (lr >= cc ? lr : cc)) * LP + d0_ + (lr >= cc ? cc : lr)];
                    }
                    double bq[4];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) bq[ks] = Ls[(d0_ + lr) * LP + c0 + 4 * ks + lk];
                    double li[4], idv[4], rs_[4], rsl_, dl_;
                    load_linv(std::false_type{}, G, li, idv, rs_, rsl_, dl_);
                    f64x4 y = f64x4{0.0, 0.0, 0.0, 0.0};                    // Y[c = lk + 4 g][r = lr] = (L^-1 A~(s+1, s)')
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) y = mfma4(li[ks], bq[ks], y);
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) c = mfma4(-y[ks], y[ks] * idv[ks], c);
#pragma unroll
                    for (int g = 0; g < 4; ++g) Ls[(d0_ + lr) * LP + d0_ + lk + 4 * g] = c[g];    // both triangles valid from here on
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            } else {
                if (s >= 1 && wave >= s) {                                   // X(wave, s-1) of the previous round, in place
#pragma unroll
                    for (int g = 0; g < 4; ++g) Ls[(16 * wave + lr) * LP + (c0 - 16) + lk + 4 * g] = xown[g];
                }
                double li[4], idv[4], rs[4], aop[4], rsl, dl;
                double ldv[4];                                               // (L D)[lr][lk + 4 g]: row lr of the factor before its scaling
#pragma unroll
                for (int g = 0; g < 4; ++g) ldv[g] = G[lr * GS + lk + 4 * g];
                load_linv(std::true_type{}, G, li, idv, rs, rsl, dl);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) aop[ks] = li[ks] * rsl;      // inv(C_ss)[lr][4 ks + lk]
                if (wave == 1) {
                    const unsigned long long neg = __ballot(!(dl > 0.0)) & 0xffffull;
                    if (neg && !bad) bad = c0 + __ffsll((long long)neg);
                }
                if (wave == (s == 0 ? 3 : 1)) {                              // the sub-panel's own outputs
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) Ls[(ir0 + lr) * LP + ic0 + 4 * ks + lk] = aop[ks];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int j = lk + 4 * g;                            // C_ss[lr][j] = (L D)[lr][j] d_j^-1/2 (j < lr), d^1/2 (j = lr)
                        if (j <= lr) Ls[(c0 + lr) * LP + c0 + j] = ldv[g] * rs[g];
                    }
                }
                if constexpr (s == 0) {
                    if (wave == 1) {
                        const f64x4 x1 = solve_tile(1, c0, aop), x2 = solve_tile(2, c0, aop), x3 = solve_tile(3, c0, aop);
                        update_tile(2, 1, x2, x1);
                        update_tile(3, 1, x3, x1);
                        xown = x1;
                    } else if (wave == 2) {
                        const f64x4 x2 = solve_tile(2, c0, aop), x3 = solve_tile(3, c0, aop);
                        update_tile(2, 2, x2, x2);
                        update_tile(3, 2, x3, x2);
                        xown = x2;
                    } else {
                        const f64x4 x3 = solve_tile(3, c0, aop);
                        update_tile(3, 3, x3, x3);
                        xown = x3;
                    }
                } else if constexpr (s == 1) {
                    if (wave == 2) {
                        const f64x4 x2 = solve_tile(2, c0, aop), x3 = solve_tile(3, c0, aop);
                        update_tile(3, 2, x3, x2);
                        xown = x2;
                    } else if (wave == 3) {
                        const f64x4 x3 = solve_tile(3, c0, aop);
                        update_tile(3, 3, x3, x3);
                        xown = x3;
                    }
                } else if constexpr (s == 2) {
                    if (wave == 3) xown = solve_tile(3, c0, aop);
                }
            }
            PIORAN_STAMP(8 * s + 3);
        }
    });
    if (tid == 64) *flag = bad;
    __syncthreads();
    bad = *flag;
    // the four inverses -> workspace (row-major 16 x 16 each), 4 entries per thread, coalesced
    {
        double v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = e * 256 + tid, sb = idx >> 8, i = (idx >> 4) & 15, j = idx & 15;
            v[e] = Ls[((sb == 3 ? 16 : 0) + i) * LP + (sb == 3 ? 32 : 16 * (sb + 1)) + j];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) ws[e * 256 + tid] = v[e];
    }
    PIORAN_STAMP(40);
    return bad;
}

// factored block: LDS -> global by 256 threads: lane = row (rows contiguous across lanes), wave = 16-column strip.
// The strictly-upper part of the slab is scratch (never read as data), so the block is stored without masking.
__device__ __forceinline__ void store_block_lower(const double* __restrict__ Ls, double* __restrict__ blk, int64_t ld, int tid)
{
    const int lane = tid & 63, h = tid >> 6;
    double v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = Ls[lane * LP + 16 * h + q];
#pragma unroll
    for (int q = 0; q < 16; ++q) blk[lane + (int64_t)(16 * h + q) * ld] = v[q];
}

// First diagonal block (no trailing update precedes it): load, factor, write back.
// (blockIdx.x = 1, 2, only launched on the one-launch-per-column path: the raw tiles (1, 0), (2, 0) -> the snapshot of step 0.)
__global__ void __launch_bounds__(256) dense_diag0_kernel(double* __restrict__ A, int64_t ld, double* __restrict__ ws,
                                                          int32_t* __restrict__ info, DenseBatch bt)
{
    A += (int64_t)blockIdx.z * bt.slab; ws += (int64_t)blockIdx.z * bt.slab; info += blockIdx.z;
    __shared__ double Ls[NB * LP];
    __shared__ int flag;
    const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
    if (blockIdx.x > 0) {
        double* sn = ws + WS_DOUBLES + (size_t)(blockIdx.x - 1) * (NB * NB);
        const double* T = A + (int64_t)blockIdx.x * NB;
#pragma unroll
        for (int q = 0; q < 16; ++q) sn[lane + (16 * h + q) * NB] = T[lane + (int64_t)(16 * h + q) * ld];
        return;
    }
    {
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = A[lane + (int64_t)(16 * h + q) * ld];
#pragma unroll
        for (int q = 0; q < 16; ++q) Ls[lane * LP + 16 * h + q] = v[q];
    }
    __syncthreads();
    const int bad = factor_block64(Ls, ws, &flag, tid);
    __syncthreads();
    store_block_lower(Ls, A, ld, tid);
    if (tid == 0 && bad && *info == 0) *info = bad;
}

// ---- panel solve, all on the matrix cores --------------------------------------------------------------------
// Rows below the (already factored) diagonal block, 16 RT rows per wave, kept TRANSPOSED in MFMA C/D layout
// (D[row = column-in-subpanel c][col = row r], so the lane index runs along memory-contiguous rows):
//   Y_s = inv(L_ss) (A_s' - sum_{s' < s} L[s][s'] Y_s')
// A operands (blocks of L, the inverses from `ws`) come straight from L2.  Key identity of the f64 layouts:
// C/D register g of lane l holds D[k = (l>>4) + 4g][col = l&15] and the B operand of k-step ks wants
// B[k = 4ks + (l>>4)][col = l&15]: with g = ks these are the same element, so a result tile is fed back as the
// next B operand with no data movement.
template <int RT>   // 16-row tiles per wave
__global__ void __launch_bounds__(256) dense_panel_kernel(double* __restrict__ A, int64_t ld, int64_t kb,
                                                          const double* __restrict__ ws, DenseBatch bt)
{
    A += (int64_t)blockIdx.z * bt.slab; ws += (int64_t)blockIdx.z * bt.slab;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const int64_t row0 = kb + NB + ((int64_t)blockIdx.x * 4 + wave) * (16 * RT);
    if (row0 + 16 * RT > ld) return;   // past the last tile of the slab (the one holding the y row)
    const double* Lb = A + kb + kb * ld;
    double* P = A + row0 + kb * ld;
    f64x4 Y[4][RT];   // [s][row tile]: D[row = c][col = r]
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int g = 0; g < 4; ++g) Y[s][rt][g] = P[(16 * rt + lr) + (int64_t)(16 * s + lk + 4 * g) * ld];
    double lop[4][4][4];   // [s][sp][ks]: -L[16s + lr][16sp + 4ks + lk]   (sp < s)
    double iop[4][4];      // [s][ks]:     inv(L_ss)[lr][4ks + lk]
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int sp = 0; sp < s; ++sp)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) lop[s][sp][ks] = -Lb[(16 * s + lr) + (int64_t)(16 * sp + 4 * ks + lk) * ld];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) iop[s][ks] = ws[(s * 16 + lr) * 16 + 4 * ks + lk];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int sp = 0; sp < s; ++sp)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) Y[s][rt] = mfma4(lop[s][sp][ks], Y[sp][rt][ks], Y[s][rt]);
        f64x4 Z[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) Z[rt] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) Z[rt] = mfma4(iop[s][ks], Y[s][rt][ks], Z[rt]);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            Y[s][rt] = Z[rt];
#pragma unroll
            for (int g = 0; g < 4; ++g) P[(16 * rt + lr) + (int64_t)(16 * s + lk + 4 * g) * ld] = Z[rt][g];
        }
    }
}

// ---- trailing update on the matrix cores --------------------------------------------------------------------
// Tile (ti, tj), ti >= tj, of 64 x 64 entries of the trailing matrix (origin j0 = kb + NB):
//   C[i][j] -= sum_p P[i][p] P[j][p],   P = A[:, kb : kb+NB]
// v_mfma_f64_16x16x4_f64: A-operand lane l holds X[row l&15][k l>>4], B-operand lane l holds Y[k l>>4][col l&15],
// result reg g of lane l is D[row (l>>4) + 4g][col l&15]  (f64 layout, cdna_hip_programming.md section 3).
// With X[row][k] = P[j][k] and Y[k][col] = P[i][k], D[row][col] = (P P^T)[i][j]: col = l&15 runs along i,
// the memory-contiguous index of the column-major slab, so C loads/stores are 128-byte segments.
// KP = 1: the trailing update of one 64-column step.  KP = 2: TWO adjacent, finished panels (columns kb .. kb + 127) applied
// in one pass to the trailing matrix that starts at kb + 128 — the read-modify-write of C, which bounds the early steps
// (each 64 x 64 x 64 tile moves 128 KB through L2 for 0.5 Mflop), happens once per 128 columns instead of once per 64.
// col0_only (KP = 1): only block column 0 of the trailing matrix (the next panel) and its diagonal tile — the narrow
// update between the two panel solves of a pair.
// One 64 x 64 tile (ti, tj), ti >= tj, of the trailing matrix with origin j0:  C -= P_i P_j',  P = A[:, pc : pc + 64 KP]  (one wavefront).
template <int KP, int NBUF = 3>   // NBUF: k-steps of operand fragments in flight
__device__ __forceinline__ void syrk_tile(double* __restrict__ A, int64_t ld, int64_t pc, int64_t j0, int ti, int tj, int lr, int lk)
{
    // Row permutation inside the tile: MFMA strip s (s = 0..3) takes the rows 32 (s >> 1) + 2 r + (s & 1), r = 0..15,
    // instead of 16 s + r.  Lane (lr, lk) then needs rows 2 lr and 2 lr + 1 of each half of the tile — ADJACENT in the
    // column-major slab — for strips 2h and 2h + 1: one 16-byte load feeds two operand fragments, and the C entries of
    // the strip pair (2h, 2h + 1) are adjacent too.  128 vector-memory instructions per tile instead of 256: with eight
    // waves per CU the texture path was as busy as the matrix pipe.  (All addresses are even multiples of 8 bytes:
    // j0, NB and ld are multiples of 64.)
    typedef double d2 __attribute__((ext_vector_type(2)));
    const double* Pi = A + (j0 + (int64_t)ti * NB) + pc * ld + 2 * lr + (int64_t)lk * ld;  // rows of the i tile
    const double* Pj = A + (j0 + (int64_t)tj * NB) + pc * ld + 2 * lr + (int64_t)lk * ld;  // rows of the j tile
    f64x4 acc[4][4];  // [jb][ib]
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) acc[jb][ib] = f64x4{0.0, 0.0, 0.0, 0.0};

    // operand fragments one k-step (4 panel columns: 4 + 4 doubles per lane) at a time, three buffers deep: the
    // loads of k-step ks+2 are issued before the 16 MFMAs (1024 issue cycles) of k-step ks.  With the 128
    // accumulator registers this stays under 256 registers per lane, so two wavefronts share a SIMD and the
    // second one hides whatever latency is left (launch bounds below).
    double xa[NBUF][4], yb[NBUF][4];   // [buffer][strip]
    auto load_kstep = [&](int ks, int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t off = 32 * h + (int64_t)(4 * ks) * ld;
            const d2 x = *reinterpret_cast<const d2*>(Pj + off);   // X[row][k] = P[j][k], rows 32h + 2lr + {0, 1}
            const d2 y = *reinterpret_cast<const d2*>(Pi + off);   // Y[k][col] = P[i][k]
            xa[buf][2 * h] = x.x; xa[buf][2 * h + 1] = x.y;
            yb[buf][2 * h] = y.x; yb[buf][2 * h + 1] = y.y;
        }
    };
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b) load_kstep(b, b);
#pragma unroll
    for (int ks = 0; ks < 16 * KP; ++ks) {
        if (ks + NBUF - 1 < 16 * KP) load_kstep(ks + NBUF - 1, (ks + NBUF - 1) % NBUF);
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
                acc[jb][ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[ks % NBUF][jb], yb[ks % NBUF][ib], acc[jb][ib], 0, 0, 0);
    }

    // C -= acc.  D row (l>>4) + 4g of strip jb is the j-row 32 (jb >> 1) + 2 (lk + 4g) + (jb & 1); D column l&15 of strip
    // ib is the i-row 32 (ib >> 1) + 2 lr + (ib & 1): the strip pair (2h, 2h + 1) is one 16-byte access.  One j strip
    // (8 accesses per lane) at a time: its loads are all in flight before the first store.
    double* C = A + (j0 + (int64_t)ti * NB) + (j0 + (int64_t)tj * NB) * ld + 2 * lr;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        d2 cv[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                cv[h][g] = *reinterpret_cast<const d2*>(C + 32 * h + (int64_t)(32 * (jb >> 1) + 2 * (lk + 4 * g) + (jb & 1)) * ld);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                d2 v = cv[h][g];
                v.x -= acc[jb][2 * h][g];
                v.y -= acc[jb][2 * h + 1][g];
                *reinterpret_cast<d2*>(C + 32 * h + (int64_t)(32 * (jb >> 1) + 2 * (lk + 4 * g) + (jb & 1)) * ld) = v;
            }
    }
}

// Half of such a tile — the j rows 32 h .. 32 h + 31 (MFMA strips 2h, 2h + 1) against all 64 i rows — per wavefront: half the matrix work
// per work item (256 KP instructions) for the launches in which whole tiles would leave SIMDs idle and make the slowest wavefront the
// launch's length (a lone 128-deep tile-wave takes 20 us; dense_step_kernel's bulk role).  The B operand is loaded by both halves.
template <int KP, int NBUF>
__device__ __forceinline__ void syrk_half_tile(double* __restrict__ A, int64_t ld, int64_t pc, int64_t j0, int ti, int tj, int h, int lr, int lk)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    const double* Pi = A + (j0 + (int64_t)ti * NB) + pc * ld + 2 * lr + (int64_t)lk * ld;           // rows of the i tile
    const double* Pj = A + (j0 + (int64_t)tj * NB) + 32 * h + pc * ld + 2 * lr + (int64_t)lk * ld;  // this half's rows of the j tile
    f64x4 acc[2][4];  // [strip 2h + jj][ib]
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) acc[jj][ib] = f64x4{0.0, 0.0, 0.0, 0.0};
    double xa[NBUF][2], yb[NBUF][4];
    auto load_kstep = [&](int ks, int buf) {
        const d2 x = *reinterpret_cast<const d2*>(Pj + (int64_t)(4 * ks) * ld);
        xa[buf][0] = x.x; xa[buf][1] = x.y;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const d2 y = *reinterpret_cast<const d2*>(Pi + 32 * hh + (int64_t)(4 * ks) * ld);
            yb[buf][2 * hh] = y.x; yb[buf][2 * hh + 1] = y.y;
        }
    };
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b) load_kstep(b, b);
#pragma unroll
    for (int ks = 0; ks < 16 * KP; ++ks) {
        if (ks + NBUF - 1 < 16 * KP) load_kstep(ks + NBUF - 1, (ks + NBUF - 1) % NBUF);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
                acc[jj][ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[ks % NBUF][jj], yb[ks % NBUF][ib], acc[jj][ib], 0, 0, 0);
    }
    double* C = A + (j0 + (int64_t)ti * NB) + (j0 + (int64_t)tj * NB) * ld + 2 * lr;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        d2 cv[2][4];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                cv[hh][g] = *reinterpret_cast<const d2*>(C + 32 * hh + (int64_t)(32 * h + 2 * (lk + 4 * g) + jj) * ld);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                d2 v = cv[hh][g];
                v.x -= acc[jj][2 * hh][g];
                v.y -= acc[jj][2 * hh + 1][g];
                *reinterpret_cast<d2*>(C + 32 * hh + (int64_t)(32 * h + 2 * (lk + 4 * g) + jj) * ld) = v;
            }
    }
}

template <int KP>
__global__ void __launch_bounds__(256, 2) dense_syrk_kernel(double* __restrict__ A, int64_t ld, int64_t kb, int64_t Mp,
                                                         double* __restrict__ ws, int32_t* __restrict__ info, int factor_next,
                                                         int col0_only, DenseBatch bt)
{
    A += (int64_t)blockIdx.z * bt.slab; ws += (int64_t)blockIdx.z * bt.slab; info += blockIdx.z;
    // Workgroups of four wavefronts.  Workgroup 0 is the critical path: its four waves share tile (0,0) — the NEXT
    // diagonal block — one 16-column strip each, keep the updated tile in LDS and factor it right away
    // (factor_block64) while the other tiles are still being updated; the next panel kernel then starts from a
    // finished diagonal block.  Every other workgroup takes four tiles, one per wavefront.
    __shared__ double Ls[NB * LP];
    __shared__ int flag;
    const int64_t j0 = kb + NB * KP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;

    if (blockIdx.x == 0) {
        f64x4 acc[4];   // [ib], column strip jb = wave
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) acc[ib] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int pp = 0; pp < KP; ++pp) {
            const double* P0 = A + j0 + (kb + NB * pp) * ld + lr + (int64_t)lk * ld;   // rows of the next diagonal block, panel columns
            double xa[16], yb[16][4];
#pragma unroll
            for (int k4 = 0; k4 < 16; ++k4) {
                xa[k4] = P0[wave * 16 + (int64_t)(4 * k4) * ld];
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) yb[k4][ib] = P0[ib * 16 + (int64_t)(4 * k4) * ld];
            }
#pragma unroll
            for (int k4 = 0; k4 < 16; ++k4)
#pragma unroll
                for (int ib = 0; ib < 4; ++ib)
                    acc[ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[k4], yb[k4][ib], acc[ib], 0, 0, 0);
        }
        double* C = A + j0 + j0 * ld + lr + (int64_t)lk * ld;
        double cv[4][4];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int g = 0; g < 4; ++g) cv[ib][g] = C[ib * 16 + (int64_t)(wave * 16 + 4 * g) * ld];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                Ls[(ib * 16 + lr) * LP + wave * 16 + lk + 4 * g] = cv[ib][g] - acc[ib][g];   // row i, column j
        __syncthreads();
        // (partial factorisation, predict_cov: the last step leaves its Schur complement unfactored)
        const int bad = factor_next ? factor_block64(Ls, ws, &flag, tid) : 0;
        __syncthreads();
        store_block_lower(Ls, A + j0 + j0 * ld, ld, tid);
        if (tid == 0 && bad && *info == 0) *info = (int32_t)(j0 + bad);
        return;
    }

    // i tiles 0..nt (the last one holds the y row Mp and 63 scratch rows of the slab), j tiles 0..nt-1, i >= j
    const int nt = (int)((Mp - j0) / NB) + 1;
    // linear tile id -> (ti, tj) with ti >= tj; tile 0 belongs to workgroup 0
    const int bid = ((int)blockIdx.x - 1) * 4 + wave + 1;
    int ti, tj;
    if (col0_only) {   // block column 0 only: tiles (1 .. nt-1, 0)
        ti = bid;
        tj = 0;
    } else {
        ti = (int)((sqrt(8.0 * bid + 1.0) - 1.0) * 0.5);
        while ((int64_t)(ti + 1) * (ti + 2) / 2 <= bid) ++ti;
        while ((int64_t)ti * (ti + 1) / 2 > bid) --ti;
        tj = bid - (int)((int64_t)ti * (ti + 1) / 2);
    }
    if (ti >= nt || tj >= nt - 1) return;   // wave-uniform; no workgroup barrier below
    syrk_tile<KP>(A, ld, kb, j0, ti, tj, lr, lk);
}

// One 64 x 64 tile per WORKGROUP (dense_step_kernel's bulk role):  C(ti, tj) -= P_i P_j',  P = A[:, pc : pc + 64].  The i rows of the panel
// (B operand of all four waves) go through LDS once, column-major with a row pitch of 80 doubles (the four k-lanes of a fragment read land
// in disjoint bank groups); wave w takes the 16 j rows 16 w .. of the tile (A operand straight from L2) and all 64 i rows: 64 matrix
// instructions per wave, every SIMD of the CU busy on the same tile.  Against one tile per wavefront: half the operand traffic, and the
// launch's work is balanced per CU, not per SIMD (a lone tile-wave leaves its SIMD's matrix pipe idle half of the time, two on a SIMD
// take 25 us together).
constexpr int BP = 80;
__device__ __forceinline__ void syrk_tile_wg(double* __restrict__ A, int64_t ld, int64_t pc, int64_t j0, int ti, int tj,
                                             double* __restrict__ Bs, int tid)
{
    const int lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
    const int64_t i0 = j0 + (int64_t)ti * NB, jr0 = j0 + (int64_t)tj * NB;
    {
        const double* Pi = A + i0 + lane + (pc + 16 * wave) * ld;
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = Pi[(int64_t)q * ld];
#pragma unroll
        for (int q = 0; q < 16; ++q) Bs[(16 * wave + q) * BP + lane] = v[q];
    }
    double xa[16];
    {
        const double* Pj = A + jr0 + 16 * wave + lr + (pc + lk) * ld;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) xa[ks] = Pj[(int64_t)(4 * ks) * ld];
    }
    double* C = A + i0 + lr + (jr0 + 16 * wave + lk) * ld;
    double cv[4][4];
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int g = 0; g < 4; ++g) cv[ib][g] = C[16 * ib + (int64_t)(4 * g) * ld];
    __syncthreads();
    f64x4 acc[4];
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) acc[ib] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) acc[ib] = mfma4(xa[ks], Bs[(4 * ks + lk) * BP + 16 * ib + lr], acc[ib]);
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int g = 0; g < 4; ++g) C[16 * ib + (int64_t)(4 * g) * ld] = cv[ib][g] - acc[ib][g];
}

// ---- one launch per block column (round 4; one matrix) ---------------------------------------------------------------------------------
// The chain above pays two kernel boundaries per 64 columns (panel, update + factor) and, in its paired steps, leaves the chip idle while
// the 13.9 us narrow launches run.  Here step k is ONE launch in which four roles run side by side, none waiting for another:
//   CRIT   workgroup 0, row tile k+1: solves its 64 rows of panel k against L_kk (four waves x 16 rows, on the matrix cores, the four
//          16 x 16 inverses read from the strictly-upper part of the diagonal block, where the factor parks them), applies panel k to the
//          next diagonal tile (k+1, k+1) and factors it at once (factor_block64).  Nothing else is on the chain.
//   DIAG2  workgroup 1: brings the diagonal tile AFTER next, (k+2, k+2), up to panel k (panel k-1 from memory, panel k from its own
//          re-solved rows of tile (k+2, k)), so that the next launch's CRIT finds it one panel short only.
//   STRIP  one workgroup per row tile i = k+2 .. nb (nb = the tile of the y row): solves its rows of panel k (final L rows), RE-solves the
//          rows of tile (k+1, k) — 40 dependent matrix instructions per wave instead of a kernel boundary — and applies panels k-1 and k
//          to tile (i, k+1): block column k+1 leaves the launch complete, ready to be panel k+1.
//   BULK   one wavefront per tile (i, j), j >= k+2 (except (k+2, k+2)): applies panel k-1, which the PREVIOUS launch finished — the bulk
//          update lags one step, so it never waits for the chain: the work that used to sit between the chain's launches rides along.
// Two tiles of block column k are solved IN PLACE by one workgroup while others re-solve them: (k+1, k) and (k+2, k).  Their raw form is
// therefore kept as a snapshot (two 64 x 64 tiles behind the slab, double-buffered by the parity of k), written by whoever produced the
// tile: the STRIP workgroups of rows k+1, k+2 of the previous launch, dense_diag0_kernel for k = 0.
// Invariant at launch k: diagonal block k factored; block column k updated through panel k-1; block columns >= k+1 through panel k-2,
// the diagonal tile (k+1, k+1) through k-1.  1 + nb launches instead of 2.5 nb; chain per step: launch + 40 + 64 MFMAs + factor.
// (rows: the 16 x 64 raw rows to solve, column stride lds — in place, or the snapshot of a tile another workgroup overwrites meanwhile)
__device__ __forceinline__ void solve_rows16_load(const double* __restrict__ A, int64_t ld, int64_t kb, const double* __restrict__ rows,
                                                  int64_t lds, int lr, int lk, f64x4 (&Y)[4], double (&lop)[4][4][4], double (&iop)[4][4])
{
    const double* Lb = A + kb + kb * ld;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int g = 0; g < 4; ++g) Y[s][g] = rows[lr + (int64_t)(16 * s + lk + 4 * g) * lds];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int sp = 0; sp < s; ++sp)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) lop[s][sp][ks] = -Lb[(16 * s + lr) + (int64_t)(16 * sp + 4 * ks + lk) * ld];
        const int ir0 = s == 3 ? 16 : 0, ic0 = s == 3 ? 32 : 16 * (s + 1);   // where factor_block64 parks inv(L_ss)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) iop[s][ks] = Lb[(ir0 + lr) + (int64_t)(ic0 + 4 * ks + lk) * ld];
    }
}
// Y_s = inv(L_ss) (A_s' - sum_{s' < s} L[s][s'] Y_s'), tiles transposed in the C/D layout (dense_panel_kernel)
__device__ __forceinline__ void solve_rows16_compute(f64x4 (&Y)[4], const double (&lop)[4][4][4], const double (&iop)[4][4])
{
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int sp = 0; sp < s; ++sp)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) Y[s] = mfma4(lop[s][sp][ks], Y[sp][ks], Y[s]);
        f64x4 Z = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) Z = mfma4(iop[s][ks], Y[s][ks], Z);
        Y[s] = Z;
    }
}

// The three solving roles.  it: row tile of this workgroup (wave w takes its rows 16 w ..); qt: the row tile whose solved rows are the other
// operand; the target tile is (it, qt).  CRIT: it = qt = k+1, no panel k-1 term, result factored.  DIAG2: it = qt = k+2, nothing of panel k
// stored.  STRIP: qt = k+1, the solved rows stored.
constexpr int SNAP_TILE = NB * NB;                        // snapshot tile: column-major 64 x 64, column stride 64
template <bool CRIT>
__device__ __forceinline__ void step_solve_role(double* __restrict__ A, int64_t ld, int64_t Mp, int k, int it, int qt, bool store_rows,
                                                int nprev, double* __restrict__ Ls, double* __restrict__ Xq, int* flag,
                                                int32_t* __restrict__ info, int tid)
{
    const int64_t kb = (int64_t)k * NB, i0 = (int64_t)it * NB, q0 = (int64_t)qt * NB;
    const int lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
    const bool has_target = qt < (int)(Mp / NB);        // a block column qt exists (else: only the y rows are left to solve)
    double* snap = A + (size_t)ld * (size_t)Mp + WS_DOUBLES;                   // [parity][which of the two tiles][64 x 64]
    const double* snap_k = snap + (size_t)(k & 1) * 2 * SNAP_TILE;             // raw (k+1, k) and (k+2, k)
    f64x4 acc[4];                                        // tile (it, qt), transposed: [jb] D[row = j][col = i], i = 16 wave + lr
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) acc[jb] = f64x4{0.0, 0.0, 0.0, 0.0};
    if constexpr (!CRIT) {
        if (has_target) {
            // the nprev finished panels k - nprev .. k - 1 this tile still lacks (the bulk update lags by up to two): all 80 operand loads
            // of a panel in flight at once (one exposure of the L2 / HBM latency), before the solve's operands take their registers
#pragma unroll 1
            for (int pp = nprev; pp >= 1; --pp) {
                const double* Pj = A + q0 + (kb - pp * NB) * ld + lr + (int64_t)lk * ld;              // rows of tile qt
                const double* Pi = A + i0 + 16 * wave + (kb - pp * NB) * ld + lr + (int64_t)lk * ld;  // this wave's 16 rows
                double xa[16][4], yb[16];
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    yb[ks] = Pi[(int64_t)(4 * ks) * ld];
#pragma unroll
                    for (int jb = 0; jb < 4; ++jb) xa[ks][jb] = Pj[16 * jb + (int64_t)(4 * ks) * ld];
                }
#pragma unroll
                for (int ks = 0; ks < 16; ++ks)
#pragma unroll
                    for (int jb = 0; jb < 4; ++jb) acc[jb] = mfma4(xa[ks][jb], yb[ks], acc[jb]);
            }
            __builtin_amdgcn_sched_barrier(0);           // (the loads below stay below: 160 + 112 operand registers do not fit together)
        }
    }
    f64x4 Yi[4];
    double lop[4][4][4], iop[4][4];
    if (store_rows)   // in place: this workgroup is the only one that reads these raw rows from the slab, and it overwrites them
        solve_rows16_load(A, ld, kb, A + i0 + 16 * wave + kb * ld, ld, lr, lk, Yi, lop, iop);
    else              // DIAG2: tile (k+2, k) is being overwritten by its STRIP workgroup
        solve_rows16_load(A, ld, kb, snap_k + SNAP_TILE + 16 * wave, NB, lr, lk, Yi, lop, iop);
    solve_rows16_compute(Yi, lop, iop);
    if (store_rows) {   // the solved rows are final entries of L
        double* P = A + i0 + 16 * wave + kb * ld;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int g = 0; g < 4; ++g) P[lr + (int64_t)(16 * s + lk + 4 * g) * ld] = Yi[s][g];
    }
    if (!has_target) return;                            // (workgroup-uniform)
    if (it == qt) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int g = 0; g < 4; ++g) Xq[(wave * 16 + s * 4 + g) * 64 + lane] = Yi[s][g];
    } else {                                            // the rows of tile (k+1, k), re-solved here (same L operands) from their snapshot
        f64x4 Yq[4];
        const double* P = snap_k + 16 * wave;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int g = 0; g < 4; ++g) Yq[s][g] = P[lr + (16 * s + lk + 4 * g) * NB];
        solve_rows16_compute(Yq, lop, iop);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int g = 0; g < 4; ++g) Xq[(wave * 16 + s * 4 + g) * 64 + lane] = Yq[s][g];
    }
    double* C = A + i0 + 16 * wave + lr + q0 * ld;
    double cv[4][4];
    if constexpr (CRIT) {                               // the diagonal tile: its load overlaps the exchange (issued any earlier it delays
                                                        // the solve's own operands: 16.5 instead of 15.5 us per step, tools/dense_roles.py)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int g = 0; g < 4; ++g) cv[jb][g] = C[(int64_t)(16 * jb + lk + 4 * g) * ld];
    }
    __syncthreads();
    // panel k: A operand X[row = j][k = c] = the solved rows of tile qt (C/D register (s, g) IS the operand of k-step (s, g)),
    // B operand Y[k = c][col = i] = this wave's own solved rows
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) acc[jb] = mfma4(Xq[(jb * 16 + s * 4 + g) * 64 + lane], Yi[s][g], acc[jb]);
    if constexpr (!CRIT) {
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int g = 0; g < 4; ++g) cv[jb][g] = C[(int64_t)(16 * jb + lk + 4 * g) * ld];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int g = 0; g < 4; ++g) cv[jb][g] -= acc[jb][g];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int g = 0; g < 4; ++g) C[(int64_t)(16 * jb + lk + 4 * g) * ld] = cv[jb][g];
        if (store_rows && it - qt <= 2) {               // rows k+2, k+3 of block column k+1: the next launch re-solves them elsewhere
            double* sn = snap + (size_t)((k + 1) & 1) * 2 * SNAP_TILE + (size_t)(it - qt - 1) * SNAP_TILE + 16 * wave + lr;
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                for (int g = 0; g < 4; ++g) sn[(16 * jb + lk + 4 * g) * NB] = cv[jb][g];
        }
    } else {
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int g = 0; g < 4; ++g) Ls[(16 * wave + lr) * LP + 16 * jb + lk + 4 * g] = cv[jb][g] - acc[jb][g];   // row i, column j
        __syncthreads();
        double* ws = A + (size_t)ld * (size_t)Mp;        // (the copy of the inverses the older kernels read; unused on this path)
        const int bad = factor_block64(Ls, ws, flag, tid);
        __syncthreads();
        store_block_lower(Ls, A + q0 + q0 * ld, ld, tid);
        if (tid == 0 && bad && *info == 0) *info = (int32_t)(q0 + bad);
    }
}

// ---- prototype (round 4, DenseOptions::old_chain == 5): the chain without kernel boundaries -------------------------------------------
// One persistent workgroup (dense_crit_chain_kernel, on a second stream) runs the CRIT role of every step; the step launches carry DIAG2,
// the strips and the bulk only.  Hand-offs through flags in device memory (release: fence + atomic add; acquire: atomic load + fence):
//   factored[k]  posted by CRIT(k-1) once block k is factored and its inverses are parked — DIAG2 and the strips of launch k wait for it;
//   ready[k]     posted by STRIP(row k+1) and DIAG2 of launch k-1 (they leave tile (k+1, k) and the diagonal tile (k+1, k+1) one panel
//                short) — CRIT(k) waits for both.
// Only the bulk-side consumers wait on the chain's flag; the chain waits for two workgroups that started a whole step earlier.  Every wait
// is bounded (kFlagSpinLimit polls): on expiry the workgroup reports info = -7 and leaves, so a lost hand-off cannot hang the device.
constexpr int kFlagSpinLimit = 1 << 19;
constexpr int FLG_FACTORED = 0, FLG_READY = 128;
__device__ __forceinline__ bool flag_wait(int* f, int need, int32_t* info, int* sh_ok, int tid, bool fence = true)
{
    if (tid == 0) {
        int it = 0;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need && ++it < kFlagSpinLimit) __builtin_amdgcn_s_sleep(1);
        *sh_ok = it < kFlagSpinLimit;
        if (it >= kFlagSpinLimit) *info = -7;
    }
    __syncthreads();
    if (fence) __threadfence();                          // acquire (every wavefront: its own view of L1 / L2)
    return *sh_ok != 0;
}
__device__ __forceinline__ void flag_post(int* f, int tid, bool fence = true)
{
    if (fence) __threadfence();                          // release: this workgroup's stores are visible device-wide before the flag is
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(f, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// What the bulk role of one launch does: panels pcb .. pcb + kp - 1 (block columns; all final) onto the tiles (ti >= tj) of the trailing matrix
// whose origin is block column org, tiles in COLUMN order (id -> (tj, ti)), ids id0 .. id0 + cnt - 1 (id 0 = tile (0, 0): DIAG2's, never listed).
constexpr int kStepBulkBuffers = 6;   // operand k-steps in flight in dense_step_kernel's bulk tiles (mostly one wavefront per SIMD there)
struct StepBulk {
    int pcb, kp, org, id0, cnt;   // cnt: work items (tiles, or half tiles when `halves`)
    int halves;                   // 1: two wavefronts per tile (syrk_half_tile)
};
template <int KP>   // depth of the bulk role's update in panels (one instantiation per depth: both tile bodies in one kernel cost 14 spilled registers)
__global__ void __launch_bounds__(256, 2) dense_step_kernel(double* __restrict__ A, int64_t ld, int64_t Mp, int k,
                                                            int32_t* __restrict__ info, int bx0, int nprev, StepBulk bulk, int* flg)
{
    __shared__ double Sh[NB * LP + 4 * 16 * 64];
    double* Ls = Sh;                     // the diagonal tile being factored (CRIT)
    double* Xq = Sh + NB * LP;           // the solved rows of the operand tile: [piece of 16 rows][s * 4 + g][lane]
    __shared__ int flag;
    __shared__ int sh_ok;
    const int nb = (int)(Mp / NB);
    const int tid = threadIdx.x;
    const int nstrip = nb - k - 1;                      // row tiles k+2 .. nb
    // Workgroup 256 of a launch is a BLANK: the dispatcher hands the first 256 workgroups one CU each and starts the second round on the first
    // CU again, i.e. beside workgroup 0 — the chain, whose sweeps then share their SIMDs' DP pipe with a neighbour's matrix instructions
    // (15.5 -> 20.7 us per step in the launches of 257 .. 512 workgroups).  The blank exits at once and the chain keeps its CU to itself
    // unless the launch has more than 512 workgroups (where the bulk bounds the step anyway).
    if (bx0 == 0 && blockIdx.x == 256) return;
    const int bx = (int)blockIdx.x - (bx0 == 0 && blockIdx.x > 256 ? 1 : 0) + bx0;   // (bx0 != 0: timing experiments, tools/dense_roles.py)
    // nprev: finished panels a tile of block column k+1 (k+2 for DIAG2) still lacks when this launch starts (the bulk role lags: 1 in the
    // one-panel schedule, 1 or 2 in the paired one)
    if (bx == 0) {
        // the chain: its wavefronts share their SIMDs with bulk tile-waves, whose fp64 matrix instructions occupy the same DP pipe for 64
        // cycles each (tools/mfma_probe.hip: an MFMA stream starves a VALU wavefront beside it) — highest issue priority for this workgroup
        __builtin_amdgcn_s_setprio(3);
        step_solve_role<true>(A, ld, Mp, k, k + 1, k + 1, true, 0, Ls, Xq, &flag, info, tid);
    } else if (bx == 1) {
        if (flg && k >= 1 && !flag_wait(flg + FLG_FACTORED + k, 1, info, &sh_ok, tid)) return;
        if (k + 2 < nb) step_solve_role<false>(A, ld, Mp, k, k + 2, k + 2, false, nprev, Ls, Xq, &flag, info, tid);
        if (flg) flag_post(flg + FLG_READY + k + 1, tid);
    } else if (bx < 2 + nstrip) {
        if (flg && k >= 1 && !flag_wait(flg + FLG_FACTORED + k, 1, info, &sh_ok, tid)) return;
        step_solve_role<false>(A, ld, Mp, k, k + bx, k + 1, true, nprev, Ls, Xq, &flag, info, tid);
        if (flg && bx == 2) flag_post(flg + FLG_READY + k + 1, tid);
    } else {
        // ---- BULK: one tile per wavefront (schedule: dense_nll_impl) -----------------------------------------------------------------------
        // Measured alternatives to this role, none kept (profiles/r04_dense_steps_wg_tiles.txt, r04_dense_roles.txt): one tile per WORKGROUP
        // (syrk_tile_wg: 1.37 instead of 1.29 ms — a workgroup has too little matrix work to hide its load -> LDS -> product -> C round trips
        // at two workgroups per CU); the tile's accumulators started from C (no read-modify-write epilogue: the tile's 128 addresses cost the
        // registers of the operand pipeline, 8 spills, bulk alone 5 % slower).
        const int64_t j0 = (int64_t)bulk.org * NB;
        const int nt = (int)((Mp - j0) / NB) + 1;       // i tiles (the last one holds the y row), j tiles 0 .. nt-2
        const int lane = tid & 63, wave = tid >> 6;
        const int w = (bx - 2 - nstrip) * 4 + wave;
        if (w >= bulk.cnt) return;                      // wave-uniform
        int id = bulk.id0 + (bulk.halves ? w >> 1 : w), tj = 0;
        while (id >= nt - tj) { id -= nt - tj; ++tj; }  // column tj holds rows tj .. nt-1
        const int ti = tj + id;
        if (bulk.halves) syrk_half_tile<KP, kStepBulkBuffers>(A, ld, (int64_t)bulk.pcb * NB, j0, ti, tj, w & 1, lane & 15, lane >> 4);
        else syrk_tile<KP, kStepBulkBuffers>(A, ld, (int64_t)bulk.pcb * NB, j0, ti, tj, lane & 15, lane >> 4);
    }
}

#ifdef PIORAN_EXPERIMENTS
// The persistent chain of the prototype: CRIT(0 .. nb-1) in one workgroup.  Measured slower than the launched chain (1.55 against 1.24 ms,
// docs/EXPERIMENTS.md section 11.7): compiled only into experiment builds (-DPIORAN_EXPERIMENTS), never into the product library.
// (launched with kChainPadLds bytes of dynamic LDS it never touches: no workgroup of a step launch then fits beside it, and the chain keeps
//  its CU's DP pipes to itself — what the blank workgroup 256 achieves for the launched chain)
constexpr int kChainPadLds = 56 * 1024;
__global__ void __launch_bounds__(256, 1) dense_crit_chain_kernel(double* __restrict__ A, int64_t ld, int64_t Mp, int32_t* __restrict__ info, int* flg, int mode)
{
    __shared__ double Sh[NB * LP + 4 * 16 * 64];
    double* Ls = Sh;
    double* Xq = Sh + NB * LP;
    __shared__ int flag;
    __shared__ int sh_ok;
    const int nb = (int)(Mp / NB);
    const int tid = threadIdx.x;
    __builtin_amdgcn_s_setprio(3);
    for (int k = 0; k < nb; ++k) {
        // mode (timing experiments; results may then be stale): bit 0 no release fence here, bit 1 no acquire fence here
        if (k >= 1 && !flag_wait(flg + FLG_READY + k, 2, info, &sh_ok, tid, !(mode & 2))) return;
        step_solve_role<true>(A, ld, Mp, k, k + 1, k + 1, true, 0, Ls, Xq, &flag, info, tid);
        flag_post(flg + FLG_FACTORED + k + 1, tid, !(mode & 1));
    }
}
#endif

__global__ void __launch_bounds__(256) dense_finish_kernel(const double* __restrict__ A, int64_t ld, int64_t N,
                                                           int64_t Mp, double* __restrict__ out,
                                                           const int32_t* __restrict__ info, DenseBatch bt)
{
    A += (int64_t)blockIdx.z * bt.slab; out += blockIdx.z; info += blockIdx.z;
    __shared__ double s1[256], s2[256];
    double ld_ = 0.0, zz = 0.0;
    for (int64_t k = threadIdx.x; k < N; k += 256) {
        ld_ += log(A[k + k * ld]);           // logdet(L.U) = sum log L_ii    :19
        const double z = A[Mp + k * ld];
        zz = fma(z, z, zz);
    }
    s1[threadIdx.x] = ld_;
    s2[threadIdx.x] = zz;
    __syncthreads();
    for (unsigned s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            s1[threadIdx.x] += s1[threadIdx.x + s];
            s2[threadIdx.x] += s2[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double nll = s1[0] + 0.5 * s2[0] + 0.5 * (double)N * 1.8378770664093454836;
        *out = *info ? (double)NAN : nll;
    }
}

// predict_direct (src/direct_solver.jl:75-119): the data y as row Mtot of the augmented slab (the panel steps turn it into
// z = L^-1 y, as in the likelihood path), and afterwards mean_m = sum_k X[m][k] z[k] with X = K(tau,t) L^-T left in the
// tau rows by the same panel steps.
__global__ void __launch_bounds__(256) dense_set_yrow_kernel(double* __restrict__ A, int64_t ld, int64_t Mtot, int64_t N,
                                                             const double* __restrict__ y)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < N) A[Mtot + k * ld] = y[k];
}

__global__ void __launch_bounds__(256) dense_predict_mean_kernel(const double* __restrict__ A, int64_t ld, int64_t Mtot,
                                                                 int64_t Mp, int64_t M, double* __restrict__ mean,
                                                                 const int32_t* __restrict__ info)
{
    __shared__ double red[256];
    const int64_t m = blockIdx.x;
    double acc = 0.0;
    for (int64_t k = threadIdx.x; k < Mp; k += 256) acc = fma(A[(Mp + m) + k * ld], A[Mtot + k * ld], acc);
    red[threadIdx.x] = acc;
    __syncthreads();
    for (unsigned s_ = 128; s_ > 0; s_ >>= 1) {
        if (threadIdx.x < s_) red[threadIdx.x] += red[threadIdx.x + s_];
        __syncthreads();
    }
    if (threadIdx.x == 0 && m < M) mean[m] = *info ? (double)NAN : red[0];
}

}  // namespace

// K must hold ld * Mp + PIORAN_DENSE_WS doubles with Mp = roundup(N, 64), ld = Mp + 64 (slab + inverse workspace + tile snapshots).
static void launch_build(int64_t N, int64_t Mp, int64_t ld, int32_t J, const double* a, const double* b, const double* c,
                         const double* d, const double* t, const double* y, const double* s2, double* K, int sorted,
                         hipStream_t stream, double mu = 0.0, double nu = 1.0, unsigned nbatch = 1, DenseBatch bt = DenseBatch{})
{
    const unsigned tiles = (unsigned)(Mp / 16);
    const int64_t nt = Mp / BT;
    const bool fast = sorted && nt > 1;
    // shared (c, d) (or one matrix): the transcendental part once per tile, the tiles themselves on the matrix cores
    bool tile_build = (nbatch == 1 || bt.cd_stride == 0) && sorted && J <= 64;
    if (tile_build && nt > 1) {
        // dense_build_fast_batch_kernel needs 2 KB x roundup4(J) of dynamic LDS (80 KB at J = 40, 128 KB at J = 64): ask once per device,
        // and take this path only where the device grants it — otherwise the entry-per-thread + per-tile pair below
        static int granted[64] = {};    // 0 unknown, 1 granted, -1 refused; racing threads at worst ask twice
        int dev = 0;
        bool ok = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
        if (ok && granted[dev] == 0) {
            granted[dev] = hipFuncSetAttribute((const void*)dense_build_fast_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               4 * 64 * BT * (int)sizeof(double)) == hipSuccess ? 1 : -1;
            if (granted[dev] < 0) (void)hipGetLastError();
        }
        tile_build = ok && granted[dev] == 1;
    }
    if (tile_build) {
        hipLaunchKernelGGL(dense_build_diag_batch_kernel, dim3(tiles, 4, (nbatch + ZC - 1) / ZC), dim3(256), 0, stream, N, Mp, ld, J, (int32_t)nbatch, a, b,
                           c, d, t, s2, y, K, bt, mu, nu);
        if (nt > 1) {
            const size_t lds = (size_t)4 * ((J + 3) & ~3) * BT * sizeof(double);
            hipLaunchKernelGGL(dense_build_fast_batch_kernel, dim3((unsigned)(nt * (nt - 1) / 2)), dim3(256), lds, stream, N, ld, J, (int32_t)nbatch, a, b,
                               c, d, t, K, bt);
        }
        return;
    }
    hipLaunchKernelGGL(dense_build_kernel, dim3(tiles, tiles, nbatch), dim3(256), 0, stream, N, Mp, ld, J, a, b, c, d, t, s2, y, K,
                       fast ? 1 : 0, mu, nu, bt);
    if (fast)
        hipLaunchKernelGGL(dense_build_fast_kernel, dim3((unsigned)(nt * (nt - 1) / 2), 1, nbatch), dim3(256), 0, stream, N, ld, J, a,
                           b, c, d, t, K, bt);
}

// (diagnostics / tuning: DenseOptions in common.h, carried by the context — tools/sweep_dense_streams.py, tools/dense_ab.py)
constexpr int kQuadThreshold = 1 << 20;        // one matrix: off
constexpr int kBatchQuadThreshold = 16;        // batched launches (tools/sweep_dense_streams.py: 0.628 -> 0.554 ms per N = 4096 factorisation, 32 per launch)

// nbatch matrices of the same size (same t, y, s2; coefficients strided, see DenseBatch): every kernel of the factorisation is
// launched ONCE with gridDim.z = nbatch, so the 64 latency-bound steps of a factorisation are paid once per batch, not once per
// matrix.  info / out: [nbatch].
static int dense_nll_impl(unsigned nbatch, DenseBatch bt, int64_t N, int32_t J, const double* a, const double* b, const double* c, const double* d,
                          const double* t, const double* y, const double* s2, double* K, hipEvent_t* phase_ev,
                          double* out, int32_t* info, int sorted, hipStream_t stream, double mu, double nu, const DenseOptions* dopt)
{
    const DenseOptions dflt{};
    const DenseOptions& dop = dopt ? *dopt : dflt;
    // phase_ev (nullptr or 3 events): recorded after the covariance build, after the factorisation loop, after the finish
    const int64_t Mp = (N + NB - 1) / NB * NB, ld = Mp + NB;
    launch_build(N, Mp, ld, J, a, b, c, d, t, y, s2, K, sorted, stream, mu, nu, nbatch, bt);
    if (phase_ev) (void)hipEventRecord(phase_ev[0], stream);
    if (hipMemsetAsync(info, 0, nbatch * sizeof(int32_t), stream) != hipSuccess) return PIORAN_ERR_HIP;
    double* ws = K + (size_t)ld * (size_t)Mp;   // WS_DOUBLES doubles right behind the slab
    // one launch per block column: from three block columns on, and while the 64-deep lagging bulk update is not bound by the traffic of
    // the trailing matrix (N = 8192: 6.6 ms against 6.2 ms on the paired chain below, whose 128-deep updates read and write C half as often)
    const bool steps = nbatch == 1 && dop.old_chain != 1 && Mp >= 3 * NB && Mp <= 6144;
    hipLaunchKernelGGL(dense_diag0_kernel, dim3(steps ? 3 : 1, 1, nbatch), dim3(256), 0, stream, K, ld, ws, info, bt);
    if (steps) {
        // one matrix: one launch per block column (dense_step_kernel).  Schedule of the lagging bulk update:
        //  * while the trailing matrix is large (more than kPairTiles tiles) the launch is bound by the traffic of the bulk tiles (6.8 TB/s at
        //    64-deep: 128 KB per tile and panel): panels go in PAIRS — launches 2m and 2m+1 share the 128-deep update B(2m) = panels (2m-2,
        //    2m-1) onto block columns >= 2m+2, first half of its tiles (column order: block column 2m+2 and tile (2m+3, 2m+3) included, which
        //    the odd launch's strips and DIAG2 read) in the even launch, the rest in the odd one; strips and DIAG2 catch up two panels / one;
        //  * from the switch launch ks (even) on, one panel per launch, 64-deep, lagging one: a lone 128-deep tile-wave (512 matrix instructions)
        //    takes 20 us whatever the chip is doing, which would be the floor of every later step (the chain is 15); the switch launch itself
        //    carries the whole of B(ks).
        // tools/dense_ab.py, profiles/r04_dense_roles.txt: pairs only 1.38 ms, one panel per launch only 1.29 ms at N = 4096.
        const int nb = (int)(Mp / NB);
        auto tiles_of = [&](int org) -> int64_t {             // tiles (ti >= tj, tj <= nt-2) of the trailing matrix with origin `org`, incl. (0, 0)
            const int64_t nt = (int64_t)(nb - org) + 1;
            return nt >= 2 ? (nt - 1) * (nt + 2) / 2 : 0;
        };
        const int64_t kPairTiles = dop.pair_tiles >= 0 ? dop.pair_tiles : 250;     // (900 up to round 5, with a chain of 15.1 us per step; at 13.3 us: tools/dense_sched_sweep.py,
                                                                                      //  N = 3000 / 4096 / 6000: 0.728 / 1.130 / 2.453 ms against 0.767 / 1.156 / 2.498)
        const int kHalfTileLimit = dop.half_tile_limit >= 0 ? dop.half_tile_limit : 1024;      // tiles per launch up to which every tile is split over two wavefronts
        int ks = 2;
        if (!dop.no_pairs) while (ks + 2 < nb && tiles_of(ks + 2) > kPairTiles) ks += 2;
        // prototype: the chain as one persistent workgroup on a second stream (see flag_wait above); the step launches start at DIAG2
#ifndef PIORAN_EXPERIMENTS
        if (dop.old_chain > 1) return PIORAN_ERR_ARG;     // timing experiments exist in experiment builds only
        int* const flg = nullptr;
#else
        const bool persist = dop.old_chain >= 5 && dop.old_chain <= 8;
        int* flg = persist ? reinterpret_cast<int*>(ws + WS_DOUBLES + 4 * SNAP_TILE) : nullptr;
        static thread_local hipStream_t chain_stream = nullptr;     // (experiment builds drive one device from one thread)
        static thread_local hipEvent_t chain_ev[2] = {nullptr, nullptr};
        if (persist) {
            if (!chain_stream) {
                if (hipStreamCreateWithFlags(&chain_stream, hipStreamNonBlocking) != hipSuccess) return PIORAN_ERR_HIP;
                for (auto& e : chain_ev)
                    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return PIORAN_ERR_HIP;
            }
            if (hipMemsetAsync(flg, 0, 1024 * sizeof(int), stream) != hipSuccess) return PIORAN_ERR_HIP;
            (void)hipEventRecord(chain_ev[0], stream);
            (void)hipStreamWaitEvent(chain_stream, chain_ev[0], 0);
            static thread_local bool chain_attr = false;
            if (!chain_attr) {
                if (hipFuncSetAttribute((const void*)dense_crit_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kChainPadLds) != hipSuccess) return PIORAN_ERR_HIP;
                chain_attr = true;
            }
            hipLaunchKernelGGL(dense_crit_chain_kernel, dim3(1), dim3(256), kChainPadLds, chain_stream, K, ld, Mp, info, flg, dop.old_chain - 5);
            (void)hipEventRecord(chain_ev[1], chain_stream);
        }
#endif
        for (int k = 0; k < nb; ++k) {
            const int nstrip = nb - k - 1;
            StepBulk bk{0, 1, 0, 0, 0, 0};
            int nprev;
            if (k > ks) {                                     // one panel per launch
                nprev = 1;
                const int64_t tot = tiles_of(k + 2);
                bk = StepBulk{k - 1, 1, k + 2, 1, (int)(tot > 1 ? tot - 1 : 0), 0};
            } else {
                nprev = k == 0 ? 0 : ((k & 1) ? 1 : 2);
                const int ke = k & ~1;
                const int64_t tot = ke >= 2 ? tiles_of(ke + 2) : 0;
                if (tot > 1) {
                    const int64_t nt = (int64_t)(nb - (ke + 2)) + 1;
                    int64_t split = (tot + 1) / 2;
                    const int64_t need = nt + (nt >= 3 ? 1 : 0);          // block column 0 of that origin and tile (1, 1)
                    if (split < need) split = need < tot ? need : tot;
                    if (k == ks) split = tot;                              // the switch launch takes all of B(ks)
                    if (k & 1) bk = StepBulk{ke - 2, 2, ke + 2, (int)split, (int)(tot - split), 0};
                    else bk = StepBulk{ke - 2, 2, ke + 2, 1, (int)(split - 1), 0};
                }
            }
            // half tiles (two wavefronts per tile) once whole tiles would leave SIMDs idle: the launch then lasts as long as its slowest wavefront
            if (!dop.no_halves && bk.cnt > 0 && bk.cnt <= kHalfTileLimit) { bk.halves = 1; bk.cnt *= 2; }
            unsigned grid = (unsigned)(2 + nstrip + (bk.cnt + 3) / 4);
            int bx0 = 0;
#ifdef PIORAN_EXPERIMENTS
            // timing experiments only (results are garbage): 2 = the critical workgroup alone, 3 = DIAG2 + the strips alone, 4 = the bulk alone
            if (dop.old_chain == 2) grid = 1;
            else if (dop.old_chain == 3) { grid = (unsigned)(1 + nstrip); bx0 = 1; }
            else if (dop.old_chain == 4) { if (grid <= (unsigned)(2 + nstrip)) continue; grid -= (unsigned)(2 + nstrip); bx0 = 2 + nstrip; }
            if (persist) { grid -= 1; bx0 = 1; }
#endif
            if (bx0 == 0 && grid > 256) ++grid;                  // the blank workgroup (see the kernel)
            if (bk.kp == 2) hipLaunchKernelGGL(dense_step_kernel<2>, dim3(grid), dim3(256), 0, stream, K, ld, Mp, k, info, bx0, nprev, bk, flg);
            else hipLaunchKernelGGL(dense_step_kernel<1>, dim3(grid), dim3(256), 0, stream, K, ld, Mp, k, info, bx0, nprev, bk, flg);
        }
#ifdef PIORAN_EXPERIMENTS
        if (persist) (void)hipStreamWaitEvent(stream, chain_ev[1], 0);
#endif
        if (phase_ev) (void)hipEventRecord(phase_ev[1], stream);
        hipLaunchKernelGGL(dense_finish_kernel, dim3(1, 1, 1), dim3(256), 0, stream, K, ld, N, Mp, out, info, bt);
        if (phase_ev) (void)hipEventRecord(phase_ev[2], stream);
        return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
    }
    auto panel = [&](int64_t kb) {
        // rows below the block: kb+NB .. Mp+63 (the y row Mp and the scratch rows of its 64-row tile)
        const int64_t below = Mp - kb;
        // one 16-row strip per wave while that still fills the chip's 1024 SIMDs at most a few times over (the step
        // is latency-bound: 40 dependent-ish MFMAs per strip); two strips per wave beyond
        if (below * nbatch <= 65536)
            hipLaunchKernelGGL(dense_panel_kernel<1>, dim3((unsigned)((below + 63) / 64), 1, nbatch), dim3(256), 0, stream, K, ld, kb, ws, bt);
        else
            hipLaunchKernelGGL(dense_panel_kernel<2>, dim3((unsigned)((below + 127) / 128), 1, nbatch), dim3(256), 0, stream, K, ld, kb, ws, bt);
    };
    // While the trailing update is bound by the traffic of C (more tile-waves than the chip holds at once), steps go in
    // PAIRS: panel k, narrow update of block column k+1 (+ its diagonal factor), panel k+1, then ONE 128-deep update of the
    // rest; afterwards (latency-bound steps) one 64-deep update per step as before.  (A batch is bound by that traffic for longer:
    // pairs down to a quarter of the single matrix's threshold.)
    const int64_t pair_threshold = nbatch > 1 ? (dop.batch_pair_threshold >= 0 ? dop.batch_pair_threshold : kPairThreshold / 4) : kPairThreshold;
    // ... and in FOURS above quad_threshold: three narrow updates (block column k+1 with one panel, k+2 with two, k+3 with three), then ONE
    // 256-deep update of the rest — the trailing matrix goes through L2 / HBM once per 256 columns.
    const int64_t quad_threshold = dop.quad_threshold >= 0 ? dop.quad_threshold : (nbatch > 1 ? kBatchQuadThreshold : kQuadThreshold);
    auto col0 = [&](auto kp, int64_t kb0, int64_t ntc) {      // block column 0 of the trailing matrix behind KP panels (+ its diagonal factor)
        hipLaunchKernelGGL(dense_syrk_kernel<decltype(kp)::value>, dim3((unsigned)(1 + (ntc - 1 + 3) / 4), 1, nbatch), dim3(256), 0, stream, K, ld, kb0, Mp,
                           ws, info, 1, 1, bt);
    };
    int64_t kb = 0;
    while (kb < Mp) {
        const int64_t nt = (Mp - kb - NB) / NB + 1;           // i tiles of the trailing matrix of step kb (incl. the y-row tile)
        if (nt > quad_threshold && kb + 4 * NB < Mp) {
            panel(kb);
            col0(std::integral_constant<int, 1>{}, kb, nt);
            panel(kb + NB);
            col0(std::integral_constant<int, 2>{}, kb, nt - 1);
            panel(kb + 2 * NB);
            col0(std::integral_constant<int, 3>{}, kb, nt - 2);
            panel(kb + 3 * NB);
            const int64_t nt4 = nt - 3;
            hipLaunchKernelGGL(dense_syrk_kernel<4>, dim3((unsigned)(1 + (nt4 * (nt4 + 1) / 2 - 1 + 3) / 4), 1, nbatch), dim3(256), 0, stream, K, ld,
                               kb, Mp, ws, info, 1, 0, bt);
            kb += 4 * NB;
        } else if (nt > pair_threshold && kb + 2 * NB < Mp) {
            panel(kb);
            hipLaunchKernelGGL(dense_syrk_kernel<1>, dim3((unsigned)(1 + (nt - 1 + 3) / 4), 1, nbatch), dim3(256), 0, stream, K, ld, kb, Mp, ws, info, 1, 1, bt);
            panel(kb + NB);
            const int64_t nt2 = nt - 1;                        // trailing matrix of the pair starts one block further
            hipLaunchKernelGGL(dense_syrk_kernel<2>, dim3((unsigned)(1 + (nt2 * (nt2 + 1) / 2 - 1 + 3) / 4), 1, nbatch), dim3(256), 0, stream, K, ld,
                               kb, Mp, ws, info, 1, 0, bt);
            kb += 2 * NB;
        } else {
            panel(kb);
            if (nt > 1)
                hipLaunchKernelGGL(dense_syrk_kernel<1>, dim3((unsigned)(1 + (nt * (nt + 1) / 2 - 1 + 3) / 4), 1, nbatch), dim3(256), 0, stream, K, ld,
                                   kb, Mp, ws, info, 1, 0, bt);
            kb += NB;
        }
    }
    if (phase_ev) (void)hipEventRecord(phase_ev[1], stream);
    hipLaunchKernelGGL(dense_finish_kernel, dim3(1, 1, nbatch), dim3(256), 0, stream, K, ld, N, Mp, out, info, bt);
    if (phase_ev) (void)hipEventRecord(phase_ev[2], stream);
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}

int pioran_dense_nll_device(int64_t N, int32_t J, const double* a, const double* b, const double* c, const double* d,
                            const double* t, const double* y, const double* s2, double* K, hipEvent_t* phase_ev,
                            double* out, int32_t* info, int sorted, hipStream_t stream, double mu, double nu, const DenseOptions* dopt)
{
    return dense_nll_impl(1, DenseBatch{}, N, J, a, b, c, d, t, y, s2, K, phase_ev, out, info, sorted, stream, mu, nu, dopt);
}

// nbatch <= 65535 matrices at K + z slab (slab >= ld Mp + 1024 doubles); a, b: [nbatch][J]; c, d: [J] (cd_stride = 0) or [nbatch][J]
// (cd_stride = J); mu, nu: device [nbatch] or nullptr (0 / 1); out, info: device [nbatch].
int pioran_dense_nll_device_batch(int64_t nbatch, int64_t N, int32_t J, const double* a, const double* b, const double* c, const double* d,
                                  int64_t cd_stride, const double* t, const double* y, const double* s2, double* K, int64_t slab,
                                  const double* mu, const double* nu, double* out, int32_t* info, int sorted, hipStream_t stream,
                                  const DenseOptions* dopt)
{
    if (nbatch < 1 || nbatch > 65535) return PIORAN_ERR_ARG;
    DenseBatch bt{slab, (int64_t)J, cd_stride, mu, nu};
    return dense_nll_impl((unsigned)nbatch, bt, N, J, a, b, c, d, t, y, s2, K, nullptr, out, info, sorted, stream, 0.0, 1.0, dopt);
}

// predict_cov (src/direct_solver.jl:28-69): K(tau,tau) - K(tau,t) (K(t,t) + diag(s2))^-1 K(t,tau) as the Schur complement
// the blocked Cholesky leaves behind when it stops after the data columns.  te / s2e: device, Mtot = Mp + Mq entries,
// [t (N) | NaN x (Mp - N) | tau (M) | NaN x (Mq - M)] with Mp, Mq = N, M rounded up to 64.  K: slab with
// ld = Mtot + 64, ld * Mtot + 1024 doubles; afterwards the lower triangle of K[Mp.., Mp..] holds the M x M result.
// y (device [N]) and mean (device [M]) may be nullptr: covariance only.
int pioran_dense_predict_cov_device(int64_t N, int64_t M, int32_t J, const double* a, const double* b, const double* c,
                                    const double* d, const double* te, const double* s2e, double* K, int32_t* info,
                                    const double* y, double* mean, hipStream_t stream)
{
    const int64_t Mp = (N + NB - 1) / NB * NB, Mq = (M + NB - 1) / NB * NB, Mtot = Mp + Mq, ld = Mtot + NB;
    if (hipMemsetAsync(K, 0, (size_t)ld * (size_t)Mtot * sizeof(double), stream) != hipSuccess) return PIORAN_ERR_HIP;
    if (hipMemsetAsync(info, 0, sizeof(int32_t), stream) != hipSuccess) return PIORAN_ERR_HIP;
    const unsigned tiles = (unsigned)(Mtot / 16);
    hipLaunchKernelGGL(dense_build_aug_kernel, dim3(tiles, tiles), dim3(256), 0, stream, Mtot, ld, J, a, b, c, d, te, s2e, K);
    if (y) hipLaunchKernelGGL(dense_set_yrow_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, K, ld, Mtot, N, y);
    double* ws = K + (size_t)ld * (size_t)Mtot;
    hipLaunchKernelGGL(dense_diag0_kernel, dim3(1), dim3(256), 0, stream, K, ld, ws, info, DenseBatch{});
    for (int64_t kb = 0; kb < Mp; kb += NB) {
        const int64_t below = Mtot - kb;
        if (below <= 65536)
            hipLaunchKernelGGL(dense_panel_kernel<1>, dim3((unsigned)((below + 63) / 64)), dim3(256), 0, stream, K, ld, kb, ws, DenseBatch{});
        else
            hipLaunchKernelGGL(dense_panel_kernel<2>, dim3((unsigned)((below + 127) / 128)), dim3(256), 0, stream, K, ld, kb, ws, DenseBatch{});
        const int64_t nt = (Mtot - kb - NB) / NB + 1;   // >= 2: the tau block is never empty
        hipLaunchKernelGGL(dense_syrk_kernel<1>, dim3((unsigned)(1 + (nt * (nt + 1) / 2 - 1 + 3) / 4)), dim3(256), 0, stream, K, ld,
                           kb, Mtot, ws, info, kb + NB < Mp ? 1 : 0, 0, DenseBatch{});
    }
    if (y && mean)
        hipLaunchKernelGGL(dense_predict_mean_kernel, dim3((unsigned)M), dim3(256), 0, stream, K, ld, Mtot, Mp, M, mean, info);
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}

void pioran_dense_dims(int64_t N, int64_t* Mp, int64_t* ld)
{
    *Mp = (N + NB - 1) / NB * NB;
    *ld = *Mp + NB;
}

int pioran_dense_build_device(int64_t N, int32_t J, const double* a, const double* b, const double* c, const double* d,
                              const double* t, const double* y, const double* s2, double* K, int sorted, hipStream_t stream)
{
    const int64_t Mp = (N + NB - 1) / NB * NB, ld = Mp + NB;
    launch_build(N, Mp, ld, J, a, b, c, d, t, y, s2, K, sorted, stream);
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}
