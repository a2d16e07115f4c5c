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
//   dense_panel      per 64-column step: every workgroup factors the 64 x 64 diagonal block in LDS (scalar
//                    16-column sub-panels + MFMA updates inside the block; reports the first non-positive pivot)
//                    and then solves X L^T = A for its rows entirely on the matrix cores (blocked, with the
//                    16 x 16 inverses), results staying in accumulator registers between sub-steps.
//   dense_syrk       trailing update A22 -= P P^T on the matrix cores: v_mfma_f64_16x16x4_f64, one
//                    64 x 64 output tile per wavefront (16 accumulators), operands straight from L2
//                    in the MFMA A/B fragment layout; the product is oriented so that the C/D
//                    fragment's lane index runs along the memory-contiguous row index.
//                    MFMA bound: N^3/3 flop at N = 4096 => 2.3e10 flop (the only dense contraction here).
//   dense_finish     logdet + z'z reduction -> +NLL.
#include "../../include/pioran_hip.h"
#include "common.h"

#include <cmath>
#include <type_traits>
#include <utility>

namespace {

constexpr int NB = 64;
using f64x4 = __attribute__((ext_vector_type(4))) double;

__global__ void __launch_bounds__(256) dense_build_kernel(int64_t N, int64_t Mp, int64_t ld, int32_t J,
                                                          const double* __restrict__ a, const double* __restrict__ b,
                                                          const double* __restrict__ c, const double* __restrict__ d,
                                                          const double* __restrict__ t, const double* __restrict__ s2,
                                                          const double* __restrict__ y, double* __restrict__ A, int diag_only)
{
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
        if (i == k) v += s2[i];                                 // src/direct_solver.jl:15
    } else {
        if (i < k) return;
        v = (i == k) ? 1.0 : 0.0;                               // identity padding
    }
    A[i + k * ld] = v;
    if (i == k) A[Mp + k * ld] = k < N ? y[k] : 0.0;            // the y row
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
                                                               const double* __restrict__ t, double* __restrict__ A)
{
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

// value of lane `src` (wave-uniform) as a scalar: two v_readlane_b32
__device__ __forceinline__ double readlane_f64(double x, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(x) to fp64 accuracy: v_rsq_f64 + two Newton steps (off the division/sqrt latency chains)
__device__ __forceinline__ double rsqrt_f64(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = fma(y, fma(-hx * y, y, 0.5), y);
    y = fma(y, fma(-hx * y, y, 0.5), y);
    return y;
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

// Factor the 64 x 64 block held row-major (padded) in LDS `Ls`, in place, by ONE wavefront; writes the four
// 16 x 16 inverses to `ws` (global).  Returns the 1-based index of the first non-positive pivot (0 = none).
//
// Per 16-column sub-panel s only the 16 x 16 diagonal tile is scalar work: every DPP row of the wave holds the
// tile (lane l&15 = tile row), the right-looking Cholesky and the triangular inverse take their cross-lane
// operands from v_mov_b64_dpp row_newbcast, and the per-column critical path is bcast -> rsqrt -> mul -> fma.
// Everything below the tile is matrix-core work through LDS: X_t = A_ts inv(L_ss)' for the tiles t > s
// (transposed C/D layout, the inverse as A operand), then the in-block update A_rc -= X_r X_c'.
__device__ __forceinline__ int factor_block64(double* __restrict__ Ls, double* __restrict__ ws, int lane)
{
    const int lr = lane & 15, lk = lane >> 4;
    int bad = 0;
    // runtime loop over the four 16-column sub-panels: one copy of the code (instruction cache stays warm after
    // the first pass), the tile loops inside are static with wave-uniform guards.
#pragma unroll 1
    for (int s = 0; s < 4; ++s) {
        const int c0 = 16 * s;
        const int ir0 = s == 3 ? 16 : 0, ic0 = s == 3 ? 32 : 16 * (s + 1);   // where inv(L_ss) is parked
        PIORAN_STAMP(8 * s + 0);
        // ---- 16 x 16 diagonal tile: right-looking Cholesky, lane lr = row (replicated in the 4 DPP rows) -----
        double r[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) r[p] = Ls[(c0 + lr) * LP + c0 + p];
        double rinv_own = 0.0;   // 1 / l_pp of this lane's row
        static_for16([&](auto Pc) {
            constexpr int p = decltype(Pc)::value;
            const double piv = bcast16<p>(r[p]);
            if (!(piv > 0.0) && !bad) bad = c0 + p + 1;
            const double rinv = rsqrt_f64(piv);           // 1 / l_pp
            r[p] = lr == p ? piv * rinv : r[p] * rinv;    // rows above p hold junk in column p (never used)
            if (lr == p) rinv_own = rinv;
            const double nrp = -r[p];
            static_for16([&](auto Qc) {
                constexpr int q = decltype(Qc)::value;
                if constexpr (q > p) r[q] = fma(bcast16<q>(r[p]), nrp, r[q]);   // a_iq -= l_ip l_qp
            });
        });
        PIORAN_STAMP(8 * s + 1);
        // ---- inverse of the tile, column-owner layout: lane j computes column j of X = inv(L) -------------------
        //   X[i][j] = (delta_ij - sum_{k<i} L[i][k] X[k][j]) / l_ii ; L[i][k] and 1/l_ii come from lane i by DPP broadcast
        double x[16];
        static_for16([&](auto Ic) {
            constexpr int i = decltype(Ic)::value;
            double acc0 = lr == i ? 1.0 : 0.0, acc1 = 0.0;
            static_for16([&](auto Kc) {
                constexpr int k = decltype(Kc)::value;
                if constexpr (k < i) {
                    if constexpr (k & 1) acc1 = fma(-bcast16<i>(r[k]), x[k], acc1);
                    else acc0 = fma(-bcast16<i>(r[k]), x[k], acc0);
                }
            });
            x[i] = (acc0 + acc1) * bcast16<i>(rinv_own);
        });
        PIORAN_STAMP(8 * s + 2);
        // publish to LDS only: L rows in place (lane = row; junk above the diagonal is never read) and the inverse
        // (lane = column) into a 16 x 16 tile of the block's strictly-upper part, which the algorithm never touches
        if (lk == 0) {
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                Ls[(c0 + lr) * LP + c0 + p] = r[p];
                Ls[(ir0 + p) * LP + ic0 + lr] = x[p];     // X[p][lr]
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        PIORAN_STAMP(8 * s + 3);
        // ---- rows below the tile on the matrix cores: Y_t = inv(L_ss) A_ts'  (D[row = c][col = r]) ---------------
        double iop[4];   // A operand of k-step ks: inv(L)[row lr][k = 4ks + lk]
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) iop[ks] = Ls[(ir0 + lr) * LP + ic0 + 4 * ks + lk];
        f64x4 Yt[4];
#pragma unroll
        for (int t = 1; t < 4; ++t) {
            Yt[t] = f64x4{0.0, 0.0, 0.0, 0.0};
            if (t > s) {   // wave-uniform
                f64x4 a;
#pragma unroll
                for (int g = 0; g < 4; ++g) a[g] = Ls[(16 * t + lr) * LP + c0 + lk + 4 * g];   // A_ts' in C/D layout
                f64x4 z = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) z = mfma4(iop[ks], a[ks], z);                    // reg ks IS the B operand
                Yt[t] = z;
#pragma unroll
                for (int g = 0; g < 4; ++g) Ls[(16 * t + lr) * LP + c0 + lk + 4 * g] = z[g];
            }
        }
        PIORAN_STAMP(8 * s + 4);
        // ---- in-block trailing update: A_rc -= X_r X_c', operands straight from the result registers ---------------
        // X_t as A operand: M[row = c][k] = Y_t[k][c] -> C/D register ks of Y_t; as B operand: B[k][col = r] -> the same.
#pragma unroll
        for (int rt = 1; rt < 4; ++rt)
#pragma unroll
            for (int ct = 1; ct <= rt; ++ct)
                if (ct > s) {   // wave-uniform (then rt > s too)
                    f64x4 c;
#pragma unroll
                    for (int g = 0; g < 4; ++g) c[g] = Ls[(16 * rt + lr) * LP + 16 * ct + lk + 4 * g];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) c = mfma4(-Yt[ct][ks], Yt[rt][ks], c);
#pragma unroll
                    for (int g = 0; g < 4; ++g) Ls[(16 * rt + lr) * LP + 16 * ct + lk + 4 * g] = c[g];
                }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        PIORAN_STAMP(8 * s + 5);
    }
    // the four inverses -> workspace (row-major 16 x 16 each), 16 entries per lane, coalesced
    {
        double v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int idx = e * 64 + lane, sb = idx >> 8, i = (idx >> 4) & 15, j = idx & 15;
            v[e] = Ls[((sb == 3 ? 16 : 0) + i) * LP + (sb == 3 ? 32 : 16 * (sb + 1)) + j];
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) ws[e * 64 + lane] = v[e];
    }
    PIORAN_STAMP(40);
    return bad;
}

// factored block: LDS -> global, lane = row, rows contiguous across lanes.  The strictly-upper part of the slab is
// scratch (never read as data), so the whole 64 x 64 block is stored without masking.
__device__ __forceinline__ void store_block_lower(const double* __restrict__ Ls, double* __restrict__ blk, int64_t ld, int lane)
{
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = Ls[lane * LP + 16 * h + q];
#pragma unroll
        for (int q = 0; q < 16; ++q) blk[lane + (int64_t)(16 * h + q) * ld] = v[q];
    }
}

// First diagonal block (no trailing update precedes it): load, factor, write back.
__global__ void __launch_bounds__(64) dense_diag0_kernel(double* __restrict__ A, int64_t ld, double* __restrict__ ws,
                                                         int32_t* __restrict__ info)
{
    __shared__ double Ls[NB * LP];
    const int lane = threadIdx.x;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = A[lane + (int64_t)(16 * h + q) * ld];
#pragma unroll
        for (int q = 0; q < 16; ++q) Ls[lane * LP + 16 * h + q] = v[q];
    }
    __syncthreads();
    const int bad = factor_block64(Ls, ws, lane);
    __syncthreads();
    store_block_lower(Ls, A, ld, lane);
    if (lane == 0 && bad && *info == 0) *info = bad;
}

// ---- panel solve, all on the matrix cores --------------------------------------------------------------------
// Rows below the (already factored) diagonal block, 32 rows per wave, kept TRANSPOSED in MFMA C/D layout
// (D[row = column-in-subpanel c][col = row r], so the lane index runs along memory-contiguous rows):
//   Y_s = inv(L_ss) (A_s' - sum_{s' < s} L[s][s'] Y_s')
// A operands (blocks of L, the inverses from `ws`) come straight from L2.  Key identity of the f64 layouts:
// C/D register g of lane l holds D[k = (l>>4) + 4g][col = l&15] and the B operand of k-step ks wants
// B[k = 4ks + (l>>4)][col = l&15]: with g = ks these are the same element, so a result tile is fed back as the
// next B operand with no data movement.
__global__ void __launch_bounds__(256) dense_panel_kernel(double* __restrict__ A, int64_t ld, int64_t kb,
                                                          const double* __restrict__ ws)
{
    constexpr int RT = 2;   // 16-row tiles per wave
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
__global__ void __launch_bounds__(64) dense_syrk_kernel(double* __restrict__ A, int64_t ld, int64_t kb, int64_t Mp,
                                                        double* __restrict__ ws, int32_t* __restrict__ info)
{
    // Lookahead: the workgroup of tile (0,0) — the NEXT diagonal block — keeps its updated tile in LDS and factors
    // it right away (factor_block64), while the other tiles are still being updated; the next panel kernel then
    // starts from a finished diagonal block.  33.8 KB of LDS per workgroup still allows 4 workgroups per CU.
    __shared__ double Ls[NB * LP];
    const int64_t j0 = kb + NB;
    // i tiles 0..nt (the last one holds the y row Mp and 63 scratch rows of the slab), j tiles 0..nt-1, i >= j
    const int nt = (int)((Mp - j0) / NB) + 1;
    // linear block id -> (ti, tj) with ti >= tj
    int bid = blockIdx.x;
    int ti = (int)((sqrt(8.0 * bid + 1.0) - 1.0) * 0.5);
    while ((int64_t)(ti + 1) * (ti + 2) / 2 <= bid) ++ti;
    while ((int64_t)ti * (ti + 1) / 2 > bid) --ti;
    const int tj = bid - (int)((int64_t)ti * (ti + 1) / 2);
    if (ti >= nt || tj >= nt - 1) return;
    const int lane = threadIdx.x;
    const int lr = lane & 15, lk = lane >> 4;
    const double* Pi = A + (j0 + (int64_t)ti * NB) + kb * ld + lr + (int64_t)lk * ld;  // rows of the i tile
    const double* Pj = A + (j0 + (int64_t)tj * NB) + kb * ld + lr + (int64_t)lk * ld;  // rows of the j tile
    f64x4 acc[4][4];  // [jb][ib]
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) acc[jb][ib] = f64x4{0.0, 0.0, 0.0, 0.0};

    // operand fragments of 4 k-steps (K = 16 columns of the panel) at a time, double-buffered in registers:
    // the loads of group g+1 are in flight while the 64 MFMAs of group g issue.
    double xa[2][4][4], yb[2][4][4];   // [buffer][k-step][16-row strip]
    auto load_group = [&](int g, int buf) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                const int64_t off = sidx * 16 + (int64_t)(16 * g + 4 * ks) * ld;
                xa[buf][ks][sidx] = Pj[off];  // X[row][k] = P[j][k]
                yb[buf][ks][sidx] = Pi[off];  // Y[k][col] = P[i][k]
            }
    };
    auto mma_group = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                for (int ib = 0; ib < 4; ++ib)
                    acc[jb][ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[buf][ks][jb], yb[buf][ks][ib], acc[jb][ib], 0, 0, 0);
    };
    load_group(0, 0);
    load_group(1, 1);
    mma_group(0);
    load_group(2, 0);
    mma_group(1);
    load_group(3, 1);
    mma_group(0);
    mma_group(1);

    // C -= acc, in two halves of 32 entries per lane: all loads of a half in flight before the first store
    double* C = A + (j0 + (int64_t)ti * NB) + (j0 + (int64_t)tj * NB) * ld + lr + (int64_t)lk * ld;
    const bool diag_next = bid == 0;   // tile (0,0)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        double cv[2][4][4];
#pragma unroll
        for (int jq = 0; jq < 2; ++jq)
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int jb = 2 * half + jq;
                    cv[jq][ib][g] = C[ib * 16 + (int64_t)(jb * 16 + 4 * g) * ld];   // D row (l>>4)+4g -> j, col l&15 -> i
                }
#pragma unroll
        for (int jq = 0; jq < 2; ++jq)
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int jb = 2 * half + jq;
                    const double v = cv[jq][ib][g] - acc[jb][ib][g];
                    if (diag_next)
                        Ls[(ib * 16 + lr) * LP + jb * 16 + lk + 4 * g] = v;   // row i, column j
                    else
                        C[ib * 16 + (int64_t)(jb * 16 + 4 * g) * ld] = v;
                }
    }
    if (diag_next) {
        __syncthreads();
        const int bad = factor_block64(Ls, ws, lane);
        __syncthreads();
        store_block_lower(Ls, A + j0 + j0 * ld, ld, lane);
        if (lane == 0 && bad && *info == 0) *info = (int32_t)(j0 + bad);
    }
}

__global__ void __launch_bounds__(256) dense_finish_kernel(const double* __restrict__ A, int64_t ld, int64_t N,
                                                           int64_t Mp, double* __restrict__ out,
                                                           const int32_t* __restrict__ info)
{
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
    for (int s = 128; s > 0; s >>= 1) {
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

}  // namespace

// K must hold ld * Mp + 1024 doubles with Mp = roundup(N, 64), ld = Mp + 64 (slab + inverse workspace).
static void launch_build(int64_t N, int64_t Mp, int64_t ld, int32_t J, const double* a, const double* b, const double* c,
                         const double* d, const double* t, const double* y, const double* s2, double* K, int sorted,
                         hipStream_t stream)
{
    const unsigned tiles = (unsigned)(Mp / 16);
    const int64_t nt = Mp / BT;
    const bool fast = sorted && nt > 1;
    hipLaunchKernelGGL(dense_build_kernel, dim3(tiles, tiles), dim3(256), 0, stream, N, Mp, ld, J, a, b, c, d, t, s2, y, K,
                       fast ? 1 : 0);
    if (fast)
        hipLaunchKernelGGL(dense_build_fast_kernel, dim3((unsigned)(nt * (nt - 1) / 2)), dim3(256), 0, stream, N, ld, J, a,
                           b, c, d, t, K);
}

int pioran_dense_nll_device(int64_t N, int32_t J, const double* a, const double* b, const double* c, const double* d,
                            const double* t, const double* y, const double* s2, double* K, double* /*work*/,
                            double* out, int32_t* info, int sorted, hipStream_t stream)
{
    const int64_t Mp = (N + NB - 1) / NB * NB, ld = Mp + NB;
    launch_build(N, Mp, ld, J, a, b, c, d, t, y, s2, K, sorted, stream);
    if (hipMemsetAsync(info, 0, sizeof(int32_t), stream) != hipSuccess) return PIORAN_ERR_HIP;
    double* ws = K + (size_t)ld * (size_t)Mp;   // WS_DOUBLES doubles right behind the slab
    hipLaunchKernelGGL(dense_diag0_kernel, dim3(1), dim3(64), 0, stream, K, ld, ws, info);
    for (int64_t kb = 0; kb < Mp; kb += NB) {
        // rows below the block: kb+NB .. Mp+63 (the y row Mp and the scratch rows of its 64-row tile)
        const int64_t below = Mp - kb;
        hipLaunchKernelGGL(dense_panel_kernel, dim3((unsigned)((below + 127) / 128)), dim3(256), 0, stream, K, ld, kb, ws);
        const int64_t nt = (Mp - kb - NB) / NB + 1;
        if (nt > 1)
            hipLaunchKernelGGL(dense_syrk_kernel, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(64), 0, stream, K, ld, kb, Mp, ws, info);
    }
    hipLaunchKernelGGL(dense_finish_kernel, dim3(1), dim3(256), 0, stream, K, ld, N, Mp, out, info);
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
