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
//   dense_potf2      64 x 64 diagonal block, ONE wavefront, row-per-lane left-looking Cholesky with the
//                    finished columns broadcast through LDS; also reports the first non-positive pivot.
//   dense_trsm       panel solve X L^T = A below the diagonal block: one row per thread, L in LDS.
//   dense_syrk       trailing update A22 -= P P^T on the matrix cores: v_mfma_f64_16x16x4_f64, one
//                    64 x 64 output tile per wavefront (16 accumulators), operands straight from L2
//                    in the MFMA A/B fragment layout; the product is oriented so that the C/D
//                    fragment's lane index runs along the memory-contiguous row index.
//                    MFMA bound: N^3/3 flop at N = 4096 => 2.3e10 flop (the only dense contraction here).
//   dense_finish     logdet + z'z reduction -> +NLL.
#include "../../include/pioran_hip.h"
#include "common.h"

#include <cmath>

namespace {

constexpr int NB = 64;
using f64x4 = __attribute__((ext_vector_type(4))) double;

__global__ void __launch_bounds__(256) dense_build_kernel(int64_t N, int64_t Mp, int64_t ld, int32_t J,
                                                          const double* __restrict__ a, const double* __restrict__ b,
                                                          const double* __restrict__ c, const double* __restrict__ d,
                                                          const double* __restrict__ t, const double* __restrict__ s2,
                                                          const double* __restrict__ y, double* __restrict__ A)
{
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

// ---- 64 x 64 diagonal block: one wavefront, lane = row --------------------------------------------------
__global__ void __launch_bounds__(64) dense_potf2_kernel(double* __restrict__ A, int64_t ld, int64_t kb,
                                                         int32_t* __restrict__ info)
{
    __shared__ double Ls[NB][NB + 1];
    const int lane = threadIdx.x;
    double* blk = A + kb + kb * ld;
    double row[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) row[q] = blk[lane + (int64_t)q * ld];
    int bad = 0;
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        // left-looking: a_ip -= sum_{q<p} l_iq l_pq ; l_pq read as an LDS broadcast
        double acc0 = row[p], acc1 = 0.0;
#pragma unroll
        for (int q = 0; q + 1 < p; q += 2) {
            acc0 = fma(-row[q], Ls[p][q], acc0);
            acc1 = fma(-row[q + 1], Ls[p][q + 1], acc1);
        }
        if (p & 1) acc0 = fma(-row[p - 1], Ls[p][p - 1], acc0);
        const double acc = acc0 + acc1;
        const double piv = __shfl(acc, p);
        if (!(piv > 0.0) && !bad) bad = p + 1;
        const double dgl = sqrt(piv);
        const double v = lane == p ? dgl : acc / dgl;
        row[p] = v;
        Ls[lane][p] = v;
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < NB; ++q)
        if (q <= lane) blk[lane + (int64_t)q * ld] = row[q];
    if (lane == 0 && bad && *info == 0) *info = (int32_t)(kb + bad);
}

// ---- panel: rows below the diagonal block (including the y row), X L^T = A ---------------------------------
__global__ void __launch_bounds__(256) dense_trsm_kernel(double* __restrict__ A, int64_t ld, int64_t kb, int64_t nrows)
{
    __shared__ double Ls[NB][NB + 1];
    __shared__ double rinv[NB];
    const double* blk = A + kb + kb * ld;
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        const int i = e & (NB - 1), q = e >> 6;
        Ls[i][q] = q <= i ? blk[i + (int64_t)q * ld] : 0.0;
    }
    __syncthreads();
    if (threadIdx.x < NB) rinv[threadIdx.x] = 1.0 / Ls[threadIdx.x][threadIdx.x];
    __syncthreads();
    const int64_t r = kb + NB + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrows) return;
    double* rowp = A + r + kb * ld;
    double x[NB];
#pragma unroll
    for (int p = 0; p < NB; ++p) x[p] = rowp[(int64_t)p * ld];
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        double acc0 = x[p], acc1 = 0.0;
#pragma unroll
        for (int q = 0; q + 1 < p; q += 2) {
            acc0 = fma(-x[q], Ls[p][q], acc0);
            acc1 = fma(-x[q + 1], Ls[p][q + 1], acc1);
        }
        if (p & 1) acc0 = fma(-x[p - 1], Ls[p][p - 1], acc0);
        x[p] = (acc0 + acc1) * rinv[p];
    }
#pragma unroll
    for (int p = 0; p < NB; ++p) rowp[(int64_t)p * ld] = x[p];
}

// ---- trailing update on the matrix cores --------------------------------------------------------------------
// Tile (ti, tj), ti >= tj, of 64 x 64 entries of the trailing matrix (origin j0 = kb + NB):
//   C[i][j] -= sum_p P[i][p] P[j][p],   P = A[:, kb : kb+NB]
// v_mfma_f64_16x16x4_f64: A-operand lane l holds X[row l&15][k l>>4], B-operand lane l holds Y[k l>>4][col l&15],
// result reg g of lane l is D[row (l>>4) + 4g][col l&15]  (f64 layout, cdna_hip_programming.md section 3).
// With X[row][k] = P[j][k] and Y[k][col] = P[i][k], D[row][col] = (P P^T)[i][j]: col = l&15 runs along i,
// the memory-contiguous index of the column-major slab, so C loads/stores are 128-byte segments.
__global__ void __launch_bounds__(64) dense_syrk_kernel(double* __restrict__ A, int64_t ld, int64_t kb, int64_t Mp)
{
    const int64_t j0 = kb + NB;
    const int nt = (int)((Mp - j0) / NB);  // tiles per side of the trailing matrix
    // linear block id -> (ti, tj) with ti >= tj
    int bid = blockIdx.x;
    int ti = (int)((sqrt(8.0 * bid + 1.0) - 1.0) * 0.5);
    while ((int64_t)(ti + 1) * (ti + 2) / 2 <= bid) ++ti;
    while ((int64_t)ti * (ti + 1) / 2 > bid) --ti;
    const int tj = bid - (int)((int64_t)ti * (ti + 1) / 2);
    if (ti >= nt) return;
    const int lane = threadIdx.x;
    const int lr = lane & 15, lk = lane >> 4;
    const double* Pi = A + (j0 + (int64_t)ti * NB) + kb * ld;  // rows of the i tile
    const double* Pj = A + (j0 + (int64_t)tj * NB) + kb * ld;  // rows of the j tile
    f64x4 acc[4][4];  // [jb][ib]
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) acc[jb][ib] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int k0 = 0; k0 < NB; k0 += 4) {
        double xa[4], yb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            xa[s] = Pj[(s * 16 + lr) + (int64_t)(k0 + lk) * ld];  // X[row][k] = P[j][k]
            yb[s] = Pi[(s * 16 + lr) + (int64_t)(k0 + lk) * ld];  // Y[k][col] = P[i][k]
        }
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
                acc[jb][ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[jb], yb[ib], acc[jb][ib], 0, 0, 0);
    }
    double* C = A + (j0 + (int64_t)ti * NB) + (j0 + (int64_t)tj * NB) * ld;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t jj = jb * 16 + lk + 4 * g;   // D row -> j
                const int64_t ii = ib * 16 + lr;           // D col -> i (contiguous)
                double* pc = C + ii + jj * ld;
                *pc -= acc[jb][ib][g];
            }
}

// y row of the trailing update: A[Mp][j] -= sum_p A[Mp][kb+p] A[j][kb+p],  j >= kb + NB
__global__ void __launch_bounds__(256) dense_yrow_kernel(double* __restrict__ A, int64_t ld, int64_t kb, int64_t Mp)
{
    __shared__ double zs[NB];
    if (threadIdx.x < NB) zs[threadIdx.x] = A[Mp + (kb + threadIdx.x) * ld];
    __syncthreads();
    const int64_t j = kb + NB + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= Mp) return;
    double acc = 0.0;
#pragma unroll 8
    for (int p = 0; p < NB; ++p) acc = fma(zs[p], A[j + (kb + p) * ld], acc);
    A[Mp + j * ld] -= acc;
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

// K slab must hold ld * Mp doubles with Mp = roundup(N, 64), ld = Mp + 64.
int pioran_dense_nll_device(int64_t N, int32_t J, const double* a, const double* b, const double* c, const double* d,
                            const double* t, const double* y, const double* s2, double* K, double* /*work*/,
                            double* out, int32_t* info, hipStream_t stream)
{
    const int64_t Mp = (N + NB - 1) / NB * NB, ld = Mp + NB;
    const unsigned tiles = (unsigned)(Mp / 16);
    hipLaunchKernelGGL(dense_build_kernel, dim3(tiles, tiles), dim3(256), 0, stream, N, Mp, ld, J, a, b, c, d, t, s2, y, K);
    if (hipMemsetAsync(info, 0, sizeof(int32_t), stream) != hipSuccess) return PIORAN_ERR_HIP;
    for (int64_t kb = 0; kb < Mp; kb += NB) {
        hipLaunchKernelGGL(dense_potf2_kernel, dim3(1), dim3(64), 0, stream, K, ld, kb, info);
        const int64_t nrows = Mp + 1;                 // rows below the block: kb+NB .. Mp (the y row)
        const int64_t below = nrows - (kb + NB);
        hipLaunchKernelGGL(dense_trsm_kernel, dim3((unsigned)((below + 255) / 256)), dim3(256), 0, stream, K, ld, kb, nrows);
        const int64_t nt = (Mp - kb - NB) / NB;
        if (nt > 0) {
            hipLaunchKernelGGL(dense_syrk_kernel, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(64), 0, stream, K, ld, kb, Mp);
            hipLaunchKernelGGL(dense_yrow_kernel, dim3((unsigned)((Mp - kb - NB + 255) / 256)), dim3(256), 0, stream, K, ld, kb, Mp);
        }
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
                              const double* t, const double* y, const double* s2, double* K, hipStream_t stream)
{
    const int64_t Mp = (N + NB - 1) / NB * NB, ld = Mp + NB;
    const unsigned tiles = (unsigned)(Mp / 16);
    hipLaunchKernelGGL(dense_build_kernel, dim3(tiles, tiles), dim3(256), 0, stream, N, Mp, ld, J, a, b, c, d, t, s2, y, K);
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}
