// Shared per-step table build: cos(d_j t_n), sin(d_j t_n), exp(-c_j (t_n - t_{n-1})).
//
// These are the 3J transcendentals per time step of src/celerite_solver.jl:52-54.  In the
// reference's `approx` flow (c_j, d_j) depend only on the spectral grid (src/psd.jl:250,266-267),
// i.e. on the data set and not on the sampled parameters, so one table serves every draw of a
// batch.  d_j * t_n reaches 1e6-1e7 rad: the device-library sincos does a full Payne-Hanek
// reduction (no fast-math, no angle-addition recurrences).  Coalesced HBM write, jp fastest.
#include "common.h"

namespace {
// record of step n (RS = 3 Rp + 2 doubles, Rp = R + 2):
//   [ v_r (r < Rp) | x_r | phi_r | y_n  sigma2_n ],  cos row: (v, x) = (cos, sin)(d t_n), sin row: (sin, cos);
//   row R = inert padding (1, 0, 1) (u = 0: never feeds a real row), row R+1 = the y row of the scan (0, 0, 1).  N + 1 records: the last one
//   repeats step N-1 so that the scan's prefetch of "step N" stays in bounds.
__global__ void __launch_bounds__(256) table_kernel(int64_t N, int32_t R, const int32_t* __restrict__ rowmap,
                                                    const double* __restrict__ t, const double* __restrict__ c,
                                                    const double* __restrict__ d, const double* __restrict__ y,
                                                    const double* __restrict__ s2, double* __restrict__ tab, int64_t RS /*record stride*/,
                                                    int64_t cd_stride, int64_t tab_draw_stride)
{
    // blockIdx.y: draw index of a batch of per-draw tables (its (c, d) at c + y * cd_stride, its table at tab + y * tab_draw_stride)
    c += (int64_t)blockIdx.y * cd_stride;
    d += (int64_t)blockIdx.y * cd_stride;
    tab += (int64_t)blockIdx.y * tab_draw_stride;
    const int32_t Rp = R + 2;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (N + 1) * Rp) return;
    const int64_t nrec = idx / Rp;
    const int32_t row = (int32_t)(idx - nrec * Rp);
    const int64_t n = nrec < N ? nrec : N - 1;
    double v = row == R ? 1.0 : 0.0, x = 0.0, ph = 1.0;   // padding (1, 0, 1) and y row (0, 0, 1)
    if (row < R && !((rowmap[row] >> 29) & 1)) {   // per-draw rows live in the per-draw table (mixed mode)
        const int32_t rm = rowmap[row];
        const int32_t term = rm & 0xfffff;
        const double tn = t[n];
        double si, co;
        sincos(d[term] * tn, &si, &co);
        const bool ks = (rm >> 30) & 1;
        v = ks ? si : co;
        x = ks ? co : si;
        ph = n > 0 ? exp(-c[term] * (tn - t[n - 1])) : 0.0;
    }
    double* rec = tab + nrec * RS;
    rec[row] = v;
    rec[Rp + row] = x;
    rec[2 * Rp + row] = ph;
    if (row == 0) {
        rec[3 * Rp] = y[n];
        rec[3 * Rp + 1] = s2[n];
    }
}
// Mixed mode: (v, x, phi) of the rows of the few terms whose (c, d) differ per draw (QPO features, src/psd.jl:15-27),
// appended to every step record of the table so that all rows share one step stride:
//   tab[n * rec_stride + rs_shared + (b * 2 npd + 2k + kind) * 3 + {v, x, phi}].
__global__ void __launch_bounds__(256) pd_table_kernel(int64_t N, int64_t B, int32_t J, int32_t npd, const int32_t* __restrict__ pd_terms,
                                                       const double* __restrict__ t, const double* __restrict__ C,
                                                       const double* __restrict__ D, double* __restrict__ tab,
                                                       int64_t rec_stride, int64_t rs_shared)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (n, b, k), k fastest
    if (idx >= (N + 1) * B * npd) return;
    const int32_t k = (int32_t)(idx % npd);
    const int64_t nb = idx / npd;
    const int64_t b = nb % B, nrec = nb / B;
    const int64_t n = nrec < N ? nrec : N - 1;
    const int32_t term = pd_terms[k];
    const double tn = t[n];
    double si, co;
    sincos(D[b * J + term] * tn, &si, &co);                                   // :52-53
    const double ph = n > 0 ? exp(-C[b * J + term] * (tn - t[n - 1])) : 0.0;  // :54
    double* rec = tab + nrec * rec_stride + rs_shared + (b * (2 * npd) + 2 * k) * 3;   // inside the step record
    rec[0] = co; rec[1] = si; rec[2] = ph;   // cos row: v = co, x = si
    rec[3] = si; rec[4] = co; rec[5] = ph;   // sin row: v = si, x = co
}

// Per-draw series of the shifted log-flux models: Y[b][n] = log(y_n - shift_b), S2[b][n] = sigma2_n / (y_n - shift_b)^2
// (docs/src/ultranest.md:199-205).  Pure streaming kernel: 16 B written per (draw, step), coalesced along n.
__global__ void __launch_bounds__(256) shift_transform_kernel(int64_t N, int64_t B, const double* __restrict__ y,
                                                              const double* __restrict__ s2, const double* __restrict__ shift,
                                                              double* __restrict__ Y, double* __restrict__ S2)
{
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t b = blockIdx.y;
    if (n >= N || b >= B) return;
    const double v = y[n] - shift[b];
    Y[b * N + n] = log(v);
    S2[b * N + n] = s2[n] / (v * v);
}
// Chain rule of the transform above: with g_Y = dL/dY_n and g_S = dL/dS2_n of draw b,
//   dL/dshift_b = sum_n ( -g_Y / (y_n - shift_b) + 2 g_S sigma2_n / (y_n - shift_b)^3 ).    One workgroup per draw.
__global__ void __launch_bounds__(256) shift_grad_kernel(int64_t N, const double* __restrict__ y, const double* __restrict__ s2,
                                                         const double* __restrict__ shift, const double* __restrict__ gY,
                                                         const double* __restrict__ gS, double* __restrict__ gshift)
{
    __shared__ double red[256];
    const int64_t b = blockIdx.x;
    const double c = shift[b];
    double acc = 0.0;
    for (int64_t n = threadIdx.x; n < N; n += 256) {
        const double v = y[n] - c, rv = 1.0 / v;
        acc += -gY[b * N + n] * rv + 2.0 * gS[b * N + n] * s2[n] * rv * rv * rv;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (unsigned s_ = 128; s_ > 0; s_ >>= 1) {
        if (threadIdx.x < s_) red[threadIdx.x] += red[threadIdx.x + s_];
        __syncthreads();
    }
    if (threadIdx.x == 0) gshift[b] = red[0];
}
}  // namespace

int pioran_launch_shift_grad(int64_t N, int64_t B, const double* y, const double* s2, const double* shift, const double* gY,
                             const double* gS, double* gshift, hipStream_t stream)
{
    if (B < 1 || B > 0x7fffffffLL) return PIORAN_ERR_ARG;
    hipLaunchKernelGGL(shift_grad_kernel, dim3((unsigned)B), dim3(256), 0, stream, N, y, s2, shift, gY, gS, gshift);
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}

int pioran_launch_pd_table(int64_t N, int64_t B, int32_t J, int32_t npd_terms, const int32_t* pd_terms, const double* t,
                           const double* C, const double* D, double* tab, int64_t rec_stride, int64_t rs_shared,
                           hipStream_t stream)
{
    const int64_t total = (N + 1) * B * npd_terms;
    const int64_t blocks = (total + 255) / 256;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return PIORAN_ERR_ARG;
    hipLaunchKernelGGL(pd_table_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, N, B, J, npd_terms, pd_terms, t, C, D, tab,
                       rec_stride, rs_shared);
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}

int pioran_launch_shift_transform(int64_t N, int64_t B, const double* y, const double* s2, const double* shift,
                                  double* Y, double* S2, hipStream_t stream)
{
    if (B > 65535 * 1024LL) return PIORAN_ERR_ARG;
    for (int64_t b0 = 0; b0 < B; b0 += 65535) {   // gridDim.y limit
        const int64_t nb = B - b0 < 65535 ? B - b0 : 65535;
        hipLaunchKernelGGL(shift_transform_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)nb), dim3(256), 0, stream, N,
                           nb, y, s2, shift + b0, Y + b0 * N, S2 + b0 * N);
    }
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}

size_t pioran_table_doubles(int64_t N, int32_t R) { return (size_t)(N + 1) * (size_t)(3 * (R + 2) + 2); }

int pioran_launch_table(int64_t N, int32_t R, const int32_t* rowmap, const double* t, const double* c,
                        const double* d, const double* y, const double* s2, double* tab, int64_t rec_stride,
                        hipStream_t stream)
{
    const int64_t total = (N + 1) * (int64_t)(R + 2);
    const int64_t blocks = (total + 255) / 256;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return PIORAN_ERR_ARG;
    if ((uint64_t)(N + 1) * (uint64_t)rec_stride * 8 > 0x7ffffff0ull) return PIORAN_ERR_UNSUPPORTED;  // 32-bit buffer offsets
    hipLaunchKernelGGL(table_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, N, R, rowmap, t, c, d, y, s2, tab, rec_stride,
                       (int64_t)0, (int64_t)0);
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}

// nb tables at once, one per draw with its own (c, d) [nb][J]: table b at tab + b * tab_draw_stride (doubles)
int pioran_launch_table_batch(int64_t N, int32_t R, int32_t J, int64_t nb, const int32_t* rowmap, const double* t, const double* C,
                              const double* D, const double* y, const double* s2, double* tab, int64_t rec_stride,
                              int64_t tab_draw_stride, hipStream_t stream)
{
    const int64_t total = (N + 1) * (int64_t)(R + 2);
    const int64_t blocks = (total + 255) / 256;
    if (blocks <= 0 || blocks > 0x7fffffffLL || nb < 1 || nb > 65535) return PIORAN_ERR_ARG;
    hipLaunchKernelGGL(table_kernel, dim3((unsigned)blocks, (unsigned)nb), dim3(256), 0, stream, N, R, rowmap, t, C, D, y, s2, tab, rec_stride,
                       (int64_t)J, tab_draw_stride);
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}


// ---- diagnostics: the FP64 FMA rate this device sustains right now -----------------------------------------------------------------
// A pure stream of independent v_fma_f64 (eight accumulators per lane, 64 instructions per loop trip) on every SIMD of the chip with
// `waves_per_simd` wavefronts each: what ANY FP64 vector kernel can reach on this box at this occupancy — the vendor peak (78.6 TFLOP/s =
// one DP FMA per SIMD every 4 cycles at 2.4 GHz) is not reachable by construction (4.3 .. 5.4 issue cycles per instruction, 1.9 .. 2.2 GHz
// under chip-wide FP64 load: tools/valu_probe.hip).  bench.py quotes the scan kernels against it beside the roofline fraction.
namespace {
__global__ void __launch_bounds__(256) fma_stream_kernel(double* out, int iters)
{
    double a0 = threadIdx.x * 1e-9 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double b = 1.0000001, c = 1e-9;
    for (int it = 0; it < iters; ++it) {
#define PIORAN_F8 "v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t" \
                  "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9\n\t"
        asm volatile(PIORAN_F8 PIORAN_F8 PIORAN_F8 PIORAN_F8 PIORAN_F8 PIORAN_F8 PIORAN_F8 PIORAN_F8
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c));
#undef PIORAN_F8
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
}  // namespace

// scratch: blocks * 256 doubles; returns the kernel's flop count through *flop (the caller times the stream)
int pioran_launch_fma_stream(int blocks, int iters, double* scratch, double* flop, hipStream_t stream)
{
    if (blocks < 1 || iters < 1 || !scratch) return PIORAN_ERR_ARG;
    hipLaunchKernelGGL(fma_stream_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, scratch, iters);
    if (flop) *flop = 2.0 * 64.0 * 256.0 * (double)blocks * (double)iters;
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}
