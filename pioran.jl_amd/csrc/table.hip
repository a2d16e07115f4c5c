// Shared per-step table build: cos(d_j t_n), sin(d_j t_n), exp(-c_j (t_n - t_{n-1})).
//
// These are the 3J transcendentals per time step of src/celerite_solver.jl:52-54.  In the
// reference's `approx` flow (c_j, d_j) depend only on the spectral grid (src/psd.jl:250,266-267),
// i.e. on the data set and not on the sampled parameters, so one table serves every draw of a
// batch.  d_j * t_n reaches 1e6-1e7 rad: the device-library sincos does a full Payne-Hanek
// reduction (no fast-math, no angle-addition recurrences).  Coalesced HBM write, jp fastest.
#include "common.h"

namespace {
__global__ void __launch_bounds__(256) table_kernel(int64_t N, int32_t J, const double* __restrict__ t,
                                                    const double* __restrict__ c,
                                                    const double* __restrict__ d, double* __restrict__ tab)
{
    const int32_t Jp = J + 2;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * Jp) return;
    const int64_t n = idx / Jp;
    const int32_t jp = (int32_t)(idx - n * Jp);
    // column J: inert padding term (1, 0, 0); column J+1: the y row of the scan (0, 0, 1)
    double co = jp == J ? 1.0 : 0.0, si = 0.0, ph = jp == J ? 0.0 : 1.0;
    if (jp < J) {
        const double tn = t[n];
        sincos(d[jp] * tn, &si, &co);
        ph = n > 0 ? exp(-c[jp] * (tn - t[n - 1])) : 0.0;
    }
    double* rec = tab + n * 3 * Jp;
    rec[jp] = co;
    rec[Jp + jp] = si;
    rec[2 * Jp + jp] = ph;
}
}  // namespace

int pioran_launch_table(int64_t N, int32_t J, const double* t, const double* c, const double* d,
                        double* tab, hipStream_t stream)
{
    const int64_t total = N * (int64_t)(J + 2);
    const int64_t blocks = (total + 255) / 256;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return PIORAN_ERR_ARG;
    hipLaunchKernelGGL(table_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, N, J, t, c, d, tab);
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}
