// Diagnostic build (never shipped): where does a time step of celerite_wide_kernel spend its cycles?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wide_probe.hip -o /tmp/wide_probe && /tmp/wide_probe
// s_memtime stamps around the phases of do_step, accumulated per phase over all steps by thread 0 of the block.
#include <hip/hip_runtime.h>
__device__ unsigned long long g_acc[8];
#define PIORAN_WSTAMP_DECL unsigned long long wacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long wprev_ = 0;
#define PIORAN_WSTAMP(i)                                                                 \
    do {                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                               \
        unsigned long long t_;                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
        __builtin_amdgcn_sched_barrier(0);                                               \
        wacc_[i] += t_ - wprev_;                                                         \
        wprev_ = t_;                                                                     \
    } while (0)
#define PIORAN_WSTAMP_FLUSH if (threadIdx.x == 0 && blockIdx.x == 0) { for (int i_ = 0; i_ < 8; ++i_) g_acc[i_] = wacc_[i_]; }
#include "../pioran.jl_amd/csrc/celerite_wide.hip"
#include <cstdio>
#include <vector>
#include <cmath>

int main()
{
    const int64_t N = 10000; const int J = 20, R = 40, Rp = R + 2; const int64_t rec = 3 * Rp + 2;
    std::vector<double> tab((N + 1) * rec), A(J, 0.05), Bc(J, 0.01);
    for (int64_t n = 0; n <= N; ++n) {
        double* r = &tab[n * rec];
        for (int j = 0; j < R; ++j) { r[j] = cos(0.01 * (j + 1) * n); r[Rp + j] = sin(0.01 * (j + 1) * n); r[2 * Rp + j] = exp(-0.01 * (j / 2 + 1)); }
        r[R] = 1; r[Rp + R] = 0; r[2 * Rp + R] = 1; r[R + 1] = 0; r[Rp + R + 1] = 0; r[2 * Rp + R + 1] = 1;
        r[3 * Rp] = sin(0.3 * n); r[3 * Rp + 1] = 0.01;
    }
    std::vector<int32_t> rm(R);
    for (int j = 0; j < R; ++j) rm[j] = (j / 2) | ((j & 1) << 30);
    double *dtab, *dA, *dB, *dout; int32_t *drm, *dst;
    hipMalloc(&dtab, tab.size() * 8); hipMalloc(&dA, J * 8); hipMalloc(&dB, J * 8); hipMalloc(&dout, 8); hipMalloc(&drm, R * 4); hipMalloc(&dst, 4);
    hipMemcpy(dtab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dA, A.data(), J * 8, hipMemcpyHostToDevice);
    hipMemcpy(dB, Bc.data(), J * 8, hipMemcpyHostToDevice); hipMemcpy(drm, rm.data(), R * 4, hipMemcpyHostToDevice);
    ScanParams p{}; p.N = N; p.J = J; p.R = R; p.standard_rows = 1; p.B = 1; p.tab = dtab; p.rowmap = drm; p.A = dA; p.Bc = dB;
    p.out = dout; p.status = dst; p.rec_stride = rec; p.npd_rows = 0;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(celerite_wide_kernel<3>, dim3(1), dim3(256), 0, 0, p); hipDeviceSynchronize(); }
    unsigned long long acc[8]; double out;
    hipMemcpyFromSymbol(acc, HIP_SYMBOL(g_acc), sizeof(acc)); hipMemcpy(&out, dout, 8, hipMemcpyDeviceToHost);
    const char* nm[7] = {"loop/other -> step start", "prefetch issue + u + S update + q share", "DPP butterfly + num + u'q share",
                         "LDS publish + barrier", "LDS read + 16-sum", "D, reciprocal, w", "logdet/quad bookkeeping"};
    unsigned long long tot = 0; for (int i = 0; i < 7; ++i) tot += acc[i];
    printf("logl = %.6f ; %.0f cycles per step (shader clock, stamps included)\n", out, (double)tot / (N - 1));
    for (int i = 0; i < 7; ++i) printf("  %-46s %7.1f\n", nm[i], (double)acc[i] / (N - 1));
    return 0;
}
