// Diagnostic build (never shipped): where does a window of celerite_block_kernel spend its cycles?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/block_probe.hip -o tools/block_probe && tools/block_probe [J]
// s_memtime stamps around the phases of a window, accumulated per phase by lane 0 of every wavefront of block 0.
#include <hip/hip_runtime.h>
__device__ unsigned long long g_acc[8][16];
#define PIORAN_BSTAMP_DECL unsigned long long wacc_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long wprev_ = 0; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wprev_)::"memory");
#define PIORAN_BSTAMP(i)                                                                 \
    do {                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                               \
        unsigned long long t_;                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
        __builtin_amdgcn_sched_barrier(0);                                               \
        wacc_[i] += t_ - wprev_;                                                         \
        wprev_ = t_;                                                                     \
    } while (0)
#define PIORAN_BSTAMP_FLUSH if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { for (int i_ = 0; i_ < 16; ++i_) g_acc[threadIdx.x >> 6][i_] = wacc_[i_]; }
#include "../pioran.jl_amd/csrc/celerite_block.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

int main(int argc, char** argv)
{
    const int64_t N = 10000; const int J = argc > 1 ? atoi(argv[1]) : 20; const int R = 2 * J;
    const int B = argc > 2 ? atoi(argv[2]) : 1;   // draws (workgroups): 512 = two per CU (the pair table then comes from global memory)
    std::vector<double> t(N), y(N), s2(N), c(J), d(J), A((size_t)J * B, 0.05), Bc((size_t)J * B, 0.01);
    for (int64_t n = 0; n < N; ++n) { t[n] = n + 0.3 * sin(1.7 * n); y[n] = sin(0.3 * n); s2[n] = 0.01; }
    for (int j = 0; j < J; ++j) { c[j] = 0.01 * (j + 1); d[j] = 0.02 * (j + 1); }
    std::vector<int32_t> rm(R);
    for (int j = 0; j < R; ++j) rm[j] = (j / 2) | ((j & 1) << 30);
    double *dt, *dy, *ds2, *dc, *dd, *dA, *dB, *dout, *btab; int32_t *drm, *dst;
    hipMalloc(&dt, N * 8); hipMalloc(&dy, N * 8); hipMalloc(&ds2, N * 8); hipMalloc(&dc, J * 8); hipMalloc(&dd, J * 8);
    hipMalloc(&dA, (size_t)J * B * 8); hipMalloc(&dB, (size_t)J * B * 8); hipMalloc(&dout, 8 * B); hipMalloc(&drm, R * 4); hipMalloc(&dst, 4 * B);
    hipMemcpy(dt, t.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(dy, y.data(), N * 8, hipMemcpyHostToDevice);
    hipMemcpy(ds2, s2.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), J * 8, hipMemcpyHostToDevice);
    hipMemcpy(dd, d.data(), J * 8, hipMemcpyHostToDevice); hipMemcpy(dA, A.data(), (size_t)J * B * 8, hipMemcpyHostToDevice);
    hipMemcpy(dB, Bc.data(), (size_t)J * B * 8, hipMemcpyHostToDevice); hipMemcpy(drm, rm.data(), R * 4, hipMemcpyHostToDevice);
    hipMalloc(&btab, pioran_block_table_doubles(N, R, J) * 8);
    pioran_launch_block_table(N, R, J, drm, dt, dc, dd, dy, ds2, btab, 0);
    ScanParams p{}; p.N = N; p.J = J; p.R = R; p.standard_rows = 1; p.B = B; p.rowmap = drm; p.A = dA; p.Bc = dB;
    p.out = dout; p.status = dst; p.npd_rows = 0;
    for (int rep = 0; rep < 2; ++rep) { pioran_launch_scan_block(p, btab, 0); hipDeviceSynchronize(); }
    unsigned long long acc[8][16]; double out;
    hipMemcpyFromSymbol(acc, HIP_SYMBOL(g_acc), sizeof(acc)); hipMemcpy(&out, dout, 8, hipMemcpyDeviceToHost);
    const char* nm[12] = {"P6 of the previous window (update)", "M' = U~'T, publish, X", "barrier 1", "chain: read M, Gram, Sigma", "chain: Sigma columns from LDS",
                          "chain: LDL' + inverse", "chain: publish L^-1, 1/D, logdet", "rescale T, A and U~ of the next window", "barrier 2",
                          "Y^' = L^-1 X', publish", "barrier 3", "tail"};
    const double nw = (double)((N + 15) / 16);
    printf("J = %d: logl = %.6f\n", J, out);
    const int nbk = (R + 1 + 15) / 16;
    for (int wv = 0; wv < (nbk >= 4 ? (nbk + 2 > 8 ? 8 : nbk + 2) : 4); ++wv) {
        unsigned long long tot = 0; for (int i = 0; i < 11; ++i) tot += acc[wv][i];
        printf("wavefront %d: %.0f cycles per window (stamps included)\n", wv, (double)tot / nw);
        for (int i = 0; i < 11; ++i) printf("  %-46s %8.1f\n", nm[i], (double)acc[wv][i] / nw);
        printf("    of the rescale / A / U~ phase: rescale T %.1f, A %.1f, U~ and V^ %.1f (the rest: DMA issue)\n", (double)acc[wv][12] / nw,
               (double)acc[wv][13] / nw, (double)acc[wv][14] / nw);
    }
    return 0;
}
