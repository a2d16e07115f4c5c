// Diagnostic build (never shipped): where does a window of celerite_tile_kernel spend its cycles?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tile_probe.hip -o tools/tile_probe && tools/tile_probe [J] [B] [drw]
// s_memtime stamps around the phases of a window, accumulated per phase by every wavefront of workgroup 0.
#include <hip/hip_runtime.h>
__device__ unsigned long long g_tacc[4][16];
#define PIORAN_TSTAMP_DECL unsigned long long wacc_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long wprev_ = 0; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wprev_)::"memory");
#define PIORAN_TSTAMP(i)                                                                 \
    do {                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                               \
        unsigned long long t_;                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
        __builtin_amdgcn_sched_barrier(0);                                               \
        wacc_[i] += t_ - wprev_;                                                         \
        wprev_ = t_;                                                                     \
    } while (0)
#define PIORAN_TSTAMP_FLUSH if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { for (int i_ = 0; i_ < 16; ++i_) g_tacc[threadIdx.x >> 6][i_] = wacc_[i_]; }
__device__ unsigned long long g_aacc[4][16];
#define PIORAN_ASTAMP2_DECL PIORAN_TSTAMP_DECL
#define PIORAN_ASTAMP2(i) PIORAN_TSTAMP(i)
#define PIORAN_ASTAMP2_FLUSH if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { for (int i_ = 0; i_ < 16; ++i_) g_aacc[threadIdx.x >> 6][i_] = wacc_[i_]; }
#include "../pioran.jl_amd/csrc/celerite_block.hip"
#include "../pioran.jl_amd/csrc/celerite_tile.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

int main(int argc, char** argv)
{
    const int64_t N = 10000; const int J = argc > 1 ? atoi(argv[1]) : 20;
    const int B = argc > 2 ? atoi(argv[2]) : 4096;
    const bool drw = argc > 3;                      // DRWCelerite-like: the second half of the terms are real (one row each)
    const int R = drw ? J + J / 2 : 2 * J;
    std::vector<double> t(N), y(N), s2(N), c(J), d(J), A((size_t)J * B, 0.05), Bc((size_t)J * B, 0.01);
    for (int64_t n = 0; n < N; ++n) { t[n] = n + 0.3 * sin(1.7 * n); y[n] = sin(0.3 * n); s2[n] = 0.01; }
    for (int j = 0; j < J; ++j) { c[j] = 0.01 * (j + 1); d[j] = (drw && j >= J / 2) ? 0.0 : 0.02 * (j + 1); }
    if (drw) for (int b = 0; b < B; ++b) for (int j = J / 2; j < J; ++j) Bc[(size_t)b * J + j] = 0.0;
    std::vector<int32_t> rm;
    for (int j = 0; j < J; ++j) { rm.push_back(j); if (!(drw && j >= J / 2)) rm.push_back(j | (1 << 30)); }
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
    double* work; hipMalloc(&work, pioran_tile_workspace_doubles(B, N) * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        int rc = pioran_launch_scan_tile(p, btab, work, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        if (rc) { printf("launch rc %d\n", rc); return 1; }
    }
    unsigned long long acc[4][16]; double out;
    hipMemcpyFromSymbol(acc, HIP_SYMBOL(g_tacc), sizeof(acc)); hipMemcpy(&out, dout, 8, hipMemcpyDeviceToHost);
    const char* nm[12] = {"loop head", "M' = U~'T", "G = U~'M (LDS transposes)", "X'", "Sigma -> columns (LDS)", "LDL' + inverse", "L^-1 -> fragments, 1/D, logdet",
                          "Y^' = L^-1 X'", "rescale + update of T", "barrier (record k + 1 landed)", "DMA issue of record k + 2", "-"};
    const double nw = (double)((N + 15) / 16);
    const double F = (double)(N - 1) * (5.5 * R * R + 18.0 * R) * B;
    printf("J = %d rows = %d B = %d: logl = %.6f; %.3f ms per launch (stamps included) = %.1f k evals/s, %.3f of 78.6 TFLOP/s\n", J, R, B, out, ms, B / ms,
           F / (ms * 1e-3) / 78.6e12);
    for (int wv = 0; wv < 4; ++wv) {
        unsigned long long tot = 0; for (int i = 0; i < 13; ++i) tot += acc[wv][i];
        printf("wavefront %d: %.0f cycles per window (stamps included)\n", wv, (double)tot / nw);
        for (int i = 0; i < 12; ++i) printf("  %-46s %8.1f\n", nm[i], (double)acc[wv][i] / nw);
        printf("  %-46s %8.1f\n", "U~ of the next window", (double)acc[wv][0] / nw);
    }
    if (argc > 4 && R <= 63) {       // the reverse mode: forward pass with T stores + reverse kernel + post-pass
        double *gw, *gtab, *ga, *gb, *gn, *gm;
        hipMalloc(&gw, pioran_tile_grad_workspace_doubles(B, N, R) * 8); hipMalloc(&gtab, pioran_block_gtab_doubles(N, R) * 8);
        hipMalloc(&ga, (size_t)J * B * 8); hipMalloc(&gb, (size_t)J * B * 8); hipMalloc(&gn, 8 * B); hipMalloc(&gm, 8 * B);
        pioran_launch_block_gtab(N, R, J, drm, dt, dc, dd, ds2, gtab, 0);
        p.gw = gw;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            int rc = pioran_launch_tile_grad(p, btab, gtab, work, ga, gb, gn, gm, nullptr, nullptr, 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            if (rc) { printf("grad launch rc %d\n", rc); return 1; }
        }
        unsigned long long aacc[4][16];
        hipMemcpyFromSymbol(aacc, HIP_SYMBOL(g_aacc), sizeof(aacc));
        const char* an[13] = {"(loop edge)", "head: C_K stage, U~, T_k upper tiles -> LDS", "M' = U~'T_k", "G, X'", "C/D table loads, Sigma -> columns", "LDL'", "L^-1 operands, K",
                              "Q' = Sigma^-1 X'", "A: Q transposes, Q'T-, P", "B: S-, pair values out", "C: U~', S-U~', W' transposes", "D1: U~-' = W'T_k, accumulators",
                              "D2: next window's loads + T- update"};
        printf("value + gradient: %.3f ms per launch of %d chains (pre-pass + forward with T stores + reverse + post-pass; stamps included)\n", ms, B);
        unsigned long long tot = 0; for (int i = 0; i < 13; ++i) tot += aacc[0][i];
        printf("reverse kernel, wavefront 0: %.0f cycles per window\n", (double)tot / nw);
        for (int i = 1; i < 13; ++i) printf("  %-58s %8.1f\n", an[i], (double)aacc[0][i] / nw);

    }
    return 0;
}
