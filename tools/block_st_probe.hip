// Diagnostic build (never shipped): where does a window of celerite_block_kernel<NB, 0, 0, 1> (the windowed FORWARD pass with the reverse mode's
// per-window stores) spend its cycles?  Same driver as tools/block_adjoint_probe.hip; the stamps are the forward kernel's.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/block_st_probe.hip -o tools/block_st_probe && tools/block_st_probe [J] [cd]
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
    const int64_t N = 10000; const int J = argc > 1 ? atoi(argv[1]) : 20; const int R = 2 * J; const bool cd = argc > 2 ? atoi(argv[2]) != 0 : true;
    std::vector<double> t(N), y(N), s2(N), c(J), d(J), A(J, 0.05), Bc(J, 0.01);
    for (int64_t n = 0; n < N; ++n) { t[n] = n + 0.3 * sin(1.7 * n); y[n] = sin(0.3 * n); s2[n] = 0.01; }
    for (int j = 0; j < J; ++j) { c[j] = 0.01 * (j + 1); d[j] = 0.02 * (j + 1); }
    if (argc > 3) {   // the bench model's spread of decay rates (7.5 decades: the fast terms underflow within a step)
        for (int j = 0; j < J; ++j) { c[j] = 2e-5 * pow(887.0 / 2e-5, (double)j / (J - 1)); d[j] = 1.7 * c[j]; }
    }
    std::vector<int32_t> rm(R);
    for (int j = 0; j < R; ++j) rm[j] = (j / 2) | ((j & 1) << 30);
    double *dt, *dy, *ds2, *dc, *dd, *dA, *dB, *dout, *btab, *gtab, *gw, *g; int32_t *drm, *dst;
    hipMalloc(&dt, N * 8); hipMalloc(&dy, N * 8); hipMalloc(&ds2, N * 8); hipMalloc(&dc, J * 8); hipMalloc(&dd, J * 8);
    hipMalloc(&dA, J * 8); hipMalloc(&dB, J * 8); hipMalloc(&dout, 8); hipMalloc(&drm, R * 4); hipMalloc(&dst, 4);
    hipMemcpy(dt, t.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(dy, y.data(), N * 8, hipMemcpyHostToDevice);
    hipMemcpy(ds2, s2.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), J * 8, hipMemcpyHostToDevice);
    hipMemcpy(dd, d.data(), J * 8, hipMemcpyHostToDevice); hipMemcpy(dA, A.data(), J * 8, hipMemcpyHostToDevice);
    hipMemcpy(dB, Bc.data(), J * 8, hipMemcpyHostToDevice); hipMemcpy(drm, rm.data(), R * 4, hipMemcpyHostToDevice);
    hipMalloc(&btab, pioran_block_table_doubles(N, R, J) * 8);
    hipMalloc(&gtab, pioran_block_gtab_doubles(N, R) * 8);
    hipMalloc(&gw, pioran_block_grad_workspace_doubles(1, N, R) * 8);
    hipMalloc(&g, (4 * J + 2) * 8);
    pioran_launch_block_table(N, R, J, drm, dt, dc, dd, dy, ds2, btab, 0);
    pioran_launch_block_gtab(N, R, J, drm, dt, dc, dd, ds2, gtab, 0);
    ScanParams p{}; p.N = N; p.J = J; p.R = R; p.standard_rows = 1; p.B = 1; p.rowmap = drm; p.A = dA; p.Bc = dB;
    p.out = dout; p.status = dst; p.npd_rows = 0; p.gw = gw; p.t = dt; p.y = dy; p.s2 = ds2; p.C = dc; p.D = dd;
    int rc = 0;
    for (int rep = 0; rep < 2; ++rep) {
        rc = pioran_launch_block_grad(p, btab, gtab, g, g + J, g + 4 * J, g + 4 * J + 1, cd ? g + 2 * J : nullptr, cd ? g + 3 * J : nullptr, 0);
        hipDeviceSynchronize();
    }
    unsigned long long acc[8][16]; double out; std::vector<double> gh(4 * J + 2);
    hipMemcpyFromSymbol(acc, HIP_SYMBOL(g_acc), sizeof(acc)); hipMemcpy(&out, dout, 8, hipMemcpyDeviceToHost);
    hipMemcpy(gh.data(), g, (4 * J + 2) * 8, hipMemcpyDeviceToHost);
    const char* nm[12] = {"P6 of the previous window (update)", "M' = U~'T, publish, X", "barrier 1", "chain: read M, Gram, Sigma", "chain: Sigma columns from LDS",
                          "chain: LDL' + inverse", "chain: publish L^-1, 1/D, logdet", "rescale T, A and U~ of the next window", "barrier 2",
                          "Y^' = L^-1 X', publish", "barrier 3", "tail"};
    const double nw = (double)((N + 15) / 16);
    printf("J = %d cd = %d rc = %d: logl = %.6f  d/da_0 = %.6e\n", J, (int)cd, rc, out, gh[0]);
    for (int wv = 0; wv < 4; ++wv) {
        unsigned long long tot = 0; for (int i = 0; i < 12; ++i) tot += acc[wv][i];
        printf("wavefront %d: %.0f cycles per window (stamps included)\n", wv, (double)tot / nw);
        for (int i = 0; i < 12; ++i) printf("  %-52s %8.1f\n", nm[i], (double)acc[wv][i] / nw);
    }
    return 0;
}
