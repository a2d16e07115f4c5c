// Diagnostic build (never shipped): how evenly do the 2048 wavefronts of the headline launch (SHO-20 shape, B = 4096) finish?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wave_spread_probe.hip -o tools/wave_spread_probe && tools/wave_spread_probe [B]
// Every wavefront records s_memrealtime (100 MHz) at its start and end and the XCC it ran on; the host prints the distribution of
// durations and of end times per XCC.  A launch lasts as long as its slowest wavefront.
#include <hip/hip_runtime.h>
__device__ unsigned long long g_w[1 << 16][4];
#define PIORAN_SSTAMP_DECL unsigned long long wt0_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wt0_)::"memory");
#define PIORAN_SSTAMP(i)
#define PIORAN_SSTAMP_FLUSH                                                                                     \
    {                                                                                                           \
        unsigned long long wt1_; unsigned xcc_, hwid_;                                                          \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wt1_)::"memory");                        \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                                     \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid_));                                     \
        if ((threadIdx.x & 63) == 0) {                                                                          \
            const int wi_ = blockIdx.x * 4 + (threadIdx.x >> 6);                                                \
            g_w[wi_][0] = wt0_; g_w[wi_][1] = wt1_; g_w[wi_][2] = xcc_; g_w[wi_][3] = hwid_;                     \
        }                                                                                                       \
    }
#include "../pioran.jl_amd/csrc/celerite_scan.hip"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv)
{
    const int64_t B = argc > 1 ? atol(argv[1]) : 4096;
    const int64_t N = 10000; const int J = 20;
    std::vector<int32_t> rm;
    for (int j = 0; j < J; ++j) { rm.push_back(j); rm.push_back(j | (1 << 30)); }
    const int R = (int)rm.size(), Rp = R + 2; const int64_t rec = 3 * Rp + 2;
    std::vector<double> tab((N + 1) * rec), A(B * J), Bc(B * J);
    for (int64_t n = 0; n <= N; ++n) {
        double* r = &tab[n * rec];
        for (int j = 0; j < R; ++j) {
            const int term = rm[j] & 0xfffff; const bool ks = (rm[j] >> 30) & 1;
            const double ph = 0.013 * (term + 1) * n;
            r[j] = ks ? sin(ph) : cos(ph); r[Rp + j] = ks ? cos(ph) : sin(ph); r[2 * Rp + j] = exp(-0.004 * (term + 1));
        }
        r[R] = 1; r[Rp + R] = 0; r[2 * Rp + R] = 1; r[R + 1] = 0; r[Rp + R + 1] = 0; r[2 * Rp + R + 1] = 1;
        r[3 * Rp] = sin(0.3 * n); r[3 * Rp + 1] = 0.01;
    }
    for (int64_t b = 0; b < B; ++b)
        for (int j = 0; j < J; ++j) { A[b * J + j] = 0.05 + 0.001 * ((b + j) % 7); Bc[b * J + j] = 0.01; }
    double *dtab, *dA, *dB, *dout; int32_t *drm, *dst;
    hipMalloc(&dtab, tab.size() * 8); hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, A.size() * 8); hipMalloc(&dout, B * 8);
    hipMalloc(&drm, R * 4); hipMalloc(&dst, B * 4);
    hipMemcpy(dtab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dB, Bc.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(drm, rm.data(), R * 4, hipMemcpyHostToDevice);
    ScanParams p{}; p.N = N; p.J = J; p.R = R; p.standard_rows = 1; p.n_complex = J; p.B = B; p.tab = dtab;
    p.rowmap = drm; p.A = dA; p.Bc = dB; p.out = dout; p.status = dst; p.rec_stride = rec;
    for (int rep = 0; rep < 3; ++rep) { pioran_launch_scan(p, 0); hipDeviceSynchronize(); }
    const int nw = (int)((B + 1) / 2);
    std::vector<unsigned long long> w((size_t)(1 << 16) * 4);
    hipMemcpyFromSymbol(w.data(), HIP_SYMBOL(g_w), w.size() * 8);
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int i = 0; i < nw; ++i) { t0 = std::min(t0, w[4 * i]); t1 = std::max(t1, w[4 * i + 1]); }
    printf("config %s, B = %ld, %d wavefronts: launch %.3f ms (first start to last end, 100 MHz clock)\n", pioran_scan_config_name(0), (long)B, nw,
           (double)(t1 - t0) * 1e-5);
    std::vector<double> dur(nw), end(nw), start(nw);
    for (int i = 0; i < nw; ++i) { dur[i] = (double)(w[4 * i + 1] - w[4 * i]) * 1e-5; end[i] = (double)(w[4 * i + 1] - t0) * 1e-5; start[i] = (double)(w[4 * i] - t0) * 1e-5; }
    auto q = [](std::vector<double> v, double f) { std::sort(v.begin(), v.end()); return v[(size_t)(f * (v.size() - 1))]; };
    printf("wavefront duration ms: min %.3f  5%% %.3f  median %.3f  95%% %.3f  max %.3f ; start spread: median %.3f max %.3f\n", q(dur, 0), q(dur, 0.05),
           q(dur, 0.5), q(dur, 0.95), q(dur, 1), q(start, 0.5), q(start, 1));
    {
        int cnt[16] = {0};
        for (int i = 0; i < nw; ++i) cnt[w[4 * i + 3] & 0xf]++;
        printf("wave slots (HW_ID[3:0]):");
        for (int i = 0; i < 16; ++i) if (cnt[i]) printf(" %d:%d", i, cnt[i]);
        printf("\n");
    }
    for (int x = 0; x < 8; ++x) {
        std::vector<double> d, e;
        for (int i = 0; i < nw; ++i) if ((int)(w[4 * i + 2] & 0xf) == x) { d.push_back(dur[i]); e.push_back(end[i]); }
        if (!d.empty()) printf("  XCC %d: %4zu wavefronts, duration median %.3f max %.3f, last end %.3f ms\n", x, d.size(), q(d, 0.5), q(d, 1), q(e, 1));
    }
    return 0;
}
