// Diagnostic build (never shipped): where does a PAIR of time steps of the two-step scan spend its cycles?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/scan_probe.hip -o /tmp/scan_probe && /tmp/scan_probe [B]
// s_memtime stamps around the phases of `pair` (celerite_scan.hip), accumulated per phase over all pairs by lane 0 of wavefront 0
// of block 0 (the stamps are scalar instructions of that wavefront's stream; every other wavefront runs the same code unstamped
// in effect: the accumulators are per wavefront and only one is flushed).  Shapes: DRWCelerite-20 (60 rows: rpl4_cbr4_nsrc4_b5a,
// one draw per wavefront) and SHO-20 (40 rows: rpl3_cbr2_nsrc7_p, two draws per wavefront); B draws (default 4096: two
// wavefronts per SIMD; 1024: one).
#include <hip/hip_runtime.h>
__device__ unsigned long long g_acc[8];
#define PIORAN_SSTAMP_DECL unsigned long long wacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long wprev_ = 0;
#define PIORAN_SSTAMP(i)                                                                 \
    do {                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                               \
        unsigned long long t_;                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
        __builtin_amdgcn_sched_barrier(0);                                               \
        wacc_[i] += t_ - wprev_;                                                         \
        wprev_ = t_;                                                                     \
    } while (0)
#define PIORAN_SSTAMP_FLUSH if (threadIdx.x == 0 && blockIdx.x == 0) { for (int i_ = 0; i_ < 8; ++i_) g_acc[i_] = wacc_[i_]; }
#include "../pioran.jl_amd/csrc/celerite_scan.hip"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

static void run(const char* name, int J, int ncomplex, int64_t B)
{
    const int64_t N = 10000;
    std::vector<int32_t> rm;
    for (int j = 0; j < J; ++j) { rm.push_back(j); if (j < ncomplex) rm.push_back(j | (1 << 30)); }
    const int R = (int)rm.size(), Rp = R + 2; const int64_t rec = 3 * Rp + 2;
    std::vector<double> tab((N + 1) * rec), A(B * J), Bc(B * J);
    for (int64_t n = 0; n <= N; ++n) {
        double* r = &tab[n * rec];
        for (int j = 0; j < R; ++j) {
            const int term = rm[j] & 0xfffff; const bool ks = (rm[j] >> 30) & 1; const bool cplx = term < ncomplex;
            const double ph = cplx ? 0.013 * (term + 1) * n : 0.0;
            r[j] = ks ? sin(ph) : cos(ph); r[Rp + j] = ks ? cos(ph) : sin(ph); r[2 * Rp + j] = exp(-0.004 * (term + 1));
        }
        r[R] = 1; r[Rp + R] = 0; r[2 * Rp + R] = 1; r[R + 1] = 0; r[Rp + R + 1] = 0; r[2 * Rp + R + 1] = 1;
        r[3 * Rp] = sin(0.3 * n); r[3 * Rp + 1] = 0.01;
    }
    for (int64_t b = 0; b < B; ++b)
        for (int j = 0; j < J; ++j) { A[b * J + j] = 0.05 + 0.001 * ((b + j) % 7); Bc[b * J + j] = j < ncomplex ? 0.01 : 0.0; }
    double *dtab, *dA, *dB, *dout; int32_t *drm, *dst;
    hipMalloc(&dtab, tab.size() * 8); hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, A.size() * 8); hipMalloc(&dout, B * 8);
    hipMalloc(&drm, R * 4); hipMalloc(&dst, B * 4);
    hipMemcpy(dtab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dB, Bc.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(drm, rm.data(), R * 4, hipMemcpyHostToDevice);
    ScanParams p{}; p.N = N; p.J = J; p.R = R; p.standard_rows = ncomplex == J ? 1 : 2; p.n_complex = ncomplex; p.B = B; p.tab = dtab;
    p.rowmap = drm; p.A = dA; p.Bc = dB; p.out = dout; p.status = dst; p.rec_stride = rec;
    for (int rep = 0; rep < 2; ++rep) { pioran_launch_scan(p, 0); hipDeviceSynchronize(); }
    unsigned long long acc[8]; double out;
    hipMemcpyFromSymbol(acc, HIP_SYMBOL(g_acc), sizeof(acc)); hipMemcpy(&out, dout, 8, hipMemcpyDeviceToHost);
    const char* nm[8] = {"update pass of T (+ loop back edge)", "coefficients, u, u~, phi products", "mat-vec pass over T",
                         "exchange of r across the DPP rows", "dot products + row sums sA, sB", "D_A, reciprocal, m_A, h_A, g partial",
                         "loads of the next record + row sum g", "D_B, reciprocal, m_B, w, log-det / quad"};
    unsigned long long tot = 0; for (int i = 0; i < 8; ++i) tot += acc[i];
    const double np = (double)(N / 2);
    printf("%s, B = %ld, config %s: logl[0] = %.6f ; %.0f cycles per pair of steps (shader clock, stamps included)\n", name, (long)B,
           pioran_scan_config_name(0), out, (double)tot / np);
    const int order[8] = {1, 2, 3, 4, 5, 6, 7, 0};
    for (int k = 0; k < 8; ++k) printf("  %-46s %8.1f\n", nm[order[k]], (double)acc[order[k]] / np);
    hipFree(dtab); hipFree(dA); hipFree(dB); hipFree(dout); hipFree(drm); hipFree(dst);
}

int main(int argc, char** argv)
{
    const int64_t B = argc > 1 ? atol(argv[1]) : 4096;
    run("DRWCelerite-20 shape (20 two-row + 20 one-row terms)", 40, 20, B);
    run("SHO-20 shape (20 two-row terms)", 20, 20, B);
    return 0;
}
