// Diagnostic build (never shipped): where does factor_block64 spend its cycles?  s_memtime stamps, shares only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/factor_probe.hip -o /tmp/factor_probe && /tmp/factor_probe
#include <hip/hip_runtime.h>
__device__ unsigned long long g_st[64];
#define PIORAN_STAMP(i)                                                                            \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        unsigned long long t_;                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                 \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (threadIdx.x == 0) g_st[i] = t_;                                                        \
    } while (0)
#include "../pioran.jl_amd/csrc/dense.hip"
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(64) probe_kernel(double* A, int64_t ld, double* ws, unsigned long long* out)
{
    __shared__ double Ls[NB * LP];
    const int lane = threadIdx.x;
    PIORAN_STAMP(50);
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = A[lane + (int64_t)(16 * h + q) * ld];
#pragma unroll
        for (int q = 0; q < 16; ++q) Ls[lane * LP + 16 * h + q] = v[q];
    }
    __syncthreads();
    PIORAN_STAMP(51);
    const int bad = factor_block64(Ls, ws, lane);
    __syncthreads();
    PIORAN_STAMP(52);
    store_block_lower(Ls, A, ld, lane);
    PIORAN_STAMP(53);
    if (lane == 0) { for (int i = 0; i < 64; ++i) out[i] = g_st[i]; out[63] = bad; }
}

int main()
{
    const int n = 64, ld = 128;
    std::vector<double> h(ld * n, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) h[i + j * ld] = (i == j ? 70.0 : 0.0) + 1.0 / (1.0 + abs(i - j));
    double *dA, *dws; unsigned long long* dout;
    hipMalloc(&dA, sizeof(double) * ld * n); hipMalloc(&dws, sizeof(double) * 2048); hipMalloc(&dout, 64 * 8);
    unsigned long long st[64];
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpy(dA, h.data(), sizeof(double) * ld * n, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, 0, dA, (int64_t)ld, dws, dout);
        hipMemcpy(st, dout, sizeof(st), hipMemcpyDeviceToHost);
    }
    auto d = [&](int a, int b) { return (long long)(st[b] - st[a]); };
    printf("load->LDS %lld | factor %lld | writeback %lld  (cycles, 100 MHz ticks? see clock) bad=%llu\n", d(50, 51), d(51, 52), d(52, 53), st[63]);
    for (int s = 0; s < 4; ++s)
        printf(" s=%d: load tile %lld | chol16 ... | inverse %lld | publish %lld | trsm-mfma %lld | update-mfma %lld\n", s,
               d(8 * s + 0, 8 * s + 1), d(8 * s + 1, 8 * s + 2), d(8 * s + 2, 8 * s + 3), s < 3 ? d(8 * s + 3, 8 * s + 4) : 0,
               s < 3 ? d(8 * s + 4, 8 * s + 5) : 0);
    printf(" (first column = tile load + 16x16 cholesky)\n");
    return 0;
}
