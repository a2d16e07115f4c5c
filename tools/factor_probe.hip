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
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) probe_kernel(double* A, int64_t ld, double* ws, unsigned long long* out)
{
    __shared__ double Ls[NB * LP];
    __shared__ int flag;
    const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
    PIORAN_STAMP(50);
    {
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = A[lane + (int64_t)(16 * h + q) * ld];
#pragma unroll
        for (int q = 0; q < 16; ++q) Ls[lane * LP + 16 * h + q] = v[q];
    }
    __syncthreads();
    PIORAN_STAMP(51);
    const int bad = factor_block64(Ls, ws, &flag, tid);
    __syncthreads();
    PIORAN_STAMP(52);
    store_block_lower(Ls, A, ld, tid);
    PIORAN_STAMP(53);
    if (tid == 0) { for (int i = 0; i < 64; ++i) out[i] = g_st[i]; out[63] = bad; }
}

int main()
{
    const int n = 64, ld = 128;
    std::vector<double> h(ld * n, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) h[i + j * ld] = i < j ? NAN : (i == j ? 70.0 : 0.0) + 1.0 / (1.0 + abs(i - j));   // (upper triangle poisoned: never read as data)
    double *dA, *dws; unsigned long long* dout;
    hipMalloc(&dA, sizeof(double) * ld * n); hipMalloc(&dws, sizeof(double) * 2048); hipMalloc(&dout, 64 * 8);
    unsigned long long st[64];
    std::vector<double> L(ld * n);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpy(dA, h.data(), sizeof(double) * ld * n, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(256), 0, 0, dA, (int64_t)ld, dws, dout);
        hipMemcpy(st, dout, sizeof(st), hipMemcpyDeviceToHost);
    }
    hipMemcpy(L.data(), dA, sizeof(double) * ld * n, hipMemcpyDeviceToHost);
    // residual of L L' against the input (lower triangle)
    double err = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double acc = 0.0;
            for (int k = 0; k <= j; ++k) acc += L[i + k * ld] * L[j + k * ld];
            err = fmax(err, fabs(acc - h[i + j * ld]));
        }
    // the four parked inverses (copied to ws): inv(C_ss) C_ss = I, exact zeros above the diagonal
    std::vector<double> W(1024);
    hipMemcpy(W.data(), dws, sizeof(double) * 1024, hipMemcpyDeviceToHost);
    double ierr = 0.0, upper = 0.0;
    for (int sb = 0; sb < 4; ++sb)
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double acc = 0.0;
                for (int k = 0; k < 16; ++k) acc += W[sb * 256 + i * 16 + k] * (k >= j ? L[(16 * sb + k) + (16 * sb + j) * ld] : 0.0);
                ierr = fmax(ierr, fabs(acc - (i == j ? 1.0 : 0.0)));
                if (j > i) upper = fmax(upper, fabs(W[sb * 256 + i * 16 + j]));
            }
    auto d = [&](int a, int b) { return (long long)(st[b] - st[a]); };
    printf("load->LDS %lld | factor %lld | writeback %lld  (shader clock cycles) bad=%llu  max|LL'-A|=%.3e  max|inv(C_ss) C_ss - I|=%.3e  max|upper of inv|=%.1e\n", d(50, 51),
           d(51, 52), d(52, 53), st[63], err, ierr, upper);
    for (int s = 0; s < 4; ++s)
        printf(" s=%d: tile load + Gauss-Jordan sweep (wave 0) %lld | raw result to LDS + barrier %lld | chain's solve + diagonal-tile update %lld\n", s,
               d(8 * s + 0, 8 * s + 1), d(8 * s + 1, 8 * s + 2), d(8 * s + 2, 8 * s + 3));
    printf(" last round's end -> return (flag, inverses to the workspace) %lld\n", d(8 * 3 + 3, 40));
    return 0;
}
