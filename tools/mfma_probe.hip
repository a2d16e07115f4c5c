// Diagnostic (never shipped): what the FP64 matrix pipe of gfx950 gives next to the FP64 vector ALU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_probe.hip -o tools/mfma_probe && tools/mfma_probe
// (1) issue interval of v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64, independent and chained accumulators;
// (2) the same MFMA stream with F v_fma_f64 placed between consecutive MFMAs of ONE wavefront (do they overlap?);
// (3) an MFMA-only wavefront and a VALU-only wavefront sharing a SIMD;
// (4) operand / result lane maps of the 4x4x4 (four-block) form, decoded with one-hot operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

#define FMA1(r) "v_fma_f64 %" #r ", %" #r ", %[b], %[c]\n\t"
#define FMA4 FMA1(0) FMA1(1) FMA1(2) FMA1(3)
#define FMA8 FMA4 FMA1(4) FMA1(5) FMA1(6) FMA1(7)

template <int F>
__device__ __forceinline__ void fillers(double& a0, double& a1, double& a2, double& a3, double& a4, double& a5, double& a6, double& a7,
                                        double b, double c)
{
    if constexpr (F == 4)
        asm volatile(FMA4 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : [b] "v"(b), [c] "v"(c));
    if constexpr (F == 8)
        asm volatile(FMA8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : [b] "v"(b), [c] "v"(c));
    if constexpr (F == 12)
        asm volatile(FMA8 FMA4 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : [b] "v"(b), [c] "v"(c));
    if constexpr (F == 16)
        asm volatile(FMA8 FMA8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : [b] "v"(b), [c] "v"(c));
}

// KIND 0: 16x16x4, 4 accumulators; 1: one accumulator; 2: 4x4x4, 8 accumulators; 3: 4x4x4 one accumulator;
// 4: 16x16x4 (4 acc) + F fillers per MFMA; 5: waves 0-3 of a block MFMA-only, waves 4-7 VALU-only (64 fma per iteration)
template <int KIND, int F>
__global__ void __launch_bounds__(1024) probe(double* out, int iters, unsigned long long* cyc, unsigned long long* real)
{
    const double x = threadIdx.x * 1e-9 + 1.0;
    double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
    const double b = 1.0000001, c = 1e-9;
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
    const double ma = x * 1e-3, mb = 1.0 - x * 1e-3;
    const int wave = threadIdx.x >> 6;
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0 || KIND == 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, c0, 0, 0, 0); fillers<F>(a0, a1, a2, a3, a4, a5, a6, a7, b, c);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, c1, 0, 0, 0); fillers<F>(a0, a1, a2, a3, a4, a5, a6, a7, b, c);
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, c2, 0, 0, 0); fillers<F>(a0, a1, a2, a3, a4, a5, a6, a7, b, c);
                c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, c3, 0, 0, 0); fillers<F>(a0, a1, a2, a3, a4, a5, a6, a7, b, c);
            }
        } else if constexpr (KIND == 1) {
#pragma unroll
            for (int k = 0; k < 16; ++k) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, c0, 0, 0, 0);
        } else if constexpr (KIND == 2) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(ma, mb, s0, 0, 0, 0); s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(ma, mb, s1, 0, 0, 0);
                s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(ma, mb, s2, 0, 0, 0); s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(ma, mb, s3, 0, 0, 0);
                s4 = __builtin_amdgcn_mfma_f64_4x4x4f64(ma, mb, s4, 0, 0, 0); s5 = __builtin_amdgcn_mfma_f64_4x4x4f64(ma, mb, s5, 0, 0, 0);
                s6 = __builtin_amdgcn_mfma_f64_4x4x4f64(ma, mb, s6, 0, 0, 0); s7 = __builtin_amdgcn_mfma_f64_4x4x4f64(ma, mb, s7, 0, 0, 0);
            }
        } else if constexpr (KIND == 3) {
#pragma unroll
            for (int k = 0; k < 16; ++k) s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(ma, mb, s0, 0, 0, 0);
        } else if constexpr (KIND == 5) {
            if (wave < 4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, c3, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) fillers<16>(a0, a1, a2, a3, a4, a5, a6, a7, b, c);
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c0.x + c0.y + c0.z + c0.w + c1.x + c2.y + c3.z
                                                 + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
    if ((threadIdx.x & 63) == 0) {
        cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
        real[blockIdx.x * (blockDim.x / 64) + wave] = r1 - r0;
    }
}

template <int KIND, int F>
void run(const char* name, int mfma_per_iter)
{
    const int iters = 4000;
    double* out; unsigned long long *cyc, *real;
    hipMalloc(&out, 8 * 1024 * 1024); hipMalloc(&cyc, 8 * 8192); hipMalloc(&real, 8 * 8192);
    std::vector<unsigned long long> hc(8192), hr(8192);
    printf("%-46s", name);
    const int cfgs[5][2] = {{256, 1}, {512, 1}, {1024, 1}, {512, 256}, {1024, 256}};
    for (auto& c : cfgs) {
        hipLaunchKernelGGL((probe<KIND, F>), dim3(c[1]), dim3(c[0]), 0, 0, out, 100, cyc, real);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<KIND, F>), dim3(c[1]), dim3(c[0]), 0, 0, out, iters, cyc, real);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const int wpb = c[0] / 64, nw = wpb * c[1];
        hipMemcpy(hc.data(), cyc, 8 * nw, hipMemcpyDeviceToHost); hipMemcpy(hr.data(), real, 8 * nw, hipMemcpyDeviceToHost);
        double sc = 0, sr = 0, sA = 0, sB = 0; int nA = 0, nB = 0;
        for (int i = 0; i < nw; ++i) {
            sc += hc[i]; sr += hr[i];
            if ((i % wpb) < 4) { sA += hc[i]; ++nA; } else { sB += hc[i]; ++nB; }
        }
        const double ghz = sc / sr * 0.1;
        if (KIND == 5) {
            // cycles per iteration of the MFMA-only waves (16 MFMA) and of the VALU-only waves (64 fma)
            printf(" | mfma-wave %6.1f cyc/16 mfma, valu-wave %6.1f cyc/64 fma, %4.2f GHz", sA / nA / iters, nB ? sB / nB / iters : 0.0, ghz);
        } else {
            const double per_simd = ms * 1e-3 * ghz * 1e9 / ((double)iters * mfma_per_iter * (c[0] / 256.0));
            printf(" | %6.2f cyc/mfma/SIMD %4.2f GHz %6.2f ms", per_simd, ghz, ms);
        }
    }
    printf("\n");
    hipFree(out); hipFree(cyc); hipFree(real);
}

// lane maps of the four-block 4x4x4 form: D = onehot(p) x onehot(q) is non-zero in exactly one lane when lane p of A and lane q of
// B meet (same block, same k)
__global__ void layout_4x4x4(int* where)
{
    const int lane = threadIdx.x;
    for (int p = 0; p < 64; ++p)
        for (int q = 0; q < 64; ++q) {
            const double a = lane == p ? 1.0 : 0.0, b = lane == q ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            if (d != 0.0) where[p * 64 + q] = lane;
        }
}

__global__ void layout_16x16x4(int* where)
{
    const int lane = threadIdx.x;
    for (int p = 0; p < 64; ++p)
        for (int q = 0; q < 64; ++q) {
            const double a = lane == p ? 1.0 : 0.0, b = lane == q ? 1.0 : 0.0;
            d4 z = {0, 0, 0, 0};
            const d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, z, 0, 0, 0);
            for (int g = 0; g < 4; ++g)
                if (d[g] != 0.0) where[p * 64 + q] = lane * 4 + g;
        }
}

int main()
{
    printf("%-46s | 4 waves (1/SIMD) | 8 waves (2/SIMD) | 16 waves (4/SIMD), one CU | whole chip 2/SIMD | whole chip 4/SIMD\n", "kind");
    run<0, 0>("mfma_f64_16x16x4, 4 accumulators", 16);
    run<1, 0>("mfma_f64_16x16x4, one accumulator (chain)", 16);
    run<2, 0>("mfma_f64_4x4x4_4b, 8 accumulators", 16);
    run<3, 0>("mfma_f64_4x4x4_4b, one accumulator (chain)", 16);
    run<4, 4>("16x16x4 + 4 v_fma_f64 per mfma (one wave)", 16);
    run<4, 8>("16x16x4 + 8 v_fma_f64 per mfma", 16);
    run<4, 12>("16x16x4 + 12 v_fma_f64 per mfma", 16);
    run<4, 16>("16x16x4 + 16 v_fma_f64 per mfma", 16);
    run<5, 0>("mfma-only waves 0-3 beside valu-only waves 4-7", 16);

    int* where; hipMalloc(&where, 4 * 4096);
    std::vector<int> h(4096);
    for (int form = 0; form < 2; ++form) {
        hipMemset(where, 0xff, 4 * 4096);
        if (form == 0) hipLaunchKernelGGL(layout_4x4x4, dim3(1), dim3(64), 0, 0, where);
        else hipLaunchKernelGGL(layout_16x16x4, dim3(1), dim3(64), 0, 0, where);
        hipMemcpy(h.data(), where, 4 * 4096, hipMemcpyDeviceToHost);
        printf("\n%s: rows = A lane p, columns = B lane q, entry = result %s (.. = no product)\n", form == 0 ? "4x4x4_4b" : "16x16x4",
               form == 0 ? "lane" : "lane*4+reg");
        for (int p = 0; p < 64; ++p) {
            printf("p=%2d:", p);
            for (int q = 0; q < 64; ++q) { if (h[p * 64 + q] < 0) printf(" .."); else printf(" %2x", h[p * 64 + q]); }
            printf("\n");
        }
    }
    return 0;
}
