// Diagnostic (never shipped): issue cost of the instructions the scan kernels are made of, on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/valu_probe.hip -o tools/valu_probe && tools/valu_probe
// Every kind is a loop of 64 independent-enough instructions; reported: shader cycles (s_memtime) per instruction per
// wavefront for 1 and 2 wavefronts per SIMD on ONE CU, and the effective shader clock (s_memtime / s_memrealtime) for a
// launch that fills the chip (2048 wavefronts) — i.e. whether the clock drops under sustained FP64 issue.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define R8(x) x x x x x x x x
#define DPPC " row_newbcast:3 row_mask:0xf bank_mask:0xf"

template <int KIND>
__global__ void __launch_bounds__(1024) probe(double* out, int iters, unsigned long long* cyc, unsigned long long* real)
{
    double a0 = threadIdx.x * 1e-9 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double b = 1.0000001, c = 1e-9, d = 0.9999999;
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {   // v_fma_f64 a = a * b + c
            asm volatile(R8("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                            "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if constexpr (KIND == 1) {   // v_mul_f64
            asm volatile(R8("v_mul_f64 %0, %0, %8\n\tv_mul_f64 %1, %1, %8\n\tv_mul_f64 %2, %2, %8\n\tv_mul_f64 %3, %3, %8\n\t"
                            "v_mul_f64 %4, %4, %8\n\tv_mul_f64 %5, %5, %8\n\tv_mul_f64 %6, %6, %8\n\tv_mul_f64 %7, %7, %8\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if constexpr (KIND == 2) {   // v_fmac_f64_dpp row_newbcast: a += bcast(c) * b
            asm volatile(R8("v_fmac_f64_dpp %0, %9, %8" DPPC "\n\tv_fmac_f64_dpp %1, %9, %8" DPPC "\n\tv_fmac_f64_dpp %2, %9, %8" DPPC "\n\t"
                            "v_fmac_f64_dpp %3, %9, %8" DPPC "\n\tv_fmac_f64_dpp %4, %9, %8" DPPC "\n\tv_fmac_f64_dpp %5, %9, %8" DPPC "\n\t"
                            "v_fmac_f64_dpp %6, %9, %8" DPPC "\n\tv_fmac_f64_dpp %7, %9, %8" DPPC "\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if constexpr (KIND == 3) {   // v_mov_b64_dpp row_newbcast
            asm volatile(R8("v_mov_b64_dpp %0, %8" DPPC "\n\tv_mov_b64_dpp %1, %8" DPPC "\n\tv_mov_b64_dpp %2, %8" DPPC "\n\tv_mov_b64_dpp %3, %8" DPPC "\n\t"
                            "v_mov_b64_dpp %4, %8" DPPC "\n\tv_mov_b64_dpp %5, %8" DPPC "\n\tv_mov_b64_dpp %6, %8" DPPC "\n\tv_mov_b64_dpp %7, %8" DPPC "\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if constexpr (KIND == 4) {   // the scan's column pair for 3 rows: 23 instructions (PairFirst + PairSecond), x3 = 69
            double pk, p0, p1, p2, q0 = 0, q1 = 0, q2 = 0;
            asm volatile(
                "s_nop 1\n\tv_mov_b64_dpp %[pk], %[d]" DPPC "\n\t"
                "v_fmac_f64_dpp %[s0], %[c], %[b]" DPPC "\n\tv_fmac_f64_dpp %[s1], %[c], %[b]" DPPC "\n\tv_fmac_f64_dpp %[s2], %[c], %[b]" DPPC "\n\t"
                "v_mul_f64 %[p0], %[d], %[pk]\n\tv_mul_f64 %[p1], %[d], %[pk]\n\tv_mul_f64 %[p2], %[d], %[pk]\n\t"
                "v_mul_f64 %[s0], %[p0], %[s0]\n\tv_mul_f64 %[s1], %[p1], %[s1]\n\tv_mul_f64 %[s2], %[p2], %[s2]\n\t"
                "v_fmac_f64_dpp %[q0], %[c], %[s0]" DPPC "\n\tv_fmac_f64_dpp %[q1], %[c], %[s1]" DPPC "\n\tv_fmac_f64_dpp %[q2], %[c], %[s2]" DPPC "\n\t"
                "s_nop 1\n\t"
                "v_fmac_f64_dpp %[s3], %[c], %[b]" DPPC "\n\tv_fmac_f64_dpp %[s4], %[c], %[b]" DPPC "\n\tv_fmac_f64_dpp %[s5], %[c], %[b]" DPPC "\n\t"
                "v_mul_f64 %[s3], %[p0], %[s3]\n\tv_mul_f64 %[s4], %[p1], %[s4]\n\tv_mul_f64 %[s5], %[p2], %[s5]\n\t"
                "v_fmac_f64_dpp %[q0], %[c], %[s3]" DPPC "\n\tv_fmac_f64_dpp %[q1], %[c], %[s4]" DPPC "\n\tv_fmac_f64_dpp %[q2], %[c], %[s5]" DPPC "\n\t"
                : [s0] "+v"(a0), [s1] "+v"(a1), [s2] "+v"(a2), [s3] "+v"(a3), [s4] "+v"(a4), [s5] "+v"(a5), [q0] "+v"(q0), [q1] "+v"(q1),
                  [q2] "+v"(q2), [pk] "=&v"(pk), [p0] "=&v"(p0), [p1] "=&v"(p1), [p2] "=&v"(p2)
                : [b] "v"(b), [c] "v"(c), [d] "v"(d));
            a6 += q0 + q1 + q2;
        } else if constexpr (KIND == 5) {   // same arithmetic with plain (non-DPP) operands: 22 VALU
            double p0, p1, p2, q0 = 0, q1 = 0, q2 = 0;
            asm volatile(
                "v_fma_f64 %[s0], %[c], %[b], %[s0]\n\tv_fma_f64 %[s1], %[c], %[b], %[s1]\n\tv_fma_f64 %[s2], %[c], %[b], %[s2]\n\t"
                "v_mul_f64 %[p0], %[d], %[d]\n\tv_mul_f64 %[p1], %[d], %[d]\n\tv_mul_f64 %[p2], %[d], %[d]\n\t"
                "v_mul_f64 %[s0], %[p0], %[s0]\n\tv_mul_f64 %[s1], %[p1], %[s1]\n\tv_mul_f64 %[s2], %[p2], %[s2]\n\t"
                "v_fma_f64 %[q0], %[c], %[s0], %[q0]\n\tv_fma_f64 %[q1], %[c], %[s1], %[q1]\n\tv_fma_f64 %[q2], %[c], %[s2], %[q2]\n\t"
                "v_fma_f64 %[s3], %[c], %[b], %[s3]\n\tv_fma_f64 %[s4], %[c], %[b], %[s4]\n\tv_fma_f64 %[s5], %[c], %[b], %[s5]\n\t"
                "v_mul_f64 %[s3], %[p0], %[s3]\n\tv_mul_f64 %[s4], %[p1], %[s4]\n\tv_mul_f64 %[s5], %[p2], %[s5]\n\t"
                "v_fma_f64 %[q0], %[c], %[s3], %[q0]\n\tv_fma_f64 %[q1], %[c], %[s4], %[q1]\n\tv_fma_f64 %[q2], %[c], %[s5], %[q2]\n\t"
                : [s0] "+v"(a0), [s1] "+v"(a1), [s2] "+v"(a2), [s3] "+v"(a3), [s4] "+v"(a4), [s5] "+v"(a5), [q0] "+v"(q0), [q1] "+v"(q1),
                  [q2] "+v"(q2), [p0] "=&v"(p0), [p1] "=&v"(p1), [p2] "=&v"(p2)
                : [b] "v"(b), [c] "v"(c), [d] "v"(d));
            a6 += q0 + q1 + q2;
        } else if constexpr (KIND == 6) {   // v_add_f64
            asm volatile(R8("v_add_f64 %0, %0, %8\n\tv_add_f64 %1, %1, %8\n\tv_add_f64 %2, %2, %8\n\tv_add_f64 %3, %3, %8\n\t"
                            "v_add_f64 %4, %4, %8\n\tv_add_f64 %5, %5, %8\n\tv_add_f64 %6, %6, %8\n\tv_add_f64 %7, %7, %8\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if constexpr (KIND == 7) {   // 32-bit DPP mov (quad_perm)
            int x0 = __double2loint(a0), x1 = __double2loint(a1), x2 = __double2loint(a2), x3 = __double2loint(a3);
            int x4 = __double2loint(a4), x5 = __double2loint(a5), x6 = __double2loint(a6), x7 = __double2loint(a7), xb = __double2loint(b);
#define QP " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            asm volatile(R8("v_mov_b32_dpp %0, %8" QP "v_mov_b32_dpp %1, %8" QP "v_mov_b32_dpp %2, %8" QP "v_mov_b32_dpp %3, %8" QP
                            "v_mov_b32_dpp %4, %8" QP "v_mov_b32_dpp %5, %8" QP "v_mov_b32_dpp %6, %8" QP "v_mov_b32_dpp %7, %8" QP)
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(xb));
#undef QP
            a0 += x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
        } else if constexpr (KIND == 8) {   // v_pk_fma_f32 (two fp32 FMAs per lane)
            asm volatile(R8("v_pk_fma_f32 %0, %0, %8, %9\n\tv_pk_fma_f32 %1, %1, %8, %9\n\tv_pk_fma_f32 %2, %2, %8, %9\n\tv_pk_fma_f32 %3, %3, %8, %9\n\t"
                            "v_pk_fma_f32 %4, %4, %8, %9\n\tv_pk_fma_f32 %5, %5, %8, %9\n\tv_pk_fma_f32 %6, %6, %8, %9\n\tv_pk_fma_f32 %7, %7, %8, %9\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if constexpr (KIND == 9) {   // v_fma_f64 with an SGPR-free constant-less three-register form, distinct dst: d_i = a_i * b + a_{i+1}
            asm volatile(R8("v_fma_f64 %0, %1, %8, %2\n\tv_fma_f64 %1, %2, %8, %3\n\tv_fma_f64 %2, %3, %8, %4\n\tv_fma_f64 %3, %4, %8, %5\n\t"
                            "v_fma_f64 %4, %5, %8, %6\n\tv_fma_f64 %5, %6, %8, %7\n\tv_fma_f64 %6, %7, %8, %0\n\tv_fma_f64 %7, %0, %8, %1\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if constexpr (KIND == 10) {   // ds_bpermute_b32 (LDS crossbar, no VALU issue)
            int idx = (threadIdx.x * 4 + 64) & 255;
            int x0 = __double2loint(a0), x1 = __double2loint(a1), x2 = __double2loint(a2), x3 = __double2loint(a3);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                x0 = __builtin_amdgcn_ds_bpermute(idx, x0); x1 = __builtin_amdgcn_ds_bpermute(idx, x1);
                x2 = __builtin_amdgcn_ds_bpermute(idx, x2); x3 = __builtin_amdgcn_ds_bpermute(idx, x3);
            }
            a0 += x0 + x1 + x2 + x3;
        } else if constexpr (KIND == 11) {   // v_rcp_f64
            asm volatile(R8("v_rcp_f64 %0, %0\n\tv_rcp_f64 %1, %1\n\tv_rcp_f64 %2, %2\n\tv_rcp_f64 %3, %3\n\t"
                            "v_rcp_f64 %4, %4\n\tv_rcp_f64 %5, %5\n\tv_rcp_f64 %6, %6\n\tv_rcp_f64 %7, %7\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if constexpr (KIND == 12) {   // v_fmac_f64_dpp alternating with plain v_mul_f64 (does the DPP cost hide behind a plain op?)
            asm volatile(R8("v_fmac_f64_dpp %0, %9, %8" DPPC "\n\tv_mul_f64 %1, %1, %8\n\tv_fmac_f64_dpp %2, %9, %8" DPPC "\n\tv_mul_f64 %3, %3, %8\n\t"
                            "v_fmac_f64_dpp %4, %9, %8" DPPC "\n\tv_mul_f64 %5, %5, %8\n\tv_fmac_f64_dpp %6, %9, %8" DPPC "\n\tv_mul_f64 %7, %7, %8\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if constexpr (KIND == 13) {   // v_fma_f64 with an SGPR multiplier (scalar broadcast operand)
            double sb = __builtin_bit_cast(double, __builtin_amdgcn_readfirstlane((int)(unsigned long long)__builtin_bit_cast(unsigned long long, b)) | 0x3ff0000000000000ull);
            asm volatile(R8("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                            "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sb), "v"(c));
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if ((threadIdx.x & 63) == 0) {
        cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
        real[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = r1 - r0;
    }
}

template <int KIND>
void run(const char* name, int per_iter)
{
    const int iters = 20000;
    double* out; unsigned long long *cyc, *real;
    hipMalloc(&out, 8 * 1024 * 1024); hipMalloc(&cyc, 8 * 8192); hipMalloc(&real, 8 * 8192);
    std::vector<unsigned long long> hc(8192), hr(8192);
    printf("%-44s", name);
    // (threads per block, blocks): 1 wave; 4 waves = 1 per SIMD; 8 = 2 per SIMD; 16 = 4 per SIMD (one CU); whole chip at 2 and 4 per SIMD
    const int cfgs[6][2] = {{64, 1}, {256, 1}, {512, 1}, {1024, 1}, {512, 256}, {1024, 256}};
    for (auto& c : cfgs) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(probe<KIND>, dim3(c[1]), dim3(c[0]), 0, 0, out, 200, cyc, real);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<KIND>, dim3(c[1]), dim3(c[0]), 0, 0, out, iters, cyc, real);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const int nw = c[0] / 64 * c[1];
        hipMemcpy(hc.data(), cyc, 8 * nw, hipMemcpyDeviceToHost); hipMemcpy(hr.data(), real, 8 * nw, hipMemcpyDeviceToHost);
        double sc = 0, sr = 0; for (int i = 0; i < nw; ++i) { sc += hc[i]; sr += hr[i]; }
        sc /= nw; sr /= nw;
        const double ninst = (double)iters * per_iter;
        // s_memrealtime ticks at 100 MHz
        // cycles per instruction per SIMD from the WALL time of the launch at the measured clock (waves of a block may be
        // spread unevenly over the SIMDs, so the per-wave s_memtime average is not the SIMD's issue interval)
        const double ghz = sc / sr * 0.1;
        const double per_simd = ms * 1e-3 * ghz * 1e9 / (ninst * (c[0] / 256.0));
        printf(" | %5.2f cyc/inst/SIMD %4.2f GHz %6.2f ms", c[0] >= 256 ? per_simd : sc / ninst, ghz, ms);
    }
    printf("\n");
    hipFree(out); hipFree(cyc); hipFree(real);
}

int main()
{
    printf("%-44s | 1 wave | 4 waves (1/SIMD) | 8 waves (2/SIMD) | 16 waves (4/SIMD), one CU | whole chip 2/SIMD | whole chip 4/SIMD\n", "kind");
    run<0>("v_fma_f64", 64);
    run<9>("v_fma_f64 (three distinct registers)", 64);
    run<13>("v_fma_f64 (SGPR multiplier)", 64);
    run<1>("v_mul_f64", 64);
    run<6>("v_add_f64", 64);
    run<2>("v_fmac_f64_dpp row_newbcast", 64);
    run<3>("v_mov_b64_dpp row_newbcast", 64);
    run<12>("v_fmac_f64_dpp / v_mul_f64 alternating", 64);
    run<7>("v_mov_b32_dpp quad_perm", 64);
    run<8>("v_pk_fma_f32", 64);
    run<11>("v_rcp_f64", 64);
    run<10>("ds_bpermute_b32", 64);
    run<4>("scan column pair, DPP-folded (22 VALU)", 22);
    run<5>("same arithmetic, plain operands (21 VALU)", 21);
    return 0;
}
