// Pieces shared by the windowed kernels (celerite_block.hip: one draw per workgroup, latency; celerite_tile.hip: one draw per
// wavefront, throughput): fragment-order table record, the 16 x 16 LDL' of a window on DPP broadcasts, LDS DMA helpers.
// (moved out of celerite_block.hip unchanged in round 5; every item keeps internal linkage)
#pragma once
#include "common.h"

#include <cmath>
#include <type_traits>

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

template <int I>
using ic = std::integral_constant<int, I>;
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (B < E) {
        f(ic<B>{});
        static_for<B + 1, E>(f);
    }
}

__device__ __forceinline__ double recip_f64(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

constexpr int KW = 16;   // time steps per window

// ---- table -----------------------------------------------------------------------------------------------------------
// One record per window k (steps 16 k + s), in doubles:
//   CVf [NB][4][64]   C_n o v_n,  A-operand fragment order: element (row 16 I + 4 ks + (lane >> 4), step lane & 15)
//   CXf [NB][4][64]   C_n o x_n,  same order                       (u = al v + be x, cos row (v, x) = (cos, sin), sin row swapped)
//   VHf [NB][4][64]   (C_K / C_n) o v_n, C/D fragment order: element (step 4 g + (lane >> 4), row 16 J + (lane & 15));
//                     the y row (row R) holds y_n
//   CK  [16 NB]       C_K (decay over the whole window; 1 for the y row, 0 for the padding rows)
//   S2w [16]          sigma2_n (1 for the padded steps of the last window)
//   E   [J][128][2]   e^{-c_t tau} (cos, sin)(d_t tau) for the pair p = n (n - 1) / 2 + j of steps j < n of the window, tau = t_n - t_j
//   (the first five padded together to a multiple of 128 doubles: the record is copied to LDS in 1 KB pieces)
__host__ __device__ inline int64_t block_rec_doubles(int NB, int J) { return ((3 * (int64_t)NB * 256 + 16 * NB + 16 + 127) & ~(int64_t)127) + (int64_t)J * 256; }

// ---- the 16 x 16 LDL' of the chain wavefront -------------------------------------------------------------------------
// In-place Gauss-Jordan form.  Lane (q, n) holds column n in m[0..15]; the four DPP rows q hold copies.  Before step P, lanes
// n >= P hold the reduced Sigma (column n, rows >= P matter), lanes n < P already hold column n of L^-1 (rows > n).  Step P:
//   d_P = m[P] of lane P ;  mult_n = -m[P]_n / d_P  (lane P: -2) ;  m[j]_n += bcast_P(m[j]) * mult_n  for j > P
// which is the rank-1 update of Sigma for n > P, the row operation on the identity for n < P, and turns lane P's own
// Sigma_jP = L_jP d_P into -L_jP d_P: column P of L^-1 SCALED BY d_P (exactly: the factor -2 is exact, and every later row
// operation is linear in the column; -1/d_P - 1 instead would leave the unscaled column but rounds 1/d_P to the grid of 1).
// One v_fmac_f64_dpp per (P, j): 120 in all.  DPP hazard (two wait states between a VALU write of a register and a DPP read of it; not
// interlocked): the DPP source of an update is the row's own register m[j], last written by the PREVIOUS step's update of that row — at least the
// seven instructions of the multiplier chain earlier — so the updates carry no s_nop (up to round 5 every group of four had one: 168 s_nop per
// window in the ISA); only the broadcast of the next pivot, which reads the register the instruction before it wrote, keeps its s_nop 1.  At the end lane n holds d_n in m[n] and d_n (L^-1)_jn in m[j], j > n; m[j]_n
// with j < n is left-over Sigma (the readers mask it).
#define PIORAN_BLK_DPP " row_newbcast:%c[p] row_mask:0xf bank_mask:0xf"
template <int P, int J0, int CNT>
__device__ __forceinline__ void ldl_rows(double (&m)[16], double mult)
{
    if constexpr (CNT >= 4) {
        asm volatile("v_fmac_f64_dpp %[c0], %[c0], %[t]" PIORAN_BLK_DPP "\n\tv_fmac_f64_dpp %[c1], %[c1], %[t]" PIORAN_BLK_DPP "\n\t"
                     "v_fmac_f64_dpp %[c2], %[c2], %[t]" PIORAN_BLK_DPP "\n\tv_fmac_f64_dpp %[c3], %[c3], %[t]" PIORAN_BLK_DPP
                     : [c0] "+v"(m[J0]), [c1] "+v"(m[J0 + 1]), [c2] "+v"(m[J0 + 2]), [c3] "+v"(m[J0 + 3])
                     : [t] "v"(mult), [p] "i"(P));
        ldl_rows<P, J0 + 4, CNT - 4>(m, mult);
    } else if constexpr (CNT >= 1) {
        asm volatile("v_fmac_f64_dpp %[c0], %[c0], %[t]" PIORAN_BLK_DPP : [c0] "+v"(m[J0]) : [t] "v"(mult), [p] "i"(P));
        ldl_rows<P, J0 + 1, CNT - 1>(m, mult);
    }
}

// One elimination step with the NEXT pivot's multiplier chain spread between the rank-1 updates of this step, which do not depend
// on it: the wavefront issues in order, so the order below is the schedule (every piece is its own asm volatile statement).
// The multiplier -m / d is formed without a finished reciprocal: r0 = v_rcp_f64(d) (24 bits), e = 1 - d r0, t0 = -m r0,
// mult = t0 (1 + e + e^2) — third order, relative error e^3 < 1e-22 before rounding — four dependent DP operations after the
// broadcast instead of the seven of "two Newton steps, then multiply".  `mult` is this step's multiplier, on return the next
// step's; D_P stays in m[P] of lane P.
__device__ __forceinline__ void blk_rcp(double& r, double d) { asm volatile("v_rcp_f64 %0, %1" : "=v"(r) : "v"(d)); }
__device__ __forceinline__ void blk_e_t0(double& e, double& t0, double r0, double d, double mrow)
{
    asm volatile("s_nop 0\n\tv_fma_f64 %0, -%2, %3, 1.0\n\tv_mul_f64 %1, -%4, %2" : "=&v"(e), "=&v"(t0) : "v"(r0), "v"(d), "v"(mrow));
}
__device__ __forceinline__ void blk_poly(double& pq, double e) { asm volatile("v_fma_f64 %0, %1, %1, %1" : "=v"(pq) : "v"(e)); }
__device__ __forceinline__ void blk_mult(double& mu_, double t0, double pq) { asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(mu_) : "v"(t0), "v"(pq)); }
template <int P>
__device__ __forceinline__ void ldl_step(double (&m)[16], double& mult, int c16)
{
    constexpr int NR = 15 - P;                       // rows below the pivot
    constexpr int NA = NR >= 1 ? 1 : 0;              // the next pivot's row first
    constexpr int NBk = NR - NA >= 4 ? 4 : NR - NA;
    constexpr int NCk = NR - NA - NBk >= 4 ? 4 : NR - NA - NBk;
    constexpr int NDk = NR - NA - NBk - NCk;
    ldl_rows<P, P + 1, NA>(m, mult);
    if constexpr (P < 15) {
        double dn, r0, e, t0, pq, mn;
        asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%c2 row_mask:0xf bank_mask:0xf" : "=v"(dn) : "v"(m[P + 1]), "i"(P + 1));
        if constexpr (NBk == 0) asm volatile("s_nop 0");
        blk_rcp(r0, dn);
        ldl_rows<P, P + 1 + NA, NBk>(m, mult);
        blk_e_t0(e, t0, r0, dn, m[P + 1]);
        ldl_rows<P, P + 1 + NA + NBk, NCk>(m, mult);
        blk_poly(pq, e);
        ldl_rows<P, P + 1 + NA + NBk + NCk, NDk>(m, mult);
        blk_mult(mn, t0, pq);
        mult = c16 == P + 1 ? -2.0 : mn;
    }
}
__device__ __forceinline__ double ldl_first_mult(double (&m)[16], int c16)
{
    double d0, r0, e, t0, pq, mn;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "=v"(d0) : "v"(m[0]));
    asm volatile("s_nop 0");
    blk_rcp(r0, d0);
    blk_e_t0(e, t0, r0, d0, m[0]);
    blk_poly(pq, e);
    blk_mult(mn, t0, pq);
    return c16 == 0 ? -2.0 : mn;
}

// doubles of the (CV, CX, VH, CK, S2w) part of a record, rounded up to whole 1 KB pieces of the LDS DMA
__host__ __device__ inline int block_tile_doubles(int NB) { return (3 * NB * 256 + 16 * NB + 16 + 127) & ~127; }

// global -> LDS without registers: 1 KB per wavefront instruction (global_load_lds_dwordx4: LDS address = uniform base + 16 lane)
__device__ __forceinline__ void dma_pieces(const double* gsrc, double* ldst, int npieces, int w, int nwaves, int lane)
{
    for (int c = w; c < npieces; c += nwaves)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + c * 128 + lane * 2),
                                         (__attribute__((address_space(3))) void*)(ldst + c * 128), 16, 0, 0);
}
// The same as inline assembly: the compiler's wait-count pass does not see an LDS write here.  With the builtin it puts s_waitcnt vmcnt(0) in
// front of LDS reads it cannot tell apart from the DMA's target (celerite_block_adjoint_kernel: in front of nearly every LDS read of one of the
// two copies of its window body — the DMA of the NEXT window's block was waited for at once).  The caller owns the ordering: every read of the
// target comes after an explicit s_waitcnt vmcnt(0) + barrier (PIORAN_BLK_BARRIER_DMA).  vmcnt stays conservative for the compiler's own
// loads: the counter retires in order, an unaccounted operation in flight can only make a wait longer.
__device__ __forceinline__ void dma_pieces_asm(const double* gsrc, double* ldst, int npieces, int w, int nwaves, int lane)
{
    const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)ldst;
    for (int c = w; c < npieces; c += nwaves) {
        const double* g = gsrc + c * 128 + lane * 2;
        const unsigned l = __builtin_amdgcn_readfirstlane(lbase + (unsigned)c * 1024u);
        // (m0 is a reserved register: saved and restored around the instruction, which reads it at issue)
        unsigned m0_save;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(m0_save)
                     : "v"(g), "s"(l)
                     : "memory");
    }
}
// workgroup barrier that publishes LDS writes but leaves LDS DMAs / global loads in flight (__syncthreads() would drain them)
#define PIORAN_BLK_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define PIORAN_BLK_BARRIER_DMA() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")

}  // namespace
