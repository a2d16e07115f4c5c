// Batched celerite log-likelihood as a register-resident scan over time steps (gfx950 / CDNA4).
//
// Replaces, for B independent parameter draws at once, the reference's
//   logl -> init_semi_separable! + solve_prec!      (src/celerite_solver.jl:12-100,115-158,312-334)
// The recurrence is the one restated in SURVEY.md appendix A; per step and per draw
//   S   <- (phi phi') o (S + D_{n-1} w w')           (:78-79,85)
//   q    = S u ;  D_n = sum(a) + sigma2_n - u'q ;  w <- (v - q) / D_n      (:80-97)
//   f   <- phi o (f + w_prev z_{n-1}) ;  z_n = y_n - u'f                    (:136-141)
//   logdet += log|D_n| ;  quad += z_n^2 / D_n        (forward-only form of :145-155,:333)
// The forward substitution is not a separate recurrence here: y is carried as ONE EXTRA ROW of S
// (row index R, with u = 0, v = y_n - mu, phi = 1).  By symmetry that row of S is f', so its
// (S u) entry is u'f, its "w" is z_n / D_n, and no cross-lane reduction is needed for z_n.
// Nothing per-step is written to HBM: U, V(W), phi, D, z of the reference are never materialised.
//
// Mapping to the hardware:
//   * one draw occupies G = 16*CBR lanes = CBR DPP rows of a 64-wide wavefront;
//   * inside a DPP row, logical lane `lam` owns the RPL rows  lam*RPL .. lam*RPL+RPL-1  of the FULL
//     (not triangular) R x R state S, for the column block of its DPP row: S lives in VGPRs;
//   * the column loop takes w_k, u_k of the row-owning lane through the DPP operand of the FMAs themselves
//     (v_fmac_f64_dpp row_newbcast; phi_k through one v_mov_b64_dpp): no LDS, no readlane, and q = S u needs no
//     cross-lane reduction inside a DPP row; u'q is summed by broadcast-accumulating the contributing lanes;
//   * with CBR > 1 every DPP row r holds the rows rotated by NSRC*r lanes, so the SAME instruction
//     stream (broadcast source lane N, register slot m) walks a DIFFERENT column block in each DPP
//     row; the partial q of the CBR column blocks are summed with ds_bpermute.
// FP64 VALU bound (no MFMA: the update is rank-1 per draw, nothing is shared across draws).
#include "common.h"
#include <vector>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

// Diagnostic hooks: compiled out in the product; tools/scan_probe.hip defines them to s_memtime accumulators.
#ifndef PIORAN_SSTAMP
#define PIORAN_SSTAMP(i)
#define PIORAN_SSTAMP_DECL
#define PIORAN_SSTAMP_FLUSH
#endif

namespace {

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

template <int CTRL>
__device__ __forceinline__ double dpp_perm(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}


// One column of the S update for RPL rows with the broadcasts FOLDED into the FMAs:
//   pk = bcast(phi_k);  S_i = (phi_i pk) * (S_i + g_i bcast(w_k));  q_i += S_i bcast(u_k)
// v_fmac_f64_dpp ... row_newbcast:N takes its src0 from lane N of the DPP row (DP-ALU DPP supports exactly this
// control on gfx950), so w_k and u_k never occupy a register or an instruction of their own.
// Hazard (VALU write of a VGPR -> DPP read of it needs 2 wait states; nothing is padded inside asm): the DPP sources of
// a pass (w, u, phi of the row-owning lane) are written BEFORE the pass and never inside it; the caller pins them in
// their registers and issues ONE s_nop 1 in front of the pass (a leading s_nop per block, as in round 1, cost 8 % of the
// instruction slots); tools/check_dpp_hazards.py verifies the listing.
#define PIORAN_DPP_STR2(N) #N
#define PIORAN_DPP_CTRL(N) " row_newbcast:" PIORAN_DPP_STR2(N) " row_mask:0xf bank_mask:0xf"
#define PD_FMAC(d, s0, s1) "v_fmac_f64_dpp %[" #d "], %[" #s0 "], %[" #s1 "]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
#define PD_MUL(d, s0, s1) "v_mul_f64 %[" #d "], %[" #s0 "], %[" #s1 "]\n\t"

template <int RPL, int N>
struct ColBlock;

template <int N>
struct ColBlock<1, N> {
    static __device__ __forceinline__ void run(double (&S)[1], double (&q)[1], const double (&g)[1],
                                               const double (&ph)[1], double ws, double us, double phs)
    {
        double pk, p0;
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
                     PD_FMAC(s0, ws, g0) PD_MUL(p0, h0, pk) PD_MUL(s0, p0, s0) PD_FMAC(q0, us, s0)
                     : [s0] "+v"(S[0]), [q0] "+v"(q[0]), [pk] "=&v"(pk), [p0] "=&v"(p0)
                     : [g0] "v"(g[0]), [h0] "v"(ph[0]), [ws] "v"(ws), [us] "v"(us), [phs] "v"(phs), [n] "i"(N));
    }
};

template <int N>
struct ColBlock<2, N> {
    static __device__ __forceinline__ void run(double (&S)[2], double (&q)[2], const double (&g)[2],
                                               const double (&ph)[2], double ws, double us, double phs)
    {
        double pk, p0, p1;
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
                     PD_FMAC(s0, ws, g0) PD_FMAC(s1, ws, g1) PD_MUL(p0, h0, pk) PD_MUL(p1, h1, pk)
                     PD_MUL(s0, p0, s0) PD_MUL(s1, p1, s1) PD_FMAC(q0, us, s0) PD_FMAC(q1, us, s1)
                     : [s0] "+v"(S[0]), [s1] "+v"(S[1]), [q0] "+v"(q[0]), [q1] "+v"(q[1]), [pk] "=&v"(pk),
                       [p0] "=&v"(p0), [p1] "=&v"(p1)
                     : [g0] "v"(g[0]), [g1] "v"(g[1]), [h0] "v"(ph[0]), [h1] "v"(ph[1]), [ws] "v"(ws), [us] "v"(us),
                       [phs] "v"(phs), [n] "i"(N));
    }
};

template <int N>
struct ColBlock<3, N> {
    static __device__ __forceinline__ void run(double (&S)[3], double (&q)[3], const double (&g)[3],
                                               const double (&ph)[3], double ws, double us, double phs)
    {
        double pk, p0, p1, p2;
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
                     PD_FMAC(s0, ws, g0) PD_FMAC(s1, ws, g1) PD_FMAC(s2, ws, g2)
                     PD_MUL(p0, h0, pk) PD_MUL(p1, h1, pk) PD_MUL(p2, h2, pk)
                     PD_MUL(s0, p0, s0) PD_MUL(s1, p1, s1) PD_MUL(s2, p2, s2)
                     PD_FMAC(q0, us, s0) PD_FMAC(q1, us, s1) PD_FMAC(q2, us, s2)
                     : [s0] "+v"(S[0]), [s1] "+v"(S[1]), [s2] "+v"(S[2]), [q0] "+v"(q[0]), [q1] "+v"(q[1]),
                       [q2] "+v"(q[2]), [pk] "=&v"(pk), [p0] "=&v"(p0), [p1] "=&v"(p1), [p2] "=&v"(p2)
                     : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [h0] "v"(ph[0]), [h1] "v"(ph[1]),
                       [h2] "v"(ph[2]), [ws] "v"(ws), [us] "v"(us), [phs] "v"(phs), [n] "i"(N));
    }
};

template <int N>
struct ColBlock<4, N> {
    static __device__ __forceinline__ void run(double (&S)[4], double (&q)[4], const double (&g)[4],
                                               const double (&ph)[4], double ws, double us, double phs)
    {
        double pk, p0, p1, p2, p3;
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
                     PD_FMAC(s0, ws, g0) PD_FMAC(s1, ws, g1) PD_FMAC(s2, ws, g2) PD_FMAC(s3, ws, g3)
                     PD_MUL(p0, h0, pk) PD_MUL(p1, h1, pk) PD_MUL(p2, h2, pk) PD_MUL(p3, h3, pk)
                     PD_MUL(s0, p0, s0) PD_MUL(s1, p1, s1) PD_MUL(s2, p2, s2) PD_MUL(s3, p3, s3)
                     PD_FMAC(q0, us, s0) PD_FMAC(q1, us, s1) PD_FMAC(q2, us, s2) PD_FMAC(q3, us, s3)
                     : [s0] "+v"(S[0]), [s1] "+v"(S[1]), [s2] "+v"(S[2]), [s3] "+v"(S[3]), [q0] "+v"(q[0]),
                       [q1] "+v"(q[1]), [q2] "+v"(q[2]), [q3] "+v"(q[3]), [pk] "=&v"(pk), [p0] "=&v"(p0),
                       [p1] "=&v"(p1), [p2] "=&v"(p2), [p3] "=&v"(p3)
                     : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [h0] "v"(ph[0]),
                       [h1] "v"(ph[1]), [h2] "v"(ph[2]), [h3] "v"(ph[3]), [ws] "v"(ws), [us] "v"(us), [phs] "v"(phs),
                       [n] "i"(N));
    }
};

template <int N>
struct ColBlock<5, N> {
    static __device__ __forceinline__ void run(double (&S)[5], double (&q)[5], const double (&g)[5],
                                               const double (&ph)[5], double ws, double us, double phs)
    {
        double pk, p0, p1, p2, p3, p4;
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
                     PD_FMAC(s0, ws, g0) PD_FMAC(s1, ws, g1) PD_FMAC(s2, ws, g2) PD_FMAC(s3, ws, g3) PD_FMAC(s4, ws, g4)
                     PD_MUL(p0, h0, pk) PD_MUL(p1, h1, pk) PD_MUL(p2, h2, pk) PD_MUL(p3, h3, pk) PD_MUL(p4, h4, pk)
                     PD_MUL(s0, p0, s0) PD_MUL(s1, p1, s1) PD_MUL(s2, p2, s2) PD_MUL(s3, p3, s3) PD_MUL(s4, p4, s4)
                     PD_FMAC(q0, us, s0) PD_FMAC(q1, us, s1) PD_FMAC(q2, us, s2) PD_FMAC(q3, us, s3) PD_FMAC(q4, us, s4)
                     : [s0] "+v"(S[0]), [s1] "+v"(S[1]), [s2] "+v"(S[2]), [s3] "+v"(S[3]), [s4] "+v"(S[4]),
                       [q0] "+v"(q[0]), [q1] "+v"(q[1]), [q2] "+v"(q[2]), [q3] "+v"(q[3]), [q4] "+v"(q[4]),
                       [pk] "=&v"(pk), [p0] "=&v"(p0), [p1] "=&v"(p1), [p2] "=&v"(p2), [p3] "=&v"(p3), [p4] "=&v"(p4)
                     : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [g4] "v"(g[4]), [h0] "v"(ph[0]),
                       [h1] "v"(ph[1]), [h2] "v"(ph[2]), [h3] "v"(ph[3]), [h4] "v"(ph[4]), [ws] "v"(ws), [us] "v"(us),
                       [phs] "v"(phs), [n] "i"(N));
    }
};
// Column PAIRS: the cos and the sin row of one celerite term have the same phi, so two consecutive columns share
// phi_k and the products phi_i * phi_k.  PairFirst = ColBlock that also returns pp[]; PairSecond reuses them
// (no v_mov_b64_dpp, no pp multiplies).
template <int RPL, int N>
struct PairFirst;
template <int RPL, int N>
struct PairSecond;

template <int N>
struct PairFirst<3, N> {
    static __device__ __forceinline__ void run(double (&S)[3], double (&q)[3], const double (&g)[3],
                                               const double (&ph)[3], double ws, double us, double phs, double (&pp)[3])
    {
        double pk;
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
                     PD_FMAC(s0, ws, g0) PD_FMAC(s1, ws, g1) PD_FMAC(s2, ws, g2)
                     PD_MUL(p0, h0, pk) PD_MUL(p1, h1, pk) PD_MUL(p2, h2, pk)
                     PD_MUL(s0, p0, s0) PD_MUL(s1, p1, s1) PD_MUL(s2, p2, s2)
                     PD_FMAC(q0, us, s0) PD_FMAC(q1, us, s1) PD_FMAC(q2, us, s2)
                     : [s0] "+v"(S[0]), [s1] "+v"(S[1]), [s2] "+v"(S[2]), [q0] "+v"(q[0]), [q1] "+v"(q[1]),
                       [q2] "+v"(q[2]), [pk] "=&v"(pk), [p0] "=&v"(pp[0]), [p1] "=&v"(pp[1]), [p2] "=&v"(pp[2])
                     : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [h0] "v"(ph[0]), [h1] "v"(ph[1]),
                       [h2] "v"(ph[2]), [ws] "v"(ws), [us] "v"(us), [phs] "v"(phs), [n] "i"(N));
    }
};
template <int N>
struct PairSecond<3, N> {
    static __device__ __forceinline__ void run(double (&S)[3], double (&q)[3], const double (&g)[3], double ws, double us,
                                               const double (&pp)[3])
    {
        asm volatile(PD_FMAC(s0, ws, g0) PD_FMAC(s1, ws, g1) PD_FMAC(s2, ws, g2)
                     PD_MUL(s0, p0, s0) PD_MUL(s1, p1, s1) PD_MUL(s2, p2, s2)
                     PD_FMAC(q0, us, s0) PD_FMAC(q1, us, s1) PD_FMAC(q2, us, s2)
                     : [s0] "+v"(S[0]), [s1] "+v"(S[1]), [s2] "+v"(S[2]), [q0] "+v"(q[0]), [q1] "+v"(q[1]), [q2] "+v"(q[2])
                     : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [p0] "v"(pp[0]), [p1] "v"(pp[1]), [p2] "v"(pp[2]),
                       [ws] "v"(ws), [us] "v"(us), [n] "i"(N));
    }
};
template <int N>
struct PairFirst<1, N> {
    static __device__ __forceinline__ void run(double (&S)[1], double (&q)[1], const double (&g)[1],
                                               const double (&ph)[1], double ws, double us, double phs, double (&pp)[1])
    {
        double pk;
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
                     PD_FMAC(s0, ws, g0)
                     PD_MUL(p0, h0, pk)
                     PD_MUL(s0, p0, s0)
                     PD_FMAC(q0, us, s0)
                     : [s0] "+v"(S[0]), [q0] "+v"(q[0]), [pk] "=&v"(pk), [p0] "=&v"(pp[0])
                     : [g0] "v"(g[0]), [h0] "v"(ph[0]), [ws] "v"(ws), [us] "v"(us), [phs] "v"(phs), [n] "i"(N));
    }
};
template <int N>
struct PairSecond<1, N> {
    static __device__ __forceinline__ void run(double (&S)[1], double (&q)[1], const double (&g)[1], double ws, double us,
                                               const double (&pp)[1])
    {
        asm volatile(PD_FMAC(s0, ws, g0)
                     PD_MUL(s0, p0, s0)
                     PD_FMAC(q0, us, s0)
                     : [s0] "+v"(S[0]), [q0] "+v"(q[0])
                     : [g0] "v"(g[0]), [p0] "v"(pp[0]), [ws] "v"(ws), [us] "v"(us), [n] "i"(N));
    }
};
template <int N>
struct PairFirst<2, N> {
    static __device__ __forceinline__ void run(double (&S)[2], double (&q)[2], const double (&g)[2],
                                               const double (&ph)[2], double ws, double us, double phs, double (&pp)[2])
    {
        double pk;
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
                     PD_FMAC(s0, ws, g0) PD_FMAC(s1, ws, g1)
                     PD_MUL(p0, h0, pk) PD_MUL(p1, h1, pk)
                     PD_MUL(s0, p0, s0) PD_MUL(s1, p1, s1)
                     PD_FMAC(q0, us, s0) PD_FMAC(q1, us, s1)
                     : [s0] "+v"(S[0]), [s1] "+v"(S[1]), [q0] "+v"(q[0]), [q1] "+v"(q[1]), [pk] "=&v"(pk), [p0] "=&v"(pp[0]), [p1] "=&v"(pp[1])
                     : [g0] "v"(g[0]), [g1] "v"(g[1]), [h0] "v"(ph[0]), [h1] "v"(ph[1]), [ws] "v"(ws), [us] "v"(us), [phs] "v"(phs), [n] "i"(N));
    }
};
template <int N>
struct PairSecond<2, N> {
    static __device__ __forceinline__ void run(double (&S)[2], double (&q)[2], const double (&g)[2], double ws, double us,
                                               const double (&pp)[2])
    {
        asm volatile(PD_FMAC(s0, ws, g0) PD_FMAC(s1, ws, g1)
                     PD_MUL(s0, p0, s0) PD_MUL(s1, p1, s1)
                     PD_FMAC(q0, us, s0) PD_FMAC(q1, us, s1)
                     : [s0] "+v"(S[0]), [s1] "+v"(S[1]), [q0] "+v"(q[0]), [q1] "+v"(q[1])
                     : [g0] "v"(g[0]), [g1] "v"(g[1]), [p0] "v"(pp[0]), [p1] "v"(pp[1]), [ws] "v"(ws), [us] "v"(us), [n] "i"(N));
    }
};
template <int N>
struct PairFirst<4, N> {
    static __device__ __forceinline__ void run(double (&S)[4], double (&q)[4], const double (&g)[4],
                                               const double (&ph)[4], double ws, double us, double phs, double (&pp)[4])
    {
        double pk;
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
                     PD_FMAC(s0, ws, g0) PD_FMAC(s1, ws, g1) PD_FMAC(s2, ws, g2) PD_FMAC(s3, ws, g3)
                     PD_MUL(p0, h0, pk) PD_MUL(p1, h1, pk) PD_MUL(p2, h2, pk) PD_MUL(p3, h3, pk)
                     PD_MUL(s0, p0, s0) PD_MUL(s1, p1, s1) PD_MUL(s2, p2, s2) PD_MUL(s3, p3, s3)
                     PD_FMAC(q0, us, s0) PD_FMAC(q1, us, s1) PD_FMAC(q2, us, s2) PD_FMAC(q3, us, s3)
                     : [s0] "+v"(S[0]), [s1] "+v"(S[1]), [s2] "+v"(S[2]), [s3] "+v"(S[3]), [q0] "+v"(q[0]), [q1] "+v"(q[1]), [q2] "+v"(q[2]), [q3] "+v"(q[3]), [pk] "=&v"(pk), [p0] "=&v"(pp[0]), [p1] "=&v"(pp[1]), [p2] "=&v"(pp[2]), [p3] "=&v"(pp[3])
                     : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [h0] "v"(ph[0]), [h1] "v"(ph[1]), [h2] "v"(ph[2]), [h3] "v"(ph[3]), [ws] "v"(ws), [us] "v"(us), [phs] "v"(phs), [n] "i"(N));
    }
};
template <int N>
struct PairSecond<4, N> {
    static __device__ __forceinline__ void run(double (&S)[4], double (&q)[4], const double (&g)[4], double ws, double us,
                                               const double (&pp)[4])
    {
        asm volatile(PD_FMAC(s0, ws, g0) PD_FMAC(s1, ws, g1) PD_FMAC(s2, ws, g2) PD_FMAC(s3, ws, g3)
                     PD_MUL(s0, p0, s0) PD_MUL(s1, p1, s1) PD_MUL(s2, p2, s2) PD_MUL(s3, p3, s3)
                     PD_FMAC(q0, us, s0) PD_FMAC(q1, us, s1) PD_FMAC(q2, us, s2) PD_FMAC(q3, us, s3)
                     : [s0] "+v"(S[0]), [s1] "+v"(S[1]), [s2] "+v"(S[2]), [s3] "+v"(S[3]), [q0] "+v"(q[0]), [q1] "+v"(q[1]), [q2] "+v"(q[2]), [q3] "+v"(q[3])
                     : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [p0] "v"(pp[0]), [p1] "v"(pp[1]), [p2] "v"(pp[2]), [p3] "v"(pp[3]), [ws] "v"(ws), [us] "v"(us), [n] "i"(N));
    }
};
template <int N>
struct PairFirst<5, N> {
    static __device__ __forceinline__ void run(double (&S)[5], double (&q)[5], const double (&g)[5],
                                               const double (&ph)[5], double ws, double us, double phs, double (&pp)[5])
    {
        double pk;
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
                     PD_FMAC(s0, ws, g0) PD_FMAC(s1, ws, g1) PD_FMAC(s2, ws, g2) PD_FMAC(s3, ws, g3) PD_FMAC(s4, ws, g4)
                     PD_MUL(p0, h0, pk) PD_MUL(p1, h1, pk) PD_MUL(p2, h2, pk) PD_MUL(p3, h3, pk) PD_MUL(p4, h4, pk)
                     PD_MUL(s0, p0, s0) PD_MUL(s1, p1, s1) PD_MUL(s2, p2, s2) PD_MUL(s3, p3, s3) PD_MUL(s4, p4, s4)
                     PD_FMAC(q0, us, s0) PD_FMAC(q1, us, s1) PD_FMAC(q2, us, s2) PD_FMAC(q3, us, s3) PD_FMAC(q4, us, s4)
                     : [s0] "+v"(S[0]), [s1] "+v"(S[1]), [s2] "+v"(S[2]), [s3] "+v"(S[3]), [s4] "+v"(S[4]), [q0] "+v"(q[0]), [q1] "+v"(q[1]), [q2] "+v"(q[2]), [q3] "+v"(q[3]), [q4] "+v"(q[4]), [pk] "=&v"(pk), [p0] "=&v"(pp[0]), [p1] "=&v"(pp[1]), [p2] "=&v"(pp[2]), [p3] "=&v"(pp[3]), [p4] "=&v"(pp[4])
                     : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [g4] "v"(g[4]), [h0] "v"(ph[0]), [h1] "v"(ph[1]), [h2] "v"(ph[2]), [h3] "v"(ph[3]), [h4] "v"(ph[4]), [ws] "v"(ws), [us] "v"(us), [phs] "v"(phs), [n] "i"(N));
    }
};
template <int N>
struct PairSecond<5, N> {
    static __device__ __forceinline__ void run(double (&S)[5], double (&q)[5], const double (&g)[5], double ws, double us,
                                               const double (&pp)[5])
    {
        asm volatile(PD_FMAC(s0, ws, g0) PD_FMAC(s1, ws, g1) PD_FMAC(s2, ws, g2) PD_FMAC(s3, ws, g3) PD_FMAC(s4, ws, g4)
                     PD_MUL(s0, p0, s0) PD_MUL(s1, p1, s1) PD_MUL(s2, p2, s2) PD_MUL(s3, p3, s3) PD_MUL(s4, p4, s4)
                     PD_FMAC(q0, us, s0) PD_FMAC(q1, us, s1) PD_FMAC(q2, us, s2) PD_FMAC(q3, us, s3) PD_FMAC(q4, us, s4)
                     : [s0] "+v"(S[0]), [s1] "+v"(S[1]), [s2] "+v"(S[2]), [s3] "+v"(S[3]), [s4] "+v"(S[4]), [q0] "+v"(q[0]), [q1] "+v"(q[1]), [q2] "+v"(q[2]), [q3] "+v"(q[3]), [q4] "+v"(q[4])
                     : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [g4] "v"(g[4]), [p0] "v"(pp[0]), [p1] "v"(pp[1]), [p2] "v"(pp[2]), [p3] "v"(pp[3]), [p4] "v"(pp[4]), [ws] "v"(ws), [us] "v"(us), [n] "i"(N));
    }
};
// ---- GENERATED by tools/gen_scan_win2.py: column blocks of the two-step form (do not edit by hand) ----
template <int RPL, int N> struct MatVec2;
template <int RPL, int N> struct Update2;
template <int RPL, int N> struct Update2First;
template <int RPL, int N> struct Update2Second;
template <int N>
struct MatVec2<1, N> {   // rA += T bcast(u~A_k), rB += T bcast(u~B_k) for one column k of this lane's rows
    static __device__ __forceinline__ void run(const double (&T)[1], double (&rA)[1], double (&rB)[1], double ua, double ub)
    {
        asm volatile(PD_FMAC(a0, ua, t0) PD_FMAC(b0, ub, t0)
                     : [a0] "+v"(rA[0]), [b0] "+v"(rB[0])
                     : [t0] "v"(T[0]), [ua] "v"(ua), [ub] "v"(ub), [n] "i"(N));
    }
};
template <int N>
struct Update2<1, N> {   // T = (phAB_i bcast(phAB_k)) T + hA_i bcast(wA_k) + mB_i bcast(wB_k)
    static __device__ __forceinline__ void run(double (&T)[1], const double (&hA)[1], const double (&mB)[1], const double (&ph)[1],
                                               double wa, double wb, double phs)
    {
        double pk, pp[1];
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
                     PD_MUL(p0, h0, pk) PD_MUL(t0, p0, t0) PD_FMAC(t0, wa, g0) PD_FMAC(t0, wb, m0)
                     : [t0] "+v"(T[0]), [pk] "=&v"(pk), [p0] "=&v"(pp[0])
                     : [g0] "v"(hA[0]), [m0] "v"(mB[0]), [h0] "v"(ph[0]), [wa] "v"(wa), [wb] "v"(wb), [phs] "v"(phs), [n] "i"(N));
    }
};
template <int N>
struct Update2First<1, N> {   // first column of a (cos, sin) pair: also hands phAB_i phAB_k to the second
    static __device__ __forceinline__ void run(double (&T)[1], const double (&hA)[1], const double (&mB)[1], const double (&ph)[1],
                                               double wa, double wb, double phs, double (&pp)[1])
    {
        double pk;
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
                     PD_MUL(p0, h0, pk) PD_MUL(t0, p0, t0) PD_FMAC(t0, wa, g0) PD_FMAC(t0, wb, m0)
                     : [t0] "+v"(T[0]), [pk] "=&v"(pk), [p0] "=&v"(pp[0])
                     : [g0] "v"(hA[0]), [m0] "v"(mB[0]), [h0] "v"(ph[0]), [wa] "v"(wa), [wb] "v"(wb), [phs] "v"(phs), [n] "i"(N));
    }
};
template <int N>
struct Update2Second<1, N> {
    static __device__ __forceinline__ void run(double (&T)[1], const double (&hA)[1], const double (&mB)[1], double wa, double wb,
                                               const double (&pp)[1])
    {
        asm volatile(PD_MUL(t0, p0, t0) PD_FMAC(t0, wa, g0) PD_FMAC(t0, wb, m0)
                     : [t0] "+v"(T[0])
                     : [g0] "v"(hA[0]), [m0] "v"(mB[0]), [p0] "v"(pp[0]), [wa] "v"(wa), [wb] "v"(wb), [n] "i"(N));
    }
};
template <int N>
struct MatVec2<2, N> {   // rA += T bcast(u~A_k), rB += T bcast(u~B_k) for one column k of this lane's rows
    static __device__ __forceinline__ void run(const double (&T)[2], double (&rA)[2], double (&rB)[2], double ua, double ub)
    {
        asm volatile(PD_FMAC(a0, ua, t0) PD_FMAC(a1, ua, t1) PD_FMAC(b0, ub, t0) PD_FMAC(b1, ub, t1)
                     : [a0] "+v"(rA[0]), [a1] "+v"(rA[1]), [b0] "+v"(rB[0]), [b1] "+v"(rB[1])
                     : [t0] "v"(T[0]), [t1] "v"(T[1]), [ua] "v"(ua), [ub] "v"(ub), [n] "i"(N));
    }
};
template <int N>
struct Update2<2, N> {   // T = (phAB_i bcast(phAB_k)) T + hA_i bcast(wA_k) + mB_i bcast(wB_k)
    static __device__ __forceinline__ void run(double (&T)[2], const double (&hA)[2], const double (&mB)[2], const double (&ph)[2],
                                               double wa, double wb, double phs)
    {
        double pk, pp[2];
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
                     PD_MUL(p0, h0, pk) PD_MUL(p1, h1, pk) PD_MUL(t0, p0, t0) PD_MUL(t1, p1, t1) PD_FMAC(t0, wa, g0) PD_FMAC(t1, wa, g1) PD_FMAC(t0, wb, m0) PD_FMAC(t1, wb, m1)
                     : [t0] "+v"(T[0]), [t1] "+v"(T[1]), [pk] "=&v"(pk), [p0] "=&v"(pp[0]), [p1] "=&v"(pp[1])
                     : [g0] "v"(hA[0]), [g1] "v"(hA[1]), [m0] "v"(mB[0]), [m1] "v"(mB[1]), [h0] "v"(ph[0]), [h1] "v"(ph[1]), [wa] "v"(wa), [wb] "v"(wb), [phs] "v"(phs), [n] "i"(N));
    }
};
template <int N>
struct Update2First<2, N> {   // first column of a (cos, sin) pair: also hands phAB_i phAB_k to the second
    static __device__ __forceinline__ void run(double (&T)[2], const double (&hA)[2], const double (&mB)[2], const double (&ph)[2],
                                               double wa, double wb, double phs, double (&pp)[2])
    {
        double pk;
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
                     PD_MUL(p0, h0, pk) PD_MUL(p1, h1, pk) PD_MUL(t0, p0, t0) PD_MUL(t1, p1, t1) PD_FMAC(t0, wa, g0) PD_FMAC(t1, wa, g1) PD_FMAC(t0, wb, m0) PD_FMAC(t1, wb, m1)
                     : [t0] "+v"(T[0]), [t1] "+v"(T[1]), [pk] "=&v"(pk), [p0] "=&v"(pp[0]), [p1] "=&v"(pp[1])
                     : [g0] "v"(hA[0]), [g1] "v"(hA[1]), [m0] "v"(mB[0]), [m1] "v"(mB[1]), [h0] "v"(ph[0]), [h1] "v"(ph[1]), [wa] "v"(wa), [wb] "v"(wb), [phs] "v"(phs), [n] "i"(N));
    }
};
template <int N>
struct Update2Second<2, N> {
    static __device__ __forceinline__ void run(double (&T)[2], const double (&hA)[2], const double (&mB)[2], double wa, double wb,
                                               const double (&pp)[2])
    {
        asm volatile(PD_MUL(t0, p0, t0) PD_MUL(t1, p1, t1) PD_FMAC(t0, wa, g0) PD_FMAC(t1, wa, g1) PD_FMAC(t0, wb, m0) PD_FMAC(t1, wb, m1)
                     : [t0] "+v"(T[0]), [t1] "+v"(T[1])
                     : [g0] "v"(hA[0]), [g1] "v"(hA[1]), [m0] "v"(mB[0]), [m1] "v"(mB[1]), [p0] "v"(pp[0]), [p1] "v"(pp[1]), [wa] "v"(wa), [wb] "v"(wb), [n] "i"(N));
    }
};
template <int N>
struct MatVec2<3, N> {   // rA += T bcast(u~A_k), rB += T bcast(u~B_k) for one column k of this lane's rows
    static __device__ __forceinline__ void run(const double (&T)[3], double (&rA)[3], double (&rB)[3], double ua, double ub)
    {
        asm volatile(PD_FMAC(a0, ua, t0) PD_FMAC(a1, ua, t1) PD_FMAC(a2, ua, t2) PD_FMAC(b0, ub, t0) PD_FMAC(b1, ub, t1) PD_FMAC(b2, ub, t2)
                     : [a0] "+v"(rA[0]), [a1] "+v"(rA[1]), [a2] "+v"(rA[2]), [b0] "+v"(rB[0]), [b1] "+v"(rB[1]), [b2] "+v"(rB[2])
                     : [t0] "v"(T[0]), [t1] "v"(T[1]), [t2] "v"(T[2]), [ua] "v"(ua), [ub] "v"(ub), [n] "i"(N));
    }
};
template <int N>
struct Update2<3, N> {   // T = (phAB_i bcast(phAB_k)) T + hA_i bcast(wA_k) + mB_i bcast(wB_k)
    static __device__ __forceinline__ void run(double (&T)[3], const double (&hA)[3], const double (&mB)[3], const double (&ph)[3],
                                               double wa, double wb, double phs)
    {
        double pk, pp[3];
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
                     PD_MUL(p0, h0, pk) PD_MUL(p1, h1, pk) PD_MUL(p2, h2, pk) PD_MUL(t0, p0, t0) PD_MUL(t1, p1, t1) PD_MUL(t2, p2, t2) PD_FMAC(t0, wa, g0) PD_FMAC(t1, wa, g1) PD_FMAC(t2, wa, g2) PD_FMAC(t0, wb, m0) PD_FMAC(t1, wb, m1) PD_FMAC(t2, wb, m2)
                     : [t0] "+v"(T[0]), [t1] "+v"(T[1]), [t2] "+v"(T[2]), [pk] "=&v"(pk), [p0] "=&v"(pp[0]), [p1] "=&v"(pp[1]), [p2] "=&v"(pp[2])
                     : [g0] "v"(hA[0]), [g1] "v"(hA[1]), [g2] "v"(hA[2]), [m0] "v"(mB[0]), [m1] "v"(mB[1]), [m2] "v"(mB[2]), [h0] "v"(ph[0]), [h1] "v"(ph[1]), [h2] "v"(ph[2]), [wa] "v"(wa), [wb] "v"(wb), [phs] "v"(phs), [n] "i"(N));
    }
};
template <int N>
struct Update2First<3, N> {   // first column of a (cos, sin) pair: also hands phAB_i phAB_k to the second
    static __device__ __forceinline__ void run(double (&T)[3], const double (&hA)[3], const double (&mB)[3], const double (&ph)[3],
                                               double wa, double wb, double phs, double (&pp)[3])
    {
        double pk;
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
                     PD_MUL(p0, h0, pk) PD_MUL(p1, h1, pk) PD_MUL(p2, h2, pk) PD_MUL(t0, p0, t0) PD_MUL(t1, p1, t1) PD_MUL(t2, p2, t2) PD_FMAC(t0, wa, g0) PD_FMAC(t1, wa, g1) PD_FMAC(t2, wa, g2) PD_FMAC(t0, wb, m0) PD_FMAC(t1, wb, m1) PD_FMAC(t2, wb, m2)
                     : [t0] "+v"(T[0]), [t1] "+v"(T[1]), [t2] "+v"(T[2]), [pk] "=&v"(pk), [p0] "=&v"(pp[0]), [p1] "=&v"(pp[1]), [p2] "=&v"(pp[2])
                     : [g0] "v"(hA[0]), [g1] "v"(hA[1]), [g2] "v"(hA[2]), [m0] "v"(mB[0]), [m1] "v"(mB[1]), [m2] "v"(mB[2]), [h0] "v"(ph[0]), [h1] "v"(ph[1]), [h2] "v"(ph[2]), [wa] "v"(wa), [wb] "v"(wb), [phs] "v"(phs), [n] "i"(N));
    }
};
template <int N>
struct Update2Second<3, N> {
    static __device__ __forceinline__ void run(double (&T)[3], const double (&hA)[3], const double (&mB)[3], double wa, double wb,
                                               const double (&pp)[3])
    {
        asm volatile(PD_MUL(t0, p0, t0) PD_MUL(t1, p1, t1) PD_MUL(t2, p2, t2) PD_FMAC(t0, wa, g0) PD_FMAC(t1, wa, g1) PD_FMAC(t2, wa, g2) PD_FMAC(t0, wb, m0) PD_FMAC(t1, wb, m1) PD_FMAC(t2, wb, m2)
                     : [t0] "+v"(T[0]), [t1] "+v"(T[1]), [t2] "+v"(T[2])
                     : [g0] "v"(hA[0]), [g1] "v"(hA[1]), [g2] "v"(hA[2]), [m0] "v"(mB[0]), [m1] "v"(mB[1]), [m2] "v"(mB[2]), [p0] "v"(pp[0]), [p1] "v"(pp[1]), [p2] "v"(pp[2]), [wa] "v"(wa), [wb] "v"(wb), [n] "i"(N));
    }
};
template <int N>
struct MatVec2<4, N> {   // rA += T bcast(u~A_k), rB += T bcast(u~B_k) for one column k of this lane's rows
    static __device__ __forceinline__ void run(const double (&T)[4], double (&rA)[4], double (&rB)[4], double ua, double ub)
    {
        asm volatile(PD_FMAC(a0, ua, t0) PD_FMAC(a1, ua, t1) PD_FMAC(a2, ua, t2) PD_FMAC(a3, ua, t3) PD_FMAC(b0, ub, t0) PD_FMAC(b1, ub, t1) PD_FMAC(b2, ub, t2) PD_FMAC(b3, ub, t3)
                     : [a0] "+v"(rA[0]), [a1] "+v"(rA[1]), [a2] "+v"(rA[2]), [a3] "+v"(rA[3]), [b0] "+v"(rB[0]), [b1] "+v"(rB[1]), [b2] "+v"(rB[2]), [b3] "+v"(rB[3])
                     : [t0] "v"(T[0]), [t1] "v"(T[1]), [t2] "v"(T[2]), [t3] "v"(T[3]), [ua] "v"(ua), [ub] "v"(ub), [n] "i"(N));
    }
};
template <int N>
struct Update2<4, N> {   // T = (phAB_i bcast(phAB_k)) T + hA_i bcast(wA_k) + mB_i bcast(wB_k)
    static __device__ __forceinline__ void run(double (&T)[4], const double (&hA)[4], const double (&mB)[4], const double (&ph)[4],
                                               double wa, double wb, double phs)
    {
        double pk, pp[4];
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
                     PD_MUL(p0, h0, pk) PD_MUL(p1, h1, pk) PD_MUL(p2, h2, pk) PD_MUL(p3, h3, pk) PD_MUL(t0, p0, t0) PD_MUL(t1, p1, t1) PD_MUL(t2, p2, t2) PD_MUL(t3, p3, t3) PD_FMAC(t0, wa, g0) PD_FMAC(t1, wa, g1) PD_FMAC(t2, wa, g2) PD_FMAC(t3, wa, g3) PD_FMAC(t0, wb, m0) PD_FMAC(t1, wb, m1) PD_FMAC(t2, wb, m2) PD_FMAC(t3, wb, m3)
                     : [t0] "+v"(T[0]), [t1] "+v"(T[1]), [t2] "+v"(T[2]), [t3] "+v"(T[3]), [pk] "=&v"(pk), [p0] "=&v"(pp[0]), [p1] "=&v"(pp[1]), [p2] "=&v"(pp[2]), [p3] "=&v"(pp[3])
                     : [g0] "v"(hA[0]), [g1] "v"(hA[1]), [g2] "v"(hA[2]), [g3] "v"(hA[3]), [m0] "v"(mB[0]), [m1] "v"(mB[1]), [m2] "v"(mB[2]), [m3] "v"(mB[3]), [h0] "v"(ph[0]), [h1] "v"(ph[1]), [h2] "v"(ph[2]), [h3] "v"(ph[3]), [wa] "v"(wa), [wb] "v"(wb), [phs] "v"(phs), [n] "i"(N));
    }
};
template <int N>
struct Update2First<4, N> {   // first column of a (cos, sin) pair: also hands phAB_i phAB_k to the second
    static __device__ __forceinline__ void run(double (&T)[4], const double (&hA)[4], const double (&mB)[4], const double (&ph)[4],
                                               double wa, double wb, double phs, double (&pp)[4])
    {
        double pk;
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
                     PD_MUL(p0, h0, pk) PD_MUL(p1, h1, pk) PD_MUL(p2, h2, pk) PD_MUL(p3, h3, pk) PD_MUL(t0, p0, t0) PD_MUL(t1, p1, t1) PD_MUL(t2, p2, t2) PD_MUL(t3, p3, t3) PD_FMAC(t0, wa, g0) PD_FMAC(t1, wa, g1) PD_FMAC(t2, wa, g2) PD_FMAC(t3, wa, g3) PD_FMAC(t0, wb, m0) PD_FMAC(t1, wb, m1) PD_FMAC(t2, wb, m2) PD_FMAC(t3, wb, m3)
                     : [t0] "+v"(T[0]), [t1] "+v"(T[1]), [t2] "+v"(T[2]), [t3] "+v"(T[3]), [pk] "=&v"(pk), [p0] "=&v"(pp[0]), [p1] "=&v"(pp[1]), [p2] "=&v"(pp[2]), [p3] "=&v"(pp[3])
                     : [g0] "v"(hA[0]), [g1] "v"(hA[1]), [g2] "v"(hA[2]), [g3] "v"(hA[3]), [m0] "v"(mB[0]), [m1] "v"(mB[1]), [m2] "v"(mB[2]), [m3] "v"(mB[3]), [h0] "v"(ph[0]), [h1] "v"(ph[1]), [h2] "v"(ph[2]), [h3] "v"(ph[3]), [wa] "v"(wa), [wb] "v"(wb), [phs] "v"(phs), [n] "i"(N));
    }
};
template <int N>
struct Update2Second<4, N> {
    static __device__ __forceinline__ void run(double (&T)[4], const double (&hA)[4], const double (&mB)[4], double wa, double wb,
                                               const double (&pp)[4])
    {
        asm volatile(PD_MUL(t0, p0, t0) PD_MUL(t1, p1, t1) PD_MUL(t2, p2, t2) PD_MUL(t3, p3, t3) PD_FMAC(t0, wa, g0) PD_FMAC(t1, wa, g1) PD_FMAC(t2, wa, g2) PD_FMAC(t3, wa, g3) PD_FMAC(t0, wb, m0) PD_FMAC(t1, wb, m1) PD_FMAC(t2, wb, m2) PD_FMAC(t3, wb, m3)
                     : [t0] "+v"(T[0]), [t1] "+v"(T[1]), [t2] "+v"(T[2]), [t3] "+v"(T[3])
                     : [g0] "v"(hA[0]), [g1] "v"(hA[1]), [g2] "v"(hA[2]), [g3] "v"(hA[3]), [m0] "v"(mB[0]), [m1] "v"(mB[1]), [m2] "v"(mB[2]), [m3] "v"(mB[3]), [p0] "v"(pp[0]), [p1] "v"(pp[1]), [p2] "v"(pp[2]), [p3] "v"(pp[3]), [wa] "v"(wa), [wb] "v"(wb), [n] "i"(N));
    }
};
template <int N>
struct MatVec2<5, N> {   // rA += T bcast(u~A_k), rB += T bcast(u~B_k) for one column k of this lane's rows
    static __device__ __forceinline__ void run(const double (&T)[5], double (&rA)[5], double (&rB)[5], double ua, double ub)
    {
        asm volatile(PD_FMAC(a0, ua, t0) PD_FMAC(a1, ua, t1) PD_FMAC(a2, ua, t2) PD_FMAC(a3, ua, t3) PD_FMAC(a4, ua, t4) PD_FMAC(b0, ub, t0) PD_FMAC(b1, ub, t1) PD_FMAC(b2, ub, t2) PD_FMAC(b3, ub, t3) PD_FMAC(b4, ub, t4)
                     : [a0] "+v"(rA[0]), [a1] "+v"(rA[1]), [a2] "+v"(rA[2]), [a3] "+v"(rA[3]), [a4] "+v"(rA[4]), [b0] "+v"(rB[0]), [b1] "+v"(rB[1]), [b2] "+v"(rB[2]), [b3] "+v"(rB[3]), [b4] "+v"(rB[4])
                     : [t0] "v"(T[0]), [t1] "v"(T[1]), [t2] "v"(T[2]), [t3] "v"(T[3]), [t4] "v"(T[4]), [ua] "v"(ua), [ub] "v"(ub), [n] "i"(N));
    }
};
template <int N>
struct Update2<5, N> {   // T = (phAB_i bcast(phAB_k)) T + hA_i bcast(wA_k) + mB_i bcast(wB_k)
    static __device__ __forceinline__ void run(double (&T)[5], const double (&hA)[5], const double (&mB)[5], const double (&ph)[5],
                                               double wa, double wb, double phs)
    {
        double pk, pp[5];
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
                     PD_MUL(p0, h0, pk) PD_MUL(p1, h1, pk) PD_MUL(p2, h2, pk) PD_MUL(p3, h3, pk) PD_MUL(p4, h4, pk) PD_MUL(t0, p0, t0) PD_MUL(t1, p1, t1) PD_MUL(t2, p2, t2) PD_MUL(t3, p3, t3) PD_MUL(t4, p4, t4) PD_FMAC(t0, wa, g0) PD_FMAC(t1, wa, g1) PD_FMAC(t2, wa, g2) PD_FMAC(t3, wa, g3) PD_FMAC(t4, wa, g4) PD_FMAC(t0, wb, m0) PD_FMAC(t1, wb, m1) PD_FMAC(t2, wb, m2) PD_FMAC(t3, wb, m3) PD_FMAC(t4, wb, m4)
                     : [t0] "+v"(T[0]), [t1] "+v"(T[1]), [t2] "+v"(T[2]), [t3] "+v"(T[3]), [t4] "+v"(T[4]), [pk] "=&v"(pk), [p0] "=&v"(pp[0]), [p1] "=&v"(pp[1]), [p2] "=&v"(pp[2]), [p3] "=&v"(pp[3]), [p4] "=&v"(pp[4])
                     : [g0] "v"(hA[0]), [g1] "v"(hA[1]), [g2] "v"(hA[2]), [g3] "v"(hA[3]), [g4] "v"(hA[4]), [m0] "v"(mB[0]), [m1] "v"(mB[1]), [m2] "v"(mB[2]), [m3] "v"(mB[3]), [m4] "v"(mB[4]), [h0] "v"(ph[0]), [h1] "v"(ph[1]), [h2] "v"(ph[2]), [h3] "v"(ph[3]), [h4] "v"(ph[4]), [wa] "v"(wa), [wb] "v"(wb), [phs] "v"(phs), [n] "i"(N));
    }
};
template <int N>
struct Update2First<5, N> {   // first column of a (cos, sin) pair: also hands phAB_i phAB_k to the second
    static __device__ __forceinline__ void run(double (&T)[5], const double (&hA)[5], const double (&mB)[5], const double (&ph)[5],
                                               double wa, double wb, double phs, double (&pp)[5])
    {
        double pk;
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
                     PD_MUL(p0, h0, pk) PD_MUL(p1, h1, pk) PD_MUL(p2, h2, pk) PD_MUL(p3, h3, pk) PD_MUL(p4, h4, pk) PD_MUL(t0, p0, t0) PD_MUL(t1, p1, t1) PD_MUL(t2, p2, t2) PD_MUL(t3, p3, t3) PD_MUL(t4, p4, t4) PD_FMAC(t0, wa, g0) PD_FMAC(t1, wa, g1) PD_FMAC(t2, wa, g2) PD_FMAC(t3, wa, g3) PD_FMAC(t4, wa, g4) PD_FMAC(t0, wb, m0) PD_FMAC(t1, wb, m1) PD_FMAC(t2, wb, m2) PD_FMAC(t3, wb, m3) PD_FMAC(t4, wb, m4)
                     : [t0] "+v"(T[0]), [t1] "+v"(T[1]), [t2] "+v"(T[2]), [t3] "+v"(T[3]), [t4] "+v"(T[4]), [pk] "=&v"(pk), [p0] "=&v"(pp[0]), [p1] "=&v"(pp[1]), [p2] "=&v"(pp[2]), [p3] "=&v"(pp[3]), [p4] "=&v"(pp[4])
                     : [g0] "v"(hA[0]), [g1] "v"(hA[1]), [g2] "v"(hA[2]), [g3] "v"(hA[3]), [g4] "v"(hA[4]), [m0] "v"(mB[0]), [m1] "v"(mB[1]), [m2] "v"(mB[2]), [m3] "v"(mB[3]), [m4] "v"(mB[4]), [h0] "v"(ph[0]), [h1] "v"(ph[1]), [h2] "v"(ph[2]), [h3] "v"(ph[3]), [h4] "v"(ph[4]), [wa] "v"(wa), [wb] "v"(wb), [phs] "v"(phs), [n] "i"(N));
    }
};
template <int N>
struct Update2Second<5, N> {
    static __device__ __forceinline__ void run(double (&T)[5], const double (&hA)[5], const double (&mB)[5], double wa, double wb,
                                               const double (&pp)[5])
    {
        asm volatile(PD_MUL(t0, p0, t0) PD_MUL(t1, p1, t1) PD_MUL(t2, p2, t2) PD_MUL(t3, p3, t3) PD_MUL(t4, p4, t4) PD_FMAC(t0, wa, g0) PD_FMAC(t1, wa, g1) PD_FMAC(t2, wa, g2) PD_FMAC(t3, wa, g3) PD_FMAC(t4, wa, g4) PD_FMAC(t0, wb, m0) PD_FMAC(t1, wb, m1) PD_FMAC(t2, wb, m2) PD_FMAC(t3, wb, m3) PD_FMAC(t4, wb, m4)
                     : [t0] "+v"(T[0]), [t1] "+v"(T[1]), [t2] "+v"(T[2]), [t3] "+v"(T[3]), [t4] "+v"(T[4])
                     : [g0] "v"(hA[0]), [g1] "v"(hA[1]), [g2] "v"(hA[2]), [g3] "v"(hA[3]), [g4] "v"(hA[4]), [m0] "v"(mB[0]), [m1] "v"(mB[1]), [m2] "v"(mB[2]), [m3] "v"(mB[3]), [m4] "v"(mB[4]), [p0] "v"(pp[0]), [p1] "v"(pp[1]), [p2] "v"(pp[2]), [p3] "v"(pp[3]), [p4] "v"(pp[4]), [wa] "v"(wa), [wb] "v"(wb), [n] "i"(N));
    }
};
template <int RPL, int N> struct MatVec3;
template <int RPL, int N> struct Update3;
template <int RPL, int N> struct Update3First;
template <int RPL, int N> struct Update3Second;
template <int N>
struct MatVec3<1, N> {   // rA += T bcast(u~A_k), rB += T bcast(u~B_k), rC += T bcast(u~C_k) for one column k of this lane's rows
    static __device__ __forceinline__ void run(const double (&T)[1], double (&rA)[1], double (&rB)[1], double (&rC)[1], double ua, double ub, double uc)
    {
        asm volatile(PD_FMAC(a0, ua, t0) PD_FMAC(b0, ub, t0) PD_FMAC(c0, uc, t0)
                     : [a0] "+v"(rA[0]), [b0] "+v"(rB[0]), [c0] "+v"(rC[0])
                     : [t0] "v"(T[0]), [ua] "v"(ua), [ub] "v"(ub), [uc] "v"(uc), [n] "i"(N));
    }
};
template <int N>
struct Update3<1, N> {   // T = (cC_i bcast(cC_k)) T + hA_i bcast(wA_k) + hB_i bcast(wB_k) + mC_i bcast(wC_k)
    static __device__ __forceinline__ void run(double (&T)[1], const double (&hA)[1], const double (&hB)[1], const double (&mC)[1], const double (&ph)[1],
                                               double wa, double wb, double wc, double phs)
    {
        double pk, pp[1];
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
                     PD_MUL(p0, h0, pk) PD_MUL(t0, p0, t0) PD_FMAC(t0, wa, g0) PD_FMAC(t0, wb, k0) PD_FMAC(t0, wc, m0)
                     : [t0] "+v"(T[0]), [pk] "=&v"(pk), [p0] "=&v"(pp[0])
                     : [g0] "v"(hA[0]), [k0] "v"(hB[0]), [m0] "v"(mC[0]), [h0] "v"(ph[0]), [wa] "v"(wa), [wb] "v"(wb), [wc] "v"(wc), [phs] "v"(phs), [n] "i"(N));
    }
};
template <int N>
struct Update3First<1, N> {   // first column of a (cos, sin) pair: also hands cC_i cC_k to the second
    static __device__ __forceinline__ void run(double (&T)[1], const double (&hA)[1], const double (&hB)[1], const double (&mC)[1], const double (&ph)[1],
                                               double wa, double wb, double wc, double phs, double (&pp)[1])
    {
        double pk;
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
                     PD_MUL(p0, h0, pk) PD_MUL(t0, p0, t0) PD_FMAC(t0, wa, g0) PD_FMAC(t0, wb, k0) PD_FMAC(t0, wc, m0)
                     : [t0] "+v"(T[0]), [pk] "=&v"(pk), [p0] "=&v"(pp[0])
                     : [g0] "v"(hA[0]), [k0] "v"(hB[0]), [m0] "v"(mC[0]), [h0] "v"(ph[0]), [wa] "v"(wa), [wb] "v"(wb), [wc] "v"(wc), [phs] "v"(phs), [n] "i"(N));
    }
};
template <int N>
struct Update3Second<1, N> {
    static __device__ __forceinline__ void run(double (&T)[1], const double (&hA)[1], const double (&hB)[1], const double (&mC)[1], double wa, double wb, double wc,
                                               const double (&pp)[1])
    {
        asm volatile(PD_MUL(t0, p0, t0) PD_FMAC(t0, wa, g0) PD_FMAC(t0, wb, k0) PD_FMAC(t0, wc, m0)
                     : [t0] "+v"(T[0])
                     : [g0] "v"(hA[0]), [k0] "v"(hB[0]), [m0] "v"(mC[0]), [p0] "v"(pp[0]), [wa] "v"(wa), [wb] "v"(wb), [wc] "v"(wc), [n] "i"(N));
    }
};
template <int N>
struct MatVec3<2, N> {   // rA += T bcast(u~A_k), rB += T bcast(u~B_k), rC += T bcast(u~C_k) for one column k of this lane's rows
    static __device__ __forceinline__ void run(const double (&T)[2], double (&rA)[2], double (&rB)[2], double (&rC)[2], double ua, double ub, double uc)
    {
        asm volatile(PD_FMAC(a0, ua, t0) PD_FMAC(a1, ua, t1) PD_FMAC(b0, ub, t0) PD_FMAC(b1, ub, t1) PD_FMAC(c0, uc, t0) PD_FMAC(c1, uc, t1)
                     : [a0] "+v"(rA[0]), [a1] "+v"(rA[1]), [b0] "+v"(rB[0]), [b1] "+v"(rB[1]), [c0] "+v"(rC[0]), [c1] "+v"(rC[1])
                     : [t0] "v"(T[0]), [t1] "v"(T[1]), [ua] "v"(ua), [ub] "v"(ub), [uc] "v"(uc), [n] "i"(N));
    }
};
template <int N>
struct Update3<2, N> {   // T = (cC_i bcast(cC_k)) T + hA_i bcast(wA_k) + hB_i bcast(wB_k) + mC_i bcast(wC_k)
    static __device__ __forceinline__ void run(double (&T)[2], const double (&hA)[2], const double (&hB)[2], const double (&mC)[2], const double (&ph)[2],
                                               double wa, double wb, double wc, double phs)
    {
        double pk, pp[2];
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
                     PD_MUL(p0, h0, pk) PD_MUL(p1, h1, pk) PD_MUL(t0, p0, t0) PD_MUL(t1, p1, t1) PD_FMAC(t0, wa, g0) PD_FMAC(t1, wa, g1) PD_FMAC(t0, wb, k0) PD_FMAC(t1, wb, k1) PD_FMAC(t0, wc, m0) PD_FMAC(t1, wc, m1)
                     : [t0] "+v"(T[0]), [t1] "+v"(T[1]), [pk] "=&v"(pk), [p0] "=&v"(pp[0]), [p1] "=&v"(pp[1])
                     : [g0] "v"(hA[0]), [g1] "v"(hA[1]), [k0] "v"(hB[0]), [k1] "v"(hB[1]), [m0] "v"(mC[0]), [m1] "v"(mC[1]), [h0] "v"(ph[0]), [h1] "v"(ph[1]), [wa] "v"(wa), [wb] "v"(wb), [wc] "v"(wc), [phs] "v"(phs), [n] "i"(N));
    }
};
template <int N>
struct Update3First<2, N> {   // first column of a (cos, sin) pair: also hands cC_i cC_k to the second
    static __device__ __forceinline__ void run(double (&T)[2], const double (&hA)[2], const double (&hB)[2], const double (&mC)[2], const double (&ph)[2],
                                               double wa, double wb, double wc, double phs, double (&pp)[2])
    {
        double pk;
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
                     PD_MUL(p0, h0, pk) PD_MUL(p1, h1, pk) PD_MUL(t0, p0, t0) PD_MUL(t1, p1, t1) PD_FMAC(t0, wa, g0) PD_FMAC(t1, wa, g1) PD_FMAC(t0, wb, k0) PD_FMAC(t1, wb, k1) PD_FMAC(t0, wc, m0) PD_FMAC(t1, wc, m1)
                     : [t0] "+v"(T[0]), [t1] "+v"(T[1]), [pk] "=&v"(pk), [p0] "=&v"(pp[0]), [p1] "=&v"(pp[1])
                     : [g0] "v"(hA[0]), [g1] "v"(hA[1]), [k0] "v"(hB[0]), [k1] "v"(hB[1]), [m0] "v"(mC[0]), [m1] "v"(mC[1]), [h0] "v"(ph[0]), [h1] "v"(ph[1]), [wa] "v"(wa), [wb] "v"(wb), [wc] "v"(wc), [phs] "v"(phs), [n] "i"(N));
    }
};
template <int N>
struct Update3Second<2, N> {
    static __device__ __forceinline__ void run(double (&T)[2], const double (&hA)[2], const double (&hB)[2], const double (&mC)[2], double wa, double wb, double wc,
                                               const double (&pp)[2])
    {
        asm volatile(PD_MUL(t0, p0, t0) PD_MUL(t1, p1, t1) PD_FMAC(t0, wa, g0) PD_FMAC(t1, wa, g1) PD_FMAC(t0, wb, k0) PD_FMAC(t1, wb, k1) PD_FMAC(t0, wc, m0) PD_FMAC(t1, wc, m1)
                     : [t0] "+v"(T[0]), [t1] "+v"(T[1])
                     : [g0] "v"(hA[0]), [g1] "v"(hA[1]), [k0] "v"(hB[0]), [k1] "v"(hB[1]), [m0] "v"(mC[0]), [m1] "v"(mC[1]), [p0] "v"(pp[0]), [p1] "v"(pp[1]), [wa] "v"(wa), [wb] "v"(wb), [wc] "v"(wc), [n] "i"(N));
    }
};
template <int N>
struct MatVec3<3, N> {   // rA += T bcast(u~A_k), rB += T bcast(u~B_k), rC += T bcast(u~C_k) for one column k of this lane's rows
    static __device__ __forceinline__ void run(const double (&T)[3], double (&rA)[3], double (&rB)[3], double (&rC)[3], double ua, double ub, double uc)
    {
        asm volatile(PD_FMAC(a0, ua, t0) PD_FMAC(a1, ua, t1) PD_FMAC(a2, ua, t2) PD_FMAC(b0, ub, t0) PD_FMAC(b1, ub, t1) PD_FMAC(b2, ub, t2) PD_FMAC(c0, uc, t0) PD_FMAC(c1, uc, t1) PD_FMAC(c2, uc, t2)
                     : [a0] "+v"(rA[0]), [a1] "+v"(rA[1]), [a2] "+v"(rA[2]), [b0] "+v"(rB[0]), [b1] "+v"(rB[1]), [b2] "+v"(rB[2]), [c0] "+v"(rC[0]), [c1] "+v"(rC[1]), [c2] "+v"(rC[2])
                     : [t0] "v"(T[0]), [t1] "v"(T[1]), [t2] "v"(T[2]), [ua] "v"(ua), [ub] "v"(ub), [uc] "v"(uc), [n] "i"(N));
    }
};
template <int N>
struct Update3<3, N> {   // T = (cC_i bcast(cC_k)) T + hA_i bcast(wA_k) + hB_i bcast(wB_k) + mC_i bcast(wC_k)
    static __device__ __forceinline__ void run(double (&T)[3], const double (&hA)[3], const double (&hB)[3], const double (&mC)[3], const double (&ph)[3],
                                               double wa, double wb, double wc, double phs)
    {
        double pk, pp[3];
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
                     PD_MUL(p0, h0, pk) PD_MUL(p1, h1, pk) PD_MUL(p2, h2, pk) PD_MUL(t0, p0, t0) PD_MUL(t1, p1, t1) PD_MUL(t2, p2, t2) PD_FMAC(t0, wa, g0) PD_FMAC(t1, wa, g1) PD_FMAC(t2, wa, g2) PD_FMAC(t0, wb, k0) PD_FMAC(t1, wb, k1) PD_FMAC(t2, wb, k2) PD_FMAC(t0, wc, m0) PD_FMAC(t1, wc, m1) PD_FMAC(t2, wc, m2)
                     : [t0] "+v"(T[0]), [t1] "+v"(T[1]), [t2] "+v"(T[2]), [pk] "=&v"(pk), [p0] "=&v"(pp[0]), [p1] "=&v"(pp[1]), [p2] "=&v"(pp[2])
                     : [g0] "v"(hA[0]), [g1] "v"(hA[1]), [g2] "v"(hA[2]), [k0] "v"(hB[0]), [k1] "v"(hB[1]), [k2] "v"(hB[2]), [m0] "v"(mC[0]), [m1] "v"(mC[1]), [m2] "v"(mC[2]), [h0] "v"(ph[0]), [h1] "v"(ph[1]), [h2] "v"(ph[2]), [wa] "v"(wa), [wb] "v"(wb), [wc] "v"(wc), [phs] "v"(phs), [n] "i"(N));
    }
};
template <int N>
struct Update3First<3, N> {   // first column of a (cos, sin) pair: also hands cC_i cC_k to the second
    static __device__ __forceinline__ void run(double (&T)[3], const double (&hA)[3], const double (&hB)[3], const double (&mC)[3], const double (&ph)[3],
                                               double wa, double wb, double wc, double phs, double (&pp)[3])
    {
        double pk;
        asm volatile("v_mov_b64_dpp %[pk], %[phs]" PIORAN_DPP_CTRL(%c[n]) "\n\t"
                     PD_MUL(p0, h0, pk) PD_MUL(p1, h1, pk) PD_MUL(p2, h2, pk) PD_MUL(t0, p0, t0) PD_MUL(t1, p1, t1) PD_MUL(t2, p2, t2) PD_FMAC(t0, wa, g0) PD_FMAC(t1, wa, g1) PD_FMAC(t2, wa, g2) PD_FMAC(t0, wb, k0) PD_FMAC(t1, wb, k1) PD_FMAC(t2, wb, k2) PD_FMAC(t0, wc, m0) PD_FMAC(t1, wc, m1) PD_FMAC(t2, wc, m2)
                     : [t0] "+v"(T[0]), [t1] "+v"(T[1]), [t2] "+v"(T[2]), [pk] "=&v"(pk), [p0] "=&v"(pp[0]), [p1] "=&v"(pp[1]), [p2] "=&v"(pp[2])
                     : [g0] "v"(hA[0]), [g1] "v"(hA[1]), [g2] "v"(hA[2]), [k0] "v"(hB[0]), [k1] "v"(hB[1]), [k2] "v"(hB[2]), [m0] "v"(mC[0]), [m1] "v"(mC[1]), [m2] "v"(mC[2]), [h0] "v"(ph[0]), [h1] "v"(ph[1]), [h2] "v"(ph[2]), [wa] "v"(wa), [wb] "v"(wb), [wc] "v"(wc), [phs] "v"(phs), [n] "i"(N));
    }
};
template <int N>
struct Update3Second<3, N> {
    static __device__ __forceinline__ void run(double (&T)[3], const double (&hA)[3], const double (&hB)[3], const double (&mC)[3], double wa, double wb, double wc,
                                               const double (&pp)[3])
    {
        asm volatile(PD_MUL(t0, p0, t0) PD_MUL(t1, p1, t1) PD_MUL(t2, p2, t2) PD_FMAC(t0, wa, g0) PD_FMAC(t1, wa, g1) PD_FMAC(t2, wa, g2) PD_FMAC(t0, wb, k0) PD_FMAC(t1, wb, k1) PD_FMAC(t2, wb, k2) PD_FMAC(t0, wc, m0) PD_FMAC(t1, wc, m1) PD_FMAC(t2, wc, m2)
                     : [t0] "+v"(T[0]), [t1] "+v"(T[1]), [t2] "+v"(T[2])
                     : [g0] "v"(hA[0]), [g1] "v"(hA[1]), [g2] "v"(hA[2]), [k0] "v"(hB[0]), [k1] "v"(hB[1]), [k2] "v"(hB[2]), [m0] "v"(mC[0]), [m1] "v"(mC[1]), [m2] "v"(mC[2]), [p0] "v"(pp[0]), [p1] "v"(pp[1]), [p2] "v"(pp[2]), [wa] "v"(wa), [wb] "v"(wb), [wc] "v"(wc), [n] "i"(N));
    }
};
// ---- END GENERATED ----
#undef PD_FMAC
#undef PD_MUL

// sum over the 16 lanes of a DPP row; every lane gets the bit-identical total
__device__ __forceinline__ double row16_sum(double x)
{
    x += dpp_perm<0xB1>(x);   // quad_perm [1,0,3,2]
    x += dpp_perm<0x4E>(x);   // quad_perm [2,3,0,1]
    x += dpp_perm<0x141>(x);  // row_half_mirror
    x += dpp_perm<0x140>(x);  // row_mirror
    return x;
}

// sum over lanes 0 .. NSRC-1 of each DPP row, delivered to all 16 lanes: acc += bcast_N(x) * 1.0 as v_fmac_f64_dpp
// row_newbcast:N — ONE 4-cycle DP instruction per contributing lane (two interleaved chains), where the butterfly above
// costs four stages of two 32-bit DPP moves (the DP ALU has no other DPP control than row_newbcast; tools/valu_probe.hip:
// 6.3 cycles per v_mov_b32_dpp) plus an add.  Lanes >= NSRC are never read, so no "contributes" mask is needed.
template <int N>
__device__ __forceinline__ void bcast_acc(double& acc, double x, double one)
{
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%c3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(one), "i"(N));
}

template <int NSRC>
__device__ __forceinline__ double row_sum_sources(double x, double one)
{
    double a0, a1 = 0.0;
    // s_nop 1: x was just written by a VALU instruction and is read through DPP (2 wait states)
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "=v"(a0) : "v"(x));
    static_for<1, NSRC>([&](auto Nc) {
        constexpr int N = decltype(Nc)::value;
        if constexpr (N & 1) bcast_acc<N>(a1, x, one);
        else bcast_acc<N>(a0, x, one);
    });
    return NSRC > 1 ? a0 + a1 : a0;
}

// u'q over one draw: every row is counted once (physical lanes l < NSRC of each of the CBR DPP rows)
// value of x in the lane at byte address `addr4` (= 4 * lane) of this wavefront: two ds_bpermute_b32 (LDS crossbar, no VALU issue)
__device__ __forceinline__ double lane_fetch(double x, int addr4)
{
    const int lo = __builtin_amdgcn_ds_bpermute(addr4, __double2loint(x));
    const int hi = __builtin_amdgcn_ds_bpermute(addr4, __double2hiint(x));
    return __hiloint2double(hi, lo);
}

// p1, p2: any lane of the DPP row across (r ^ 1, r ^ 2) — after the row sum all 16 lanes of a row hold its total
// GS (round 3) trades exchange rounds for broadcast-accumulate instructions: every DPP row of a draw holds ALL the draw's
// row slots (rotated), so once a row vector is the same in every DPP row (after the exchange of r = T u~) a sum over
// rows needs no exchange at all — GS = 2: all 16 lanes of the DPP row, no ds_bpermute round; GS = 1 (four DPP rows per
// draw): the 2 NSRC lanes that this DPP row and its neighbour r ^ 1 contribute, then ONE round with the row r ^ 2.
// One draw per wavefront leaves two wavefronts per SIMD at most, and each dependent ds_bpermute round trip (~200 cycles)
// is then exposed; GS = 0 (NSRC lanes + log2(CBR) rounds) stays the choice where the SIMD is issue-bound anyway.
template <int CBR, int NSRC, int GS = 0>
__device__ __forceinline__ double group_sum(double x, bool contributes, double one, int p1, int p2)
{
    if constexpr (GS == 2 && CBR >= 2) {
        return row_sum_sources<16>(x, one);
    } else if constexpr (GS == 1 && CBR == 4) {
        x = row_sum_sources<2 * NSRC>(x, one);
        return x + lane_fetch(x, p2);
    } else
    if constexpr (NSRC <= 10) {
        x = row_sum_sources<NSRC>(x, one);
    } else {
        if (!contributes) x = 0.0;
        x = row16_sum(x);
    }
    if constexpr (CBR >= 2) x += lane_fetch(x, p1);
    if constexpr (CBR >= 4) x += lane_fetch(x, p2);
    return x;
}

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double buf_load_f64(__amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
    return __hiloint2double((int)v.y, (int)v.x);
}

// 1/x to fp64 accuracy without the IEEE division sequence (no denormal scaling needed: D_n is O(1e-6..1e6)):
// v_rcp_f64 seed + two Newton steps; 0 -> inf, NaN -> NaN, so failures still surface in `status`.
__device__ __forceinline__ double recip_f64(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

template <int RPL>
struct StepIn {           // what one time step reads: per own row (v, x, phi) + the shared y_n, sigma2_n
    double v[RPL], x[RPL], ph[RPL], y, s2;
};

// NPB > 0 selects the "two-row terms first, one-row terms after" slot layout (DRWCelerite: n complex terms and n real
// ones, src/psd.jl:264-275): every DPP row's block of RB slots holds NPB column PAIRS (rows of NPB complex terms), then
// RB - 1 - 2 NPB single rows (real terms), then one spare slot (padding; the y row in the last block).  All blocks
// look alike, so the phi_i phi_k product of a pair is formed once per pair in every DPP row of the wavefront.
// WIN2: the two-step form (below, after the slot set-up) instead of the step-by-step recurrence.
// YC (round 3, two-step form only): y is NOT one of the row slots — every slot can be a real row, so the shape holds one more
// row (64 in the 64-row shape: the reference benchmark's j = 32, benchmark/benchmarks.jl:16-18, which used to pay the 80-row
// shape).  The y row of T, f = T[y][.], rides as RPL more entries per lane (the lane's own rows), its products f'u~A, f'u~B cost
// two more row sums per pair (independent of the pass over T) and z_n = y_n - mu - f'u~ falls out in every lane.
// WIN3 (round 3, experiment): the THREE-step form — the same re-association with a window of three steps: per entry of T three
// mat-vec FMAs, ONE Hadamard scaling and three rank-1 FMAs per three steps (2.5 instead of 2.75 instructions per entry and step),
// paid for with three more row sums per window (g_AB, g_AC, g_BC) and the h vectors that carry m_A, m_B forward.
// WS = 2 (round 4; two-step form, y as a vector, GS = 2): ONE DRAW OVER TWO WAVEFRONTS.  80 rows at five rows per lane leave one
// wavefront 100 entries of T per lane — 200 of 256 registers before a single row vector: the two-step form does not fit and the shape ran
// step by step with AGPR copies at a quarter of the FP64 roof.  Here the draw's column blocks are spread over the 8 DPP rows of a PAIR of
// wavefronts (NSRC * 8 <= 16 source lanes; a workgroup is that pair = one draw, so its barrier couples nothing else): 50 entries per lane, the same loop body as
// the 48-row shapes.  Per pair of steps ONE exchange crosses the pair — the partial products r = T u~ (2 RPL doubles per lane) through
// LDS behind a workgroup barrier, double-buffered by the parity of the pair so that one barrier per exchange suffices; every row sum
// after it is taken over the 16 lanes of a DPP row (GS = 2: once r is complete every DPP row of both wavefronts holds ALL row slots,
// rotated), so no scalar ever crosses.
template <int RPL, int CBR, int NSRC, bool SHARED_TAB, int MINW = 1, bool PAIRED = false, bool MIXED = false, int NPB = 0, bool WIN2 = false, int GS = 0, bool YC = false,
          bool WIN3 = false, int WS = 1>
__global__ void __launch_bounds__(256, MINW) celerite_scan_kernel(const ScanParams p)
{
    static_assert(WS == 1 || (WS == 2 && WIN2 && YC && GS == 2 && CBR == 4 && SHARED_TAB), "one draw over two wavefronts: two-step form, y as a vector, row sums over all 16 lanes");
    static_assert(!YC || (WIN2 && !MIXED && NPB == 0), "y as a separate vector: two-step form, no per-draw rows, no block layout");
    static_assert(!WIN3 || (!WIN2 && !YC && !MIXED && NPB == 0 && SHARED_TAB && RPL <= 3), "three-step form: shared table, plain or paired layout");
    static_assert(NPB == 0 || (!PAIRED && 2 * NPB < NSRC * RPL && RPL % 2 == 0),
                  "block layout: unpaired base; both columns of a pair must sit in one source lane");
    static_assert(NSRC * CBR * WS <= 16, "source lanes must fit a DPP row");
    constexpr int G = 16 * CBR;          // lanes per draw (WS = 2: per wavefront of the draw's pair)
    constexpr int EPW = 64 / G;          // draws per wavefront
    constexpr int NC = NSRC * RPL;       // columns held per lane
    constexpr int YLAM = NSRC * CBR * WS - 1; // the y row is the LAST row slot: logical lane YLAM, slot RPL-1
    constexpr int YS = RPL - 1;
    // Columns nobody reads (round 3).  A column whose slot has u = 0 in EVERY DPP row of the wavefront contributes nothing to S u,
    // and an entry of S is read by nothing else, so such a column is neither updated nor kept in registers: the spare slot that
    // closes every block of the block layout and of a paired layout with an odd block (padding, or the y row in the last block);
    // with one DPP row per draw the y row's own slot — the last one — and, paired with an even block, the padding slot beside it
    // (capacity() leaves it free).  With several DPP rows per draw the last slot of the other blocks is a real row.
    constexpr int NDEAD = NPB > 0 ? 1 : (PAIRED && ((NSRC * RPL) & 1)) ? 1 : (YC || CBR > 1) ? 0 : (PAIRED ? 2 : 1);
    constexpr int NCL = NC - NDEAD;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // known to be wave-uniform: with one draw per wavefront
                                                                         // (CBR = 4) mu, nu, sum(a) then live in SGPRs
    const int e = EPW == 1 ? 0 : lane / G;
    [[maybe_unused]] const int half = WS == 2 ? (wave & 1) : 0;   // which wavefront of the draw's pair
    const int r = ((lane % G) >> 4) + (WS == 2 ? 4 * half : 0);   // DPP row inside the draw = column block
    const int l = lane & 15;
    const int lam = (l + NSRC * r) & 15;     // logical lane: which rows this lane owns
    const bool contributes = l < NSRC;       // each row is counted once in u'q
    const bool isy = !YC && lam == YLAM;     // this lane's slot YS is the y row
    const int64_t b_raw = WS == 2 ? (int64_t)blockIdx.x : ((int64_t)blockIdx.x * 4 + wave) * EPW + e;   // WS = 2: workgroups of TWO wavefronts = one draw
    const bool active = b_raw < p.B;
    const int64_t b = active ? b_raw : p.B - 1;

    const int J = p.J, R = p.R, Rp = R + 2;  // table rows R (inert pad) and R+1 (y row)
    const int64_t N = p.N;

    // slot -> row.  Unpaired: slot index = row index, the slots after the last real row are padding and the very
    // last slot is the y row.  PAIRED (every term has both rows; R = 2J): the columns of each DPP row's block are
    // processed two at a time, so real rows must sit pair-aligned inside a block: with RB = NSRC*RPL slots per
    // block, an odd RB leaves one "single" slot at the end of each block (padding, or the y row in the last block).
    constexpr int RB = NSRC * RPL;
    constexpr int RBR = PAIRED ? (RB & ~1) : RB;   // real-row slots per block
    int trow[RPL];                           // row of the shared table this slot reads
    double al[RPL], be[RPL];                 // u = al * v + be * x
    [[maybe_unused]] double cc[RPL], dd[RPL];
    [[maybe_unused]] bool ksin[RPL];
    [[maybe_unused]] int pdoff[RPL];         // >= 0: this slot's row is a per-draw row (mixed mode): its index in the per-draw block
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        pdoff[i] = -1;
        const int slot = lam * RPL + i;
        int j;   // real row index, or >= R for a special slot
        if constexpr (NPB > 0) {
            constexpr int NSB = RB - 1 - 2 * NPB;      // single rows per block
            constexpr int NCR = 2 * NPB * CBR;         // rows of the two-row terms (n_complex = NPB * CBR, checked on the host)
            const int blk = slot / RB, c = slot - blk * RB;
            if (c < 2 * NPB) j = blk * 2 * NPB + c;
            else if (c < RB - 1) j = NCR + blk * NSB + (c - 2 * NPB);
            else j = R;                                // spare slot: padding (the y row in the last block)
            if (j > R) j = R;
        } else if constexpr (PAIRED) {
            const int blk = slot / RB, c = slot - blk * RB;
            j = c < RBR ? blk * RBR + c : R;   // the single slot of an odd block is never a real row
        } else {
            j = slot;
        }
        if (j < R) {
            const int rm = p.rowmap[j];
            const int term = rm & 0xfffff;
            const bool ks = (rm >> 30) & 1;
            if (MIXED && ((rm >> 29) & 1)) pdoff[i] = (int)(b * p.npd_rows) + ((rm >> 20) & 0x1ff);   // b: draw within this launch
            const double a = p.A[b * J + term], bb = p.Bc[b * J + term];
            // cos row: v = co, x = si, u = a co + b si ; sin row: v = si, x = co, u = a si - b co   (:59-63)
            trow[i] = j;
            al[i] = a;
            be[i] = ks ? -bb : bb;
            ksin[i] = ks;
            if constexpr (!SHARED_TAB) {
                cc[i] = p.C[b * J + term];
                dd[i] = p.D[b * J + term];
            }
        } else {
            // inert padding row (u = 0, v = 1, phi = 1), or the y row (u = 0, v = y_n - mu, phi = 1)
            trow[i] = (isy && i == YS) ? R + 1 : R;
            al[i] = 0.0;
            be[i] = 0.0;
            ksin[i] = false;
            if constexpr (!SHARED_TAB) { cc[i] = 0.0; dd[i] = 0.0; }
        }
    }
    double suma = 0.0;  // :21
    for (int j = 0; j < J; ++j) suma += p.A[b * J + j];
    const double mu = p.mu ? p.mu[b] : 0.0;
    const double nu = p.nu ? p.nu[b] : 1.0;
    const bool own_series = p.Y != nullptr;   // per-draw y / sigma2 (wave-uniform)
    [[maybe_unused]] const double* yv = own_series ? p.Y + b * N : p.y;     // per-draw (c, d) path only
    [[maybe_unused]] const double* sv = own_series ? p.S2 + b * N : p.s2;

    [[maybe_unused]] int p1 = 0, p2 = 0;  // byte addresses (ds_bpermute) of the lanes holding the same rows in the other column blocks
    if constexpr (CBR >= 2) p1 = 4 * (e * G + ((r ^ 1) & 3) * 16 + ((lam - NSRC * (r ^ 1)) & 15));
    if constexpr (CBR >= 4) p2 = 4 * (e * G + ((r ^ 2) & 3) * 16 + ((lam - NSRC * (r ^ 2)) & 15));
    // WS = 2: the lane of the PARTNER wavefront that holds the same rows (column block r ^ 4: same DPP row index there)
    [[maybe_unused]] const int px = (r & 3) * 16 + ((lam - NSRC * (r ^ 4)) & 15);

    // Table record of step n: [v x Rp | x x Rp | phi x Rp | y_n, sigma2_n | per-draw block]; N + 1 records (the last
    // one is a readable dummy so the prefetch of step n + 1 needs no bounds test).  The per-draw block (mixed mode:
    // rows of the few terms whose (c, d) differ per draw) is [draw][row][v, x, phi] and sits INSIDE the step record,
    // so every row has the same step stride.  Read with buffer loads: per-lane byte offsets in VGPRs (constant),
    // step offset in an SGPR -> no vector address arithmetic and no branches.
    const int RS = 3 * Rp + 2;
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double*>(SHARED_TAB ? p.tab : p.t), 0, 0x7ffffffc, 0x00020000);
    // Per-lane byte offsets of v, x, phi.  MIXED launches (some rows are per-draw) need all three per lane; otherwise
    // x and phi are the v offset plus a wave-uniform constant, which can either live in two more VGPRs per row (no
    // scalar adds per load: faster where registers allow, RPL <= 3) or be added to the scalar step offset.
    constexpr bool LANE_OFFS = MIXED || (RPL <= 3 && !WIN3);   // (the three-step form needs every register: offsets through the scalar side)
    [[maybe_unused]] int vo_v[RPL], vo_x[LANE_OFFS ? RPL : 1], vo_p[LANE_OFFS ? RPL : 1];
    [[maybe_unused]] const int step_bytes = (int)p.rec_stride * 8;
    // y_n, sigma2_n of the step: the shared series sit in the step record (columns 3 Rp, 3 Rp + 1); per-draw series are
    // [B][N] arrays read through a buffer resource based at the wavefront's FIRST draw (wave-uniform base in SGPRs,
    // per-lane draw offset in one VGPR, step offset in an SGPR): the same two loads either way, no 64-bit per-lane
    // pointers and no branch in the loop.  The prefetch of step N reads one element past a draw's series: the next draw's
    // first element, or zero from the bounds check of the resource at the very end of the array — never used.
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rs_y = rs, rs_s = rs;
    [[maybe_unused]] int vo_y = 3 * Rp * 8, vo_s = 3 * Rp * 8 + 8, y_step = step_bytes;
    if constexpr (SHARED_TAB) {
        if (own_series) {
            int64_t b0 = WS == 2 ? (int64_t)blockIdx.x : ((int64_t)blockIdx.x * 4 + wave) * EPW;
            b0 = b0 < p.B ? b0 : p.B - 1;
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b0), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b0 >> 32));
            const int64_t b0u = (int64_t)(((uint64_t)hi << 32) | lo);
            const int64_t rem = (p.B - b0u) * N * 8;
            const int recs = rem > 0x7ffffff0LL ? 0x7ffffff0 : (int)rem;
            rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p.Y + b0u * N), 0, recs, 0x00020000);
            rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p.S2 + b0u * N), 0, recs, 0x00020000);
            vo_y = vo_s = (int)((b - b0u) * N * 8);
            y_step = 8;
        }
    }
    if constexpr (SHARED_TAB) {
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            if (MIXED && pdoff[i] >= 0) {
                vo_v[i] = (RS + pdoff[i] * 3) * 8;
                if constexpr (LANE_OFFS) { vo_x[i] = vo_v[i] + 8; vo_p[i] = vo_v[i] + 16; }
            } else {
                vo_v[i] = trow[i] * 8;
                if constexpr (LANE_OFFS) { vo_x[i] = vo_v[i] + Rp * 8; vo_p[i] = vo_v[i] + 2 * Rp * 8; }
            }
        }
    }

    // one draw per wavefront (two-step form): scalar sources of (y_n, sigma2_n), see load_step
    [[maybe_unused]] const double* ybase1 = own_series ? p.Y + b * N : p.tab + 3 * Rp;
    [[maybe_unused]] const double* sbase1 = own_series ? p.S2 + b * N : p.tab + 3 * Rp + 1;
    [[maybe_unused]] const int ystr1 = own_series ? 1 : (int)p.rec_stride;
    [[maybe_unused]] const int ylast1 = own_series ? (int)N - 1 : (int)N;
    auto load_step = [&](int64_t n, StepIn<RPL>& in) {
        if constexpr (SHARED_TAB) {
            if constexpr (WIN2 || WIN3) n = n < N ? n : N;   // the table holds N + 1 records
            const int soff = (int)n * step_bytes;
            if constexpr (WS != 2) {   // (WS = 2: the rows' (v, x, phi) come through the LDS ring of the two-step loop, not through registers)
#pragma unroll
                for (int i = 0; i < RPL; ++i) {
                    in.v[i] = buf_load_f64(rs, vo_v[i], soff);
                    if constexpr (LANE_OFFS) {
                        in.x[i] = buf_load_f64(rs, vo_x[i], soff);
                        in.ph[i] = buf_load_f64(rs, vo_p[i], soff);
                    } else {
                        in.x[i] = buf_load_f64(rs, vo_v[i], soff + Rp * 8);
                        in.ph[i] = buf_load_f64(rs, vo_v[i], soff + 2 * Rp * 8);
                    }
                }
            }
            if constexpr (WIN2 && EPW == 1) {
                // one draw per wavefront: y_n, sigma2_n are wave-uniform -> scalar loads, no vector registers.  Branch-free
                // (round 3): base, stride and last index are chosen once; a conditional LOAD in the loop body is a real branch,
                // and with it came per-block register copies (105 v_mov_b64 and ~160 scalar instructions per pair of steps)
                const int ny = (int)n < ylast1 ? (int)n : ylast1;
                if constexpr (WS == 2) {
                    // (the workgroup barrier in this shape's loop is a fence: a plain load would no longer be provably unclobbered and would
                    // become a VECTOR load, whose s_waitcnt vmcnt then also waits for the LDS copies in flight — the series are read-only
                    // here, so read them through the constant address space: scalar loads whatever stands between)
                    typedef const __attribute__((address_space(4))) double* cptr;
                    in.y = ((cptr)(uintptr_t)ybase1)[ny * ystr1];
                    in.s2 = ((cptr)(uintptr_t)sbase1)[ny * ystr1];
                } else {
                    in.y = ybase1[ny * ystr1];
                    in.s2 = sbase1[ny * ystr1];
                }
            } else {
                const int ysoff = (int)n * y_step;
                in.y = buf_load_f64(rs_y, vo_y, ysoff);
                in.s2 = buf_load_f64(rs_s, vo_s, ysoff);
            }
        } else {
            const int64_t nn = n < N ? n : N - 1;
            const double tn = p.t[nn];
            const double dt = nn > 0 ? tn - p.t[nn - 1] : 0.0;
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                if (trow[i] < R) {
                    double s_, c_;
                    sincos(dd[i] * tn, &s_, &c_);  // :52-53
                    in.v[i] = ksin[i] ? s_ : c_;
                    in.x[i] = ksin[i] ? c_ : s_;
                    in.ph[i] = exp(-cc[i] * dt);   // :54
                } else {
                    in.v[i] = trow[i] == R ? 1.0 : 0.0;
                    in.x[i] = 0.0;
                    in.ph[i] = 1.0;
                }
            }
            in.y = yv[nn];
            in.s2 = sv[nn];
        }
    };


    // ---- three-step form ------------------------------------------------------------------------------------------------
    // Steps A = m + 1, B = m + 2, C = m + 3 on the state T = S_m + D_m w_m w_m' (cA = phA, cB = phA phB, cC = phA phB phC):
    //   u~s = cs o us;  rs = T u~s  (one pass over T, three FMAs per entry);  sigma_s = u~s'rs
    //   D_A = dA - sigma_A;                                   mA = vA - cA o rA
    //   hAB = phB o mA, g_AB = hAB'uB;  hAC = phC o hAB, g_AC = hAC'uC
    //   D_B = dB - sigma_B - g_AB^2 / D_A;                    mB = vB - cB o rB - hAB g_AB / D_A
    //   hBC = phC o mB, g_BC = hBC'uC
    //   D_C = dC - sigma_C - g_AC^2 / D_A - g_BC^2 / D_B;     mC = vC - cC o rC - hAC g_AC / D_A - hBC g_BC / D_B
    //   T <- (cC cC') o T + hAC hAC' / D_A + hBC hBC' / D_B + mC mC' / D_C
    // The series starts with NV = (3 - N mod 3) mod 3 void steps (D = 1, m = 0).
    if constexpr (WIN3) {
        double T[NC][RPL];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int i = 0; i < RPL; ++i) T[c][i] = 0.0;
        __shared__ double sh_coef3[2 * RPL * 256];   // the row coefficients (al, be) wait in LDS (lane-private slots): 4 RPL registers
#pragma unroll
        for (int i = 0; i < RPL; ++i) { sh_coef3[(2 * i) * 256 + threadIdx.x] = al[i]; sh_coef3[(2 * i + 1) * 256 + threadIdx.x] = be[i]; }
        double one = 1.0;
        asm volatile("" : "+v"(one));
        const double ysel = isy ? 1.0 : 0.0;
        double Pm = 1.0, quad = 0.0;
        int Pe = 0;
        bool nonpd = false;
        StepIn<RPL> sa, sb, sc;
        auto triple = [&](int64_t nA, auto nvc, auto firstc) __attribute__((always_inline)) {
            constexpr int NV = decltype(nvc)::value;
            constexpr bool FIRST = decltype(firstc)::value;
            double uB[RPL], uC[RPL], tA[RPL], tB[RPL], tC[RPL], cB[RPL], cC[RPL], rA[RPL], rB[RPL], rC[RPL];
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const double ali = sh_coef3[(2 * i) * 256 + threadIdx.x], bei = sh_coef3[(2 * i + 1) * 256 + threadIdx.x];
                const double uA = ali * sa.v[i] + bei * sa.x[i];
                uB[i] = ali * sb.v[i] + bei * sb.x[i];
                uC[i] = ali * sc.v[i] + bei * sc.x[i];
                cB[i] = sa.ph[i] * sb.ph[i];
                cC[i] = cB[i] * sc.ph[i];
                tA[i] = sa.ph[i] * uA;
                tB[i] = cB[i] * uB[i];
                tC[i] = cC[i] * uC[i];
                rA[i] = 0.0; rB[i] = 0.0; rC[i] = 0.0;
            }
            sa.v[YS] = fma(ysel, sa.y - mu, sa.v[YS]);
            sb.v[YS] = fma(ysel, sb.y - mu, sb.v[YS]);
            sc.v[YS] = fma(ysel, sc.y - mu, sc.v[YS]);
            const double dA = fma(nu, sa.s2, suma), dB = fma(nu, sb.s2, suma), dC = fma(nu, sc.s2, suma);
#pragma unroll
            for (int i = 0; i < RPL; ++i) asm volatile("" : "+v"(tA[i]), "+v"(tB[i]), "+v"(tC[i]));
            asm volatile("s_nop 1");
            constexpr int NCM = NCL;   // (the columns that are ever read: see NDEAD)
            static_for<0, NCM>([&](auto Cc) {
                constexpr int c = decltype(Cc)::value;
                MatVec3<RPL, c / RPL>::run(T[c], rA, rB, rC, tA[c % RPL], tB[c % RPL], tC[c % RPL]);
            });
            if constexpr (CBR >= 2) {
#pragma unroll
                for (int i = 0; i < RPL; ++i) { rA[i] += lane_fetch(rA[i], p1); rB[i] += lane_fetch(rB[i], p1); rC[i] += lane_fetch(rC[i], p1); }
            }
            if constexpr (CBR >= 4) {
#pragma unroll
                for (int i = 0; i < RPL; ++i) { rA[i] += lane_fetch(rA[i], p2); rB[i] += lane_fetch(rB[i], p2); rC[i] += lane_fetch(rC[i], p2); }
            }
            double spA = 0.0, spB = 0.0, spC = 0.0;
#pragma unroll
            for (int i = 0; i < RPL; ++i) { spA += tA[i] * rA[i]; spB += tB[i] * rB[i]; spC += tC[i] * rC[i]; }
            const double sA = group_sum<CBR, NSRC, GS>(spA, contributes, one, p1, p2);
            const double sB = group_sum<CBR, NSRC, GS>(spB, contributes, one, p1, p2);
            const double sC = group_sum<CBR, NSRC, GS>(spC, contributes, one, p1, p2);
            const double DA = NV >= 1 ? 1.0 : dA - sA;                                    // :92
            const double rDA = recip_f64(DA);
            double hAC[RPL], hBC[RPL], mC[RPL], wA[RPL], wB[RPL], wC[RPL], hAB[RPL];
            double spAB = 0.0, spAC = 0.0, zA = 0.0;
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const double mA = NV >= 1 ? 0.0 : fma(-sa.ph[i], rA[i], sa.v[i]);         // v - q      :89
                if (i == YS) zA = mA;
                hAB[i] = sb.ph[i] * mA;
                spAB = fma(hAB[i], uB[i], spAB);
                hAC[i] = sc.ph[i] * hAB[i];
                spAC = fma(hAC[i], uC[i], spAC);
            }
            load_step(nA + 3, sa);
            const double gAB = group_sum<CBR, NSRC, GS>(spAB, contributes, one, p1, p2);
            const double gAC = group_sum<CBR, NSRC, GS>(spAC, contributes, one, p1, p2);
            const double yAB = gAB * rDA, yAC = gAC * rDA;
            const double DB = NV >= 2 ? 1.0 : dB - sB - gAB * yAB;
            const double rDB = recip_f64(DB);
            double spBC = 0.0, zB = 0.0;
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const double mB = NV >= 2 ? 0.0 : fma(-hAB[i], yAB, fma(-cB[i], rB[i], sb.v[i]));
                if (i == YS) zB = mB;
                hBC[i] = sc.ph[i] * mB;
                spBC = fma(hBC[i], uC[i], spBC);
            }
            load_step(nA + 4, sb);
            const double gBC = group_sum<CBR, NSRC, GS>(spBC, contributes, one, p1, p2);
            const double yBC = gBC * rDB;
            const double DC = dC - sC - gAC * yAC - gBC * yBC;
            const double rDC = recip_f64(DC);
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                mC[i] = fma(-hBC[i], yBC, fma(-hAC[i], yAC, fma(-cC[i], rC[i], sc.v[i])));
                wA[i] = hAC[i] * rDA;
                wB[i] = hBC[i] * rDB;
                wC[i] = mC[i] * rDC;
            }
            load_step(nA + 5, sc);
            const double zC = mC[YS];
            nonpd |= !(DA > 0.0) | !(DB > 0.0) | !(DC > 0.0);
            // log(D[1]) :126 keeps the sign of the series' FIRST step, log(abs(D[n])) :140 for the others
            Pm *= ((FIRST && NV == 0) ? DA : fabs(DA)) * ((FIRST && NV == 1) ? DB : fabs(DB)) * ((FIRST && NV == 2) ? DC : fabs(DC));
            {
                int ex;
                Pm = frexp(Pm, &ex);
                Pe += ex;
            }
            quad = fma(zC * zC, rDC, fma(zB * zB, rDB, fma(zA * zA, rDA, quad)));        // z_n^2 / D_n  (== y'K^-1 y, :333)
#pragma unroll
            for (int i = 0; i < RPL; ++i) asm volatile("" : "+v"(wA[i]), "+v"(wB[i]), "+v"(wC[i]), "+v"(cC[i]));
            asm volatile("s_nop 1");
            if constexpr (PAIRED) {
                static_for<0, NCL / 2>([&](auto Pc) {
                    constexpr int c = 2 * decltype(Pc)::value;
                    double pp[RPL];
                    Update3First<RPL, c / RPL>::run(T[c], hAC, hBC, mC, cC, wA[c % RPL], wB[c % RPL], wC[c % RPL], cC[c % RPL], pp);
                    Update3Second<RPL, (c + 1) / RPL>::run(T[c + 1], hAC, hBC, mC, wA[(c + 1) % RPL], wB[(c + 1) % RPL], wC[(c + 1) % RPL], pp);
                });
                // (an odd block's closing column is never read — see the two-step form — and is not kept)
            } else {
                static_for<0, NCL>([&](auto Cc) {
                    constexpr int c = decltype(Cc)::value;
                    Update3<RPL, c / RPL>::run(T[c], hAC, hBC, mC, cC, wA[c % RPL], wB[c % RPL], wC[c % RPL], cC[c % RPL]);
                });
            }
        };
        auto void_step = [&](StepIn<RPL>& s_) {
#pragma unroll
            for (int i = 0; i < RPL; ++i) { s_.v[i] = 0.0; s_.x[i] = 0.0; s_.ph[i] = 1.0; }
            s_.y = mu;
            s_.s2 = 0.0;
        };
        int64_t n;
        const int nv = (int)((3 - N % 3) % 3);
        if (nv == 0) {
            load_step(0, sa); load_step(1, sb); load_step(2, sc);
            triple(0, ic<0>{}, std::true_type{});
            n = 3;
        } else if (nv == 1) {
            void_step(sa); load_step(0, sb); load_step(1, sc);
            triple(-1, ic<1>{}, std::true_type{});
            n = 2;
        } else {
            void_step(sa); void_step(sb); load_step(0, sc);
            triple(-2, ic<2>{}, std::true_type{});
            n = 1;
        }
        for (; n + 2 < N; n += 3) triple(n, ic<0>{}, std::false_type{});
        if (active && isy && r == 0) {
            const double logdet = log(Pm) + (double)Pe * 0.6931471805599453094;
            const double res = -0.5 * logdet - 0.5 * (double)N * 1.8378770664093454836 - 0.5 * quad;
            p.out[b] = res;
            if (p.status) p.status[b] = !isfinite(res) ? 2 : (nonpd ? 1 : 0);
        }
        return;
    }


    // ---- two-step form --------------------------------------------------------------------------------------------------
    // State T = S_m + D_m w_m w_m' (time-m coordinates).  Steps A = m + 1 and B = m + 2 are taken together:
    //   u~A = phA o uA,  u~B = phA phB o uB;   rA = T u~A,  rB = T u~B            one pass over T, two FMAs per entry
    //   D_A = dA - u~A'rA;   mA = vA - phA o rA;   hA = phB o mA;   g = hA'uB
    //   D_B = dB - u~B'rB - g^2 / D_A;   mB = vB - phAB o rB - hA g / D_A
    //   T  <- (phAB phAB') o T + hA hA' / D_A + mB mB' / D_B                       one scaling per entry and PAIR of steps
    // i.e. 5.5 instead of 7 instructions per entry and two steps (the Hadamard scaling by phi phi', which the step-by-step
    // form applies every step, :78-79,85, is applied once per pair), one more row reduction (g).  Same D_n, z_n as the
    // reference up to rounding; T = 0 before the first step, so there is no special first row; a series of odd length
    // starts with the pair (void, step 0).
    if constexpr (WIN2) {
        double T[NC][RPL];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int i = 0; i < RPL; ++i) T[c][i] = 0.0;
        // RPL >= 4: the row coefficients (al, be) wait in LDS (lane-private slots) instead of 4 RPL registers
        constexpr bool COEF_LDS = RPL >= 4;
        // (WS = 2: one copy per LOGICAL lane — the eight DPP rows of the draw hold the same rows, rotated: 1.3 KB instead of 10)
        constexpr int CTH = WS == 2 ? 16 : 256;
        const int cix = WS == 2 ? lam : (int)threadIdx.x;
        __shared__ double sh_coef[COEF_LDS ? 2 * RPL * CTH : 1];
        if constexpr (COEF_LDS) {
#pragma unroll
            for (int i = 0; i < RPL; ++i) { sh_coef[(2 * i) * CTH + cix] = al[i]; sh_coef[(2 * i + 1) * CTH + cix] = be[i]; }
        }
        double one = 1.0;
        asm volatile("" : "+v"(one));
        [[maybe_unused]] const double ysel = isy ? 1.0 : 0.0;
        double Pm = 1.0, quad = 0.0;
        int Pe = 0;
        bool nonpd = false;
        PIORAN_SSTAMP_DECL
        StepIn<RPL> sa, sb;
        __shared__ double sh_x[WS == 2 ? 2 * 2 * 2 * RPL * 64 : 1];   // [parity][wavefront of the pair][2 RPL values][lane]
        [[maybe_unused]] int xpar = 0;
        // WS = 2: the per-step inputs (v, x, phi of this lane's rows, two steps) never occupy registers.  Each wavefront of the pair copies
        // ONE record of a pair of steps into LDS two pairs ahead (LDS DMA, 256 B per instruction: no registers, any alignment), the
        // workgroup barrier of the pair in between is what publishes it, and the values are read where they are used (twice: before the
        // pass over T and after the exchange) — 6 RPL doubles per lane less than the prefetched register sets of the other shapes, which
        // is what lets the two-step body of five rows per lane stay clear of scratch (a reload's s_waitcnt vmcnt(0) would otherwise wait
        // for a table record's round trip from L2 every pair of steps).
        // LDS layout of a record: the three sections (v, x, phi: R + 2 <= 82 doubles each) at a FIXED pitch of 656 bytes, so that every read
        // is `row address + immediate` whatever R is; a slot holds the two records of a pair of steps.
        constexpr int RSEC = 656, RREC = 3 * RSEC, RSLOT = 2 * RREC;   // bytes
        // THREE SEPARATE arrays, the slot a compile-time constant of every pair body (the loop below is unrolled by three): the compiler
        // orders an LDS read behind a preceding LDS DMA (s_waitcnt vmcnt(0)) unless it can see that they touch different objects — with one
        // array and a run-time slot every pair waited for the copy it had just issued (3.8 instead of ~3 us per pair of steps).
        __shared__ double sh_ring0[WS == 2 ? RSLOT / 8 : 1], sh_ring1[WS == 2 ? RSLOT / 8 : 1], sh_ring2[WS == 2 ? RSLOT / 8 : 1];
        [[maybe_unused]] auto ring_ptr = [&](auto slotc) __attribute__((always_inline)) -> double* {
            constexpr int SL = decltype(slotc)::value;
            if constexpr (SL == 0) return sh_ring0; else if constexpr (SL == 1) return sh_ring1; else return sh_ring2;
        };
        [[maybe_unused]] auto ring_dma = [&](auto slotc, int64_t n) __attribute__((always_inline)) {
            if constexpr (WS == 2) {
                const int64_t nn = n < 0 ? 0 : (n < N ? n : N);       // (the table holds N + 1 records)
                const char* rec = (const char*)(p.tab + nn * p.rec_stride);
                const int lim = Rp * 8 - 4;
                char* dst = (char*)ring_ptr(slotc) + half * RREC;
#pragma unroll
                for (int sct = 0; sct < 3; ++sct)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        int off = c * 256 + lane * 4;
                        off = off < lim ? off : lim;                   // past the section: a harmless repeat of its last word
                        if (c < 2 || lane < (RSEC - 512) / 4)          // (the third piece stops at the section's pitch: the next section is another copy's)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rec + sct * Rp * 8 + off),
                                                             (__attribute__((address_space(3))) void*)(dst + sct * RSEC + c * 256), 4, 0, 0);
                    }
            }
        };
        [[maybe_unused]] double fy[RPL];   // YC: the y row of T for this lane's rows
#pragma unroll
        for (int i = 0; i < RPL; ++i) fy[i] = 0.0;
        // (WS = 2: f waits in lane-private LDS slots between its two uses of a pair — ten more registers for the pass over T)
        __shared__ double sh_fy[WS == 2 ? RPL * 128 : 1];
        if constexpr (WS == 2) {
#pragma unroll
            for (int i = 0; i < RPL; ++i) sh_fy[i * 128 + threadIdx.x] = 0.0;
        }
        // VOIDA: step A does not exist (D = 1, m = 0, no contribution); FIRST: step A is the first of the series (log D_1, :126)
        auto pair = [&](int64_t nA, auto voidc, auto firstc, auto slotc) __attribute__((always_inline)) {
            constexpr bool VOIDA = decltype(voidc)::value, FIRST = decltype(firstc)::value;
            [[maybe_unused]] constexpr int SL = decltype(slotc)::value;   // WS = 2: ring slot of this pair
            PIORAN_SSTAMP(0);
            double uB[RPL], tA[RPL], tB[RPL], pAB[RPL], rA[RPL], rB[RPL];
            // per-step inputs: the prefetched register sets, or (WS = 2) the ring slot of this pair, read at the point of use
            [[maybe_unused]] const char* ra_[RPL];    // this pair's slot + the lane's row offset: every read below is ra_[i] + an immediate
            if constexpr (WS == 2) {
#pragma unroll
                for (int i = 0; i < RPL; ++i) ra_[i] = (const char*)ring_ptr(slotc) + vo_v[i];
            }
            auto A_v = [&](int i) { if constexpr (WS == 2) return VOIDA ? 0.0 : *(const double*)(ra_[i]); else return sa.v[i]; };
            auto A_x = [&](int i) { if constexpr (WS == 2) return VOIDA ? 0.0 : *(const double*)(ra_[i] + RSEC); else return sa.x[i]; };
            auto A_ph = [&](int i) { if constexpr (WS == 2) return VOIDA ? 1.0 : *(const double*)(ra_[i] + 2 * RSEC); else return sa.ph[i]; };
            auto B_v = [&](int i) { if constexpr (WS == 2) return *(const double*)(ra_[i] + RREC); else return sb.v[i]; };
            auto B_x = [&](int i) { if constexpr (WS == 2) return *(const double*)(ra_[i] + RREC + RSEC); else return sb.x[i]; };
            auto B_ph = [&](int i) { if constexpr (WS == 2) return *(const double*)(ra_[i] + RREC + 2 * RSEC); else return sb.ph[i]; };
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                double ali = al[i], bei = be[i];
                if constexpr (COEF_LDS) { ali = sh_coef[(2 * i) * CTH + cix]; bei = sh_coef[(2 * i + 1) * CTH + cix]; }
                const double aph = A_ph(i);
                const double uA = ali * A_v(i) + bei * A_x(i);
                uB[i] = ali * B_v(i) + bei * B_x(i);
                pAB[i] = aph * B_ph(i);
                tA[i] = aph * uA;
                tB[i] = pAB[i] * uB[i];
                rA[i] = 0.0;
                rB[i] = 0.0;
            }
            [[maybe_unused]] double ryA = 0.0, ryB = 0.0;
            if constexpr (YC) {   // f'u~A, f'u~B: row vectors only, independent of the pass over T below
                double spyA = 0.0, spyB = 0.0;
#pragma unroll
                for (int i = 0; i < RPL; ++i) {
                    const double fi = WS == 2 ? sh_fy[i * 128 + threadIdx.x] : fy[i];
                    spyA = fma(fi, tA[i], spyA); spyB = fma(fi, tB[i], spyB);
                }
                ryA = group_sum<CBR, NSRC, GS>(spyA, contributes, one, p1, p2);
                ryB = group_sum<CBR, NSRC, GS>(spyB, contributes, one, p1, p2);
            } else if constexpr (COEF_LDS) {   // (register-starved shapes: a select on the lane mask instead of the multiplier ysel)
                sa.v[YS] = isy ? sa.y - mu : sa.v[YS];
                sb.v[YS] = isy ? sb.y - mu : sb.v[YS];
            } else {
                sa.v[YS] = fma(ysel, sa.y - mu, sa.v[YS]);
                sb.v[YS] = fma(ysel, sb.y - mu, sb.v[YS]);
            }
            // DPP hazard (a VALU write of a DPP source needs two wait states before the DPP read): the sources of a pass (u~A, u~B
            // here; wA, wB, phAB below) are written BEFORE the pass and never inside it, so one s_nop in front of the pass covers
            // a producer scheduled directly before it; the column blocks carry none (tools/check_dpp_hazards.py checks the listing)
            // (the "+v" operands pin every source in its register before the pass: the compiler may not sink a producer between blocks)
#pragma unroll
            for (int i = 0; i < RPL; ++i) asm volatile("" : "+v"(tA[i]), "+v"(tB[i]));
            PIORAN_SSTAMP(1);
            asm volatile("s_nop 1");
            // the single slot that closes every block of a paired or block layout is padding or the y row in EVERY DPP row:
            // u = 0 there, so the column contributes nothing to T u~ (its own row of T is still needed: it is updated below)
            constexpr int NCM = NCL;
            static_for<0, NCM>([&](auto Cc) {
                constexpr int c = decltype(Cc)::value;
                MatVec2<RPL, c / RPL>::run(T[c], rA, rB, tA[c % RPL], tB[c % RPL]);
            });
            PIORAN_SSTAMP(2);
            if constexpr (CBR >= 2) {
#pragma unroll
                for (int i = 0; i < RPL; ++i) { rA[i] += lane_fetch(rA[i], p1); rB[i] += lane_fetch(rB[i], p1); }
            }
            if constexpr (CBR >= 4) {
#pragma unroll
                for (int i = 0; i < RPL; ++i) { rA[i] += lane_fetch(rA[i], p2); rB[i] += lane_fetch(rB[i], p2); }
            }
            if constexpr (WS == 2) {   // the other wavefront's four column blocks
                double* xo = sh_x + (xpar * 2 + half) * (2 * RPL * 64);
                const double* xi = sh_x + (xpar * 2 + (half ^ 1)) * (2 * RPL * 64);
#pragma unroll
                for (int i = 0; i < RPL; ++i) { xo[(2 * i) * 64 + lane] = rA[i]; xo[(2 * i + 1) * 64 + lane] = rB[i]; }
                __syncthreads();           // (also: the records copied during the previous pair are complete and visible)
#pragma unroll
                for (int i = 0; i < RPL; ++i) { rA[i] += xi[(2 * i) * 64 + px]; rB[i] += xi[(2 * i + 1) * 64 + px]; }
                xpar ^= 1;
                // this wavefront's record of the pair after next: its slot was last read before this barrier
                ring_dma(ic<(SL + 2) % 3>{}, nA + 4 + half);
            }
            PIORAN_SSTAMP(3);
            double spA = 0.0, spB = 0.0;
#pragma unroll
            for (int i = 0; i < RPL; ++i) { spA += tA[i] * rA[i]; spB += tB[i] * rB[i]; }
            if constexpr (WS == 2) {   // u_B and phi_A phi_B again from the ring: ten registers each that need not live through the pass over T
#pragma unroll
                for (int i = 0; i < RPL; ++i) {
                    asm volatile("" : "+v"(tA[i]), "+v"(tB[i]));      // (ends the first definitions' live ranges here)
                    const double ali = sh_coef[(2 * i) * CTH + cix], bei = sh_coef[(2 * i + 1) * CTH + cix];
                    uB[i] = ali * B_v(i) + bei * B_x(i);
                    pAB[i] = A_ph(i) * B_ph(i);
                }
            }
            const double sA = group_sum<CBR, NSRC, GS>(spA, contributes, one, p1, p2);
            const double sB = group_sum<CBR, NSRC, GS>(spB, contributes, one, p1, p2);
            PIORAN_SSTAMP(4);
            const double DA = VOIDA ? 1.0 : fma(nu, sa.s2, suma) - sA;                  // :92
            const double rDA = recip_f64(DA);
            double hA[RPL], mB[RPL], wA[RPL], wB[RPL], spg = 0.0;
            double zA = 0.0;
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const double mA = VOIDA ? 0.0 : fma(-A_ph(i), rA[i], A_v(i));          // v - q      :89
                if (!YC && i == YS) zA = mA;
                hA[i] = B_ph(i) * mA;
                spg = fma(hA[i], uB[i], spg);
            }
            if constexpr (YC) zA = VOIDA ? 0.0 : (sa.y - mu) - ryA;                      // z_A = y_A - mu - f'u~A   (phi_y = 1)
            [[maybe_unused]] const double ymB = sb.y - mu;
            PIORAN_SSTAMP(5);
            load_step(nA + 2, sa);                                                       // step A's record is consumed
            const double g = group_sum<CBR, NSRC, GS>(spg, contributes, one, p1, p2);
            PIORAN_SSTAMP(6);
            const double gr = g * rDA;
            const double DB = fma(nu, sb.s2, suma) - sB - g * gr;
            const double rDB = recip_f64(DB);
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                mB[i] = fma(-hA[i], gr, fma(-pAB[i], rB[i], B_v(i)));
                wA[i] = hA[i] * rDA;
                wB[i] = mB[i] * rDB;
            }
            load_step(nA + 3, sb);
            double zB;                                                                   // y row: z_n = y_n - u'f      :141
            if constexpr (YC) {
                zB = ymB - ryB - zA * gr;                                                // the y row's (h_A)_y = z_A
#pragma unroll
                for (int i = 0; i < RPL; ++i) {
                    if constexpr (WS == 2) sh_fy[i * 128 + threadIdx.x] = fma(zB, wB[i], fma(zA, wA[i], pAB[i] * sh_fy[i * 128 + threadIdx.x]));
                    else fy[i] = fma(zB, wB[i], fma(zA, wA[i], pAB[i] * fy[i]));
                }
            } else {
                zB = mB[YS];
            }
            nonpd |= !(DA > 0.0) | !(DB > 0.0);
            Pm *= (FIRST ? DA : fabs(DA)) * ((VOIDA && FIRST) ? DB : fabs(DB));          // log(D[1]) :126, log(abs(D[n])) :140
            {
                int ex;
                Pm = frexp(Pm, &ex);
                Pe += ex;
            }
            quad = fma(zB * zB, rDB, fma(zA * zA, rDA, quad));                           // z_n^2 / D_n  (== y'K^-1 y, :333)
            // ---- T <- (phAB phAB') o T + hA wA' + mB wB' ----
#pragma unroll
            for (int i = 0; i < RPL; ++i) asm volatile("" : "+v"(wA[i]), "+v"(wB[i]), "+v"(pAB[i]));
            PIORAN_SSTAMP(7);
            asm volatile("s_nop 1");
            if constexpr (NPB > 0) {
                static_for<0, NPB>([&](auto Pc) {
                    constexpr int c = 2 * decltype(Pc)::value;
                    double pp[RPL];
                    Update2First<RPL, c / RPL>::run(T[c], hA, mB, pAB, wA[c % RPL], wB[c % RPL], pAB[c % RPL], pp);
                    Update2Second<RPL, (c + 1) / RPL>::run(T[c + 1], hA, mB, wA[(c + 1) % RPL], wB[(c + 1) % RPL], pp);
                });
                static_for<2 * NPB, NCL>([&](auto Cc) {   // (the spare slot's column is never read: not updated, not kept)
                    constexpr int c = decltype(Cc)::value;
                    Update2<RPL, c / RPL>::run(T[c], hA, mB, pAB, wA[c % RPL], wB[c % RPL], pAB[c % RPL]);
                });
            } else if constexpr (PAIRED) {
                static_for<0, NCL / 2>([&](auto Pc) {
                    constexpr int c = 2 * decltype(Pc)::value;
                    double pp[RPL];
                    Update2First<RPL, c / RPL>::run(T[c], hA, mB, pAB, wA[c % RPL], wB[c % RPL], pAB[c % RPL], pp);
                    Update2Second<RPL, (c + 1) / RPL>::run(T[c + 1], hA, mB, wA[(c + 1) % RPL], wB[(c + 1) % RPL], pp);
                });
                // the closing column of an odd block multiplies u = 0 in every DPP row (NCM above): nothing ever reads it, so it is
                // neither updated nor kept in registers (round 3)
            } else {
                static_for<0, NCL>([&](auto Cc) {
                    constexpr int c = decltype(Cc)::value;
                    Update2<RPL, c / RPL>::run(T[c], hA, mB, pAB, wA[c % RPL], wB[c % RPL], pAB[c % RPL]);
                });
            }
        };
        int64_t n;
        if constexpr (WS == 2) {   // the first two pairs' records, published by a barrier before the first read
            const int64_t nA0 = (N & 1) ? -1 : 0;
            ring_dma(ic<0>{}, nA0 + half);
            ring_dma(ic<1>{}, nA0 + 2 + half);
            __syncthreads();
        }
        if (N & 1) {   // (void, step 0), then (1, 2), (3, 4), ...
            load_step(0, sb);
#pragma unroll
            for (int i = 0; i < RPL; ++i) { sa.v[i] = 0.0; sa.x[i] = 0.0; sa.ph[i] = 1.0; }
            sa.y = mu;
            sa.s2 = 0.0;
            pair(-1, std::true_type{}, std::true_type{}, ic<0>{});
            n = 1;
        } else {       // (0, 1), (2, 3), ...
            load_step(0, sa);
            load_step(1, sb);
            pair(0, std::false_type{}, std::true_type{}, ic<0>{});
            n = 2;
        }
        if constexpr (WS == 2) {   // ring slots 1, 2, 0, 1, 2, 0, ... as compile-time constants
            for (; n + 5 < N; n += 6) {
                pair(n, std::false_type{}, std::false_type{}, ic<1>{});
                pair(n + 2, std::false_type{}, std::false_type{}, ic<2>{});
                pair(n + 4, std::false_type{}, std::false_type{}, ic<0>{});
            }
            if (n + 1 < N) {
                pair(n, std::false_type{}, std::false_type{}, ic<1>{});
                n += 2;
                if (n + 1 < N) { pair(n, std::false_type{}, std::false_type{}, ic<2>{}); n += 2; }
            }
        } else {
            for (; n + 1 < N; n += 2) pair(n, std::false_type{}, std::false_type{}, ic<0>{});
        }
        PIORAN_SSTAMP(0);
        PIORAN_SSTAMP_FLUSH
        if (active && (YC ? ((lane % G) == 0 && half == 0) : (isy && r == 0))) {
            const double logdet = log(Pm) + (double)Pe * 0.6931471805599453094;
            const double res = -0.5 * logdet - 0.5 * (double)N * 1.8378770664093454836 - 0.5 * quad;
            p.out[b] = res;
            if (p.status) p.status[b] = !isfinite(res) ? 2 : (nonpd ? 1 : 0);
        }
        return;
    }

    StepIn<RPL> bufA, bufB;
    load_step(0, bufA);

    // ---- first row, :27-42 and :126-128 ----
    double S[NC][RPL];   // [column][own row]
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < RPL; ++i) S[c][i] = 0.0;
    double w[RPL];
    double Dn = fma(nu, bufA.s2, suma);
    double one = 1.0;
    asm volatile("" : "+v"(one));   // keep 1.0 in a VGPR pair (DPP multiplier operand of the row sums)
    double rD = recip_f64(Dn);
    const double ysel = isy ? 1.0 : 0.0;     // the y row's table entry is v = 0: v_y = ysel (y_n - mu) + v, no selects
    bufA.v[YS] = fma(ysel, bufA.y - mu, bufA.v[YS]);   // z_1 = y_1      :128
    double num[RPL];                          // (v - q) of the last step = D_n W_n, the `dn` of :73 up to one rounding
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        num[i] = bufA.v[i];
        w[i] = bufA.v[i] * rD;
    }
    double Pm = Dn;      // running product of |D| (sign of D_1 kept: log of a negative D_1 is NaN, :126)
    int Pe = 0;
    {
        int ex;
        Pm = frexp(Pm, &ex);
        Pe += ex;
    }
    double quad = bufA.v[YS] * bufA.v[YS] * rD;   // meaningful in the y-row lanes only
    bool nonpd = !(Dn > 0.0);

    // one time step: consumes `in` (loaded one step earlier), prefetches step n + 1 into `nxt`
    auto do_step = [&](int64_t n, StepIn<RPL>& in, StepIn<RPL>& nxt, auto renorm) {
        load_step(n + 1, nxt);   // independent of the recurrence
        double u[RPL], g[RPL], qt[RPL];
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            u[i] = al[i] * in.v[i] + be[i] * in.x[i];
            g[i] = num[i];                          // dn = D[n-1] * V[j,n-1]   :73
            qt[i] = 0.0;
        }
        in.v[YS] = fma(ysel, in.y - mu, in.v[YS]);

        // ---- S update + q = S u over this DPP row's column block ----
#pragma unroll
        for (int i = 0; i < RPL; ++i) asm volatile("" : "+v"(w[i]), "+v"(u[i]), "+v"(in.ph[i]));   // DPP sources: final before the pass
        asm volatile("s_nop 1");
        if constexpr (NPB > 0) {
            // block layout: the first 2 NPB columns of the block are the (cos, sin) pairs of NPB two-row terms
            // (phi_i phi_k formed once per pair), the rest single rows of one-row terms and the spare slot
            static_for<0, NPB>([&](auto Pc) {
                constexpr int c = 2 * decltype(Pc)::value;
                double pp[RPL];
                PairFirst<RPL, c / RPL>::run(S[c], qt, g, in.ph, w[c % RPL], u[c % RPL], in.ph[c % RPL], pp);
                PairSecond<RPL, (c + 1) / RPL>::run(S[c + 1], qt, g, w[(c + 1) % RPL], u[(c + 1) % RPL], pp);
            });
            static_for<2 * NPB, NCL>([&](auto Cc) {   // (the spare slot's column multiplies u = 0: never read, not kept)
                constexpr int c = decltype(Cc)::value;
                ColBlock<RPL, c / RPL>::run(S[c], qt, g, in.ph, w[c % RPL], u[c % RPL], in.ph[c % RPL]);
            });
        } else if constexpr (PAIRED) {
            // column pairs (c, c+1), c even: same phi_k, so phi_i * phi_k is formed once per pair
            static_for<0, NCL / 2>([&](auto Pc) {
                constexpr int c = 2 * decltype(Pc)::value;
                double pp[RPL];
                PairFirst<RPL, c / RPL>::run(S[c], qt, g, in.ph, w[c % RPL], u[c % RPL], in.ph[c % RPL], pp);
                PairSecond<RPL, (c + 1) / RPL>::run(S[c + 1], qt, g, w[(c + 1) % RPL], u[(c + 1) % RPL], pp);
            });
            // (the single slot at the end of an odd block — padding or the y row — multiplies u = 0: its column is never read)
        } else {
            static_for<0, NCL>([&](auto Cc) {
                constexpr int c = decltype(Cc)::value;
                ColBlock<RPL, c / RPL>::run(S[c], qt, g, in.ph, w[c % RPL], u[c % RPL], in.ph[c % RPL]);
            });
        }
        if constexpr (CBR >= 2) {
#pragma unroll
            for (int i = 0; i < RPL; ++i) qt[i] += lane_fetch(qt[i], p1);
        }
        if constexpr (CBR >= 4) {
#pragma unroll
            for (int i = 0; i < RPL; ++i) qt[i] += lane_fetch(qt[i], p2);
        }
        double sp = 0.0;
#pragma unroll
        for (int i = 0; i < RPL; ++i) sp += u[i] * qt[i];        // u'Su                       :83,88
        const double s = group_sum<CBR, NSRC>(sp, contributes, one, p1, p2);

        Dn = fma(nu, in.s2, suma) - s;                           // :92  (nu = 1 without a per-draw scale: exact)
        rD = recip_f64(Dn);
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            num[i] = in.v[i] - qt[i];                            // :89
            w[i] = num[i] * rD;                                  // :96
        }
        const double z = num[YS];                                // y row: z_n = y_n - u'f      :141
        nonpd |= !(Dn > 0.0);
        Pm *= fabs(Dn);                                          // log(abs(D[n]))  :140
        if constexpr (decltype(renorm)::value) {                 // mantissa/exponent split every second step
            int ex;
            Pm = frexp(Pm, &ex);
            Pe += ex;
        }
        quad = fma(z * z, rD, quad);                             // z_n^2 / D_n  (== y'K^-1 y, :333)
    };

    if (N > 1) load_step(1, bufB);
    int64_t n = 1;
    for (; n + 1 < N; n += 2) {      // ping-pong: no register rotation between steps
        do_step(n, bufB, bufA, std::false_type{});
        do_step(n + 1, bufA, bufB, std::true_type{});
    }
    if (n < N) do_step(n, bufB, bufA, std::true_type{});

    if (active && isy && r == 0) {
        const double logdet = log(Pm) + (double)Pe * 0.6931471805599453094;
        const double res = -0.5 * logdet - 0.5 * (double)N * 1.8378770664093454836 - 0.5 * quad;
        p.out[b] = res;
        if (p.status) p.status[b] = !isfinite(res) ? 2 : (nonpd ? 1 : 0);
    }
}

using LaunchFn = void (*)(const ScanParams&, dim3, hipStream_t);
// Every launch of this file goes through SCAN_LAUNCH: with g_occ_query set (pioran_scan_pass_draws) the SAME selection logic asks the
// runtime how many wavefronts of the selected instantiation a CU holds instead of launching it.
thread_local int* g_occ_query = nullptr;
#define SCAN_LAUNCH(KERN, grid, threads, st, p)                                                         \
    do {                                                                                               \
        if (g_occ_query) {                                                                             \
            int nb_ = 0;                                                                               \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_, KERN, threads, 0) != hipSuccess) nb_ = 0; \
            *g_occ_query = nb_ * ((threads) / 64);                                                     \
        } else {                                                                                       \
            hipLaunchKernelGGL(KERN, grid, dim3(threads), 0, st, p);                                   \
        }                                                                                              \
    } while (0)
thread_local bool g_last_win3 = false;   // the calling thread's last launch used the three-step form (diagnostics)

// which form of the row sums (group_sum's GS): the context option "gsum" (0, 1, 2) when set, else automatic
int gsum_mode(const ScanParams& p, int cbr)
{
    if (p.opt && p.opt->gsum >= 0) return p.opt->gsum > 2 ? 2 : p.opt->gsum;
    (void)cbr;
    return 0;
}

// GSV: the configuration also exists with the row sums of group_sum's GS = 1 / 2 (fewer exchange rounds); picked per launch
// (gsum_mode below); two-step form, shared table without per-draw rows only
template <int RPL, int CBR, int NSRC, int MINW = 1, bool PAIRED = false, bool GSV = false>
void launch_cfg(const ScanParams& p, dim3 grid, hipStream_t st)
{
    if constexpr (RPL == 2 || RPL == 3) {
        // three-step form: measured faster where its row vectors fit the register file — two rows per lane from 13 source lanes on
        // (R = 24 .. 31: 2 .. 6 % at B = 4096, tools/sweep_win3.py); with three rows per lane it spills (SHO-20: 18 % slower) and with
        // one the extra row sums outweigh the saved scalings (10 .. 30 % slower): those run it only when the option asks
        const bool auto3 = RPL == 2 && CBR == 1 && NSRC >= 13 && !(p.opt && (p.opt->no_win3 || p.opt->no_win2 || p.opt->win2)) && p.B > 2048;
        if (((p.opt && p.opt->win3) || auto3) && p.tab && p.npd_rows == 0) {
            SCAN_LAUNCH((celerite_scan_kernel<RPL, CBR, NSRC, true, 2, PAIRED, false, 0, false, 0, false, true>), grid, 256, st, p);
            g_last_win3 = true;
            return;
        }
    }
    if constexpr (GSV) {
        const int gs = gsum_mode(p, CBR);
        if (gs && p.tab && p.npd_rows == 0 && !(p.opt && p.opt->no_win2)) {
            if (gs == 2) SCAN_LAUNCH((celerite_scan_kernel<RPL, CBR, NSRC, true, MINW, PAIRED, false, 0, true, 2>), grid, 256, st, p);
            else SCAN_LAUNCH((celerite_scan_kernel<RPL, CBR, NSRC, true, MINW, PAIRED, false, 0, true, CBR == 4 ? 1 : 2>), grid, 256, st, p);
            return;
        }
    }
    // two-step form: faster wherever its working set fits the register file (tools/sweep_win2.py: RPL <= 3: +10 .. 24 %;
    // RPL = 4: +4 .. 6 % with a few spilled registers); with RPL = 5 it loses 7 %, so there it runs only when forced
    // (context option "win2")
    if ((RPL <= 4 && !(p.opt && p.opt->no_win2)) || (p.opt && p.opt->win2)) {
        if (p.tab && p.npd_rows > 0)
            SCAN_LAUNCH((celerite_scan_kernel<RPL, CBR, NSRC, true, MINW, PAIRED, true, 0, true>), grid, 256, st, p);
        else if (p.tab)
            SCAN_LAUNCH((celerite_scan_kernel<RPL, CBR, NSRC, true, MINW, PAIRED, false, 0, true>), grid, 256, st, p);
        else
            SCAN_LAUNCH((celerite_scan_kernel<RPL, CBR, NSRC, false, MINW, PAIRED, false, 0, true>), grid, 256, st, p);
        return;
    }
    if (p.tab && p.npd_rows > 0)
        SCAN_LAUNCH((celerite_scan_kernel<RPL, CBR, NSRC, true, MINW, PAIRED, true>), grid, 256, st, p);
    else if (p.tab)
        SCAN_LAUNCH((celerite_scan_kernel<RPL, CBR, NSRC, true, MINW, PAIRED>), grid, 256, st, p);
    else
        SCAN_LAUNCH((celerite_scan_kernel<RPL, CBR, NSRC, false, MINW, PAIRED>), grid, 256, st, p);
}

// block layout (NPB pairs per block): shared-table launches without per-draw rows only
template <int RPL, int CBR, int NSRC, int MINW, int NPB>
void launch_blocked(const ScanParams& p, dim3 grid, hipStream_t st)
{
    const int gs = gsum_mode(p, CBR);
    if (gs && !(p.opt && p.opt->no_win2)) {
        if (gs == 2) SCAN_LAUNCH((celerite_scan_kernel<RPL, CBR, NSRC, true, MINW, false, false, NPB, true, 2>), grid, 256, st, p);
        else SCAN_LAUNCH((celerite_scan_kernel<RPL, CBR, NSRC, true, MINW, false, false, NPB, true, 1>), grid, 256, st, p);
        return;
    }
    if ((RPL <= 4 && !(p.opt && p.opt->no_win2)) || (p.opt && p.opt->win2))
        SCAN_LAUNCH((celerite_scan_kernel<RPL, CBR, NSRC, true, MINW, false, false, NPB, true>), grid, 256, st, p);
    else
        SCAN_LAUNCH((celerite_scan_kernel<RPL, CBR, NSRC, true, MINW, false, false, NPB>), grid, 256, st, p);
}

struct ScanConfig {
    const char* name;
    int rpl, cbr, nsrc;
    LaunchFn fn;
    bool paired = false;   // needs the standard row map (every term has both rows)
    bool autopick = true;  // false: only selectable by name (measured slower than the default of its row range)
    int npb = 0;           // > 0: block layout for "n_complex = npb * cbr two-row terms, then one-row terms" (standard_rows == 2)
    bool ycol = false;     // y rides as a separate vector (kernel template YC): every slot is a row; two-step form, no per-draw rows
    int ws = 1;            // wavefronts per draw (2: the draw's column blocks over a pair of wavefronts, kernel template WS)
    // real rows it holds: one slot carries y (unless ycol); a paired config with an odd block keeps one single slot per block
    int capacity() const
    {
        const int rb = rpl * nsrc;
        if (ycol) return cbr * ws * (paired ? (rb & ~1) : rb);
        return paired ? cbr * (rb & ~1) - ((rb & 1) ? 0 : 1) : rpl * cbr * nsrc - 1;
    }
};

template <int RPL, int CBR, int NSRC, int MINW, bool PAIRED>
void launch_ycol(const ScanParams& p, dim3 grid, hipStream_t st)
{
    SCAN_LAUNCH((celerite_scan_kernel<RPL, CBR, NSRC, true, MINW, PAIRED, false, 0, true, 0, true>), grid, 256, st, p);
}

#ifdef PIORAN_EXPERIMENTS
template <int RPL, int NSRC, bool PAIRED>
void launch_ycol_w2(const ScanParams& p, dim3 grid, hipStream_t st)
{
    SCAN_LAUNCH((celerite_scan_kernel<RPL, 4, NSRC, true, 2, PAIRED, false, 0, true, 2, true, false, 2>), grid, 128, st, p);
}
#endif

#define CFG(RPL, CBR, NSRC) {"rpl" #RPL "_cbr" #CBR "_nsrc" #NSRC, RPL, CBR, NSRC, &launch_cfg<RPL, CBR, NSRC>}
#define CFG_P(RPL, CBR, NSRC, MINW) {"rpl" #RPL "_cbr" #CBR "_nsrc" #NSRC "_p", RPL, CBR, NSRC, &launch_cfg<RPL, CBR, NSRC, MINW, true>, true}
// preference order: first entry whose capacity >= R wins (unless the context option "scan_config" names another).
// Every entry uses the DPP-folded column blocks (ColBlock / PairFirst / PairSecond); the compiler-scheduled builtin
// variants of round 1 were dropped when the DPP-folded ones became the faster choice in every row range
// (SHO-30: 140 k vs 116 k evals/s, SHO-39: 60 k vs 52 k; profiles/r02_scan_variants.txt).
const ScanConfig kConfigs[] = {
    // (the _y entries: y as a separate vector, one more row than the shape otherwise holds — R = 16, 32, 48, 64 are the reference
    //  benchmark's j = 8, 16, (24,) 32, benchmark/benchmarks.jl:16-18; shared-table launches only)
    CFG(1, 1, 5),  CFG(1, 1, 9),  CFG(1, 1, 13), CFG(1, 1, 16),           // R <= 15
    {"rpl1_cbr1_nsrc16_y", 1, 1, 16, &launch_ycol<1, 1, 16, 1, false>, false, true, 0, true},   // R = 16
    CFG(2, 1, 9),  CFG(2, 1, 11), CFG(2, 1, 13), CFG(2, 1, 15), CFG(2, 1, 16),  // R <= 31
    {"rpl2_cbr1_nsrc16_y", 2, 1, 16, &launch_ycol<2, 1, 16, 1, false>, false, true, 0, true},   // R = 32
    CFG(3, 2, 6),  CFG(3, 2, 7),  CFG(3, 2, 8),                           // R <= 47
    {"rpl3_cbr2_nsrc8_y", 3, 2, 8, &launch_ycol<3, 2, 8, 1, false>, false, true, 0, true},      // R = 48
    {"rpl4_cbr4_nsrc4", 4, 4, 4, &launch_cfg<4, 4, 4, 2, false, true>},   // R <= 63: 256 registers/lane, 2 waves per SIMD
    {"rpl4_cbr4_nsrc4_y", 4, 4, 4, &launch_ycol<4, 4, 4, 2, false>, false, true, 0, true},   // R = 64: y as a separate vector
    CFG(5, 4, 4),                                                         // R <= 79
    {"rpl5_cbr4_nsrc4_y", 5, 4, 4, &launch_ycol<5, 4, 4, 1, false>, false, true, 0, true},   // R = 80 (SHO-40: the dense configuration's model)
    // 65 .. 80 rows, shared table: one draw over a PAIR of wavefronts (round 4; kernel template WS = 2): 50 entries of T per lane in the
    // two-step form, no scratch, no AGPR copies — and exactly the speed of the one-wavefront shapes above at 80 rows (54.6 k against 54 k
    // evaluations per second for SHO-40, B = 4096), slower below (the stream is the same for 65 .. 80 rows; SHO-39 runs at 64 k on
    // rpl5_cbr4_nsrc4_p).  589 VALU instructions per wavefront and pair of steps, of which only 280 are the passes over T: the row-vector
    // work of five rows per lane, the five 16-lane row sums and the exchange are paid by BOTH wavefronts (profiles/r04_r80.txt).  Selected
    // by name only (context option scan_config); the measured answer to "split the draw over two wavefronts" — in experiment builds
    // (-DPIORAN_EXPERIMENTS) only since round 5: from 49 rows on large batches run the windowed form on the matrix cores (celerite_tile.hip:
    // 99 k evaluations per second for SHO-40).
#ifdef PIORAN_EXPERIMENTS
    {"rpl5_cbr4_nsrc2_w2_y", 5, 4, 2, &launch_ycol_w2<5, 2, false>, false, false, 0, true, 2},
#endif
    // column-paired variants (standard row map only; picked automatically by pick_config when applicable)
    {"rpl3_cbr2_nsrc7_p", 3, 2, 7, &launch_cfg<3, 2, 7, 1, true, true>, true}, CFG_P(3, 2, 8, 1), CFG_P(3, 2, 6, 1),
    CFG_P(1, 1, 5, 1), CFG_P(1, 1, 9, 1), CFG_P(1, 1, 13, 1), CFG_P(1, 1, 16, 1),
    CFG_P(2, 1, 9, 1), CFG_P(2, 1, 11, 1), CFG_P(2, 1, 13, 1), CFG_P(2, 1, 15, 1), CFG_P(2, 1, 16, 1),
    {"rpl4_cbr4_nsrc4_p", 4, 4, 4, &launch_cfg<4, 4, 4, 2, true, true>, true},
    {"rpl4_cbr4_nsrc4_yp", 4, 4, 4, &launch_ycol<4, 4, 4, 2, true>, true, true, 0, true}, CFG_P(5, 4, 4, 1),
    {"rpl1_cbr1_nsrc16_yp", 1, 1, 16, &launch_ycol<1, 1, 16, 1, true>, true, true, 0, true},
    {"rpl2_cbr1_nsrc16_yp", 2, 1, 16, &launch_ycol<2, 1, 16, 1, true>, true, true, 0, true},
    {"rpl3_cbr2_nsrc8_yp", 3, 2, 8, &launch_ycol<3, 2, 8, 1, true>, true, true, 0, true},
    {"rpl5_cbr4_nsrc4_yp", 5, 4, 4, &launch_ycol<5, 4, 4, 1, true>, true, true, 0, true},
#ifdef PIORAN_EXPERIMENTS
    {"rpl5_cbr4_nsrc2_w2_yp", 5, 4, 2, &launch_ycol_w2<5, 2, true>, true, false, 0, true, 2},
#endif
    // DRWCelerite with 20 components: 20 complex + 20 real terms = 5 pairs + 5 singles + 1 spare in each of the 4 blocks
    {"rpl4_cbr4_nsrc4_b5a", 4, 4, 4, &launch_blocked<4, 4, 4, 2, 5>, false, true, 5},
    // alternatives kept for tuning runs (selected by name)
    {"rpl3_cbr4_nsrc4", 3, 4, 4, &launch_cfg<3, 4, 4, 1, false, true>}, CFG(2, 2, 8),
};
#undef CFG
#undef CFG_P
#ifdef PIORAN_EXPERIMENTS
constexpr int kNumPreferred = 20;
#else
constexpr int kNumPreferred = 19;
#endif

// y-as-a-vector shapes exist in the two-step form for launches without per-draw rows only
bool ycol_usable(const ScanConfig& c, const ScanParams* p)
{
    if (!c.ycol) return true;
    // (per-draw (c, d) launches evaluate the transcendentals in the kernel: at four rows per lane the two-step form then spills —
    // 19.5 k instead of 42.8 k evaluations per second at j = 32 — so those keep the 80-row step-by-step shape)
    return p && p->tab && p->npd_rows == 0 && !(p->opt && p->opt->no_win2);
}

// does the block layout of `c` hold this row structure?  (shared table, no per-draw rows)
bool blocked_fits(const ScanConfig& c, const ScanParams& p)
{
    if (c.npb == 0) return true;
    const int rb = c.rpl * c.nsrc, nsb = rb - 1 - 2 * c.npb;
    return p.standard_rows == 2 && p.tab && p.npd_rows == 0 && p.n_complex == c.npb * c.cbr &&
           p.R - 2 * p.n_complex <= nsb * c.cbr;
}

const ScanConfig* pick_config(int R, bool standard_rows, const ScanParams* p = nullptr)
{
    const ScanOptions* opt = p ? p->opt : nullptr;
    const bool no_paired = opt && opt->no_paired;
    if (const char* env = (opt && opt->scan_config[0]) ? opt->scan_config : nullptr) {
        for (const auto& c : kConfigs)
            if (!std::strcmp(env, c.name) && c.capacity() >= R && (!c.paired || standard_rows) && (c.npb == 0 || (p && blocked_fits(c, *p))) &&
                ycol_usable(c, p))
                return &c;
    }
    if (p && !no_paired) {
        for (const auto& c : kConfigs)
            if (c.npb > 0 && c.autopick && blocked_fits(c, *p)) return &c;
    }
    const ScanConfig* best = nullptr;
    for (int i = 0; i < kNumPreferred; ++i)
        if (kConfigs[i].autopick && kConfigs[i].capacity() >= R && ycol_usable(kConfigs[i], p)) { best = &kConfigs[i]; break; }
    if (standard_rows && !no_paired) {
        // a paired variant of the same shape (or the smallest paired one that fits) wins when the row map allows it
        for (const auto& c : kConfigs)
            if (c.paired && c.npb == 0 && c.autopick && c.capacity() >= R && ycol_usable(c, p) &&
                (!best || c.rpl * c.cbr * c.nsrc <= best->rpl * best->cbr * best->nsrc))
                return &c;
    }
    return best;
}

}  // namespace

int pioran_scan_supported_rows() { return 79; }
// one more with a shared table and no per-draw rows (y as a separate vector): 80 rows = SHO-40
int pioran_scan_supported_rows_shared() { return 80; }

static thread_local const char* g_last_config = "none";

const char* pioran_scan_config_name(int R)
{
    if (R <= 0) {                                          // the configuration the last launch of this thread ran on
        static thread_local char buf[64];
        std::snprintf(buf, sizeof(buf), "%s%s", g_last_config, g_last_win3 ? "+win3" : "");
        return buf;
    }
    const ScanConfig* c = pick_config(R, (R & 1) == 0);   // diagnostics: assumes the standard row map for even R
    return c ? c->name : "fallback";
}

static const ScanConfig* launch_config(const ScanParams& p);

// Draws that ONE full pass of the throughput layout holds for this launch (every SIMD of the chip with as many wavefronts as the selected
// instantiation's registers and LDS allow); 0 if unknown.  A launch of k such passes plus a remainder runs the remainder as a second,
// mostly empty pass of full-length wavefronts: capi.hip sends small remainders to the windowed kernel on a second stream instead.
static const ScanConfig* launch_config(const ScanParams& p)
{
    const ScanConfig* c = pick_config(p.R, p.standard_rows == 1, &p);
    if (!c) return nullptr;
    // Mid-size batches: up to 2048 draws the four-draws-per-wavefront shapes of this row range put at most one wavefront on
    // half of the chip's 1024 SIMDs, and the launch takes as long as ONE wavefront needs for the series; two draws per
    // wavefront (rpl2_cbr2_nsrc8: half the columns per lane) is then the faster walk (tools/sweep_midbatch.py, N = 1e4,
    // R = 30: 4.2 instead of 5.3 ms for 768 .. 2048 draws; above 2048 draws the denser shape wins again: 5.3 vs 7.3 ms)
    if (!(p.opt && p.opt->scan_config[0]) && p.B <= 2048 && p.R >= 24 && p.R <= 31 && c->cbr == 1) {
        for (const auto& alt : kConfigs)
            if (!std::strcmp(alt.name, "rpl2_cbr2_nsrc8")) c = &alt;
    }
    return c;
}

int64_t pioran_scan_pass_draws(const ScanParams& p, int* waves_per_simd)
{
    if (waves_per_simd) *waves_per_simd = 0;
    const ScanConfig* c = launch_config(p);
    if (!c) return 0;
    int waves_per_cu = 0, ncu = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    // The answer depends on the device and on which instantiation the selection logic picks (configuration, batch class, options): asked
    // once per such key and thread (the dispatchers of capi.hip call this on every large launch).
    struct Key { const ScanConfig* c; int dev; unsigned flags; int waves_per_cu, ncu; };
    const unsigned flags = (p.B > 2048 ? 1u : 0u) | (p.tab ? 2u : 0u) | (p.npd_rows > 0 ? 4u : 0u) |
                           (p.opt ? ((p.opt->win3 ? 8u : 0u) | (p.opt->no_win3 ? 16u : 0u) | (p.opt->win2 ? 32u : 0u) | (p.opt->no_win2 ? 64u : 0u) |
                                     ((unsigned)(p.opt->gsum + 1) << 8)) : 0u);
    static thread_local std::vector<Key> cache;
    bool hit = false;
    for (const Key& k : cache)
        if (k.c == c && k.dev == dev && k.flags == flags) { waves_per_cu = k.waves_per_cu; ncu = k.ncu; hit = true; break; }
    if (!hit) {
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        const bool keep_win3 = g_last_win3;      // a query is not a launch: the diagnostics keep describing the last launch
        g_occ_query = &waves_per_cu;
        c->fn(p, dim3(1), nullptr);
        g_occ_query = nullptr;
        g_last_win3 = keep_win3;
        (void)hipGetLastError();
        if (waves_per_cu > 0 && ncu > 0) cache.push_back(Key{c, dev, flags, waves_per_cu, ncu});
    }
    const int epw = 64 / (16 * c->cbr);
    if (waves_per_cu <= 0 || ncu <= 0) return 0;
    if (waves_per_simd) *waves_per_simd = waves_per_cu / 4;
    return c->ws == 2 ? (int64_t)ncu * (waves_per_cu / 2) : (int64_t)ncu * waves_per_cu * epw;
}

int pioran_launch_scan(const ScanParams& p, hipStream_t stream)
{
    const ScanConfig* c = launch_config(p);
    if (!c) return PIORAN_ERR_UNSUPPORTED;
    const int epw = 64 / (16 * c->cbr);
    const int64_t per_block = c->ws == 2 ? 1 : 4 * epw;
    const int64_t blocks = (p.B + per_block - 1) / per_block;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return PIORAN_ERR_ARG;
    g_last_config = c->name;
    g_last_win3 = false;
    c->fn(p, dim3((unsigned)blocks), stream);
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}
