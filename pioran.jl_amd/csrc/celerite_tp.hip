// Time-parallel evaluation of the celerite log-likelihood for a HANDFUL of draws (round 5; the reference's own benchmark is one evaluation at a
// time: benchmark/benchmarks.jl:76-91, BASELINE configs[1]).  The step-by-step and windowed kernels run one draw as a serial chain of N steps on one
// CU; this family cuts the series into segments that run on different CUs.
//
// Form (numpy prototype with the derivation and the accuracy study: tools/time_parallel_proto.py, profiles/r05_time_parallel_proto.txt).  A celerite
// kernel is the covariance of a linear-Gaussian state-space model — two state components per term with transition F = e^{-c dt} Rot(d dt), observation
// h = (1, 0), stationary covariance P_inf = [[a, -b], [-b, a]]; one component for a term with b = d = 0 — and the recurrence of
// src/celerite_solver.jl:12-100 (init_semi_separable!) with the forward half of solve_prec! (:115-142) is its Kalman filter in the coordinates
// S_n = P_inf - P_n: D_n is the innovation variance, z_n the innovation, log L = -1/2 sum (log |D_n| + z_n^2 / D_n) - N/2 log 2 pi (:312-334).
// The filter parallelises over time with the associative elements of Sarkka & Garcia-Fernandez (IEEE TAC 66 (2021) 299): a = (A, b, C, eta, J),
//     p(x_k | y_k, x_k-1) = N(A x_k-1 + b, C),   p(y_k | x_k-1) ~ N_information(eta, J),
//     a_i (x) a_j:  A = A_j M A_i,  b = A_j M (b_i + C_i eta_j) + b_j,  C = A_j M C_i A_j' + C_j,   M = (I + C_i J_j)^-1,
//                   eta = A_i' M' (eta_j - J_j b_i) + eta_i,  J = A_i' M' J_j A_i + J_i.
// Phases (kernels below):
//   0  tp_records_kernel    per step and state row: the transition, the process noise Q = P_inf - F P_inf F' in closed form (no difference of nearly
//                           equal numbers at small c dt), the gain of the single-step element — everything that does not depend on the state;
//   1  tp_element_kernel    one workgroup per (draw, segment): the segment's element, composed step by step.  A single step's J_j is rank one, so M
//                           is a Sherman-Morrison correction and a composition is rotations of adjacent rows / columns, rank-one updates and
//                           matrix-vector sums: O(R^2) per step (about 30 R^2 flop against the recurrence's 5.5 R^2), two barriers per step;
//   2  tp_boundary_kernel   one workgroup per draw, sequential over the segments: the filtered state (m, P) at every segment boundary, one R x R
//                           solve with partial pivoting and two products each;
//   3  tp_filter_kernel     one workgroup per (draw, segment): the ordinary filter from the boundary state — sum log |D_n| and sum z_n^2 / D_n of the
//                           segment; tp_finish_kernel adds the segments up in a fixed order.
// Layout of a matrix in phases 1 and 3: lane = row (R <= 64 state rows: two-row terms first, then the one-row terms), wavefront w of the four owns
// the column PAIRS 4 s + w; a pair is the two rows of a two-row term or two one-row terms.
#include <hip/hip_runtime.h>
#include <cstdint>

#include "common.h"

int pioran_tp_scan_rows(int RP);
double pioran_tp_scan_tol(const ScanOptions* opt);

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

struct __attribute__((aligned(16))) TpRec { double al, be, g, K, qd, qo; };         // per (draw, step, row): three 16-byte units (al, be), (g, K), (qd, qo)
struct __attribute__((aligned(32))) TpStep { double s, y, s2, yos; };                // per (draw, step): h Q h' + sigma2, y - mu, nu sigma2, y / s

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every global load in flight (s_waitcnt vmcnt(0)): the records and the
// element matrices are loaded a chunk / a boundary AHEAD on purpose, and each barrier made them arrive first — ~2 us per load point (6 us per boundary at
// four rows).  All data that passes between the threads of these kernels passes through LDS.
#define TP_SYNC()                                       \
    do {                                                \
        if constexpr (TW == 1) asm volatile("" ::: "memory"); \
        else TP_BARRIER();                              \
    } while (0)
#define TP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

constexpr int TP_ELEM_DOUBLES = (3 * 64 + 2) * 64;     // A' | C | J (64 x 64 each, row-major, rows = lanes), b, eta
constexpr int TP_BND_DOUBLES = 65 * 64;                // m | P (64 x 64 row-major)
constexpr double kTpScanTol = 1e-3;                    // largest accepted distance, on the scale of the innovation variance, between the scan's boundary states and the sequentially propagated ones (tp_filter_kernel)
// Experiment builds only (-DPIORAN_TP_STAMP, tools/tp_combine_stamps.sh; never in the product library): s_memtime stamps of the phases of one combination
// (the workgroup of target blockIdx.x == gridDim.x - 1, draw 0), read back through pioran_tp_read_stamps.
#ifdef PIORAN_TP_STAMP
__device__ unsigned long long tp_stamp_buf[32];
#define TP_STAMP(i) do { if (blockIdx.x == gridDim.x - 1 && blockIdx.y == 0 && threadIdx.x == 0) tp_stamp_buf[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TP_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ double tp_readlane(double x, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l), hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}

template <int CTRL>
__device__ __forceinline__ double tp_dpp(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// the neighbour lane of a row pair (lane ^ 1): quad_perm [1, 0, 3, 2]
__device__ __forceinline__ double tp_partner(double x) { return tp_dpp<0xB1>(x); }

// sums over the wavefront, two at a time: four DPP rounds inside the rows of 16 lanes, then the four row totals as scalar operands
// (ds_bpermute-based shuffles cost ~700 cycles for the same: the first version of these kernels spent a third of a step there)
template <bool ROW0 = false>      // ROW0: every contributing lane sits in the first row of 16 lanes (up to 16 state rows): the row total is the total
__device__ __forceinline__ void tp_sum2(double& a, double& b)
{
    a += tp_dpp<0xB1>(a);  b += tp_dpp<0xB1>(b);
    a += tp_dpp<0x4E>(a);  b += tp_dpp<0x4E>(b);
    a += tp_dpp<0x141>(a); b += tp_dpp<0x141>(b);
    a += tp_dpp<0x140>(a); b += tp_dpp<0x140>(b);
    if constexpr (ROW0) {
        a = tp_readlane(a, 0);
        b = tp_readlane(b, 0);
    } else {
        a = (tp_readlane(a, 0) + tp_readlane(a, 16)) + (tp_readlane(a, 32) + tp_readlane(a, 48));
        b = (tp_readlane(b, 0) + tp_readlane(b, 16)) + (tp_readlane(b, 32) + tp_readlane(b, 48));
    }
}

// 1 / x to fp64 accuracy: v_rcp_f64 and two Newton steps (as in celerite_scan.hip)
__device__ __forceinline__ double tp_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// max over the wavefront of non-negative 32-bit keys (the high words of |x|: monotonic for non-negative doubles), DPP folded into v_max_u32
__device__ __forceinline__ unsigned tp_max_u32(unsigned a)
{
    a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0xB1, 0xf, 0xf, true));
    a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x4E, 0xf, 0xf, true));
    a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x141, 0xf, 0xf, true));
    a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x140, 0xf, 0xf, true));
    const unsigned r0 = __builtin_amdgcn_readlane((int)a, 0), r1 = __builtin_amdgcn_readlane((int)a, 16), r2 = __builtin_amdgcn_readlane((int)a, 32),
                   r3 = __builtin_amdgcn_readlane((int)a, 48);
    return max(max(r0, r1), max(r2, r3));
}

__device__ __forceinline__ double tp_max(double a)
{
    a = fmax(a, tp_dpp<0xB1>(a));
    a = fmax(a, tp_dpp<0x4E>(a));
    a = fmax(a, tp_dpp<0x141>(a));
    a = fmax(a, tp_dpp<0x140>(a));
    return fmax(fmax(tp_readlane(a, 0), tp_readlane(a, 16)), fmax(tp_readlane(a, 32), tp_readlane(a, 48)));
}

// ---- phase 0 ---------------------------------------------------------------------------------------------------------------------------------
// kind: 0 / 1 first / second row of a two-row term, 2 one-row term, 3 padding.  Step 0 is the prior as "filtered state before the first step":
// dt = 0 gives F = I and Q = 0 by the same formulas.
__global__ void __launch_bounds__(64) tp_records_kernel(int64_t N, int RP, int J, const int32_t* __restrict__ row_term, const int32_t* __restrict__ row_kind,
                                                        const double* __restrict__ t, const double* __restrict__ y, const double* __restrict__ s2,
                                                        const double* __restrict__ Y, const double* __restrict__ S2, const double* __restrict__ A,
                                                        const double* __restrict__ Bc, const double* __restrict__ C, const double* __restrict__ D,
                                                        const double* __restrict__ mu, const double* __restrict__ nu, TpRec* __restrict__ rec,
                                                        TpStep* __restrict__ stp)
{
    const int64_t n = blockIdx.x, b = blockIdx.y;
    const int r = threadIdx.x;
    const double dt = n > 0 ? t[n] - t[n - 1] : 0.0;
    TpRec o{0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    double hq = 0.0, qh = 0.0;
    if (r < RP) {
        const int kind = row_kind[r];
        if (kind != 3) {
            const int j = row_term[r];
            const double c = C[j], d = kind == 2 ? 0.0 : D[j], a = A[b * J + j], bb = kind == 2 ? 0.0 : Bc[b * J + j];
            const double e = exp(-c * dt);
            double sn, cs;
            sincos(d * dt, &sn, &cs);
            const double si = e * sn;
            o.al = e * cs;
            o.be = kind == 0 ? -si : (kind == 1 ? si : 0.0);
            const double gam = -expm1(-2.0 * c * dt);
            o.qd = fma(a, gam, 2.0 * bb * o.al * o.be);
            o.qo = -bb * fma(2.0 * o.be, o.be, gam);
            o.g = kind == 1 ? -o.be : o.al;
            qh = kind == 1 ? o.qo : o.qd;
            hq = kind == 1 ? 0.0 : o.qd;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) hq += __shfl_xor(hq, off);
    const double s2v = (S2 ? S2[b * N + n] : s2[n]) * (nu ? nu[b] : 1.0);
    const double yv = (Y ? Y[b * N + n] : y[n]) - (mu ? mu[b] : 0.0);
    const double s = hq + s2v;
    o.K = qh / s;
    if (r < RP) rec[(b * N + n) * RP + r] = o;
    if (r == 0) stp[b * N + n] = TpStep{s, yv, s2v, yv / s};
}

// The records reach a workgroup a CHUNK of steps ahead: while chunk c runs from one LDS buffer, the threads hold chunk c + 1 in registers (loaded a
// chunk earlier) and copy it to the other buffer in the chunk's last step, then issue the loads of chunk c + 2.  (Records loaded where they were used
// cost 1.7 us per step; two steps ahead still left 0.75 us per step at four state rows: one cold HBM line per step, 1.5 us away.)
template <int NWV>
struct TpStage {
    static constexpr int CH = NWV == 1 ? 16 : 8, RMAX = NWV == 1 ? 16 : 64, T = 64 * NWV, NU = CH * RMAX * 3 / T;
    double2 r[NU], s;
    __device__ __forceinline__ void load(const TpRec* rec0, const TpStep* stp0, int steps, int RP, int t)
    {
        const double2* a = reinterpret_cast<const double2*>(rec0);
        const double2* c = reinterpret_cast<const double2*>(stp0);
        const int units = steps * RP * 3;
#pragma unroll
        for (int i = 0; i < NU; ++i) r[i] = t + T * i < units ? a[t + T * i] : double2{0.0, 0.0};
        s = t < 2 * steps ? c[t] : double2{0.0, 0.0};
    }
    __device__ __forceinline__ void store(double2* rdst, double2* sdst, int RP, int t) const
    {
#pragma unroll
        for (int i = 0; i < NU; ++i)
            if (t + T * i < CH * RP * 3) rdst[t + T * i] = r[i];
        if (t < 2 * CH) sdst[t] = s;
    }
};

// ---- phase 3 ---------------------------------------------------------------------------------------------------------------------------------
// The records of a step reach the wavefronts one step ahead: every lane loads its row's record of step n + 2 while step n runs, wavefront 0 copies the
// record of step n + 1 into LDS before the step's barrier, and the column entries (wave-uniform) are LDS broadcast reads — no global load sits on
// the chain of a step (the first version loaded them where it used them: 1.7 us per step).
template <int NP, int NWV>     // NWV wavefronts per workgroup (1: up to 16 rows, no barrier at all; 4: up to 64), NP column pairs per wavefront: RP = 2 NP NWV
__global__ void __launch_bounds__(64 * NWV) tp_filter_kernel(int64_t N, int RP, int nseg, int64_t L, const int32_t* __restrict__ row_kind,
                                                             const TpRec* __restrict__ rec, const TpStep* __restrict__ stp, const double* __restrict__ bnd,
                                                             double* __restrict__ part, double* __restrict__ sval, double* __restrict__ disc, int check)
{
    using Stage = TpStage<NWV>;
    constexpr int CH = Stage::CH;
    __shared__ double red[2][NWV][64];
    __shared__ double2 rbuf[2][CH * Stage::RMAX * 3], sbuf[2][CH * 2];
    __shared__ double sring[256];
    __shared__ double2 uni[NWV][32];       // a lane vector back as wave-uniform column pairs (one 16-byte broadcast read per pair; v_readlane: two per value)          // D_n of the last (up to) 256 steps: their logarithms are taken 256 at a time by all threads, off the chain
    const int lane = threadIdx.x & 63, w = NWV == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int seg = blockIdx.x;
    const int64_t b = blockIdx.y;
    const int64_t n0 = seg * L, n1 = n0 + L < N ? n0 + L : N;
    const int len = (int)(n1 - n0);
    double ldsum = 0.0;
    int bad = 0;
    const double* bs = bnd + (b * nseg + seg) * TP_BND_DOUBLES;
    double m = bs[lane];
    double P[NP][2], hc[NP][2];
    const TpRec* rb = rec + (b * N + n0) * RP;
    const TpStep* sb = stp + b * N + n0;
    const int rl = lane < RP ? lane : RP - 1;        // (lanes past the rows: a live row's record; their state is zero and stays zero)
    const int mykind = row_kind[rl];
    const double hh = (lane < RP && (mykind == 0 || mykind == 2)) ? 1.0 : 0.0;
#pragma unroll
    for (int s = 0; s < NP; ++s) {
        const int c0 = 2 * (NWV * s + w);
        P[s][0] = bs[64 + lane * 64 + c0];
        P[s][1] = bs[64 + lane * 64 + c0 + 1];
        const int k0 = row_kind[c0], k1 = row_kind[c0 + 1];
        hc[s][0] = (k0 == 0 || k0 == 2) ? 1.0 : 0.0;
        hc[s][1] = (k1 == 0 || k1 == 2) ? 1.0 : 0.0;
    }
    const bool odd = lane & 1;
    Stage stg;
    stg.load(rb, sb, len < CH ? len : CH, RP, threadIdx.x);
    stg.store(rbuf[0], sbuf[0], RP, threadIdx.x);
    {
        const int rest = len - CH;
        stg.load(rb + (int64_t)CH * RP, sb + CH, rest < 0 ? 0 : (rest < CH ? rest : CH), RP, threadIdx.x);
    }
    TP_BARRIER();
    double quad = 0.0;
    double Slast = 1.0;                  // the innovation variance of the segment's last step: the scale of the scan's check below
    for (int k = 0; k < len; ++k) {
        const int ci = k / CH, si = k % CH, buf = ci & 1;
        const double2* rr = rbuf[buf] + si * RP * 3;
        const double2 mab = rr[rl * 3], mq = rr[rl * 3 + 2];
        const double2 s0 = sbuf[buf][si * 2];
        const double al = mab.x, be = mab.y;
        m = fma(al, m, be * tp_partner(m));
        double ph = 0.0;
#pragma unroll
        for (int s = 0; s < NP; ++s) {
            const int jj = NWV * s + w, c0 = 2 * jj;
            const double2 ab0 = rr[c0 * 3], ab1 = rr[c0 * 3 + 3];
            const double p0 = P[s][0], p1 = P[s][1];
            const double r0 = fma(al, p0, be * tp_partner(p0)), r1 = fma(al, p1, be * tp_partner(p1));
            double v0 = fma(ab0.x, r0, ab0.y * r1), v1 = fma(ab1.x, r1, ab1.y * r0);
            const bool mine = (lane >> 1) == jj;
            v0 += mine ? (odd ? mq.y : mq.x) : 0.0;
            v1 += mine ? (odd ? mq.x : mq.y) : 0.0;
            P[s][0] = v0;
            P[s][1] = v1;
            ph = fma(v0, hc[s][0], fma(v1, hc[s][1], ph));
        }
        double Ph = ph;
        if (si == CH - 1) {
            stg.store(rbuf[buf ^ 1], sbuf[buf ^ 1], RP, threadIdx.x);
            const int rest = len - (ci + 2) * CH;
            stg.load(rb + (int64_t)(ci + 2) * CH * RP, sb + (ci + 2) * CH, rest < 0 ? 0 : (rest < CH ? rest : CH), RP, threadIdx.x);
        }
        if constexpr (NWV > 1) {
            red[k & 1][w][lane] = ph;
            TP_BARRIER();
            Ph = (red[k & 1][0][lane] + red[k & 1][1][lane]) + (red[k & 1][2][lane] + red[k & 1][3][lane]);
        }
        reinterpret_cast<double*>(uni[w])[lane] = Ph;
        double2 phc[NP];
#pragma unroll
        for (int s = 0; s < NP; ++s) phc[s] = uni[w][NWV * s + w];
        double sS = hh * Ph, sm = hh * m;
        tp_sum2<NWV == 1>(sS, sm);
        const double S = sbuf[buf][si * 2 + 1].x + sS, v = s0.y - sm;
        const double iS = tp_rcp(S);
        Slast = S;
        if (threadIdx.x == 0) sring[k & 255] = S;
        quad = fma(v * v, iS, quad);
        const double K = Ph * iS;
        m = fma(K, v, m);
#pragma unroll
        for (int s = 0; s < NP; ++s) {
            const double2 pc = phc[s];
            P[s][0] = fma(-K, pc.x, P[s][0]);
            P[s][1] = fma(-K, pc.y, P[s][1]);
        }
        if ((k & 255) == 255 || k == len - 1) {
            TP_BARRIER();
            const int base = k & ~255, cnt = k - base + 1;
            for (int i = threadIdx.x; i < cnt; i += 64 * NWV) {
                const double Sv = sring[i];
                ldsum += (n0 + base + i == 0) ? log(Sv) : log(fabs(Sv));       // (src/celerite_solver.jl:126, 140: log D_1, log |D_n|)
                bad |= !(Sv > 0.0);
            }
            TP_BARRIER();
        }
    }
    // the segment's sums: sum over the threads' shares of log |D_n| (fixed order: the share of a thread is fixed by the segment's length)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        ldsum += __shfl_xor(ldsum, off);
        bad |= __shfl_xor(bad, off);
    }
    if constexpr (NWV > 1) {
        if (lane == 0) { red[0][w][0] = ldsum; red[0][w][1] = (double)bad; }
        TP_BARRIER();
        ldsum = (red[0][0][0] + red[0][1][0]) + (red[0][2][0] + red[0][3][0]);
        bad = (red[0][0][1] + red[0][1][1] + red[0][2][1] + red[0][3][1]) > 0.0;
    }
    if (threadIdx.x == 0) {
        double* o = part + (b * nseg + seg) * 4;
        o[0] = quad; o[1] = ldsum; o[2] = (double)bad; o[3] = 0.0;
    }
    // The check of the boundary states (round 6; disc != nullptr: behind the scan AND behind the walk).  The filter has just carried the state it started from at
    // boundary `seg` through the segment: that is the state at boundary seg + 1 by the step-by-step arithmetic.  Its distance from the state the next segment's
    // workgroup starts from (computed by phase 2 from the segments' ELEMENTS) goes to the draw's maximum, on the scale that log L feels — the INNOVATION VARIANCE S
    // of the segment's last step (a step's term of log L moves by ~ dS / S and dv^2 / S, with dS, dv sums of entries of dP, dm): |dP| / S and |dm| / sqrt(S),
    // absolute differences.  If every boundary's distance is below tol, phase 2's states are within nseg tol of the sequential ones (induction from the exact prior):
    // a residual, not a guess.  The caller's repair pass goes by it: kTpScanTol = 1e-3.
    // Why it is needed, and the threshold (oracle as reference, positive definite draws: tools/tp_scan_accept.py, tp_scan_metrics.py, tp_walk_accuracy.py;
    // profiles/r06_time_parallel_scan.txt sections 11 - 13): elements of long segments and composites of elements are not always well conditioned.  On the
    // DRWCelerite models a few prior draws per thousand come out of the SCAN alone wrong by 1e-6 .. O(1), and two of 160 out of the WALK alone by 4e-7 / 8e-7
    // (2e-6 at 32 segments) where the serial chain holds 4e-10; none of ~6000 draws of the SHO models.  Every one of those draws has a distance above 0.1; at 1e-3
    // the worst accepted draw of 7000 is 9e-10 off and 1 .. 3 % (SHO) / 6 .. 8 % (DRWCelerite) of the prior draws are repaired.  (tp_check = 1: the first form of this
    // check, relative to the state's largest entry; 3: an estimate of log L's relative error from the distance — tools.)
    if (disc && seg + 1 < nseg) {
        const double* b2 = bs + TP_BND_DOUBLES;
        double dP = 0.0, sP = 0.0, dm = 0.0, sm = 0.0;
        if (lane < RP) {
#pragma unroll
            for (int s = 0; s < NP; ++s) {
                const int c0 = 2 * (NWV * s + w);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const double o = b2[64 + lane * 64 + c0 + k], dd = fabs(P[s][k] - o);
                    dP = fmax(dP, dd >= 0.0 ? dd : __builtin_inf()); sP = fmax(sP, fabs(o));       // (fmax drops a NaN: a state that is not a number must fail)
                }
            }
            const double o = b2[lane];
            dm = fabs(m - o); dm = dm >= 0.0 ? dm : __builtin_inf(); sm = fabs(o);
        }
        dP = tp_max(dP); sP = tp_max(sP); dm = tp_max(dm); sm = tp_max(sm);
        if constexpr (NWV > 1) {
            TP_BARRIER();
            if (lane == 0) { red[0][w][0] = dP; red[0][w][1] = sP; red[0][w][2] = dm; red[0][w][3] = sm; }
            TP_BARRIER();
            dP = fmax(fmax(red[0][0][0], red[0][1][0]), fmax(red[0][2][0], red[0][3][0])); sP = fmax(fmax(red[0][0][1], red[0][1][1]), fmax(red[0][2][1], red[0][3][1]));
            dm = fmax(fmax(red[0][0][2], red[0][1][2]), fmax(red[0][2][2], red[0][3][2])); sm = fmax(fmax(red[0][0][3], red[0][1][3]), fmax(red[0][2][3], red[0][3][3]));
        }
        if (threadIdx.x == 0) {
            const double Sa = fabs(Slast);
            double rel = fmax(dP / (Sa > 0.0 ? Sa : 1.0), dm / (Sa > 0.0 ? sqrt(Sa) : 1.0));
            if (check == 1) {            // (the first form: relative to the state's largest entries; tools/tp_scan_metrics.py holds the two against each other)
                const double mscale = fmax(sm, sqrt(sP));
                rel = fmax(dP / (sP > 0.0 ? sP : 1.0), dm / (mscale > 0.0 ? mscale : 1.0));
            }
            if (!(rel >= 0.0)) rel = __builtin_inf();
            atomicMax(reinterpret_cast<unsigned long long*>(disc + b), (unsigned long long)__double_as_longlong(rel));
        }
    }
}

// log L = -1/2 sum log |D_n| - N/2 log 2 pi - 1/2 sum z_n^2 / D_n from the segments' sums, in a fixed order
__global__ void __launch_bounds__(64) tp_finish_kernel(int64_t N, int nseg, int64_t B, const double* __restrict__ part, double* __restrict__ out,
                                                       int32_t* __restrict__ status, double* __restrict__ disc, int mode)
{
    // one wavefront per draw (round 6: with up to 256 segments the one-thread loop was 34 us of a 600 us evaluation): lane l adds the segments l, l + 64, ...,
    // then a butterfly over the lanes — the same order on every call
    const int64_t b = blockIdx.x;
    const int lane = threadIdx.x;
    double ld = 0.0, q = 0.0, bad = 0.0;
    for (int s2 = lane; s2 < nseg; s2 += 64) {
        const double* o = part + (b * nseg + s2) * 4;
        q += o[0]; ld += o[1]; bad += o[2];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        q += __shfl_xor(q, off); ld += __shfl_xor(ld, off); bad += __shfl_xor(bad, off);
    }
    if (lane != 0) return;
    const double res = -0.5 * ld - 0.5 * (double)N * 1.8378770664093454836 - 0.5 * q;
    out[b] = res;
    if (status) status[b] = !isfinite(res) ? 2 : (bad > 0.0 ? 1 : 0);
    // tools only (tp_check = 3; tools/tp_scan_metrics.py): the check as an ESTIMATE of log L's relative error, distance x sqrt(N) / |log L| (the terms' errors have random signs)
    if (disc && mode == 3) disc[b] = disc[b] * sqrt((double)N) / fmax(fabs(res), 1.0);
}

// ---- phase 1 ---------------------------------------------------------------------------------------------------------------------------------
// A is kept TRANSPOSED (At: lane = column of A): then every product is an in-lane column rotation / a rank-one update with one lane-indexed and one
// register-indexed (wave-uniform) factor / a matrix-vector sum over the wavefront's columns — tools/time_parallel_proto.py::lane_form_element is this
// loop in numpy, checked against the textbook composition.  Records as in the filter kernel (one step ahead through LDS).
template <int NP, int NWV>
__global__ void __launch_bounds__(64 * NWV) tp_element_kernel(int64_t N, int RP, int nseg, int64_t L, const int32_t* __restrict__ row_kind,
                                                              const TpRec* __restrict__ rec, const TpStep* __restrict__ stp, double* __restrict__ elem)
{
    using Stage = TpStage<NWV>;
    constexpr int CH = Stage::CH;
    __shared__ double red[2][2][NWV][64], red2[2][2][NWV][64];     // by step parity: no wavefront is more than one barrier ahead of another
    __shared__ double2 rbuf[2][CH * Stage::RMAX * 3], sbuf[2][CH * 2];
    __shared__ double2 uni[NWV][3][32];    // u, A'g, Y h back as wave-uniform column pairs (one 16-byte broadcast read per pair; v_readlane: two per value)
    const int lane = threadIdx.x & 63, w = NWV == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int seg = blockIdx.x;
    const int64_t b = blockIdx.y;
    const int64_t n0 = seg * L, n1 = n0 + L < N ? n0 + L : N;
    const int len = (int)(n1 - n0);
    const TpRec* rb = rec + (b * N + n0) * RP;
    const TpStep* sb = stp + b * N + n0;
    const int rl = lane < RP ? lane : RP - 1;
    const bool rowok = lane < RP;
    const int mykind = row_kind[rl];
    const double mh = (rowok && (mykind == 0 || mykind == 2)) ? 1.0 : 0.0;
    double At[NP][2], C[NP][2], Jm[NP][2], hc[NP][2];
#pragma unroll
    for (int s = 0; s < NP; ++s) {
        const int c0 = 2 * (NWV * s + w);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            At[s][k] = (lane == c0 + k && lane < RP) ? 1.0 : 0.0;
            C[s][k] = 0.0;
            Jm[s][k] = 0.0;
            const int kc = row_kind[c0 + k];
            hc[s][k] = (kc == 0 || kc == 2) ? 1.0 : 0.0;
        }
    }
    double bv = 0.0, eta = 0.0;
    const bool odd = lane & 1;
    Stage stg;
    stg.load(rb, sb, len < CH ? len : CH, RP, threadIdx.x);
    stg.store(rbuf[0], sbuf[0], RP, threadIdx.x);
    {
        const int rest = len - CH;
        stg.load(rb + (int64_t)CH * RP, sb + CH, rest < 0 ? 0 : (rest < CH ? rest : CH), RP, threadIdx.x);
    }
    TP_BARRIER();
    for (int kk = 0; kk < len; ++kk) {
        const int ci = kk / CH, si = kk % CH, buf = ci & 1;
        const double2* rr = rbuf[buf] + si * RP * 3;
        const double2 mab = rr[rl * 3], mgk = rr[rl * 3 + 1], mq = rr[rl * 3 + 2];
        const double2 s0 = sbuf[buf][si * 2], s1 = sbuf[buf][si * 2 + 1];
        const double st_s = s0.x, st_y = s0.y, st_yos = s1.y;
        const double mal = mab.x, mbe = mab.y;
        const double mg = rowok ? mgk.x : 0.0, Kr = rowok ? mgk.y : 0.0;
        double al_[NP][2], be_[NP][2], K_[NP][2];
        double up = 0.0, ap = 0.0;
#pragma unroll
        for (int s = 0; s < NP; ++s) {
            const int c0 = 2 * (NWV * s + w);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const double2 ab = rr[(c0 + k) * 3], gk = rr[(c0 + k) * 3 + 1];
                al_[s][k] = ab.x; be_[s][k] = ab.y; K_[s][k] = gk.y;
                up = fma(C[s][k], gk.x, up);
                ap = fma(At[s][k], gk.x, ap);
            }
        }
        double u = up, ag = ap;
        if (si == CH - 1) {
            stg.store(rbuf[buf ^ 1], sbuf[buf ^ 1], RP, threadIdx.x);
            const int rest = len - (ci + 2) * CH;
            stg.load(rb + (int64_t)(ci + 2) * CH * RP, sb + (ci + 2) * CH, rest < 0 ? 0 : (rest < CH ? rest : CH), RP, threadIdx.x);
        }
        if constexpr (NWV > 1) {
            red[kk & 1][0][w][lane] = up;
            red[kk & 1][1][w][lane] = ap;
            TP_BARRIER();
            u = (red[kk & 1][0][0][lane] + red[kk & 1][0][1][lane]) + (red[kk & 1][0][2][lane] + red[kk & 1][0][3][lane]);
            ag = (red[kk & 1][1][0][lane] + red[kk & 1][1][1][lane]) + (red[kk & 1][1][2][lane] + red[kk & 1][1][3][lane]);
        }
        reinterpret_cast<double*>(uni[w][0])[lane] = u;
        reinterpret_cast<double*>(uni[w][1])[lane] = ag;
        double2 uc[NP], agc[NP];
#pragma unroll
        for (int s = 0; s < NP; ++s) { uc[s] = uni[w][0][NWV * s + w]; agc[s] = uni[w][1][NWV * s + w]; }
        double gu = mg * u, gb = mg * bv;
        tp_sum2<NWV == 1>(gu, gb);
        const double delta = st_s + gu, idel = tp_rcp(delta);
        const double agd = ag * idel, ud = u * idel;
        double xf[NP][2], yv[NP][2];
        double hfx = 0.0, yh = 0.0;
#pragma unroll
        for (int s = 0; s < NP; ++s) {
            const double u0 = uc[s].x, u1 = uc[s].y;
            const double x0 = fma(-agd, u0, At[s][0]), x1 = fma(-agd, u1, At[s][1]);
            xf[s][0] = fma(al_[s][0], x0, be_[s][0] * x1);
            xf[s][1] = fma(al_[s][1], x1, be_[s][1] * x0);
            hfx = fma(xf[s][0], hc[s][0], fma(xf[s][1], hc[s][1], hfx));
            const double c_0 = fma(-ud, u0, C[s][0]), c_1 = fma(-ud, u1, C[s][1]);
            const double r0 = fma(mal, c_0, mbe * tp_partner(c_0)), r1 = fma(mal, c_1, mbe * tp_partner(c_1));
            yv[s][0] = fma(al_[s][0], r0, be_[s][0] * r1);
            yv[s][1] = fma(al_[s][1], r1, be_[s][1] * r0);
            yh = fma(yv[s][0], hc[s][0], fma(yv[s][1], hc[s][1], yh));
        }
        double hFX = hfx, Yh = yh;
        if constexpr (NWV > 1) {
            red2[kk & 1][0][w][lane] = hfx;
            red2[kk & 1][1][w][lane] = yh;
            TP_BARRIER();
            hFX = (red2[kk & 1][0][0][lane] + red2[kk & 1][0][1][lane]) + (red2[kk & 1][0][2][lane] + red2[kk & 1][0][3][lane]);
            Yh = (red2[kk & 1][1][0][lane] + red2[kk & 1][1][1][lane]) + (red2[kk & 1][1][2][lane] + red2[kk & 1][1][3][lane]);
        }
        double bb = fma(u, st_yos, bv);
        const double gbb = fma(gu, st_yos, gb);
        bb = fma(-u, gbb * idel, bb);
        const double Fb = fma(mal, bb, mbe * tp_partner(bb));
        reinterpret_cast<double*>(uni[w][2])[lane] = Yh;
        double2 yhc2[NP];
#pragma unroll
        for (int s = 0; s < NP; ++s) yhc2[s] = uni[w][2][NWV * s + w];
        double hYh = mh * Yh, hFb = mh * Fb;
        tp_sum2<NWV == 1>(hYh, hFb);
#pragma unroll
        for (int s = 0; s < NP; ++s) {
            const int jj = NWV * s + w;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const double Kc = K_[s][k], yhc = k == 0 ? yhc2[s].x : yhc2[s].y;
                At[s][k] = fma(-hFX, Kc, xf[s][k]);
                double cn = yv[s][k] - Kr * yhc - Yh * Kc + Kr * Kc * hYh - Kr * (Kc * st_s);
                cn += (lane >> 1) == jj ? ((odd == (k == 1)) ? mq.x : mq.y) : 0.0;
                C[s][k] = cn;
                Jm[s][k] = fma(agd, k == 0 ? agc[s].x : agc[s].y, Jm[s][k]);
            }
        }
        eta = fma(ag, (st_y - gb) * idel, eta);
        bv = fma(Kr, st_y - hFb, Fb);
    }
    double* e = elem + (b * nseg + seg) * TP_ELEM_DOUBLES;
#pragma unroll
    for (int s = 0; s < NP; ++s) {
        const int c0 = 2 * (NWV * s + w);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            e[lane * 64 + c0 + k] = At[s][k];
            e[4096 + lane * 64 + c0 + k] = C[s][k];
            e[8192 + lane * 64 + c0 + k] = Jm[s][k];
        }
    }
    if (w == 0) { e[12288 + lane] = bv; e[12288 + 64 + lane] = eta; }
}

// ---- phase 2 ---------------------------------------------------------------------------------------------------------------------------------
// (m, P) at the start of segment p + 1 from (m, P) at the start of segment p and the element of segment p:
//     [z | Z] = (I + P J)^-1 [m + P eta | P],   m' = A z + b,   P' = A Z A' + C.
// One workgroup per draw.  W = I + P J and the right-hand sides in LDS; Gauss-Jordan elimination with partial pivoting WITHOUT row exchanges (the
// pivot of column k is the largest entry among the rows not used yet; the row keeps its place and remembers its column): one barrier per pivot.
// (I + P J is the identity plus a product of two symmetric positive semi-definite matrices: not symmetric, eigenvalues >= 1.)
// Thread (lane = row, wavefront = every fourth column).  The products run over the RP live rows only.
template <int TW, int RT>      // wavefronts (4; 1: no barrier at all, LDS traffic of one wavefront is served in order); RT 16-row tiles: 16 (RT - 1) < RP <= 16 RT
__global__ void __launch_bounds__(64 * TW) tp_boundary_kernel(int RP, int nseg, int J, const int32_t* __restrict__ row_term, const int32_t* __restrict__ row_kind,
                                                          const double* __restrict__ A_, const double* __restrict__ Bc_, const double* __restrict__ elem,
                                                          double* __restrict__ bnd, const double* __restrict__ disc, double tol)
{
    // (disc: behind the scan form of this phase, tp_combine_kernel — only the draws whose scan failed its check are walked)
    if (disc && !(disc[blockIdx.x] > tol)) return;
    extern __shared__ double lds[];
    constexpr int R16 = 16 * RT;                       // the products run on 16 x 16 tiles: rows and columns RP .. R16 - 1 of every matrix stay zero
    constexpr int S1 = R16 + 1, LW = 2 * R16 + 3;      // odd strides
    double* X = lds;                     // [R16][LW]: [W, later A Z | right-hand sides z (1), Z (RP)]
    double* Pm = X + R16 * LW;           // [R16][S1]: P, later Z, then P'
    double* JL = Pm + R16 * S1;          // J, then (the same buffer) A' as stored: AL[k][r] = A[r][k] — J is dead once W = I + P J is formed
    double* AL = JL;
    double* mv = JL + R16 * S1;          // [64] m, later z
    double* ev = mv + 64;                // [64] eta
    double* bl = ev + 64;                // [64] b
    double* fneg = bl + 64;              // [TW][4][64]: the block's multipliers, negated, per wavefront (A operand of the rank-4 update)
    constexpr int T = 64 * TW;
    const int tid = threadIdx.x, lane = tid & 63, w = TW == 1 ? 0 : __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t b = blockIdx.x;
    double* bs = bnd + b * nseg * TP_BND_DOUBLES;
    const int NC = 2 * RP + 1;           // live columns of [W | z | Z], packed: W 0 .. RP-1, z RP, Z RP+1 .. 2 RP
    for (int i = tid; i < R16 * LW + 2 * R16 * S1; i += T) lds[i] = 0.0;
    // One R x R x R product on the matrix cores: tile (I, Jt) of the result by wavefront (I RT + Jt) mod 4, operands straight from LDS
    // (A operand: row 16 I + (lane & 15), k = 4 ks + (lane >> 4); B operand: k, column 16 Jt + (lane & 15); result register g: row 4 g + (lane >> 4)).
    // (Scalar products — one output per thread, two LDS reads per FMA — took 9 us each at 40 rows, the LDS's bandwidth.)
    const int li = lane & 15, lk = lane >> 4;
    auto gemm = [&](auto aop, auto bop, auto store) __attribute__((always_inline)) {
        for (int tI = w; tI < RT * RT; tI += TW) {
            const int I = tI / RT, Jt = tI - I * RT;
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
            // (every operand read of the tile in flight before its first matrix instruction: one read pair at a time cost 4.6 us per product, four
            //  pairs at a time 2.4; RT <= 4 at the 64 rows this kernel takes)
            double av[RT][4], bv[RT][4];
#pragma unroll
            for (int k4 = 0; k4 < RT; ++k4) {
                const int kq = k4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    av[k4][i] = aop(16 * I + li, 16 * kq + 4 * i + lk);
                    bv[k4][i] = bop(16 * kq + 4 * i + lk, 16 * Jt + li);
                }
            }
#pragma unroll
            for (int k4 = 0; k4 < RT; ++k4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[k4][i], bv[k4][i], acc, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = 16 * I + 4 * g + lk, c = 16 * Jt + li;
                if (r < RP && c < RP) store(r, c, acc[g]);
            }
        }
    };
    TP_SYNC();
    // entry i = tid + T q of an RP x RP matrix: (row, col) = (i / RP, i % RP) — consecutive lanes, consecutive columns
    const int nq = (RP * RP + T - 1) / T;      // <= 16 at the 64 rows four wavefronts take, 4 at the 16 of one
    constexpr int NQ = (R16 * R16 + T - 1) / T;
    int row[NQ], col[NQ];
    bool ok[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i = tid + T * q;
        ok[q] = q < nq && i < RP * RP;
        row[q] = ok[q] ? i / RP : 0;
        col[q] = ok[q] ? i % RP : 0;
    }
    // An element's matrices come through registers, each loaded well ahead of its use (the compiler waits with vmcnt(0) at every use of a loaded
    // register, i.e. for EVERY load in flight: a load issued just before another one's use costs its full latency, ~2 us, there): J, eta, b of
    // element p + 1 and C of element p at the top of iteration p (J is used at the next top, C at this iteration's end), A' of element p + 1 behind
    // the point where A' of element p goes to LDS (after the first product).
    double rj[NQ], ra[NQ], rc[NQ], reta = 0.0, rb = 0.0;
    auto fetch_mat = [&](double (&dst)[NQ], const double* src) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) dst[q] = ok[q] ? src[row[q] * 64 + col[q]] : 0.0;
    };
    if (nseg > 1) {
        const double* e0 = elem + b * nseg * TP_ELEM_DOUBLES;
        fetch_mat(rj, e0 + 8192);
        fetch_mat(ra, e0);
        if (tid < RP) { reta = e0[12288 + 64 + tid]; rb = e0[12288 + tid]; }
    }
    double pn[NQ], mnew = 0.0;
    auto publish = [&](int pb) __attribute__((always_inline)) {
        double* bo = bs + (int64_t)pb * TP_BND_DOUBLES;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (ok[q]) bo[64 + row[q] * 64 + col[q]] = pn[q];
        if (tid < RP) bo[tid] = mnew;
    };
    // the prior: m = 0, P = P_inf (the boundary states' rows and columns past RP are zero: the launcher clears the buffer)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (ok[q]) {
            const int r = row[q], c = col[q];
            double v = 0.0;
            const int kr = row_kind[r], kc = row_kind[c];
            if (kr != 3 && kc != 3) {
                if (r == c) v = A_[b * J + row_term[r]];
                else if ((r ^ 1) == c && kr < 2 && kc < 2) v = -Bc_[b * J + row_term[r]];
            }
            Pm[r * S1 + c] = v;
            bs[64 + r * 64 + c] = v;
        }
    }
    if (tid < 64) mv[tid] = 0.0;
    TP_SYNC();
    for (int p = 0; p + 1 < nseg; ++p) {
        const double* e = elem + (b * nseg + p) * TP_ELEM_DOUBLES;
        // (J, A' and C of this element sit in registers since the previous iteration; J and A' share one LDS buffer — five R x R matrices and the
        //  solve's columns do not fit 160 KB at 64 rows —, C is added from its registers; the next element's loads go out once all three are consumed)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (ok[q]) JL[row[q] * S1 + col[q]] = rj[q];
        if (tid < RP) { ev[tid] = reta; bl[tid] = rb; }
        fetch_mat(rc, e + 4096);
        if (p + 2 < nseg) {
            fetch_mat(rj, e + TP_ELEM_DOUBLES + 8192);
            if (tid < RP) { reta = e[TP_ELEM_DOUBLES + 12288 + 64 + tid]; rb = e[TP_ELEM_DOUBLES + 12288 + tid]; }
        }
        if (p > 0) publish(p);       // (the state this iteration starts from: its stores, too, have the iteration — vmcnt counts them)
        TP_SYNC();
        // W = I + P J, z = m + P eta, Z = P
        gemm([&](int r, int kk) { return Pm[r * S1 + kk]; }, [&](int kk, int c) { return JL[kk * S1 + c]; },
             [&](int r, int c, double v) { X[r * LW + c] = v + (r == c ? 1.0 : 0.0); });
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (ok[q]) X[row[q] * LW + RP + 1 + col[q]] = Pm[row[q] * S1 + col[q]];
        if (tid < RP) {
            double acc = mv[tid];
#pragma unroll 8
            for (int k = 0; k < RP; ++k) acc = fma(Pm[tid * S1 + k], ev[k], acc);
            X[tid * LW + RP] = acc;
        }
        TP_SYNC();
        // Gauss-Jordan elimination with partial pivoting, FOUR pivots per barrier: every wavefront factors the 4-column panel itself (lane = row, the
        // panel in registers, pivot rows by v_readlane: no LDS, no barrier), then the rank-4 update of its share of the other columns with the four
        // pivot rows as they stand after the earlier pivots of the block (u_j = row pr_j - sum_{i<j} f_i[pr_j] u_i), one barrier per block.
        // (One pivot per barrier with its column through LDS: 1.1 us per pivot, 43 of a boundary's 77 us at 40 rows.)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (ok[q]) AL[row[q] * S1 + col[q]] = ra[q];          // (J is dead: the barrier behind W = I + P J has passed)
        if (p + 2 < nseg) fetch_mat(ra, e + TP_ELEM_DOUBLES);
        bool used = lane >= RP;
        int mycol = 0;
        double mypiv = 1.0;
        const int lr = lane < RP ? lane : 0;
        for (int k0 = 0; k0 < RP; k0 += 4) {          // (RP is a multiple of 8 here)
            double xp[4], f[4];
            int pr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) xp[j] = X[lr * LW + k0 + j];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // the pivot: a row whose |x| agrees with the column's largest in its high word (within 2^-20 of it)
                const unsigned cand = used ? 0u : ((unsigned)__double2hiint(xp[j]) & 0x7fffffffu) + 1u;
                const unsigned mx = tp_max_u32(cand);
                const unsigned long long bal = __ballot(cand == mx);
                pr[j] = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(bal));
                const double ipv = tp_rcp(tp_readlane(xp[j], pr[j]));
                f[j] = (lane == pr[j] || lane >= RP) ? 0.0 : xp[j] * ipv;
                if (lane == pr[j]) { used = true; mycol = k0 + j; mypiv = ipv; }
#pragma unroll
                for (int jj = j + 1; jj < 4; ++jj) xp[jj] = fma(-f[j], tp_readlane(xp[jj], pr[j]), xp[jj]);
            }
            const double f01 = tp_readlane(f[0], pr[1]), f02 = tp_readlane(f[0], pr[2]), f03 = tp_readlane(f[0], pr[3]);
            const double f12 = tp_readlane(f[1], pr[2]), f13 = tp_readlane(f[1], pr[3]), f23 = tp_readlane(f[2], pr[3]);
            // the rank-4 update of the other columns on the matrix cores: X[:, cols] -= F U, F = (f_0 .. f_3) (rows x 4), U = the four pivot rows as they
            // stand after the earlier pivots of the block (4 x cols).  A wavefront owns whole column tiles of 16 (every row tile of them): the pivot
            // rows of a column are read before any wavefront writes that column.  (As vector FMAs, lane = row: 14 of a boundary's 33 us at 40 rows.)
#pragma unroll
            for (int j = 0; j < 4; ++j) fneg[(w * 4 + j) * 64 + lane] = -f[j];
            for (int ct = w; k0 + 4 + 16 * ct < NC; ct += TW) {
                const int c0 = k0 + 4 + 16 * ct, col = c0 + li, cc = col < NC ? col : NC - 1;
                const double x0 = X[pr[0] * LW + cc], x1 = X[pr[1] * LW + cc], x2 = X[pr[2] * LW + cc], x3 = X[pr[3] * LW + cc];
                const double u0 = x0, u1 = fma(-f01, u0, x1), u2 = fma(-f12, u1, fma(-f02, u0, x2)), u3 = fma(-f23, u2, fma(-f13, u1, fma(-f03, u0, x3)));
                const double ub = lk == 0 ? u0 : (lk == 1 ? u1 : (lk == 2 ? u2 : u3));
                f64x4 acc[RT];
                double fa[RT];
#pragma unroll
                for (int It = 0; It < RT; ++It) {
                    const int Iq = It;
                    fa[It] = fneg[(w * 4 + lk) * 64 + 16 * Iq + li];
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[It][g] = X[(16 * Iq + 4 * g + lk) * LW + cc];
                }
#pragma unroll
                for (int It = 0; It < RT; ++It) {
                    {
                        acc[It] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[It], ub, acc[It], 0, 0, 0);
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            if (col < NC && 16 * It + 4 * g + lk < RP) X[(16 * It + 4 * g + lk) * LW + col] = acc[It][g];
                    }
                }
            }
            TP_SYNC();
        }
        // row `lane` solved column mycol: [z | Z][mycol] = its right-hand sides / its pivot.  Z into Pm (P is dead), z into mv.
        if (lane < RP) {
            for (int c = w; c < RP; c += TW) Pm[mycol * S1 + c] = X[lane * LW + RP + 1 + c] * mypiv;
            if (w == 0) mv[mycol] = X[lane * LW + RP] * mypiv;
        }
        TP_SYNC();
        // m' = A z + b; T = A Z (into X, columns 0 .. RP-1); P' = T A' + C.  A[r][k] = AL[k][r].
        if (tid < RP) {
            double acc = bl[tid];
#pragma unroll 8
            for (int k = 0; k < RP; ++k) acc = fma(AL[k * S1 + tid], mv[k], acc);
            mnew = acc;
        }
        gemm([&](int r, int kk) { return AL[kk * S1 + r]; }, [&](int kk, int c) { return Pm[kk * S1 + c]; },
             [&](int r, int c, double v) { X[r * LW + c] = v; });
        TP_SYNC();
        if (tid < RP) mv[tid] = mnew;
        // (the A operand runs over columns up to R16 - 1 of X: past RP sit z and Z, multiplied by the zero rows of A')
        gemm([&](int r, int kk) { return X[r * LW + kk]; }, [&](int kk, int c) { return AL[kk * S1 + c]; },
             [&](int r, int c, double v) { Pm[r * S1 + c] = v; });
        TP_SYNC();
        // symmetrise (the two products round differently); published at the top of the next iteration
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (ok[q]) pn[q] = 0.5 * (Pm[row[q] * S1 + col[q]] + Pm[col[q] * S1 + row[q]]) + rc[q];       // (C is symmetric: the element kernel symmetrises nothing,
                                                                                                          //  its C is symmetric by construction up to rounding)
        TP_SYNC();
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (ok[q]) Pm[row[q] * S1 + col[q]] = pn[q];
        TP_SYNC();
    }
    if (nseg > 1) publish(nseg - 1);
}

// ---- phase 2 as a SCAN (round 6) ---------------------------------------------------------------------------------------------------------------
// The boundary walk above is sequential over the segments (25 us per boundary at 40 rows: 724 of SHO-20's 1450 us at N = 1e4) while the elements it
// consumes are associative.  Here: a Kogge-Stone scan over the scan indices 0 (the prior, as the element A = 0, b = 0, C = P_inf) and p = 1 .. nseg-1
// (the element of segment p-1): level l combines  out[p] = in[p - 2^l] (x) in[p]  for every p >= 2^l, one workgroup each on its own CU; after level l
// the indices below 2^(l+1) are complete prefixes, and the (b, C) of a complete prefix IS the filtered state (m, P) at the boundary in front of segment
// p (its A is 0).  ceil(log2 nseg) launches instead of nseg - 1 dependent steps.
//   a_i (x) a_j (i the earlier):  M = (I + C_i J_j)^-1;  b = A_j M (b_i + C_i eta_j) + b_j;  C = A_j (M C_i) A_j' + C_j        <- the boundary step with (m, P) := (b_i, C_i)
//                                 A = A_j (M A_i);  J = A_i' J_j (M A_i) + J_i  (M' J_j = J_j M);  eta = A_i' (v - J_j (M C_i) v) + eta_i,  v = eta_j - J_j b_i  (M' = I - J_j M C_i)
// i.e. the solve of the boundary step with RP more right-hand sides (A_i), three more products and a few matrix-vector sums.  A complete prefix as the
// left operand (its A is 0: "prefix mode") needs none of the extras and its result goes straight to the boundary states; its (b, C) are read from there.
// One workgroup of four wavefronts per (draw, target); the pieces — products on the matrix cores tile by tile, Gauss-Jordan with partial pivoting without
// row exchanges, four pivots per barrier, rank-4 updates on the matrix cores — are tp_boundary_kernel's.  16 RT >= RP, RP a multiple of 8, RT <= 3 (the
// three-block right-hand side does not fit 160 KB of LDS at 64 rows: those stay on the walk).
template <int RT, int TW>
__global__ void __launch_bounds__(64 * TW) tp_combine_kernel(int RP, int nseg, int J, int stride, const int32_t* __restrict__ row_term,
                                                         const int32_t* __restrict__ row_kind, const double* __restrict__ A_, const double* __restrict__ Bc_,
                                                         const double* __restrict__ ein, double* __restrict__ eout, double* __restrict__ bnd,
                                                         double* __restrict__ disc)
{
    // disc != nullptr ("verify", stride = 1 with every left operand taken as a complete prefix): the boundary step of the sequential walk from the SCAN's
    // state at boundary p - 1 with the raw element of segment p - 1, compared with the scan's state at boundary p: a safety net under the scan's
    // combinations of incomplete elements, which are NOT as stable as the sequential filter (0.8 % of 2300 draws of four models come out of the scan alone wrong by
    // 1e-8 .. 1e-3: tools/tp_scan_metrics.py).  This launch is the check of the WALK-REPAIR mode (option tp_walk_repair; discrepancy relative to the state's largest
    // entries, the draws above the threshold are walked by tp_boundary_kernel).  The product path does not use it: there the filter kernel checks AND corrects
    // (tp_filter_kernel: two sweeps, distances on the scale of the innovation variance) and what still fails goes to the serial-chain kernel (capi.hip tp_dispatch).
    extern __shared__ double lds[];
    constexpr int R16 = 16 * RT, S1 = R16 + 1, LW = 3 * R16 + 3, T = 64 * TW;
    double* X = lds;                     // [R16][LW]: [W, later A_j Z | z (1), Z (RP), later J_j (M A_i) | A_i -> M A_i (RP)]
    double* Pm = X + R16 * LW;           // [R16][S1]: C_i, later Z = M C_i, then C (unsymmetrised), then J (unsymmetrised)
    double* JL = Pm + R16 * S1;          // J_j
    double* AL = JL + R16 * S1;          // A_j' as stored: AL[k][r] = A_j[r][k]; at the end A_i' the same way
    double* WA = AL + R16 * S1;          // M A_i
    double* mv = WA + R16 * S1;          // [64] b_i, later z
    double* ev = mv + 64;                // eta_j
    double* bl = ev + 64;                // b_j
    double* bi = bl + 64;                // b_i (kept)
    double* vv = bi + 64;                // v = eta_j - J_j b_i
    double* zz = vv + 64;                // Z v
    double* w2 = zz + 64;                // v - J_j Z v
    double* fneg = w2 + 64;              // [TW][4][64]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t b = blockIdx.y;
    const int p = (int)blockIdx.x + stride, i = p - stride;
    const bool prior = i == 0, full = !disc && i >= stride;  // the left operand: the prior | a complete prefix (in bnd) | an incomplete element (in ein)
    const double* ej = ein + (b * nseg + (p - 1)) * TP_ELEM_DOUBLES;
    const double* ei_ = ein + (b * nseg + (i > 0 ? i - 1 : 0)) * TP_ELEM_DOUBLES;
    double* bs = bnd + b * nseg * TP_BND_DOUBLES;
    const double* bsi = bs + (int64_t)i * TP_BND_DOUBLES;
    const int NC = full ? 3 * RP + 1 : 2 * RP + 1;          // live columns of [W | z | Z (| M A_i)]
    for (int q = tid; q < R16 * LW + 4 * R16 * S1 + 7 * 64; q += T) lds[q] = 0.0;
    const int li = lane & 15, lk = lane >> 4;
    auto gemm = [&](auto aop, auto bop, auto store) __attribute__((always_inline)) {
        for (int tI = w; tI < RT * RT; tI += TW) {
            const int I = tI / RT, Jt = tI - I * RT;
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
            double av[RT][4], bv[RT][4];
#pragma unroll
            for (int k4 = 0; k4 < RT; ++k4)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    av[k4][q4] = aop(16 * I + li, 16 * k4 + 4 * q4 + lk);
                    bv[k4][q4] = bop(16 * k4 + 4 * q4 + lk, 16 * Jt + li);
                }
#pragma unroll
            for (int k4 = 0; k4 < RT; ++k4)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[k4][q4], bv[k4][q4], acc, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = 16 * I + 4 * g + lk, c = 16 * Jt + li;
                if (r < RP && c < RP) store(r, c, acc[g]);
            }
        }
    };
    TP_STAMP(0);
    TP_BARRIER();
    TP_STAMP(1);
    // ---- operands -----------------------------------------------------------------------------------------------------------------------------
    for (int q = tid; q < RP * RP; q += T) {
        const int r = q / RP, c = q % RP;
        double v;
        if (prior) {                                         // P_inf (tp_boundary_kernel's prior)
            v = 0.0;
            const int kr = row_kind[r], kc = row_kind[c];
            if (kr != 3 && kc != 3) {
                if (r == c) v = A_[b * J + row_term[r]];
                else if ((r ^ 1) == c && kr < 2 && kc < 2) v = -Bc_[b * J + row_term[r]];
            }
            if (p == 1 && !disc) bs[64 + r * 64 + c] = v;    // (level 0: the prior's boundary state, which the filter of segment 0 starts from)
        } else {
            v = full ? ei_[4096 + r * 64 + c] : bsi[64 + r * 64 + c];
        }
        Pm[r * S1 + c] = v;
        JL[r * S1 + c] = ej[8192 + r * 64 + c];
        AL[r * S1 + c] = ej[r * 64 + c];
    }
    if (tid < RP) {
        const double m0 = prior ? 0.0 : (full ? ei_[12288 + tid] : bsi[tid]);
        mv[tid] = m0; bi[tid] = m0;
        ev[tid] = ej[12288 + 64 + tid];
        bl[tid] = ej[12288 + tid];
        if (prior && p == 1 && !disc) bs[tid] = 0.0;
    }
    TP_BARRIER();
    TP_STAMP(2);
    // ---- W = I + C_i J_j, z = b_i + C_i eta_j, Z = C_i (, A_i) ----------------------------------------------------------------------------------
    gemm([&](int r, int kk) { return Pm[r * S1 + kk]; }, [&](int kk, int c) { return JL[kk * S1 + c]; },
         [&](int r, int c, double v) { X[r * LW + c] = v + (r == c ? 1.0 : 0.0); });
    for (int q = tid; q < RP * RP; q += T) {
        const int r = q / RP, c = q % RP;
        X[r * LW + RP + 1 + c] = Pm[r * S1 + c];
        if (full) X[r * LW + 2 * RP + 1 + c] = ei_[c * 64 + r];          // A_i[r][c] (stored transposed)
    }
    if (tid < RP) {
        double acc = mv[tid];
#pragma unroll 8
        for (int k = 0; k < RP; ++k) acc = fma(Pm[tid * S1 + k], ev[k], acc);
        X[tid * LW + RP] = acc;
    }
    TP_BARRIER();
    TP_STAMP(3);
    // ---- Gauss-Jordan elimination, four pivots per barrier (tp_boundary_kernel) ------------------------------------------------------------------
    bool used = lane >= RP;
    int mycol = 0;
    double mypiv = 1.0;
    const int lr = lane < RP ? lane : 0;
    for (int k0 = 0; k0 < RP; k0 += 4) {
        double xp[4], f[4];
        int pr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xp[j] = X[lr * LW + k0 + j];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned cand = used ? 0u : ((unsigned)__double2hiint(xp[j]) & 0x7fffffffu) + 1u;
            const unsigned mx = tp_max_u32(cand);
            const unsigned long long bal = __ballot(cand == mx);
            pr[j] = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(bal));
            const double ipv = tp_rcp(tp_readlane(xp[j], pr[j]));
            f[j] = (lane == pr[j] || lane >= RP) ? 0.0 : xp[j] * ipv;
            if (lane == pr[j]) { used = true; mycol = k0 + j; mypiv = ipv; }
#pragma unroll
            for (int jj = j + 1; jj < 4; ++jj) xp[jj] = fma(-f[j], tp_readlane(xp[jj], pr[j]), xp[jj]);
        }
        if (k0 == 0) TP_STAMP(10);
        const double f01 = tp_readlane(f[0], pr[1]), f02 = tp_readlane(f[0], pr[2]), f03 = tp_readlane(f[0], pr[3]);
        const double f12 = tp_readlane(f[1], pr[2]), f13 = tp_readlane(f[1], pr[3]), f23 = tp_readlane(f[2], pr[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) fneg[(w * 4 + j) * 64 + lane] = -f[j];
        if (k0 == 0) TP_STAMP(11);
        for (int ct = w; k0 + 4 + 16 * ct < NC; ct += TW) {
            const int c0 = k0 + 4 + 16 * ct, col = c0 + li, cc = col < NC ? col : NC - 1;
            const double x0 = X[pr[0] * LW + cc], x1 = X[pr[1] * LW + cc], x2 = X[pr[2] * LW + cc], x3 = X[pr[3] * LW + cc];
            const double u0 = x0, u1 = fma(-f01, u0, x1), u2 = fma(-f12, u1, fma(-f02, u0, x2)), u3 = fma(-f23, u2, fma(-f13, u1, fma(-f03, u0, x3)));
            const double ub = lk == 0 ? u0 : (lk == 1 ? u1 : (lk == 2 ? u2 : u3));
            f64x4 acc[RT];
            double fa[RT];
#pragma unroll
            for (int It = 0; It < RT; ++It) {
                fa[It] = fneg[(w * 4 + lk) * 64 + 16 * It + li];
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[It][g] = X[(16 * It + 4 * g + lk) * LW + cc];
            }
#pragma unroll
            for (int It = 0; It < RT; ++It) {
                acc[It] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[It], ub, acc[It], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (col < NC && 16 * It + 4 * g + lk < RP) X[(16 * It + 4 * g + lk) * LW + col] = acc[It][g];
            }
        }
        if (k0 == 0) TP_STAMP(12);
        TP_BARRIER();
        if (k0 == 0) TP_STAMP(13);
    }
    TP_STAMP(4);
    // row `lane` solved column mycol: Z into Pm (C_i is dead), z into mv, M A_i into WA
    if (lane < RP) {
        for (int c = w; c < RP; c += TW) {
            Pm[mycol * S1 + c] = X[lane * LW + RP + 1 + c] * mypiv;
            if (full) WA[mycol * S1 + c] = X[lane * LW + 2 * RP + 1 + c] * mypiv;
        }
        if (w == 0) mv[mycol] = X[lane * LW + RP] * mypiv;
    }
    TP_BARRIER();
    TP_STAMP(5);
    // ---- vectors: b = A_j z + b_j;  v = eta_j - J_j b_i, Z v, v - J_j Z v (general mode) -----------------------------------------------------------
    double mnew = 0.0;
    if (tid < RP) {
        double acc = bl[tid];
#pragma unroll 8
        for (int k = 0; k < RP; ++k) acc = fma(AL[k * S1 + tid], mv[k], acc);
        mnew = acc;
        if (full) {
            double a2 = ev[tid];
#pragma unroll 8
            for (int k = 0; k < RP; ++k) a2 = fma(-JL[tid * S1 + k], bi[k], a2);
            vv[tid] = a2;
        }
    }
    // T = A_j Z (into X, columns 0 .. RP-1; W is dead)
    gemm([&](int r, int kk) { return AL[kk * S1 + r]; }, [&](int kk, int c) { return Pm[kk * S1 + c]; },
         [&](int r, int c, double v) { X[r * LW + c] = v; });
    if (full) {
        // J_j (M A_i) into X, columns RP+1 .. 2 RP (the copy of Z there is dead);  A = A_j (M A_i), stored transposed, straight to the output
        gemm([&](int r, int kk) { return JL[r * S1 + kk]; }, [&](int kk, int c) { return WA[kk * S1 + c]; },
             [&](int r, int c, double v) { X[r * LW + RP + 1 + c] = v; });
        double* eo = eout + (b * nseg + (p - 1)) * TP_ELEM_DOUBLES;
        gemm([&](int k, int s_) { return WA[s_ * S1 + k]; }, [&](int s_, int r) { return AL[s_ * S1 + r]; },
             [&](int k, int r, double v) { eo[k * 64 + r] = v; });
    }
    TP_BARRIER();
    if (full && tid < RP) {
        double a2 = 0.0;
#pragma unroll 8
        for (int k = 0; k < RP; ++k) a2 = fma(Pm[tid * S1 + k], vv[k], a2);
        zz[tid] = a2;
    }
    TP_BARRIER();
    if (full && tid < RP) {
        double a2 = vv[tid];
#pragma unroll 8
        for (int k = 0; k < RP; ++k) a2 = fma(-JL[tid * S1 + k], zz[k], a2);
        w2[tid] = a2;
    }
    TP_BARRIER();
    TP_STAMP(6);
    // C = T A_j' + C_j (Z is dead: into Pm), symmetrised on the way out
    gemm([&](int r, int kk) { return X[r * LW + kk]; }, [&](int kk, int c) { return kk < RP ? AL[kk * S1 + c] : 0.0; },
         [&](int r, int c, double v) { Pm[r * S1 + c] = v; });
    TP_BARRIER();
    if (disc) {
        // verify: |scan - step| over the state, relative to the state's largest entries
        double* bo = bs + (int64_t)p * TP_BND_DOUBLES;
        double dP = 0.0, sP = 0.0, dm = 0.0, sm = 0.0;
        for (int q = tid; q < RP * RP; q += T) {
            const int r = q / RP, c = q % RP;
            const double v = 0.5 * (Pm[r * S1 + c] + Pm[c * S1 + r]) + ej[4096 + r * 64 + c], o = bo[64 + r * 64 + c];
            const double dd = fabs(v - o);
            dP = fmax(dP, dd >= 0.0 ? dd : __builtin_inf()); sP = fmax(sP, fabs(o));      // (fmax drops a NaN: a state that is not a number must fail the check)
        }
        if (tid < RP) { const double o = bo[tid]; dm = fabs(mnew - o); dm = dm >= 0.0 ? dm : __builtin_inf(); sm = fabs(o); }
        dP = tp_max(dP); sP = tp_max(sP); dm = tp_max(dm); sm = tp_max(sm);
        if (lane == 0) { fneg[w] = dP; fneg[TW + w] = sP; fneg[2 * TW + w] = dm; fneg[3 * TW + w] = sm; }
        TP_BARRIER();
        if (tid == 0) {
            double DP = 0.0, SP = 0.0, DM = 0.0, SM = 0.0;
            for (int x = 0; x < TW; ++x) { DP = fmax(DP, fneg[x]); SP = fmax(SP, fneg[TW + x]); DM = fmax(DM, fneg[2 * TW + x]); SM = fmax(SM, fneg[3 * TW + x]); }
            // (the mean is measured against its own scale and the standard deviation the covariance implies: a mean that is tiny by symmetry must not trip it)
            double rel = DP / (SP > 0.0 ? SP : 1.0);
            const double mscale = fmax(SM, sqrt(SP));
            rel = fmax(rel, DM / (mscale > 0.0 ? mscale : 1.0));
            if (!(rel >= 0.0)) rel = 1.0;                    // NaN: a draw that is not positive definite — the walk reports it the way it always did
            atomicMax(reinterpret_cast<unsigned long long*>(disc + b), (unsigned long long)__double_as_longlong(rel));
        }
        return;
    }
    {
        double* eo = eout + (b * nseg + (p - 1)) * TP_ELEM_DOUBLES;
        double* bo = bs + (int64_t)p * TP_BND_DOUBLES;
        for (int q = tid; q < RP * RP; q += T) {
            const int r = q / RP, c = q % RP;
            const double v = 0.5 * (Pm[r * S1 + c] + Pm[c * S1 + r]) + ej[4096 + r * 64 + c];
            if (full) eo[4096 + r * 64 + c] = v; else bo[64 + r * 64 + c] = v;
        }
        if (tid < RP) { if (full) eo[12288 + tid] = mnew; else bo[tid] = mnew; }
    }
    TP_STAMP(7);
    if (!full) return;                                       // (workgroup-uniform)
    TP_BARRIER();
    // ---- J = A_i' (J_j M A_i) + J_i, eta = A_i' (v - J_j Z v) + eta_i: A_i' into AL (A_j is dead) ---------------------------------------------------
    for (int q = tid; q < RP * RP; q += T) {
        const int r = q / RP, c = q % RP;
        AL[r * S1 + c] = ei_[r * 64 + c];
    }
    TP_BARRIER();
    gemm([&](int r, int kk) { return AL[r * S1 + kk]; }, [&](int kk, int c) { return kk < RP ? X[kk * LW + RP + 1 + c] : 0.0; },
         [&](int r, int c, double v) { Pm[r * S1 + c] = v; });
    double enew = 0.0;
    if (tid < RP) {
        double a2 = ei_[12288 + 64 + tid];
#pragma unroll 8
        for (int k = 0; k < RP; ++k) a2 = fma(AL[tid * S1 + k], w2[k], a2);
        enew = a2;
    }
    TP_BARRIER();
    {
        double* eo = eout + (b * nseg + (p - 1)) * TP_ELEM_DOUBLES;
        for (int q = tid; q < RP * RP; q += T) {
            const int r = q / RP, c = q % RP;
            eo[8192 + r * 64 + c] = 0.5 * (Pm[r * S1 + c] + Pm[c * S1 + r]) + ei_[8192 + r * 64 + c];
        }
        if (tid < RP) eo[12288 + 64 + tid] = enew;
    }
    TP_STAMP(8);
}

// The same combination for up to FOUR tiles of 16 rows (49 .. 64 state rows: DRWCelerite-20 is 60), where tp_combine_kernel's five matrices do not fit
// 160 KB of LDS.  Only the right-hand-side block X and one matrix stay in LDS (142 KB at 64 rows); A_j, J_j, C_j, A_i come from the elements in global
// memory straight into the matrix-core operands (32 KB each, L2-resident: they were written by the launch before), Z and M A_i stay where the elimination
// left them — row l of X holds row mycol(l) of the solution — and are read through the inverse permutation `rowof`.  J_j is read as its transpose (it is
// symmetric up to rounding) so that the lanes of a load run along a row.  Same three modes (prior / complete prefix / incomplete element) and the same
// verification mode as tp_combine_kernel; tests hold the two against each other at 16 .. 48 rows (option tp_scan_lean).
template <int RT, int TW>
__global__ void __launch_bounds__(64 * TW) tp_combine_lean_kernel(int RP, int nseg, int J, int stride, const int32_t* __restrict__ row_term,
                                                              const int32_t* __restrict__ row_kind, const double* __restrict__ A_, const double* __restrict__ Bc_,
                                                              const double* __restrict__ ein, double* __restrict__ eout, double* __restrict__ bnd,
                                                              double* __restrict__ disc)
{
    extern __shared__ double lds[];
    constexpr int R16 = 16 * RT, S1 = R16 + 1, LW = 3 * R16 + 3, T = 64 * TW;
    double* X = lds;                     // [R16][LW]: [W, later A_j Z, later J (unsymmetrised) | z (1) | C_i -> Z (RP), later C (unsymmetrised) | A_i -> M A_i (RP)]
    double* Pm = X + R16 * LW;           // [R16][S1]: J_j (M A_i)
    double* mv = Pm + R16 * S1;          // [64] b_i, later z
    double* ev = mv + 64;                // eta_j
    double* bl = ev + 64;                // b_j
    double* bi = bl + 64;                // b_i (kept)
    double* vv = bi + 64;                // v = eta_j - J_j b_i
    double* zz = vv + 64;                // Z v
    double* w2 = zz + 64;                // v - J_j Z v
    int* rowof = reinterpret_cast<int*>(w2 + 64);   // [64] the row of X that holds row k of the solution
    double* fneg = w2 + 64 + 32;         // [TW][4][64]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t b = blockIdx.y;
    const int p = (int)blockIdx.x + stride, i = p - stride;
    const bool prior = i == 0, full = !disc && i >= stride;  // the left operand: the prior | a complete prefix (in bnd) | an incomplete element (in ein: "full" combination)
    const double* ej = ein + (b * nseg + (p - 1)) * TP_ELEM_DOUBLES;
    const double* ei_ = ein + (b * nseg + (i > 0 ? i - 1 : 0)) * TP_ELEM_DOUBLES;
    double* bs = bnd + b * nseg * TP_BND_DOUBLES;
    const double* bsi = bs + (int64_t)i * TP_BND_DOUBLES;
    const int NC = full ? 3 * RP + 1 : 2 * RP + 1;          // live columns of [W | z | Z (| M A_i)]
    const int ZC = RP + 1, MC = 2 * RP + 1;                  // first column of Z / of M A_i in X
    for (int q = tid; q < R16 * LW + R16 * S1 + 7 * 64 + 32; q += T) lds[q] = 0.0;
    const int li = lane & 15, lk = lane >> 4;
    // one 16 x 16 tile of a product per wavefront and round; operands through aop(row, k) / bop(k, col), which return 0 outside RP themselves
    auto gemm = [&](auto aop, auto bop, auto store) __attribute__((always_inline)) {
        for (int tI = w; tI < RT * RT; tI += TW) {
            const int I = tI / RT, Jt = tI - I * RT;
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
            double av[RT][4], bv[RT][4];
#pragma unroll
            for (int k4 = 0; k4 < RT; ++k4)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    av[k4][q4] = aop(16 * I + li, 16 * k4 + 4 * q4 + lk);
                    bv[k4][q4] = bop(16 * k4 + 4 * q4 + lk, 16 * Jt + li);
                }
#pragma unroll
            for (int k4 = 0; k4 < RT; ++k4)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[k4][q4], bv[k4][q4], acc, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = 16 * I + 4 * g + lk, c = 16 * Jt + li;
                if (r < RP && c < RP) store(r, c, acc[g]);
            }
        }
    };
    auto in = [&](int r, int c) { return r < RP && c < RP; };
    TP_STAMP(0);
    TP_BARRIER();
    TP_STAMP(1);
    // ---- operands: C_i into the Z block, A_i into the last block ------------------------------------------------------------------------------------
    for (int q = tid; q < RP * RP; q += T) {
        const int r = q / RP, c = q % RP;
        double v;
        if (prior) {                                         // P_inf (tp_boundary_kernel's prior)
            v = 0.0;
            const int kr = row_kind[r], kc = row_kind[c];
            if (kr != 3 && kc != 3) {
                if (r == c) v = A_[b * J + row_term[r]];
                else if ((r ^ 1) == c && kr < 2 && kc < 2) v = -Bc_[b * J + row_term[r]];
            }
            if (p == 1 && !disc) bs[64 + r * 64 + c] = v;    // (level 0: the prior's boundary state, which the filter of segment 0 starts from)
        } else {
            v = full ? ei_[4096 + r * 64 + c] : bsi[64 + r * 64 + c];
        }
        X[r * LW + ZC + c] = v;
        if (full) X[r * LW + MC + c] = ei_[c * 64 + r];      // A_i[r][c] (stored transposed)
    }
    if (tid < RP) {
        const double m0 = prior ? 0.0 : (full ? ei_[12288 + tid] : bsi[tid]);
        mv[tid] = m0; bi[tid] = m0;
        ev[tid] = ej[12288 + 64 + tid];
        bl[tid] = ej[12288 + tid];
        if (prior && p == 1 && !disc) bs[tid] = 0.0;
    }
    TP_BARRIER();
    TP_STAMP(2);
    // ---- W = I + C_i J_j, z = b_i + C_i eta_j ------------------------------------------------------------------------------------------------------
    gemm([&](int r, int kk) { return in(r, kk) ? X[r * LW + ZC + kk] : 0.0; }, [&](int kk, int c) { return in(kk, c) ? ej[8192 + kk * 64 + c] : 0.0; },
         [&](int r, int c, double v) { X[r * LW + c] = v + (r == c ? 1.0 : 0.0); });
    if (tid < RP) {
        double acc = mv[tid];
#pragma unroll 8
        for (int k = 0; k < RP; ++k) acc = fma(X[tid * LW + ZC + k], ev[k], acc);
        X[tid * LW + RP] = acc;
    }
    TP_BARRIER();
    TP_STAMP(3);
    // ---- Gauss-Jordan elimination, four pivots per barrier (tp_boundary_kernel) ------------------------------------------------------------------
    bool used = lane >= RP;
    int mycol = 0;
    double mypiv = 1.0;
    const int lr = lane < RP ? lane : 0;
    for (int k0 = 0; k0 < RP; k0 += 4) {
        double xp[4], f[4];
        int pr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xp[j] = X[lr * LW + k0 + j];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned cand = used ? 0u : ((unsigned)__double2hiint(xp[j]) & 0x7fffffffu) + 1u;
            const unsigned mx = tp_max_u32(cand);
            const unsigned long long bal = __ballot(cand == mx);
            pr[j] = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(bal));
            const double ipv = tp_rcp(tp_readlane(xp[j], pr[j]));
            f[j] = (lane == pr[j] || lane >= RP) ? 0.0 : xp[j] * ipv;
            if (lane == pr[j]) { used = true; mycol = k0 + j; mypiv = ipv; }
#pragma unroll
            for (int jj = j + 1; jj < 4; ++jj) xp[jj] = fma(-f[j], tp_readlane(xp[jj], pr[j]), xp[jj]);
        }
        const double f01 = tp_readlane(f[0], pr[1]), f02 = tp_readlane(f[0], pr[2]), f03 = tp_readlane(f[0], pr[3]);
        const double f12 = tp_readlane(f[1], pr[2]), f13 = tp_readlane(f[1], pr[3]), f23 = tp_readlane(f[2], pr[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) fneg[(w * 4 + j) * 64 + lane] = -f[j];
        for (int ct = w; k0 + 4 + 16 * ct < NC; ct += TW) {
            const int c0 = k0 + 4 + 16 * ct, col = c0 + li, cc = col < NC ? col : NC - 1;
            const double x0 = X[pr[0] * LW + cc], x1 = X[pr[1] * LW + cc], x2 = X[pr[2] * LW + cc], x3 = X[pr[3] * LW + cc];
            const double u0 = x0, u1 = fma(-f01, u0, x1), u2 = fma(-f12, u1, fma(-f02, u0, x2)), u3 = fma(-f23, u2, fma(-f13, u1, fma(-f03, u0, x3)));
            const double ub = lk == 0 ? u0 : (lk == 1 ? u1 : (lk == 2 ? u2 : u3));
            f64x4 acc[RT];
            double fa[RT];
#pragma unroll
            for (int It = 0; It < RT; ++It) {
                fa[It] = fneg[(w * 4 + lk) * 64 + 16 * It + li];
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[It][g] = X[(16 * It + 4 * g + lk) * LW + cc];
            }
#pragma unroll
            for (int It = 0; It < RT; ++It) {
                acc[It] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[It], ub, acc[It], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (col < NC && 16 * It + 4 * g + lk < RP) X[(16 * It + 4 * g + lk) * LW + col] = acc[It][g];
            }
        }
        TP_BARRIER();
    }
    TP_STAMP(4);
    // row `lane` solved column mycol: scaled in place, found again through rowof; z into mv
    if (lane < RP) {
        for (int c = w; c < NC - RP; c += TW) X[lane * LW + RP + c] *= mypiv;
        if (w == 0) rowof[mycol] = lane;
    }
    TP_BARRIER();
    if (tid < RP) mv[tid] = X[rowof[tid] * LW + RP];
    auto Zs = [&](int kk, int c) { return in(kk, c) ? X[rowof[kk] * LW + ZC + c] : 0.0; };        // Z = M C_i
    auto MA = [&](int kk, int c) { return in(kk, c) ? X[rowof[kk] * LW + MC + c] : 0.0; };        // M A_i
    TP_BARRIER();
    TP_STAMP(5);
    // ---- vectors: b = A_j z + b_j;  v = eta_j - J_j b_i, Z v, v - J_j Z v (full combination) ---------------------------------------------------------
    double mnew = 0.0;
    if (tid < RP) {
        double acc = bl[tid];
#pragma unroll 8
        for (int k = 0; k < RP; ++k) acc = fma(ej[k * 64 + tid], mv[k], acc);
        mnew = acc;
        if (full) {
            double a2 = ev[tid];
#pragma unroll 8
            for (int k = 0; k < RP; ++k) a2 = fma(-ej[8192 + k * 64 + tid], bi[k], a2);
            vv[tid] = a2;
        }
    }
    // T = A_j Z (into X, columns 0 .. RP-1; W is dead)
    gemm([&](int r, int kk) { return in(r, kk) ? ej[kk * 64 + r] : 0.0; }, Zs, [&](int r, int c, double v) { X[r * LW + c] = v; });
    if (full) {
        // J_j (M A_i) into Pm;  A = A_j (M A_i), stored transposed, straight to the output
        gemm([&](int r, int kk) { return in(r, kk) ? ej[8192 + kk * 64 + r] : 0.0; }, MA, [&](int r, int c, double v) { Pm[r * S1 + c] = v; });
        double* eo = eout + (b * nseg + (p - 1)) * TP_ELEM_DOUBLES;
        gemm([&](int k, int s_) { return MA(s_, k); }, [&](int s_, int r) { return in(s_, r) ? ej[s_ * 64 + r] : 0.0; },
             [&](int k, int r, double v) { eo[k * 64 + r] = v; });
    }
    TP_BARRIER();
    if (full && tid < RP) {
        double a2 = 0.0;
        const int ro = rowof[tid];
#pragma unroll 8
        for (int k = 0; k < RP; ++k) a2 = fma(X[ro * LW + ZC + k], vv[k], a2);
        zz[tid] = a2;
    }
    TP_BARRIER();
    if (full && tid < RP) {
        double a2 = vv[tid];
#pragma unroll 8
        for (int k = 0; k < RP; ++k) a2 = fma(-ej[8192 + k * 64 + tid], zz[k], a2);
        w2[tid] = a2;
    }
    TP_STAMP(6);
    // C = T A_j' + C_j (Z is dead — every wavefront is past its last read of it: into its block), symmetrised on the way out
    gemm([&](int r, int kk) { return in(r, kk) ? X[r * LW + kk] : 0.0; }, [&](int kk, int c) { return in(kk, c) ? ej[kk * 64 + c] : 0.0; },
         [&](int r, int c, double v) { X[r * LW + ZC + c] = v; });
    TP_BARRIER();
    if (disc) {
        // verify: |scan - step| over the state, relative to the state's largest entries
        double* bo = bs + (int64_t)p * TP_BND_DOUBLES;
        double dP = 0.0, sP = 0.0, dm = 0.0, sm = 0.0;
        for (int q = tid; q < RP * RP; q += T) {
            const int r = q / RP, c = q % RP;
            const double v = 0.5 * (X[r * LW + ZC + c] + X[c * LW + ZC + r]) + ej[4096 + r * 64 + c], o = bo[64 + r * 64 + c];
            const double dd = fabs(v - o);
            dP = fmax(dP, dd >= 0.0 ? dd : __builtin_inf()); sP = fmax(sP, fabs(o));      // (fmax drops a NaN: a state that is not a number must fail the check)
        }
        if (tid < RP) { const double o = bo[tid]; dm = fabs(mnew - o); dm = dm >= 0.0 ? dm : __builtin_inf(); sm = fabs(o); }
        dP = tp_max(dP); sP = tp_max(sP); dm = tp_max(dm); sm = tp_max(sm);
        if (lane == 0) { fneg[w] = dP; fneg[TW + w] = sP; fneg[2 * TW + w] = dm; fneg[3 * TW + w] = sm; }
        TP_BARRIER();
        if (tid == 0) {
            double DP = 0.0, SP = 0.0, DM = 0.0, SM = 0.0;
            for (int x = 0; x < TW; ++x) { DP = fmax(DP, fneg[x]); SP = fmax(SP, fneg[TW + x]); DM = fmax(DM, fneg[2 * TW + x]); SM = fmax(SM, fneg[3 * TW + x]); }
            double rel = DP / (SP > 0.0 ? SP : 1.0);
            const double mscale = fmax(SM, sqrt(SP));
            rel = fmax(rel, DM / (mscale > 0.0 ? mscale : 1.0));
            if (!(rel >= 0.0)) rel = 1.0;                    // (NaN: see tp_combine_kernel)
            atomicMax(reinterpret_cast<unsigned long long*>(disc + b), (unsigned long long)__double_as_longlong(rel));
        }
        return;
    }
    {
        double* eo = eout + (b * nseg + (p - 1)) * TP_ELEM_DOUBLES;
        double* bo = bs + (int64_t)p * TP_BND_DOUBLES;
        for (int q = tid; q < RP * RP; q += T) {
            const int r = q / RP, c = q % RP;
            const double v = 0.5 * (X[r * LW + ZC + c] + X[c * LW + ZC + r]) + ej[4096 + r * 64 + c];
            if (full) eo[4096 + r * 64 + c] = v; else bo[64 + r * 64 + c] = v;
        }
        if (tid < RP) { if (full) eo[12288 + tid] = mnew; else bo[tid] = mnew; }
    }
    TP_STAMP(7);
    if (!full) return;                                       // (workgroup-uniform)
    // ---- J = A_i' (J_j M A_i) + J_i, eta = A_i' (v - J_j Z v) + eta_i  (T is dead since the barrier above: into its block) ------------------------------
    gemm([&](int r, int kk) { return in(r, kk) ? ei_[r * 64 + kk] : 0.0; }, [&](int kk, int c) { return in(kk, c) ? Pm[kk * S1 + c] : 0.0; },
         [&](int r, int c, double v) { X[r * LW + c] = v; });
    double enew = 0.0;
    if (tid < RP) {
        double a2 = ei_[12288 + 64 + tid];
#pragma unroll 8
        for (int k = 0; k < RP; ++k) a2 = fma(ei_[tid * 64 + k], w2[k], a2);
        enew = a2;
    }
    TP_BARRIER();
    {
        double* eo = eout + (b * nseg + (p - 1)) * TP_ELEM_DOUBLES;
        for (int q = tid; q < RP * RP; q += T) {
            const int r = q / RP, c = q % RP;
            eo[8192 + r * 64 + c] = 0.5 * (X[r * LW + c] + X[c * LW + r]) + ei_[8192 + r * 64 + c];
        }
        if (tid < RP) eo[12288 + 64 + tid] = enew;
    }
    TP_STAMP(8);
}

// The same for two or four state rows (the reference benchmark grid's j = 2: benchmark/benchmarks.jl:16-18), ONE THREAD per draw, everything in
// registers, fully unrolled; partial pivoting by conditional row exchanges.  A boundary is ~550 dependent-chain-free instructions (1.3 us) where the
// workgroup kernel spends 4.2 us in LDS round trips.
template <int R>
__global__ void __launch_bounds__(64) tp_boundary_small_kernel(int nseg, int J, int64_t B, const int32_t* __restrict__ row_term,
                                                               const int32_t* __restrict__ row_kind, const double* __restrict__ A_,
                                                               const double* __restrict__ Bc_, const double* __restrict__ elem, double* __restrict__ bnd)
{
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    double* bs = bnd + b * nseg * TP_BND_DOUBLES;
    double P[R][R], m[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        m[r] = 0.0;
#pragma unroll
        for (int c = 0; c < R; ++c) {
            double v = 0.0;
            const int kr = row_kind[r], kc = row_kind[c];
            if (kr != 3 && kc != 3) {
                if (r == c) v = A_[b * J + row_term[r]];
                else if ((r ^ 1) == c && kr < 2 && kc < 2) v = -Bc_[b * J + row_term[r]];
            }
            P[r][c] = v;
            bs[64 + r * 64 + c] = v;
        }
    }
    struct Elem { double A[R][R], C[R][R], Jm[R][R], b[R], eta[R]; };
    auto load = [&](Elem& E, const double* e) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int c = 0; c < R; ++c) {
                E.A[r][c] = e[c * 64 + r];              // A[r][c] = At[c][r]
                E.C[r][c] = e[4096 + r * 64 + c];
                E.Jm[r][c] = e[8192 + r * 64 + c];
            }
            E.b[r] = e[12288 + r];
            E.eta[r] = e[12288 + 64 + r];
        }
    };
    // Three register sets in rotation, the loop unrolled by three (no copies between the sets: a copy is a use, and the compiler waits for every
    // load in flight at a use): element p + 2 is loaded while element p is applied — a boundary is ~1 us of arithmetic, a load ~2 us away.
    Elem S0, S1, S2;
    if (nseg > 1) load(S0, elem + b * nseg * TP_ELEM_DOUBLES);
    S1 = S0;
    if (nseg > 2) load(S1, elem + (b * nseg + 1) * TP_ELEM_DOUBLES);
    S2 = S0;
    auto step = [&](const Elem& cur, Elem& fill, int p) __attribute__((always_inline)) {
        if (p + 3 < nseg) load(fill, elem + (b * nseg + p + 2) * TP_ELEM_DOUBLES);
        double X[R][2 * R + 1];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double z = m[r];
#pragma unroll
            for (int c = 0; c < R; ++c) {
                double acc = r == c ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < R; ++k) acc = fma(P[r][k], cur.Jm[k][c], acc);
                X[r][c] = acc;
                X[r][R + 1 + c] = P[r][c];
                z = fma(P[r][c], cur.eta[c], z);
            }
            X[r][R] = z;
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
#pragma unroll
            for (int i = k + 1; i < R; ++i) {
                const bool sw = fabs(X[i][k]) > fabs(X[k][k]);
#pragma unroll
                for (int c = k; c < 2 * R + 1; ++c) {
                    const double a0 = X[k][c], a1 = X[i][c];
                    X[k][c] = sw ? a1 : a0;
                    X[i][c] = sw ? a0 : a1;
                }
            }
            const double ipv = tp_rcp(X[k][k]);
#pragma unroll
            for (int c = k + 1; c < 2 * R + 1; ++c) X[k][c] *= ipv;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                if (i != k) {
                    const double f = X[i][k];
#pragma unroll
                    for (int c = k + 1; c < 2 * R + 1; ++c) X[i][c] = fma(-f, X[k][c], X[i][c]);
                }
            }
        }
        double T[R][R], mn[R], Pn[R][R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double acc = cur.b[r];
#pragma unroll
            for (int k = 0; k < R; ++k) acc = fma(cur.A[r][k], X[k][R], acc);
            mn[r] = acc;
#pragma unroll
            for (int c = 0; c < R; ++c) {
                double t = 0.0;
#pragma unroll
                for (int k = 0; k < R; ++k) t = fma(cur.A[r][k], X[k][R + 1 + c], t);
                T[r][c] = t;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int c = 0; c < R; ++c) {
                double acc = cur.C[r][c];
#pragma unroll
                for (int k = 0; k < R; ++k) acc = fma(T[r][k], cur.A[c][k], acc);
                Pn[r][c] = acc;
            }
        double* bo = bs + (int64_t)(p + 1) * TP_BND_DOUBLES;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            m[r] = mn[r];
            bo[r] = mn[r];
#pragma unroll
            for (int c = 0; c < R; ++c) {
                P[r][c] = 0.5 * (Pn[r][c] + Pn[c][r]);
                bo[64 + r * 64 + c] = P[r][c];
            }
        }
    };
    for (int p = 0;;) {
        if (p + 1 >= nseg) break;
        step(S0, S2, p); ++p;
        if (p + 1 >= nseg) break;
        step(S1, S0, p); ++p;
        if (p + 1 >= nseg) break;
        step(S2, S1, p); ++p;
    }
}

// Six to sixteen state rows: ONE WAVEFRONT per draw, lane = row, every matrix row in the lane's registers; the wave-uniform operands (a row of J, of Z,
// a column of A', the pivot row of the elimination) come from the owning lane by v_readlane — no LDS round trip, no barrier (LDS only for the two
// permutations at the end: rows back into natural order, the transpose of the symmetrisation).
template <int R>
__global__ void __launch_bounds__(64) tp_boundary_wave_kernel(int nseg, int J, const int32_t* __restrict__ row_term, const int32_t* __restrict__ row_kind,
                                                              const double* __restrict__ A_, const double* __restrict__ Bc_, const double* __restrict__ elem,
                                                              double* __restrict__ bnd, const double* __restrict__ disc, double tol)
{
    if (disc && !(disc[blockIdx.x] > tol)) return;
    __shared__ double tr[16][17];
    __shared__ int inv[16];
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x;
    const bool live = lane < R;
    const int r = live ? lane : 0;
    double* bs = bnd + b * nseg * TP_BND_DOUBLES;
    double P[R], m = 0.0;
    {
        const int kr = row_kind[r];
#pragma unroll
        for (int c = 0; c < R; ++c) {
            double v = 0.0;
            const int kc = row_kind[c];
            if (live && kr != 3 && kc != 3) {
                if (r == c) v = A_[b * J + row_term[r]];
                else if ((r ^ 1) == c && kr < 2 && kc < 2) v = -Bc_[b * J + row_term[r]];
            }
            P[c] = v;
            if (live) bs[64 + r * 64 + c] = v;
        }
    }
    struct Elem { double A[R], C[R], Jm[R], b, eta; };      // this lane's ROW of A, C, J (C, J symmetric: read as columns, coalesced)
    auto load = [&](Elem& E, const double* e) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
            E.A[k] = e[k * 64 + r];                 // A[r][k] = At[k][r]
            E.C[k] = e[4096 + k * 64 + r];
            E.Jm[k] = e[8192 + k * 64 + r];
        }
        E.b = e[12288 + r];
        E.eta = e[12288 + 64 + r];
    };
    Elem cur, nxt;
    if (nseg > 1) load(cur, elem + b * nseg * TP_ELEM_DOUBLES);
    nxt = cur;
    for (int p = 0; p + 1 < nseg; ++p) {
        if (p + 2 < nseg) load(nxt, elem + (b * nseg + p + 1) * TP_ELEM_DOUBLES);
        // X = [W | z | Z] with W = I + P J, z = m + P eta, Z = P: this lane's row
        double X[2 * R + 1];
        {
            double z = m;
#pragma unroll
            for (int c = 0; c < R; ++c) X[c] = lane == c ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const double pk = P[k];
                z = fma(pk, tp_readlane(cur.eta, k), z);
#pragma unroll
                for (int c = 0; c < R; ++c) X[c] = fma(pk, tp_readlane(cur.Jm[c], k), X[c]);      // J[k][c]: lane k's entry c
            }
            X[R] = z;
#pragma unroll
            for (int c = 0; c < R; ++c) X[R + 1 + c] = P[c];
        }
        bool used = !live;
        int mycol = 0;
        double mypiv = 1.0;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double cand = used ? -1.0 : fabs(X[k]);
            double mx = cand;
            mx = fmax(mx, tp_dpp<0xB1>(mx));
            mx = fmax(mx, tp_dpp<0x4E>(mx));
            mx = fmax(mx, tp_dpp<0x141>(mx));
            mx = fmax(mx, tp_dpp<0x140>(mx));
            mx = tp_readlane(mx, 0);                       // (the live lanes sit in the first row of 16)
            const unsigned long long bal = __ballot(cand == mx);
            const int pr = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(bal));
            const double ipv = tp_rcp(tp_readlane(X[k], pr));
            const double f = (lane == pr || !live) ? 0.0 : X[k] * ipv;
            if (lane == pr) { used = true; mycol = k; mypiv = ipv; }
#pragma unroll
            for (int c = k + 1; c < 2 * R + 1; ++c) X[c] = fma(-f, tp_readlane(X[c], pr), X[c]);
        }
        // the row this lane solved is row `mycol` of [z | Z]: back into natural order through LDS
        if (live) inv[mycol] = lane;
        const int src = inv[r];
        double zn = __shfl(X[R] * mypiv, src);
        double Z[R];
#pragma unroll
        for (int c = 0; c < R; ++c) Z[c] = __shfl(X[R + 1 + c] * mypiv, src);
        // m' = A z + b;  T = A Z;  P' = T A' + C
        double mn = cur.b, T[R], Pn[R];
#pragma unroll
        for (int c = 0; c < R; ++c) T[c] = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double ak = cur.A[k];
            mn = fma(ak, tp_readlane(zn, k), mn);
#pragma unroll
            for (int c = 0; c < R; ++c) T[c] = fma(ak, tp_readlane(Z[c], k), T[c]);
        }
#pragma unroll
        for (int c = 0; c < R; ++c) {
            double acc = cur.C[c];
#pragma unroll
            for (int k = 0; k < R; ++k) acc = fma(T[k], tp_readlane(cur.A[k], c), acc);          // A[c][k]: lane c's entry k
            Pn[c] = acc;
        }
        // symmetrise (the two products round differently): the transpose through LDS
        if (live) {
#pragma unroll
            for (int c = 0; c < R; ++c) tr[r][c] = Pn[c];
        }
        double* bo = bs + (int64_t)(p + 1) * TP_BND_DOUBLES;
#pragma unroll
        for (int c = 0; c < R; ++c) {
            P[c] = live ? 0.5 * (Pn[c] + tr[c][r]) : 0.0;
            if (live) bo[64 + r * 64 + c] = P[c];
        }
        m = live ? mn : 0.0;
        if (live) bo[r] = mn;
        cur = nxt;
    }
}

template <int NP, int NWV>
int tp_launch(const ScanParams& p, int RP, int nseg, int64_t L, const int32_t* row_term, const int32_t* row_kind, double* work, hipStream_t stream, int scan)
{
    const int64_t B = p.B, N = p.N;
    TpRec* rec = reinterpret_cast<TpRec*>(work);
    TpStep* stp = reinterpret_cast<TpStep*>(work + (size_t)B * N * RP * 6);
    double* sval = work + (size_t)B * N * RP * 6 + (size_t)B * N * 4;
    double* elem = sval + (size_t)B * N;
    double* elem2 = elem + (size_t)B * nseg * TP_ELEM_DOUBLES;       // the scan's two element buffers (its levels ping-pong; the raw elements stay for the check)
    double* elem3 = elem2 + (size_t)B * nseg * TP_ELEM_DOUBLES;
    double* bnd = elem3 + (size_t)B * nseg * TP_ELEM_DOUBLES;
    double* part = bnd + (size_t)B * nseg * TP_BND_DOUBLES;
    double* disc = part + (size_t)B * nseg * 4;                      // [B] per draw: the scan's check (product path: the estimate of log L's relative error; walk-repair mode: the state discrepancy)
    if (hipMemsetAsync(bnd, 0, (size_t)B * nseg * TP_BND_DOUBLES * sizeof(double), stream) != hipSuccess) return PIORAN_ERR_HIP;
    hipLaunchKernelGGL(tp_records_kernel, dim3((unsigned)N, (unsigned)B), dim3(64), 0, stream, N, RP, p.J, row_term, row_kind, p.t, p.y, p.s2, p.Y, p.S2, p.A,
                       p.Bc, p.C, p.D, p.mu, p.nu, rec, stp);
    if (nseg > 1)
        hipLaunchKernelGGL((tp_element_kernel<NP, NWV>), dim3((unsigned)(nseg - 1), (unsigned)B), dim3(64 * NWV), 0, stream, N, RP, nseg, L, row_kind,
                           (const TpRec*)rec, (const TpStep*)stp, elem);
    const size_t r16 = (size_t)((RP + 15) / 16) * 16, lds2 = (r16 * (2 * r16 + 3) + 2 * r16 * (r16 + 1) + 192 + 1024) * sizeof(double);
    static size_t granted[8][64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return PIORAN_ERR_HIP;
    double* filter_disc = nullptr;
    if (scan == 4) {
        // the boundary WALK with the filter's check behind it (round 6, late): the walk's states are sequential, but a long segment's element is not better conditioned than
        // a composite of the scan — on prior draws of DRWCelerite-10 the walk alone is off by up to 8e-7 (2e-6 at 32 segments) where the serial chain holds 4e-10
        // (tools/tp_walk_accuracy.py) — so its draws are checked and repaired like the scan's
        filter_disc = disc;
        if (hipMemsetAsync(disc, 0, (size_t)B * sizeof(double), stream) != hipSuccess) return PIORAN_ERR_HIP;
    }
    if (scan && scan != 4 && nseg >= 2 && pioran_tp_scan_rows(RP)) {
        // phase 2 as a scan: ceil(log2 nseg) launches of tp_combine_kernel, one workgroup per (draw, target)
        const int rt = (RP + 15) / 16;
        const bool lean = rt == 4 || (p.opt && p.opt->tp_scan_lean);          // (four tiles of rows: only the lean form fits the LDS)
        const size_t r16s = (size_t)rt * 16;
        // wavefronts per combination: eight from 17 rows on (three tiles of rows: nine product tiles and eight column tiles per elimination round — 48.9 -> 40.2 us;
        // four tiles, lean: sixteen and twelve)
        const int tw = rt >= 2 && !(p.opt && p.opt->tp_scan_waves == 4) ? 8 : 4;
        const size_t ldsc = lean ? (r16s * (3 * r16s + 3) + r16s * (r16s + 1) + 7 * 64 + 32 + 256 * tw) * sizeof(double)
                                 : (r16s * (3 * r16s + 3) + 4 * r16s * (r16s + 1) + 7 * 64 + 256 * tw) * sizeof(double);
        const void* fnc = nullptr;
#define TP_PICK(KERNEL) (rt == 1 ? (const void*)KERNEL<1, 4> : (rt == 2 ? (tw == 8 ? (const void*)KERNEL<2, 8> : (const void*)KERNEL<2, 4>) : (rt == 3 ? (tw == 8 ? (const void*)KERNEL<3, 8> : (const void*)KERNEL<3, 4>) : (tw == 8 ? (const void*)KERNEL<4, 8> : (const void*)KERNEL<4, 4>))))
        if (lean) fnc = TP_PICK(tp_combine_lean_kernel);
        else if (rt <= 3) fnc = rt == 1 ? (const void*)tp_combine_kernel<1, 4> : (rt == 2 ? (tw == 8 ? (const void*)tp_combine_kernel<2, 8> : (const void*)tp_combine_kernel<2, 4>) : (tw == 8 ? (const void*)tp_combine_kernel<3, 8> : (const void*)tp_combine_kernel<3, 4>));
#undef TP_PICK
        if (!fnc) return PIORAN_ERR_UNSUPPORTED;
        static size_t granted_c[4][5][64] = {};
        const int gi = (lean ? 2 : 0) + (tw == 8 ? 1 : 0);
        if (ldsc > granted_c[gi][rt][dev]) {
            if (hipFuncSetAttribute(fnc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsc) != hipSuccess) return PIORAN_ERR_HIP;
            granted_c[gi][rt][dev] = ldsc;
        }
        const double tol = p.opt && p.opt->tp_scan_tol != 0.0 ? p.opt->tp_scan_tol : 1e-6;      // (walk-repair mode: of the verification launch's state discrepancy)
        if (hipMemsetAsync(disc, 0, (size_t)B * sizeof(double), stream) != hipSuccess) return PIORAN_ERR_HIP;
        auto combine = [&](const dim3& gr, int stride, const double* src, double* dst, double* dsc) {
#define TP_COMBINE(KERNEL, WV) hipLaunchKernelGGL(KERNEL, gr, dim3(64 * WV), ldsc, stream, RP, nseg, p.J, stride, row_term, row_kind, p.A, p.Bc, src, dst, bnd, dsc)
#define TP_COMBINE_RT(KERNEL, RR) do { if (tw == 8) TP_COMBINE((KERNEL<RR, 8>), 8); else TP_COMBINE((KERNEL<RR, 4>), 4); } while (0)
            if (lean) {
                if (rt == 1) TP_COMBINE((tp_combine_lean_kernel<1, 4>), 4); else if (rt == 2) TP_COMBINE_RT(tp_combine_lean_kernel, 2);
                else if (rt == 3) TP_COMBINE_RT(tp_combine_lean_kernel, 3); else TP_COMBINE_RT(tp_combine_lean_kernel, 4);
            } else {
                if (rt == 1) TP_COMBINE((tp_combine_kernel<1, 4>), 4); else if (rt == 2) TP_COMBINE_RT(tp_combine_kernel, 2); else TP_COMBINE_RT(tp_combine_kernel, 3);
            }
#undef TP_COMBINE_RT
#undef TP_COMBINE
        };
        const double* src = elem;
        double* dst = elem2;
        for (int stride = 1; stride < nseg; stride *= 2) {
            combine(dim3((unsigned)(nseg - stride), (unsigned)B), stride, src, dst, nullptr);
            src = dst;
            dst = dst == elem2 ? elem3 : elem2;
        }
        // the check and what happens to the draws that fail it
        if (scan == 2) {
            // the caller repairs the draws that fail the check itself (capi.hip tp_dispatch: the serial-chain kernel with ScanParams::only_if), so the check can
            // wait for the filter, which makes it on its way (tp_filter_kernel's `disc`) — no verification launch.  Otherwise: one boundary step per boundary
            // from the scan's states, all at once (tp_combine_kernel's verify mode), and the walk for the draws that fail it, before the filter runs
            filter_disc = disc;
        } else if (RP <= 16) {
            combine(dim3((unsigned)(nseg - 1), (unsigned)B), 1, elem, nullptr, disc);
#define TP_WAVE_CASE(RR) case RR: hipLaunchKernelGGL((tp_boundary_wave_kernel<RR>), dim3((unsigned)B), dim3(64), 0, stream, nseg, p.J, row_term, row_kind, p.A, p.Bc, (const double*)elem, bnd, (const double*)disc, tol); break;
            switch (RP) { TP_WAVE_CASE(8) TP_WAVE_CASE(16) default: return PIORAN_ERR_UNSUPPORTED; }
#undef TP_WAVE_CASE
        } else {
            combine(dim3((unsigned)(nseg - 1), (unsigned)B), 1, elem, nullptr, disc);
            const void* fn = rt == 2 ? (const void*)tp_boundary_kernel<4, 2> : (rt == 3 ? (const void*)tp_boundary_kernel<4, 3> : (const void*)tp_boundary_kernel<4, 4>);
            if (lds2 > granted[rt][dev]) {
                if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess) return PIORAN_ERR_HIP;
                granted[rt][dev] = lds2;
            }
            if (rt == 2)
                hipLaunchKernelGGL((tp_boundary_kernel<4, 2>), dim3((unsigned)B), dim3(256), lds2, stream, RP, nseg, p.J, row_term, row_kind, p.A, p.Bc, (const double*)elem, bnd, (const double*)disc, tol);
            else if (rt == 3)
                hipLaunchKernelGGL((tp_boundary_kernel<4, 3>), dim3((unsigned)B), dim3(256), lds2, stream, RP, nseg, p.J, row_term, row_kind, p.A, p.Bc, (const double*)elem, bnd, (const double*)disc, tol);
            else
                hipLaunchKernelGGL((tp_boundary_kernel<4, 4>), dim3((unsigned)B), dim3(256), lds2, stream, RP, nseg, p.J, row_term, row_kind, p.A, p.Bc, (const double*)elem, bnd, (const double*)disc, tol);
        }
    } else if (RP == 2)
        hipLaunchKernelGGL((tp_boundary_small_kernel<2>), dim3((unsigned)((B + 63) / 64)), dim3(64), 0, stream, nseg, p.J, B, row_term, row_kind, p.A, p.Bc, (const double*)elem, bnd);
    else if (RP == 4)
        hipLaunchKernelGGL((tp_boundary_small_kernel<4>), dim3((unsigned)((B + 63) / 64)), dim3(64), 0, stream, nseg, p.J, B, row_term, row_kind, p.A, p.Bc, (const double*)elem, bnd);
    else if (RP <= 16) {
#define TP_WAVE_CASE(RR) case RR: hipLaunchKernelGGL((tp_boundary_wave_kernel<RR>), dim3((unsigned)B), dim3(64), 0, stream, nseg, p.J, row_term, row_kind, p.A, p.Bc, (const double*)elem, bnd, (const double*)nullptr, 0.0); break;
        switch (RP) {
            TP_WAVE_CASE(6) TP_WAVE_CASE(8) TP_WAVE_CASE(10) TP_WAVE_CASE(12) TP_WAVE_CASE(14) TP_WAVE_CASE(16)
            default: return PIORAN_ERR_UNSUPPORTED;
        }
#undef TP_WAVE_CASE
    } else {
        // four wavefronts, RT = 2 .. 4 tiles of 16 rows (24 .. 64 state rows)
        const int rt = (RP + 15) / 16;
        const void* fn = rt == 2 ? (const void*)tp_boundary_kernel<4, 2> : (rt == 3 ? (const void*)tp_boundary_kernel<4, 3> : (const void*)tp_boundary_kernel<4, 4>);
        if (lds2 > granted[rt][dev]) {
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess) return PIORAN_ERR_HIP;
            granted[rt][dev] = lds2;
        }
        if (rt == 2)
            hipLaunchKernelGGL((tp_boundary_kernel<4, 2>), dim3((unsigned)B), dim3(256), lds2, stream, RP, nseg, p.J, row_term, row_kind, p.A, p.Bc, (const double*)elem, bnd, (const double*)nullptr, 0.0);
        else if (rt == 3)
            hipLaunchKernelGGL((tp_boundary_kernel<4, 3>), dim3((unsigned)B), dim3(256), lds2, stream, RP, nseg, p.J, row_term, row_kind, p.A, p.Bc, (const double*)elem, bnd, (const double*)nullptr, 0.0);
        else
            hipLaunchKernelGGL((tp_boundary_kernel<4, 4>), dim3((unsigned)B), dim3(256), lds2, stream, RP, nseg, p.J, row_term, row_kind, p.A, p.Bc, (const double*)elem, bnd, (const double*)nullptr, 0.0);
    }
    const int check = p.opt ? p.opt->tp_check : 0;
    hipLaunchKernelGGL((tp_filter_kernel<NP, NWV>), dim3((unsigned)nseg, (unsigned)B), dim3(64 * NWV), 0, stream, N, RP, nseg, L, row_kind,
                       (const TpRec*)rec, (const TpStep*)stp, (const double*)bnd, part, sval, filter_disc, check);
    hipLaunchKernelGGL(tp_finish_kernel, dim3((unsigned)B), dim3(64), 0, stream, N, nseg, B, (const double*)part, p.out, p.status, check != 1 ? filter_disc : (double*)nullptr, check);      // (tp_check = 2 / 3: the raw distance on the innovation scale / the estimate alone: tools)
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}

}  // namespace

const double* pioran_tp_disc(const double* work, int64_t B, int64_t N, int RP, int nseg)
{
    return work + (size_t)B * N * RP * 6 + (size_t)B * N * 5 + (size_t)B * nseg * (3 * (size_t)TP_ELEM_DOUBLES + TP_BND_DOUBLES + 4);
}
double pioran_tp_scan_tol(const ScanOptions* opt) { return opt && opt->tp_scan_tol != 0.0 ? opt->tp_scan_tol : kTpScanTol; }   // (negative: every draw is repaired — tests)

#ifdef PIORAN_TP_STAMP
extern "C" int pioran_tp_read_stamps(unsigned long long* out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(tp_stamp_buf), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -1;
}
#endif

int pioran_tp_supported_rows() { return 64; }     // (lane = state row in the element and filter kernels)
// state rows (padded) whose boundary phase can run as a scan (tp_combine_kernel up to three tiles of 16 rows, tp_combine_lean_kernel at four: a multiple of 8)
int pioran_tp_scan_rows(int RP) { return RP >= 8 && RP <= 64 && RP % 8 == 0; }

// state rows as the kernels want them: a multiple of 2 up to 12 rows (one wavefront per segment), of 8 above (four; at 16 rows four are 9 % ahead of one)
int pioran_tp_padded_rows(int rows) { return rows <= 12 ? (rows + 1) & ~1 : (rows + 7) & ~7; }

size_t pioran_tp_workspace_doubles(int64_t B, int64_t N, int RP, int nseg)
{
    return (size_t)B * N * RP * 6 + (size_t)B * N * 5 + (size_t)B * nseg * (3 * (size_t)TP_ELEM_DOUBLES + TP_BND_DOUBLES + 4) + (size_t)B;
}

// RP = pioran_tp_padded_rows(rows) state rows in the layout of row_term / row_kind (device arrays, [RP]; kind 3 = padding); nseg segments of L steps
// (the last one shorter)
int pioran_launch_tp(const ScanParams& p, int RP, int nseg, int64_t L, const int32_t* row_term, const int32_t* row_kind, double* work, hipStream_t stream, int scan)
{
    if (RP < 2 || RP > 64 || RP != pioran_tp_padded_rows(RP) || nseg < 1 || L < 2 || (int64_t)nseg * L < p.N || (int64_t)(nseg - 1) * L >= p.N || p.B < 1 ||
        p.B > 65535 || p.N > 0x7fffffffLL)
        return PIORAN_ERR_UNSUPPORTED;
    if (RP <= 12) {
        switch (RP / 2) {
            case 1: return tp_launch<1, 1>(p, RP, nseg, L, row_term, row_kind, work, stream, scan);
            case 2: return tp_launch<2, 1>(p, RP, nseg, L, row_term, row_kind, work, stream, scan);
            case 3: return tp_launch<3, 1>(p, RP, nseg, L, row_term, row_kind, work, stream, scan);
            case 4: return tp_launch<4, 1>(p, RP, nseg, L, row_term, row_kind, work, stream, scan);
            case 5: return tp_launch<5, 1>(p, RP, nseg, L, row_term, row_kind, work, stream, scan);
            case 6: return tp_launch<6, 1>(p, RP, nseg, L, row_term, row_kind, work, stream, scan);
            case 7: return tp_launch<7, 1>(p, RP, nseg, L, row_term, row_kind, work, stream, scan);
            case 8: return tp_launch<8, 1>(p, RP, nseg, L, row_term, row_kind, work, stream, scan);
        }
        return PIORAN_ERR_UNSUPPORTED;
    }
    switch (RP / 8) {
        case 2: return tp_launch<2, 4>(p, RP, nseg, L, row_term, row_kind, work, stream, scan);
        case 3: return tp_launch<3, 4>(p, RP, nseg, L, row_term, row_kind, work, stream, scan);
        case 4: return tp_launch<4, 4>(p, RP, nseg, L, row_term, row_kind, work, stream, scan);
        case 5: return tp_launch<5, 4>(p, RP, nseg, L, row_term, row_kind, work, stream, scan);
        case 6: return tp_launch<6, 4>(p, RP, nseg, L, row_term, row_kind, work, stream, scan);
        case 7: return tp_launch<7, 4>(p, RP, nseg, L, row_term, row_kind, work, stream, scan);
        case 8: return tp_launch<8, 4>(p, RP, nseg, L, row_term, row_kind, work, stream, scan);
    }
    return PIORAN_ERR_UNSUPPORTED;
}
