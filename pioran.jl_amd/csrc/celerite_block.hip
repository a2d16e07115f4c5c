// Windowed ("block") form of the celerite factorisation for SMALL batches (gfx950): one draw per workgroup of four
// wavefronts, sixteen time steps per window, the O(R^2) work of a window on the matrix cores.
//
// Same mathematics as init_semi_separable! + the forward half of solve_prec! (src/celerite_solver.jl:12-100,115-142; logl
// :312-334), re-associated.  The reference (and celerite_scan.hip / celerite_wide.hip) walk the time steps one by one:
//   S_n = (phi phi') o (S_{n-1} + D_{n-1} w w') ;  q = S_n u_n ;  D_n = sum(a) + sigma2_n - u_n'q ;  w_n = (v_n - q) / D_n
// (:69-98): a chain of N dependent steps, each with a reduction and a reciprocal — 1400 cycles per step in the latency layout.
// Here the state is T_m = S_m + D_m w_m w_m' (time-m coordinates) and K = 16 steps m+1 .. m+K are eliminated together:
//   C_n   = prod_{i=m+1..n} phi_i                       (cumulative decay from the window base; table)
//   u~_n  = C_n o u_n ,  U~ = [u~_n]                    R x K
//   M     = T U~                                        R x K     (MFMA)      the past, seen from every step of the window
//   G     = U~' M                                       K x K     (MFMA)
//   Sigma = A - G ,  A_jn = k(t_n - t_j), A_nn = sum(a) + sigma2_n            the window's own covariance block (kappa,
//                                                                             src/acvf.jl:138-140) minus what the past explains
//   Sigma = L D L'                                      16 x 16 dense LDL' — D holds exactly the reference's D_n (:92)
//   X     = V^ - C_K o M ,  V^_n = (C_K / C_n) o v_n    R x K     (v scaled forward to the window end)
//   Y^    = X L^-T                                      R x K     (MFMA)      = [ w_n D_n ] in window-end coordinates
//   T    <- (C_K C_K') o T + Y^ D^-1 Y^'                R x R     (MFMA)
// and, with y carried as one more row (u = 0, v = y_n - mu, phi = 1; as in the other kernels), z_n = Y^_{y,n}, so that
//   log L = -1/2 sum log|D_n| - 1/2 sum Y^_{y,n}^2 / D_n - N/2 log(2 pi).
// Every scaling factor is a product of phi's (<= 1): nothing is ever divided by a decay, so terms whose phi underflows inside a
// window (c_j dt >> 700) are exact zeros, not NaNs.  The only place where a pairwise decay C_n / C_j (j < n) is needed is the
// window's own block A, and that is the plain kernel function: a shared table holds e^{-c tau}(cos, sin)(d tau) per (pair, term),
// each draw contracts it with its (a_j, b_j).
//
// Work split (R + 1 <= 16 NB rows, NB <= 4): wavefront w < NB owns block column w of T (C/D register layout of
// v_mfma_f64_16x16x4_f64: register g of block (I, J) = element (16 I + 4 g + (lane >> 4), 16 J + (lane & 15))).  Those registers
// are the B operand of M' = U~' T and the accumulator of the update without any data movement; M' comes out in the layout in
// which X, Y^ and the update's operands are needed; each owner adds its own Gram partial U~_w' M_w (one transposing LDS round
// trip inside the wavefront).  One more wavefront (the last: 3, or 4 when NB = 4) runs the chain: Sigma from the partials, the
// 16 x 16 LDL' as an in-place Gauss-Jordan with every column replicated in the four DPP rows (pivot broadcasts and rank-1
// updates are v_mov_b64_dpp / v_fmac_f64_dpp row_newbcast: no LDS inside the factorisation; L^-1 comes out of the same
// instructions).  Three workgroup barriers per window.  While the chain runs, the owners rescale T, form A and U~ of the NEXT
// window and copy record k + 2 of the table into LDS (global_load_lds_dwordx4: the table is stored in fragment order — one
// 8-byte element per lane and MFMA register — so a record is a plain copy in 1 KB pieces, lands a whole window ahead, costs no
// registers, and every operand read is a conflict-free ds_read_b64).
// Numerical notes: the only division-like operation is 1 / D_n (v_rcp_f64 + two Newton steps, 1 ulp); a non-positive D_n is
// carried on like the reference does (log|D_n|, :140; status 1), NaN / inf surface as status 2.
#include "common.h"
#include "window_common.h"

#include <cmath>
#include <type_traits>

// Diagnostic hooks: compiled out in the product; tools/block_probe.hip defines them to s_memtime accumulators.
#ifndef PIORAN_ASTAMP
// (celerite_block_adjoint_kernel: the phase boundaries are scheduling fences in the product as well — left to itself the compiler moves the
//  loads that are issued in one phase to land during the next back to their uses: 6.2 instead of 4.7 ms per 625 windows)
#define PIORAN_ASTAMP(i) __builtin_amdgcn_sched_barrier(0)
#define PIORAN_ASTAMP_DECL
#define PIORAN_ASTAMP_FLUSH
#endif
#ifndef PIORAN_BSTAMP
#define PIORAN_BSTAMP(i)
#define PIORAN_BSTAMP_DECL
#define PIORAN_BSTAMP_FLUSH
#endif

namespace {


__global__ void __launch_bounds__(256) block_table_kernel(int64_t N, int32_t R, int32_t J, int32_t NB, const int32_t* __restrict__ rowmap,
                                                          const double* __restrict__ t, const double* __restrict__ c,
                                                          const double* __restrict__ d, const double* __restrict__ y,
                                                          const double* __restrict__ s2, double* __restrict__ tab)
{
    const int64_t RSB = block_rec_doubles(NB, J);
    const int64_t NW = (N + KW - 1) / KW;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= NW * RSB) return;
    const int64_t k = idx / RSB;
    int64_t e = idx - k * RSB;
    const int64_t n0 = k * KW;
    const int64_t nlast = n0 + KW - 1 < N ? n0 + KW - 1 : N - 1;
    const double tb = k > 0 ? t[n0 - 1] : t[0];    // window base (irrelevant for k = 0: T = 0)
    const double te = t[nlast];                    // window end
    double val = 0.0;
    const int64_t nfrag = (int64_t)NB * 256;
    if (e < 3 * nfrag) {
        const int sec = (int)(e / nfrag);
        const int f = (int)(e - sec * nfrag);
        const int blk = f >> 8, reg = (f >> 6) & 3, lane = f & 63;
        int row, s;
        if (sec < 2) { row = 16 * blk + 4 * reg + (lane >> 4); s = lane & 15; }
        else { s = 4 * reg + (lane >> 4); row = 16 * blk + (lane & 15); }
        const int64_t n = n0 + s;
        if (n < N) {
            if (row < R) {
                const int32_t rm = rowmap[row];
                const int32_t term = rm & 0xfffff;
                const bool ks = (rm >> 30) & 1;
                const double tn = t[n];
                double si, co;
                sincos(d[term] * tn, &si, &co);                                  // :52-53
                if (sec == 2) val = (ks ? si : co) * exp(-c[term] * (te - tn));
                else val = ((sec == 0) == ks ? si : co) * exp(-c[term] * (tn - tb));   // sec 0: v, sec 1: x
            } else if (row == R && sec == 2) {
                val = y[n];
            }
        }
    } else if ((e -= 3 * nfrag) < 16 * NB) {
        const int row = (int)e;
        if (row < R) val = exp(-c[rowmap[row] & 0xfffff] * (te - tb));
        else if (row == R) val = 1.0;
    } else if ((e -= 16 * NB) < 16) {
        const int64_t n = n0 + e;
        val = n < N ? s2[n] : 1.0;
    } else if ((e -= 16) < RSB - (int64_t)J * 256 - 3 * nfrag - 16 * NB - 16) {
        val = 0.0;   // padding of the tile part
    } else {
        e -= RSB - (int64_t)J * 256 - 3 * nfrag - 16 * NB - 16;
        const int term = (int)(e >> 8), p = (int)((e >> 1) & 127), h = (int)(e & 1);
        if (p < 120) {
            int nn = 1;
            while ((nn + 1) * nn / 2 <= p) ++nn;       // p = nn (nn - 1) / 2 + jj, jj < nn
            const int jj = p - nn * (nn - 1) / 2;
            if (n0 + nn < N) {
                // (cos, sin)(d tau) by angle addition from the SAME rounded (cos, sin)(d t) that the u / v fragments above are
                // made of, not from d * tau directly: the window's own covariance block A must be consistent with the Gram
                // block G = U~' T U~ it is subtracted from.  The phase d * t_n (up to 1e4 .. 1e7 rad) is rounded to ~1e-12 rad;
                // the step-by-step recurrence (and the reference) use those rounded phases everywhere, which is a harmless
                // perturbation of the time stamps — mixing them with the exactly-differenced d * tau left an O(1e-12 a)
                // mismatch in A - G, i.e. 1e-8 relative in D_n where near-coincident time stamps make D_n ~ 1e-4 a
                // (tools/explain_outliers.py).
                const double tn = t[n0 + nn], tj = t[n0 + jj];
                double sn, cn, sj, cj;
                sincos(d[term] * tn, &sn, &cn);
                sincos(d[term] * tj, &sj, &cj);
                const double trig = h ? fma(sn, cj, -cn * sj) : fma(cn, cj, sn * sj);
                val = exp(-c[term] * (tn - tj)) * trig;
            }
        }
    }
    tab[idx] = val;
}

// The same record, one workgroup per (window, draw): the transcendentals a window needs are evaluated ONCE — (cos, sin)(d t_n) and the
// two decays per (term, step), e^{-c tau} per (term, pair) — and every entry is a product of those (round 3; block_table_kernel above
// re-evaluates two sincos and an exponential per ENTRY: 44 us per table at N = 1e4, J = 20; this one 8x fewer instructions).  Same
// expressions on the same arguments, so the two tables are bit-identical (tests/test_gpu_parity.py compares them).  blockIdx.y = draw
// of a batch of per-draw tables: (c, d) at c + y J, table at tab + y tab_draw_stride.
__global__ void __launch_bounds__(256) block_table_window_kernel(int64_t N, int32_t R, int32_t J, int32_t NB, const int32_t* __restrict__ rowmap,
                                                                 const double* __restrict__ t, const double* __restrict__ c,
                                                                 const double* __restrict__ d, const double* __restrict__ y,
                                                                 const double* __restrict__ s2, double* __restrict__ tab, int64_t cd_stride,
                                                                 int64_t tab_draw_stride)
{
    __shared__ double cs[64 * 16], sn[64 * 16], Cn[64 * 16], Hn[64 * 16], ckt[64], tt[16];
    c += (int64_t)blockIdx.y * cd_stride;
    d += (int64_t)blockIdx.y * cd_stride;
    tab += (int64_t)blockIdx.y * tab_draw_stride;
    const int64_t RSB = block_rec_doubles(NB, J);
    const int64_t k = blockIdx.x;
    const int64_t n0 = k * KW;
    const int64_t nlast = n0 + KW - 1 < N ? n0 + KW - 1 : N - 1;
    const double tb = k > 0 ? t[n0 - 1] : t[0];    // window base (irrelevant for k = 0: T = 0)
    const double te = t[nlast];                    // window end
    for (int it = threadIdx.x; it < J * 16; it += 256) {
        const int term = it >> 4, s = it & 15;
        const int64_t n = n0 + s;
        double si = 0.0, co = 0.0, cn = 0.0, hn = 0.0;
        if (n < N) {
            const double tn = t[n];
            sincos(d[term] * tn, &si, &co);                                  // :52-53
            cn = exp(-c[term] * (tn - tb));
            hn = exp(-c[term] * (te - tn));
        }
        cs[it] = co; sn[it] = si; Cn[it] = cn; Hn[it] = hn;
        if (s == 0) ckt[term] = exp(-c[term] * (te - tb));
    }
    if (threadIdx.x < 16) tt[threadIdx.x] = n0 + threadIdx.x < N ? t[n0 + threadIdx.x] : 0.0;
    __syncthreads();
    const int64_t nfrag = (int64_t)NB * 256;
    const int64_t tile_end = RSB - (int64_t)J * 256;
    double* rec = tab + k * RSB;
    for (int64_t e0 = threadIdx.x; e0 < RSB; e0 += 256) {
        int64_t e = e0;
        double val = 0.0;
        if (e < 3 * nfrag) {
            const int sec = (int)(e / nfrag);
            const int f = (int)(e - sec * nfrag);
            const int blk = f >> 8, reg = (f >> 6) & 3, lane = f & 63;
            int row, s;
            if (sec < 2) { row = 16 * blk + 4 * reg + (lane >> 4); s = lane & 15; }
            else { s = 4 * reg + (lane >> 4); row = 16 * blk + (lane & 15); }
            const int64_t n = n0 + s;
            if (n < N) {
                if (row < R) {
                    const int32_t rm = rowmap[row];
                    const int it = (rm & 0xfffff) * 16 + s;
                    const bool ks = (rm >> 30) & 1;
                    if (sec == 2) val = (ks ? sn[it] : cs[it]) * Hn[it];
                    else val = ((sec == 0) == ks ? sn[it] : cs[it]) * Cn[it];   // sec 0: v, sec 1: x
                } else if (row == R && sec == 2) {
                    val = y[n];
                }
            }
        } else if ((e -= 3 * nfrag) < 16 * NB) {
            const int row = (int)e;
            if (row < R) val = ckt[rowmap[row] & 0xfffff];
            else if (row == R) val = 1.0;
        } else if ((e -= 16 * NB) < 16) {
            const int64_t n = n0 + e;
            val = n < N ? s2[n] : 1.0;
        } else if (e0 < tile_end) {
            val = 0.0;   // padding of the tile part
        } else {
            e = e0 - tile_end;
            const int term = (int)(e >> 8), p = (int)((e >> 1) & 127), h = (int)(e & 1);
            if (p < 120) {
                int nn = 1;
                while ((nn + 1) * nn / 2 <= p) ++nn;       // p = nn (nn - 1) / 2 + jj, jj < nn
                const int jj = p - nn * (nn - 1) / 2;
                if (n0 + nn < N) {
                    const double cn_ = cs[term * 16 + nn], sn_ = sn[term * 16 + nn], cj = cs[term * 16 + jj], sj = sn[term * 16 + jj];
                    const double trig = h ? fma(sn_, cj, -cn_ * sj) : fma(cn_, cj, sn_ * sj);   // angle addition from the rounded phases (see above)
                    val = exp(-c[term] * (tt[nn] - tt[jj])) * trig;
                }
            }
        }
        rec[e0] = val;
    }
}

template <int NB>
struct BlockSharedT {               // exchanges between the wavefronts of a workgroup
    double MG[NB][16 * 18];         // per owner wavefront: its block of M' [step][row, stride 18] for the transposing read-back, then
                                    // (same storage) its partial Gram U~_J' M_J as fragments [g][lane]
    double Ab[2][NB >= 4 ? 2 : 1][256];   // A of window k in Ab[k & 1][0] (from four block columns on: the two term-parity partial sums [0] + [1])
    double Li[16 * 18];             // D_k (L^-1)_ik at [k * 18 + i], i > k; D_k at [k * 18 + k]; the rest is not L^-1 (readers mask).
                                    // Before the elimination the chain wavefront uses the same storage for Sigma [j][n] (its own
                                    // layout change; the readers of L^-1 of the previous window are two barriers behind)
    double Yt[NB * 256];            // Y^' fragments [J][g][lane]
    double fin[8];
    double2 ab[64];                 // (a_t, b_t) of this draw
    double2 albe[NB > 4 ? 16 * NB : 64];   // per row: u = al v + be x
    double ys[2][32];               // per-draw series: (y_n, sigma2_n) of window k in ys[k & 1][0..15 | 16..31]
};
constexpr int kBlockMaxTerms = 64;

// Per-draw rows (mixed mode: a few terms whose (c, d) differ per draw — QPO features on an approx continuum, src/psd.jl:254-261;
// round 3).  The host puts those rows LAST (rows R - 2 npd .. R - 1, cos row then sin row of each term; capi.hip build_rowmap), the
// shared table holds nothing useful for them, and the chain wavefront — idle between barrier 2 of a window and barrier 1 of the
// next — forms their entries of the record of window k + 2 in LDS from the per-draw (cos, sin)(d t_n) (a compact table written by
// pd_trig_kernel, so that the entries use the same rounded phases everywhere: see block_table_kernel) and the time stamps, both
// staged by LDS DMA a window earlier: four exponentials per lane and window.  The readers of a record (load_u, load_vh, form_A)
// take these rows' values from here instead of the tile.
constexpr int kBlockMaxPdTerms = 2;
struct BlockPd {
    double cv[2][2 * kBlockMaxPdTerms][16];   // [window parity][per-draw row][step]: C_n o v_n
    double cx[2][2 * kBlockMaxPdTerms][16];   //                                       C_n o x_n
    double vh[2][2 * kBlockMaxPdTerms][16];   //                                       (C_K / C_n) o v_n
    double ck[2][2 * kBlockMaxPdTerms];       // C_K
    // (followed in LDS by double2 E[2][npd terms][128]: e^{-c tau} (cos, sin)(d tau) per pair of the window — sized by the launch)
    double stage[2][24 + 32 * kBlockMaxPdTerms];   // [parity]: t[16 k - 2 .. 16 k + 17] (20, padded to 24), then (cos, sin)(d t_n) x 16 per term
    double2 ab[kBlockMaxPdTerms];             // (a, b) of the per-draw terms
    double c[kBlockMaxPdTerms];
    int term[kBlockMaxPdTerms];
};


// ---- the kernel --------------------------------------------------------------------------------------------------------
// LDS: two tile buffers (records k, k + 1), one E buffer, the exchange block.  Record k + 2 is copied in (LDS DMA) while the chain
// of window k runs and is first read after barrier 1 of window k + 1: a whole window of latency cover, no registers.
// Wavefronts: NB owners of a block column of T + the chain wavefront (with NB = 4 a fifth wavefront, so that the chain never
// queues behind an owner's MFMA work on the same instruction stream); never fewer than four (the pair contraction uses three).
// EDBL: E has two LDS buffers as well (when it fits): then every piece of record k + 2 may be copied by any wavefront once barrier 1
// of window k has passed, and the copies are spread over all wavefronts (a piece costs its issuer ~150 cycles); with a single E
// buffer a wavefront refills exactly the E pieces that only it reads.
// With four block columns (always a single E buffer) the workgroup is filled up to eight wavefronts with COPY wavefronts that do
// nothing but issue the LDS DMA of record k + 2 between the barriers (a piece costs its issuer ~150 cycles, ~100 pieces per window: too
// much for the computing wavefronts' slack).
// EM (E mode): 0 = one LDS buffer for the pair table E, 1 = two (EDBL above), 2 = none — form_A reads E of the next window straight
// from global memory (L2: every workgroup of the launch reads the same 256 J doubles at about the same time), which takes 41 KB
// (J = 20) out of the workgroup's LDS and two thirds out of its LDS DMA: two workgroups then share a CU (round 3; batches above 256 draws).
// PDM (per-draw rows): 0 none; 1 a helper wavefront prepares them (up to three block columns: five wavefronts, one workgroup per CU);
// 2 the chain wavefront does, in its idle stretch (four block columns; and above 256 draws, where two workgroups of FOUR wavefronts
// share a CU — with EM = 2 — and that is worth more than the helper)
// ST (gradient, round 3): the forward pass of the windowed reverse mode — per window it leaves in p.gw what the reverse pass
// (celerite_block_adjoint_kernel) needs: the state T at the window start, M', Q' = Sigma^-1 X' (both in C/D order and, Q, in A-operand
// order) and K = Sigma^-1 (block_grad_ws_doubles gives the layout).  The value it returns is bit-identical to the plain kernel's.
// (the last 320: the chain wavefront's D_k and D_k (L^-1)_ik, sh.Li as it stands — the simulation applies L with it)
__host__ __device__ constexpr int64_t block_grad_ws_doubles(int NB) { return (int64_t)NB * NB * 256 + 3 * (int64_t)NB * 256 + 256 + 320; }
// The prediction (ST = 2: Q in A-operand order) and the simulation (ST = 3: Q in C/D order, then sh.Li) keep only what they read, packed
// (late round 4; before, they used the reverse mode's 41 KB per window with 6 .. 9 KB of it written: 8.3 GB for 256 draws at N = 1e4, J = 20).
__host__ __device__ constexpr int64_t block_ws_doubles(int NB, int ST) { return ST == 2 ? (int64_t)NB * 256 : (ST == 3 ? (int64_t)NB * 256 + 320 : block_grad_ws_doubles(NB)); }
__host__ __device__ constexpr int block_ws_off_q(int NB, int ST) { return ST == 3 ? 0 : NB * NB * 256 + NB * 256; }
__host__ __device__ constexpr int block_ws_off_qf(int NB, int ST) { return ST == 2 ? 0 : NB * NB * 256 + 2 * NB * 256; }
__host__ __device__ constexpr int block_ws_off_li(int NB, int ST) { return ST == 3 ? NB * 256 : NB * NB * 256 + 3 * NB * 256 + 256; }
// Between a wavefront's LDS writes and its OWN reads of the same addresses (or the other way round) the LDS's in-order service of a wavefront is the ordering;
// only the compiler has to be kept from moving them (a counter wait stood here before)
#ifndef PIORAN_BLK_NOWAIT
#define PIORAN_BLK_NOWAIT 1      // 256 draws of SHO-20: 1.859 -> 1.834 ms, same box
#endif
#if PIORAN_BLK_NOWAIT
#define PIORAN_BLK_SAMEWAVE() asm volatile("" ::: "memory")
#else
#define PIORAN_BLK_SAMEWAVE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif
template <int NB, int EM, int PDM = 0, int ST = 0>   // ST: 0 no stores, 1 everything the reverse pass needs, 2 Q in A-operand order (prediction), 3 Q in C/D order and L^-1, D (simulation)
__global__ void __launch_bounds__(NB < 4 ? (PDM == 1 ? 320 : 256) : 512, (NB < 4 && PDM != 1 && (!ST || (ST == 1 && EM == 2))) ? 2 : 1) celerite_block_kernel(const ScanParams p, const double* __restrict__ btab)
{
    constexpr bool PD = PDM != 0;
    [[maybe_unused]] constexpr int64_t GWS = block_ws_doubles(NB, ST);
    [[maybe_unused]] constexpr int OFF_M = NB * NB * 256, OFF_Q = block_ws_off_q(NB, ST), OFF_QF = block_ws_off_qf(NB, ST), OFF_K = NB * NB * 256 + 3 * NB * 256,
                                   OFF_LI = block_ws_off_li(NB, ST);
    constexpr bool EDBL = EM == 1, EGLOB = EM == 2;
    constexpr int NCW = NB < 4 ? 4 : NB + 1;   // computing wavefronts: owners + chain (five and six block columns, round 4: 64 .. 95 rows,
                                               // value only — the reference grid's j = 32 is 64 rows + y)
    constexpr bool COPYW = NB >= 4;         // copy wavefronts (these shapes fill a CU's LDS with one workgroup anyway)
    static_assert(NB <= 6, "eight wavefronts: owners + chain + at least one copy wavefront");
    // per-draw rows, up to three block columns: one more wavefront forms their record entries (a whole window of time for ~4 exponentials
    // per lane: never on the critical path; on the chain wavefront the same work delayed barrier 1 of every window — 5.1 instead of
    // 3.3 us per window).  With four block columns the chain wavefront does it (the workgroup is full: copy wavefronts).
    constexpr bool HELPW = PDM == 1;
    static_assert(PDM != 1 || NB < 4, "the helper wavefront exists up to three block columns");
    constexpr int NWV = COPYW ? 8 : NCW + (HELPW ? 1 : 0);    // wavefronts per workgroup
    constexpr int CH = NCW - 1;             // the chain wavefront
    constexpr int TS = 3 * NB * 256 + 16 * NB + 16, TSP = (TS + 127) & ~127;
    extern __shared__ double lds_[];
    if constexpr (ST == 0) {
        if (p.only_if && !(p.only_if[blockIdx.x] > p.only_if_tol)) return;     // (the repair pass behind the time-parallel scan: workgroup-uniform, before any barrier)
    }
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform, and known to be
    const int lane = tid & 63, q = lane >> 4, c16 = lane & 15;
    const int64_t b = blockIdx.x;
    const int64_t N = p.N;
    const int J = p.J, R = p.R;
    const int64_t NW = (N + KW - 1) / KW;
    const int64_t RSB = TSP + 256 * (int64_t)J;
    btab += b * p.tab_draw_stride;                   // per-draw tables (every term's (c, d) per draw): 0 for the shared table
    double* const tileb = lds_;
    double* const Eb = lds_ + 2 * TSP;               // E(k) lives in Eb + (EDBL ? (k & 1) * 256 J : 0)   (EGLOB: not in LDS)
    const int ebs = EDBL ? 256 * J : 0;
    const int e_lds = EGLOB ? 0 : (EDBL ? 2 : 1) * 256 * J;   // doubles of LDS the pair table takes
    using BlockShared = BlockSharedT<NB>;
    BlockShared& sh = *reinterpret_cast<BlockShared*>(lds_ + 2 * TSP + e_lds);
    [[maybe_unused]] BlockPd& pd = *reinterpret_cast<BlockPd*>(reinterpret_cast<char*>(&sh) + sizeof(BlockShared));   // (allocated only for PD launches)
    [[maybe_unused]] double2* const pdE = reinterpret_cast<double2*>(reinterpret_cast<char*>(&pd) + sizeof(BlockPd));
    [[maybe_unused]] const int npd = PD ? p.npd_rows / 2 : 0, npdr = PD ? p.npd_rows : 0;
    [[maybe_unused]] const int pdr0 = p.R - npdr;                  // first per-draw row
    const bool chain = w == CH;
    [[maybe_unused]] const bool pdw = HELPW ? w == NCW : chain;   // the wavefront that prepares the per-draw rows
    const bool owner = w < NB;                                    // this wavefront owns block column w of T
    const bool has_u = owner || chain;
    const int Jy = R >> 4, ry = R & 15;                           // block column / lane column of the y row
    const bool ycol = owner && w == Jy && c16 == ry;
    const double mu = p.mu ? p.mu[b] : 0.0;
    const double nu = p.nu ? p.nu[b] : 1.0;
    const bool has_nu = p.nu != nullptr;
    const double* __restrict__ Ab_ = p.A + b * J;
    const double* __restrict__ Bb_ = p.Bc + b * J;
    double suma = 0.0;  // :21
    for (int j = 0; j < J; ++j) suma += Ab_[j];

    dma_pieces(btab, tileb, TSP / 128, w, NWV, lane);
    if constexpr (!EGLOB) dma_pieces(btab + TSP, Eb, 2 * J, w, NWV, lane);
    if (tid < J) sh.ab[tid] = double2{Ab_[tid], Bb_[tid]};

    // u = al v + be x per row (:59-63)
    if (tid < (NB > 4 ? 16 * NB : 64)) {
        double a = 0.0, bb = 0.0;
        if (tid < R) {
            const int rm = p.rowmap[tid];
            const int term = rm & 0xfffff;
            a = Ab_[term];
            bb = ((rm >> 30) & 1) ? -Bb_[term] : Bb_[term];
        }
        sh.albe[tid] = double2{a, bb};
    }

    if constexpr (PD) {
        if (tid < npd) {
            const int term = p.rowmap[pdr0 + 2 * tid] & 0xfffff;
            pd.term[tid] = term;
            pd.ab[tid] = double2{Ab_[term], Bb_[term]};
            pd.c[tid] = p.pd_C[b * J + term];
        }
    }

    d4 T[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) T[I] = d4{0.0, 0.0, 0.0, 0.0};
    double Uf[NB][4];
    double vh[4], ckc = 0.0, ckr[NB][4];

    // ---- per-draw rows: staging (LDS DMA) and the record entries of one window (chain wavefront) ----
    int pnn[2] = {1, 1}, pjj[2] = {0, 0};   // the two pairs (jj < nn) of this lane: p = lane, lane + 64
    if constexpr (PD) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pq = lane + 64 * h;
            int n_ = 1;
            while ((n_ + 1) * n_ / 2 <= pq) ++n_;
            pnn[h] = n_;
            pjj[h] = pq - n_ * (n_ - 1) / 2;
        }
    }
    [[maybe_unused]] auto stage_dma = [&](int64_t K) __attribute__((always_inline)) {
        double* dst = pd.stage[K & 1];
        const int64_t n0 = K * KW;
        if (lane < 10) {   // st[j] = t[n0 - 2 + j]: 16 bytes per lane from a 16-byte aligned address inside the series (what falls
                           // outside [0, N) is never used: the window base of window 0, steps past the end)
            int64_t j0 = n0 - 2 + 2 * lane;
            const int64_t jmax = (N - 1) & ~(int64_t)1;
            j0 = j0 < 0 ? 0 : (j0 > jmax ? jmax : j0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.t + j0), (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
        if (lane < 16) {
            for (int i = 0; i < npd; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.pd_trig + ((b * npd + i) * p.pd_npad + n0 + lane) * 2),
                                                 (__attribute__((address_space(3))) void*)(dst + 24 + 32 * i), 16, 0, 0);
        }
    };
    [[maybe_unused]] auto compute_pd = [&](int64_t K) __attribute__((always_inline)) {
        const double* st = pd.stage[K & 1];
        const int64_t n0 = K * KW;
        const int last = (int)((n0 + KW - 1 < N ? n0 + KW - 1 : N - 1) - n0);
        const double tb = K > 0 ? st[1] : st[2], te = st[2 + last];      // window base (irrelevant for K = 0: T = 0), window end
        for (int i = 0; i < npd; ++i) {
            const double cdec = pd.c[i];
            const double* tr = st + 24 + 32 * i;
            if (lane < 16) {
                double cvc = 0.0, cxc = 0.0, vhc = 0.0, cvs = 0.0, cxs = 0.0, vhs = 0.0;   // cos row (v, x) = (cos, sin); sin row swapped
                if (n0 + lane < N) {
                    const double tn = st[2 + lane], co = tr[2 * lane], si = tr[2 * lane + 1];
                    const double Cn = exp(-cdec * (tn - tb)), Hn = exp(-cdec * (te - tn));
                    cvc = co * Cn; cxc = si * Cn; vhc = co * Hn;
                    cvs = si * Cn; cxs = co * Cn; vhs = si * Hn;
                }
                pd.cv[K & 1][2 * i][lane] = cvc; pd.cx[K & 1][2 * i][lane] = cxc; pd.vh[K & 1][2 * i][lane] = vhc;
                pd.cv[K & 1][2 * i + 1][lane] = cvs; pd.cx[K & 1][2 * i + 1][lane] = cxs; pd.vh[K & 1][2 * i + 1][lane] = vhs;
            } else if (lane == 16) {
                const double ck = exp(-cdec * (te - tb));
                pd.ck[K & 1][2 * i] = ck;
                pd.ck[K & 1][2 * i + 1] = ck;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int pq = lane + 64 * h;
                if (pq < 120) {
                    double2 e = double2{0.0, 0.0};
                    if (n0 + pnn[h] < N) {
                        const double tn = st[2 + pnn[h]], tj = st[2 + pjj[h]];
                        const double cn = tr[2 * pnn[h]], sn = tr[2 * pnn[h] + 1], cj = tr[2 * pjj[h]], sj = tr[2 * pjj[h] + 1];
                        const double dec = exp(-cdec * (tn - tj));
                        e.x = dec * fma(cn, cj, sn * sj);          // (cos, sin)(d tau) by angle addition from the rounded phases, as block_table_kernel
                        e.y = dec * fma(sn, cj, -cn * sj);
                    }
                    pdE[((K & 1) * npd + i) * 128 + pq] = e;
                }
            }
        }
    };

    auto load_u = [&](int64_t k) __attribute__((always_inline)) {
        const double* tl = tileb + (k & 1) * TSP;
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int f = (I * 4 + ks) * 64 + lane;
                const double2 cf = sh.albe[16 * I + 4 * ks + q];
                double cvv = tl[f], cxv = tl[NB * 256 + f];
                if constexpr (PD) {
                    const int r = 16 * I + 4 * ks + q - pdr0;                     // (wave-uniform up to q: four rows per (I, ks))
                    if (16 * I + 4 * ks + 3 >= pdr0 && r >= 0 && r < npdr) { cvv = pd.cv[k & 1][r][c16]; cxv = pd.cx[k & 1][r][c16]; }
                }
                Uf[I][ks] = fma(cf.x, cvv, cf.y * cxv);
            }
    };
    auto load_vh = [&](int64_t k) __attribute__((always_inline)) {
        const double* tl = tileb + (k & 1) * TSP;
        if (owner) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                double v = tl[2 * NB * 256 + (w * 4 + g) * 64 + lane];
                if constexpr (PD) {
                    const int r = 16 * w + c16 - pdr0;
                    if (r >= 0 && r < npdr) v = pd.vh[k & 1][r][4 * g + q];
                }
                if (ycol) {
                    const int64_t n = k * KW + 4 * g + q;
                    if (p.Y) v = sh.ys[k & 1][4 * g + q];
                    v = n < N ? v - mu : 0.0;              // z_n = y_n - u'f   :141
                }
                vh[g] = v;
            }
            ckc = tl[3 * NB * 256 + 16 * w + c16];
            if constexpr (PD) {
                const int r = 16 * w + c16 - pdr0;
                if (r >= 0 && r < npdr) ckc = pd.ck[k & 1][r];
            }
#pragma unroll
            for (int I = 0; I < NB; ++I)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    ckr[I][g] = tl[3 * NB * 256 + 16 * I + 4 * g + q];
                    if constexpr (PD) {
                        const int r = 16 * I + 4 * g + q - pdr0;
                        if (16 * I + 4 * g + 3 >= pdr0 && r >= 0 && r < npdr) ckr[I][g] = pd.ck[k & 1][r];
                    }
                }
        }
    };
    // A of window k (kappa, src/acvf.jl:138-140, on the window's own pairs) -> sh.Ab[k & 1].  Thread pp < 120 of a pair half owns
    // the pair (jj < nn), threads 120 .. 135 the diagonal.  NB <= 3: wavefronts 0 .. 2, all terms.  NB = 4 (SPLIT4; the E buffer
    // is single and a wavefront refills what it alone reads): wavefronts 0 .. 3 = (pair half w & 1) x (terms of parity w >> 1),
    // two partial sums that the chain adds; E piece 2 t + h then belongs to wavefront h + 2 (t & 1).
    constexpr bool SPLIT4 = NB >= 4;
    const int pp = SPLIT4 ? 64 * (w & 1) + lane : 64 * w + lane;
    const int tpar = SPLIT4 ? (w >> 1) : 0, tstep = SPLIT4 ? 2 : 1;
    int nn = 1;
    while ((nn + 1) * nn / 2 <= pp) ++nn;
    const int jj = pp - nn * (nn - 1) / 2;
    auto form_A = [&](int64_t k) __attribute__((always_inline)) {
        if (w < (SPLIT4 ? 4 : 3)) {
            double* Ad = sh.Ab[k & 1][tpar];
            if (pp < 120) {
                const double2* E = EGLOB ? reinterpret_cast<const double2*>(btab + k * RSB + TSP) + pp
                                         : reinterpret_cast<const double2*>(Eb + (k & 1) * ebs) + pp;
                double acc0 = 0.0, acc1 = 0.0;
                int t = tpar;
                constexpr int UN = NB == 5 ? 6 : (NB == 6 ? 4 : 10);   // reads in flight per chunk (two chunks at J = 20; five and six block columns are short of registers)
                for (; t + (UN - 1) * tstep < J; t += UN * tstep) {
                    double2 e[UN], cf[UN];
#pragma unroll
                    for (int i = 0; i < UN; ++i) { e[i] = E[(t + i * tstep) * 128]; cf[i] = sh.ab[t + i * tstep]; }
#pragma unroll
                    for (int i = 0; i < UN; ++i) { acc0 = fma(cf[i].x, e[i].x, acc0); acc1 = fma(cf[i].y, e[i].y, acc1); }
                }
                if constexpr (EGLOB) {
                    // the remainder in ONE batch as well: from global memory every dependent read is a round trip of its own
                    // (J = 16: six of them made 512 draws take 3.2 instead of 2.6 ms)
                    if (t < J) {
                        double2 e[UN], cf[UN];
#pragma unroll
                        for (int i = 0; i < UN; ++i) {
                            const int ti = t + i * tstep;
                            e[i] = E[(ti < J ? ti : t) * 128];
                            cf[i] = ti < J ? sh.ab[ti] : double2{0.0, 0.0};
                        }
#pragma unroll
                        for (int i = 0; i < UN; ++i) { acc0 = fma(cf[i].x, e[i].x, acc0); acc1 = fma(cf[i].y, e[i].y, acc1); }
                    }
                } else {
                    for (; t < J; t += tstep) {
                        const double2 e = E[t * 128], cf = sh.ab[t];
                        acc0 = fma(cf.x, e.x, acc0);
                        acc1 = fma(cf.y, e.y, acc1);
                    }
                }
                if constexpr (PD) {
                    if (tpar == 0)
                        for (int i = 0; i < npd; ++i) {
                            const double2 e = pdE[((k & 1) * npd + i) * 128 + pp], cf = pd.ab[i];
                            acc0 = fma(cf.x, e.x, acc0);
                            acc1 = fma(cf.y, e.y, acc1);
                        }
                }
                const double acc = acc0 + acc1;
                Ad[jj * 16 + nn] = acc;
                Ad[nn * 16 + jj] = acc;
            } else if (pp < (SPLIT4 ? 128 : 136)) {
                // diagonal: NB <= 3: s = pp - 120 (wavefronts 1 and 2); SPLIT4: wavefront 1 -> s = 0 .. 7, wavefront 3 -> s = 8 .. 15,
                // each also zeroing the same entries of the other partial sum
                const int s = SPLIT4 ? pp - 120 + 8 * tpar : pp - 120;
                const int64_t n = k * KW + s;
                double v = 1.0;
                if (n < N) {
                    const double s2n = p.S2 ? sh.ys[k & 1][16 + s] : tileb[(k & 1) * TSP + 3 * NB * 256 + 16 * NB + s];
                    v = suma + (has_nu ? nu * s2n : s2n);    // :92
                }
                Ad[s * 16 + s] = v;
                if constexpr (SPLIT4) sh.Ab[k & 1][tpar ^ 1][s * 16 + s] = 0.0;
            }
        }
    };

    // per-draw series (y, sigma2) [B][N]: fetched by wavefront 2 into a register a whole window before it is staged in LDS (a
    // plain global load consumed at once — or on the critical path — would expose its HBM latency, and wait for the LDS DMAs
    // issued after it as well): window k + 3 is loaded and window k + 2 staged while the chain of window k runs
    double series_reg = 0.0;
    auto fetch_series = [&](int64_t k) __attribute__((always_inline)) {
        if (p.Y && w == 2 && lane < 32) {
            const int64_t n = k * KW + (lane & 15);
            const double* src = lane < 16 ? p.Y : p.S2;
            series_reg = n < N ? src[b * N + n] : 0.0;
        }
    };
    auto stage_series = [&](int64_t k) __attribute__((always_inline)) {
        if (p.Y && w == 2 && lane < 32) sh.ys[k & 1][lane] = series_reg;
    };
    fetch_series(0);
    stage_series(0);
    fetch_series(1);
    stage_series(1);
    fetch_series(2);
    if constexpr (PD) {
        if (pdw) {
            stage_dma(0);
            if (NW > 1) stage_dma(1);
        }
    }
    PIORAN_BLK_BARRIER_DMA();          // record 0 has landed
    if constexpr (PD) {
        // the shared table's E entries of a per-draw term mean nothing: its (a, b) leave the shared contraction of form_A
        if (tid < npd) sh.ab[pd.term[tid]] = double2{0.0, 0.0};
        if (pdw) {
            compute_pd(0);
            if (NW > 1) compute_pd(1);
        }
        PIORAN_BLK_BARRIER();
        if (pdw && NW > 2) stage_dma(2);
    }
    if (has_u) load_u(0);
    load_vh(0);
    form_A(0);
    double quad = 0.0;                 // meaningful in the y-row lanes of the wavefront that owns block column Jy
    double Pm = 1.0;                   // running product of |D| (sign of D_1 kept: :126), chain wavefront
    int Pe = 0;
    bool nonpd = false;
    PIORAN_BLK_BARRIER();              // every wavefront is done with E(0)
    if (NW > 1) {
        dma_pieces(btab + RSB, tileb + TSP, TSP / 128, w, NWV, lane);
        if constexpr (!EGLOB) dma_pieces(btab + RSB + TSP, Eb + ebs, 2 * J, w, NWV, lane);
    }

    PIORAN_BSTAMP_DECL
    for (int64_t k = 0; k < NW; ++k) {
        PIORAN_BSTAMP(0);
        // ---- M' = U~' T, X' = V^' - C_K o M' ------------------------------------------------------------------------
        double x[4];
        if (owner) {
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int I = 0; I < NB; ++I)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Uf[I][ks], T[I][ks], acc, 0, 0, 0);
            if constexpr (ST == 1) {
                double* gwk = p.gw + (b * NW + k) * GWS;
#pragma unroll
                for (int I = 0; I < NB; ++I)
#pragma unroll
                    for (int g = 0; g < 4; ++g) gwk[((w * NB + I) * 4 + g) * 64 + lane] = T[I][g];
#pragma unroll
                for (int g = 0; g < 4; ++g) gwk[OFF_M + (w * 4 + g) * 64 + lane] = acc[g];
            }
            double* mg = sh.MG[w];
#pragma unroll
            for (int g = 0; g < 4; ++g) mg[(4 * g + q) * 18 + c16] = acc[g];
            PIORAN_BLK_SAMEWAVE();
            double mb[4];   // the same block transposed: M [row 16 w + 4 ks + q][step c16], the B operand of U~_w' M_w
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) mb[ks] = mg[c16 * 18 + 4 * ks + q];
#pragma unroll
            for (int g = 0; g < 4; ++g) x[g] = fma(-ckc, acc[g], vh[g]);
            double uw[4];   // U~ fragments of this wavefront's own row block
            static_for<0, NB>([&](auto Ic) __attribute__((always_inline)) {
                constexpr int I = decltype(Ic)::value;
                if (w == I) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) uw[ks] = Uf[I][ks];
                }
            });
            d4 G = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) G = __builtin_amdgcn_mfma_f64_16x16x4f64(uw[ks], mb[ks], G, 0, 0, 0);
            PIORAN_BLK_SAMEWAVE();   // the read-back comes first: the block's storage is reused
#pragma unroll
            for (int g = 0; g < 4; ++g) mg[g * 64 + lane] = G[g];
        }
        PIORAN_BSTAMP(1);
        PIORAN_BLK_BARRIER_DMA();   // B1: M' published; record k + 1 has landed
        PIORAN_BSTAMP(2);
        if constexpr (HELPW) {
            // per-draw rows of record k + 2: its staging has landed (this barrier), buffer k & 1 held window k's values (last read before
            // this barrier: the owners in window k - 1, the chain wavefront's load_u(k) at its end); first read after barrier 1 of
            // window k + 1.  Then the staging of window k + 3 (its buffer was consumed one window ago).
            if (pdw && k + 2 < NW) {
                compute_pd(k + 2);
                if (k + 3 < NW) stage_dma(k + 3);
            }
        }
        // ---- chain: Sigma = A - U~' M, LDL', L^-1 ---------------------------------------------------------------------
        if (chain) {
            const double* Ad = sh.Ab[k & 1][0];
            double sg[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                double gs = sh.MG[0][g * 64 + lane];
#pragma unroll
                for (int I = 1; I < NB; ++I) gs += sh.MG[I][g * 64 + lane];
                double av = Ad[(4 * g + q) * 16 + c16];
                if constexpr (SPLIT4) av += sh.Ab[k & 1][1][(4 * g + q) * 16 + c16];
                sg[g] = av - gs;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) sh.Li[(4 * g + q) * 16 + c16] = sg[g];
            PIORAN_BLK_SAMEWAVE();
            PIORAN_BSTAMP(3);
            double m[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) m[j] = sh.Li[j * 16 + c16];
            PIORAN_BSTAMP(4);
            double mult = ldl_first_mult(m, c16);
            static_for<0, 16>([&](auto Pc) __attribute__((always_inline)) { ldl_step<decltype(Pc)::value>(m, mult, c16); });
            PIORAN_BSTAMP(5);
            if (q == 0) {   // lane n holds column n of L^-1 in m[j], j > n; the rest of m is left-over Sigma (the readers mask it)
                double2* dst = reinterpret_cast<double2*>(sh.Li + c16 * 18);
#pragma unroll
                for (int j = 0; j < 16; j += 2) dst[j / 2] = double2{m[j], m[j + 1]};
            }
        }
        PIORAN_BSTAMP(6);
        // ---- meanwhile: rescale T, A and U~ of the next window ---------------------------------------------------------
        if (owner) {
#pragma unroll
            for (int I = 0; I < NB; ++I)
#pragma unroll
                for (int g = 0; g < 4; ++g) T[I][g] *= ckr[I][g] * ckc;
        }
        PIORAN_BSTAMP(12);
        if (k + 1 < NW) {
            form_A(k + 1);
            PIORAN_BSTAMP(13);
            if (owner && !chain) load_u(k + 1);
            load_vh(k + 1);
            PIORAN_BSTAMP(14);
        }
        if (k + 2 < NW) {   // slot k & 1: window k's values were consumed before barrier 2 of window k - 1
            stage_series(k + 2);
            fetch_series(k + 3);
        }
        // record k + 2: tile k (buffer k & 1) has no reader left since barrier 1.  One E buffer: the E pieces a wavefront refills
        // are the ones only it reads (piece 2 t + h = term t, pairs 64 h .. 64 h + 63), just consumed by its own form_A above.
        // Two E buffers: the target held E(k), last read before barrier 2 of window k - 1.
        const int np_tile = TSP / 128, np_all = np_tile + (EGLOB ? 0 : 2 * J), np_chain = np_all / 4;
        auto copy_piece = [&](int c) __attribute__((always_inline)) {
            const double* src = btab + (k + 2) * RSB + c * 128 + lane * 2;
            double* dst = c < np_tile ? tileb + (k & 1) * TSP + c * 128 : Eb + (k & 1) * ebs + (c - np_tile) * 128;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        };
        if (k + 2 < NW && !chain) {
            if constexpr ((EDBL || EGLOB) && !COPYW) {
                // contiguous shares; wavefronts 0 and 1 carry the pair contraction, the others take twice as much
                const int rest = np_all - np_chain, unit = rest / (NWV == 4 ? 4 : 6);
                const int lo = np_chain + (w == 0 ? 0 : w == 1 ? unit : w == 2 ? 2 * unit : 4 * unit);
                const int hi = w == 0 ? np_chain + unit : w == 1 ? np_chain + 2 * unit : (w == 2 && NWV == 5) ? np_chain + 4 * unit : np_all;
                for (int c = lo; c < hi; ++c) copy_piece(c);
            } else if constexpr (!COPYW) {
                if (w < 2)
                    for (int t = 0; t < J; ++t) copy_piece(np_tile + 2 * t + w);
            }
        }
        PIORAN_BSTAMP(7);
        PIORAN_BLK_BARRIER();   // B2: L^-1, 1/D published
        PIORAN_BSTAMP(8);
        const int nt_chain = 0;   // with copy wavefronts the chain wavefront issues none (measured: 0, 6, 12 pieces: 2.61, 2.67, 2.64 ms)
        // the chain wavefront idles until the next barrier 1: its share of record k + 2 (issued right away: the copies then have
        // the whole Y^ / update / M' stretch to land; issued after its own LDS reads below they land too late for barrier 1)
        if (chain && k + 2 < NW) {
            for (int c = 0; c < (COPYW ? nt_chain : ((EDBL || EGLOB) ? np_chain : np_tile)); ++c) copy_piece(c);
        }
        if constexpr (PD && !HELPW) {
            // per-draw rows of record k + 2 (its staging landed before barrier 1 of this window; buffer k & 1 held window k's values,
            // last read before this barrier), then the staging of window k + 3
            if (chain && k + 2 < NW) {
                compute_pd(k + 2);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (k + 3 < NW) stage_dma(k + 3);
            }
        }
        // copy wavefronts (one E buffer): everything else, E(k + 1) and tile k being consumed since this barrier; a quarter of the
        // share before barrier 3, which follows within ~1000 cycles, the rest after it
        int c_copy = 0, c_step = 1, c_mid = 0;
        if constexpr (COPYW) {
            if (w >= NCW) {
                c_step = NWV - NCW;
                c_copy = nt_chain + (w - NCW);
                c_mid = c_copy + ((np_all - c_copy) / c_step / 4) * c_step;
                if (k + 2 < NW)
                    for (; c_copy < c_mid; c_copy += c_step) copy_piece(c_copy);
            }
        }
        // ---- Y^' = L^-1 X' --------------------------------------------------------------------------------------------
        double ysc[4];
        if (owner) {
            double li[4], idv[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int kk = 4 * ks + q;                             // L^-1 [i = c16][k = kk]: unit lower triangular
                const double lv = sh.Li[kk * 18 + c16];
                idv[ks] = recip_f64(sh.Li[kk * 18 + kk]);             // D_n sits on the diagonal (same reciprocal as the chain's)
                li[ks] = kk < c16 ? lv * idv[ks] : (kk == c16 ? 1.0 : 0.0);   // the column arrives scaled by D_kk
            }
            d4 yt = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) yt = __builtin_amdgcn_mfma_f64_16x16x4f64(li[ks], x[ks], yt, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                sh.Yt[(w * 4 + g) * 64 + lane] = yt[g];
                ysc[g] = yt[g] * idv[g];
                if (w == Jy) quad = fma(yt[g], ysc[g], quad);     // z_n^2 / D_n (== y'K^-1 y, :333)
            }
            if constexpr (ST) {
                // Q' = L^-T D^-1 Y^' = Sigma^-1 X' (steps x rows of this block); A operand (m = c16, k = kk): (L^-1)[kk][m] / D_kk
                const double idm = recip_f64(sh.Li[c16 * 18 + c16]);
                d4 qv = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int kk = 4 * ks + q;
                    const double lt = (kk > c16 ? sh.Li[c16 * 18 + kk] * idm : (kk == c16 ? 1.0 : 0.0)) * idv[ks];
                    qv = __builtin_amdgcn_mfma_f64_16x16x4f64(lt, yt[ks], qv, 0, 0, 0);
                }
                double* gwk = p.gw + (b * NW + k) * GWS;
                double* mg = sh.MG[w];                    // (its Gram partial was consumed before barrier 2)
                if constexpr (ST != 2) {     // C/D order: the reverse pass and the simulation
#pragma unroll
                    for (int g = 0; g < 4; ++g) gwk[OFF_Q + (w * 4 + g) * 64 + lane] = qv[g];
                }
                if constexpr (ST != 3) {     // A-operand order (one transposing LDS round trip): the reverse pass and the back-substitution
#pragma unroll
                    for (int g = 0; g < 4; ++g) mg[(4 * g + q) * 18 + c16] = qv[g];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) gwk[OFF_QF + (w * 4 + ks) * 64 + lane] = mg[c16 * 18 + 4 * ks + q];   // Q [row 16 w + 4 ks + q][step c16]
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
        }
        if constexpr (ST) {
            if (chain) {   // K = Sigma^-1 = L^-T D^-1 L^-1: both operands are the same per-lane expression (A: (m, k) = (c16, kk); B: (k, n) = (kk, c16))
                double* gwk = p.gw + (b * NW + k) * GWS;
                if constexpr (ST == 1) {
                    const double idm = recip_f64(sh.Li[c16 * 18 + c16]);
                    d4 kv = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const int kk = 4 * ks + q;
                        const double lb = kk > c16 ? sh.Li[c16 * 18 + kk] * idm : (kk == c16 ? 1.0 : 0.0);
                        const double la = lb * recip_f64(sh.Li[kk * 18 + kk]);
                        kv = __builtin_amdgcn_mfma_f64_16x16x4f64(la, lb, kv, 0, 0, 0);
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) gwk[OFF_K + g * 64 + lane] = kv[g];
                }
                if constexpr (ST != 2) {
#pragma unroll
                    for (int i = lane; i < 16 * 18; i += 64) gwk[OFF_LI + i] = sh.Li[i];
                }
            }
        }
        PIORAN_BSTAMP(9);
        PIORAN_BLK_BARRIER();   // B3: Y^' published
        PIORAN_BSTAMP(10);
        if constexpr (COPYW) {
            if (w >= NCW && k + 2 < NW)
                for (; c_copy < np_all; c_copy += c_step) copy_piece(c_copy);
        }
        // ---- T += Y^ D^-1 Y^' -----------------------------------------------------------------------------------------
        if (owner) {
            double ya[NB][4];
#pragma unroll
            for (int I = 0; I < NB; ++I)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) ya[I][ks] = sh.Yt[(I * 4 + ks) * 64 + lane];
#pragma unroll
            for (int I = 0; I < NB; ++I)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) T[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[I][ks], ysc[ks], T[I], 0, 0, 0);
        }
        if (chain) {   // off the critical path: log-determinant bookkeeping, U~ of the next window
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const double dj = sh.Li[j * 18 + j];
                nonpd |= !(dj > 0.0);
                Pm *= (k == 0 && j == 0) ? dj : fabs(dj);    // log(D[1]) :126, log(abs(D[n])) :140
                if ((j & 3) == 3) {
                    int ex;
                    Pm = frexp(Pm, &ex);
                    Pe += ex;
                }
            }
            if (k + 1 < NW) load_u(k + 1);
        }
    }

    PIORAN_BSTAMP(11);
    PIORAN_BSTAMP_FLUSH
    // ---- result ------------------------------------------------------------------------------------------------------------
    if (ycol) sh.fin[q] = quad;
    PIORAN_BLK_BARRIER();
    if (chain && lane == 0) {
        const double qs = (sh.fin[0] + sh.fin[1]) + (sh.fin[2] + sh.fin[3]);
        const double logdet = log(Pm) + (double)Pe * 0.6931471805599453094;
        const double res = -0.5 * logdet - 0.5 * (double)N * 1.8378770664093454836 - 0.5 * qs;
        p.out[b] = res;
        if (p.status) p.status[b] = !isfinite(res) ? 2 : (nonpd ? 1 : 0);
    }
}


// ---- windowed reverse mode (round 3) --------------------------------------------------------------------------------------
// Gradient of log L with respect to (a_j, b_j, mu, nu) through the windowed form, one window of 16 steps at a time, walking the
// windows backwards.  With K = Sigma^-1, Q' = K X' (both left by the forward pass, ST above) the window's adjoint needs no adjoint
// of the LDL' factorisation (tools/block_adjoint_proto.py is the numpy prototype, checked against the complex-step oracle):
//   T' = (cK cK') o T + X K X',  l_k = -1/2 log det Sigma - 1/2 x_y' K x_y             (forward, steps x rows written with a prime)
//   X-' = 2 Q' T-  (- Q'[:, y] in the y column);     S- = -1/2 K - Q' T- Q + 1/2 q_y q_y'
//   M-' = -cK o X-' - S- U~';      U~-' = -S- M' + M-' T;      T- <- (cK cK') o T- + 1/2 (U~ M-' + M- U~')
//   d/dal_r += sum_n U~-'[n][r] (C v)[n][r], d/dbe_r likewise with x;  d/dmu -= sum_n X-'[n][y];  d/dsum(a) += tr S-;
//   d/dnu += sum_n S-_nn sigma2_n;  d/da_t += 2 sum_pairs S-_jn E_t,p.cos,  d/db_t += 2 sum_pairs S-_jn E_t,p.sin
// One workgroup per draw; wavefront w < NB owns block column w of T- (same register layout as T in the forward kernel), one more
// wavefront assembles S-.  Six GEMM stages per window (60 MFMAs per owner at three block columns against the forward's 32), no
// factorisation: the reverse window is owner-bound.
// Table of the reverse pass (block_gtable_window_kernel), per window: C o v and C o x in C/D order ([block][g][lane]: step 4 g + (lane >> 4),
// row 16 block + (lane & 15)), then C_K [16 NB], sigma2 [16], then — for d/d(c, d) — (C_K / C) o v and (C_K / C) o x in the same order and
// the window's times [t_n x 16 | t_base | t_end | pad].
__host__ __device__ inline int64_t block_gtab_doubles(int NB) { return 4 * (int64_t)NB * 256 + 16 * NB + 16 + 24; }

__global__ void __launch_bounds__(256) block_gtable_window_kernel(int64_t N, int32_t R, int32_t J, int32_t NB, const int32_t* __restrict__ rowmap,
                                                                  const double* __restrict__ t, const double* __restrict__ c,
                                                                  const double* __restrict__ d, const double* __restrict__ s2, double* __restrict__ tab,
                                                                  int64_t cd_stride, int64_t tab_draw_stride)
{
    __shared__ double cs[64 * 16], sn[64 * 16], Cn[64 * 16], Hn[64 * 16], ckt[64];
    c += (int64_t)blockIdx.y * cd_stride;          // blockIdx.y: draw of a batch of per-draw tables
    d += (int64_t)blockIdx.y * cd_stride;
    tab += (int64_t)blockIdx.y * tab_draw_stride;
    const int64_t GS = block_gtab_doubles(NB);
    const int64_t k = blockIdx.x;
    const int64_t n0 = k * KW;
    const int64_t nlast = n0 + KW - 1 < N ? n0 + KW - 1 : N - 1;
    const double tb = k > 0 ? t[n0 - 1] : t[0];
    const double te = t[nlast];
    for (int it = threadIdx.x; it < J * 16; it += 256) {
        const int term = it >> 4, s = it & 15;
        const int64_t n = n0 + s;
        double si = 0.0, co = 0.0, cn = 0.0, hn = 0.0;
        if (n < N) {
            const double tn = t[n];
            sincos(d[term] * tn, &si, &co);
            cn = exp(-c[term] * (tn - tb));
            hn = exp(-c[term] * (te - tn));
        }
        cs[it] = co; sn[it] = si; Cn[it] = cn; Hn[it] = hn;
        if (s == 0) ckt[term] = exp(-c[term] * (te - tb));
    }
    __syncthreads();
    const int64_t nfrag = (int64_t)NB * 256;
    double* rec = tab + k * GS;
    for (int64_t e0 = threadIdx.x; e0 < GS; e0 += 256) {
        int64_t e = e0;
        double val = 0.0;
        if (e < 2 * nfrag) {
            const int sec = (int)(e / nfrag);
            const int f = (int)(e - sec * nfrag);
            const int blk = f >> 8, reg = (f >> 6) & 3, lane = f & 63;
            const int s = 4 * reg + (lane >> 4), row = 16 * blk + (lane & 15);
            if (n0 + s < N && row < R) {
                const int32_t rm = rowmap[row];
                const int it = (rm & 0xfffff) * 16 + s;
                const bool ks = (rm >> 30) & 1;
                val = ((sec == 0) == ks ? sn[it] : cs[it]) * Cn[it];   // sec 0: C o v, sec 1: C o x
            }
        } else if ((e -= 2 * nfrag) < 16 * NB) {
            const int row = (int)e;
            if (row < R) val = ckt[rowmap[row] & 0xfffff];
            else if (row == R) val = 1.0;
        } else if ((e -= 16 * NB) < 16) {
            val = n0 + e < N ? s2[n0 + e] : 0.0;
        } else if ((e -= 16) < 2 * nfrag) {
            const int sec = (int)(e / nfrag);
            const int f = (int)(e - sec * nfrag);
            const int blk = f >> 8, reg = (f >> 6) & 3, lane = f & 63;
            const int s = 4 * reg + (lane >> 4), row = 16 * blk + (lane & 15);
            if (n0 + s < N && row < R) {
                const int32_t rm = rowmap[row];
                const int it = (rm & 0xfffff) * 16 + s;
                const bool ks = (rm >> 30) & 1;
                val = ((sec == 0) == ks ? sn[it] : cs[it]) * Hn[it];   // sec 0: (C_K / C) o v, sec 1: (C_K / C) o x
            }
        } else {
            e -= 2 * nfrag;
            val = e < 16 ? (n0 + e < N ? t[n0 + e] : te) : (e == 16 ? tb : (e == 17 ? te : 0.0));
        }
        rec[e0] = val;
    }
}

template <int NB>
struct BlockAdjShared {
    double tileU[2][NB][288];   // U~' of window parity: [block][step * 18 + row]
    double tileM[2][NB][288];   // M-'
    double scr[NB][288];        // per owner: transposing scratch
    double P[NB][256];          // partial Q' T- Q of each owner, C/D fragments [g][lane]
    double Srm[256];            // S- row-major
    double qy[16];
    double2 albe[64];
    double ra[64], rb[64];      // per-term reductions
    double rc[64], rd[64];
    double tt[2][16];
    double rs[8];
    unsigned char pn[120], pj[120];
};

// CD: also d/d(c_j, d_j) (QPO features, CARMA kernels and free Celerite terms under NUTS).  Rows: c enters through the three decays
// (C in U~, C_K / C in V^, C_K in the rescaling and in X'), d through (v, x) = (cos, sin)(d t_n); terms: both enter the window's own
// covariance through the pair table, d/dc E = -tau E, d/dd (E.cos, E.sin) = tau (-E.sin, E.cos):
//   d/dc_r -= sum_n U~-'[n][r] U~'[n][r] (t_n - t_b) + sum_n X-'[n][r] V^'[n][r] (t_e - t_n) + cK-_r cK_r (t_e - t_b),
//             cK-_r = 2 sum_j T-'_rj cK_j T_rj - sum_n X-'[n][r] M'[n][r]
//   d/dd_r += s_r sum_n t_n (U~-'[n][r] (al_r C x - be_r C v)[n][r] + X-'[n][r] ((C_K / C) x)[n][r]),  s_r = -1 (cos row), +1 (sin row)
//   d/dc_t -= 2 sum_pairs S-_jn tau (a_t E.cos + b_t E.sin);   d/dd_t += 2 sum_pairs S-_jn tau (b_t E.cos - a_t E.sin)
template <int NB, bool CD, int TPT /*contraction threads per term: 8 (up to 32 terms) or 4*/>
__global__ void __launch_bounds__(256, 1) celerite_block_adjoint_kernel(const ScanParams p, const double* __restrict__ btab,
                                                                                      const double* __restrict__ gtab, double* __restrict__ grad_a,
                                                                                      double* __restrict__ grad_b, double* __restrict__ grad_nu,
                                                                                      double* __restrict__ grad_mu, double* __restrict__ grad_c,
                                                                                      double* __restrict__ grad_d)
{
    constexpr int CH = 3;      // four wavefronts (512 registers each: the hand-pipelined loads live in them); with four block columns
                               // the last owner also assembles S-
    constexpr int64_t GWS = block_grad_ws_doubles(NB);
    constexpr int OFF_M = NB * NB * 256, OFF_Q = OFF_M + NB * 256, OFF_QF = OFF_Q + NB * 256, OFF_K = OFF_QF + NB * 256;
    constexpr int TSP = (3 * NB * 256 + 16 * NB + 16 + 127) & ~127;
    constexpr int64_t GS = 4 * (int64_t)NB * 256 + 16 * NB + 16 + 24;
    constexpr int OFF_H = 2 * NB * 256 + 16 * NB + 16, OFF_TM = OFF_H + 2 * NB * 256;
    __shared__ BlockAdjShared<NB> sh;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, q = lane >> 4, c16 = lane & 15;
    const int64_t b = blockIdx.x;
    const int64_t N = p.N;
    const int J = p.J, R = p.R;
    const int64_t NW = (N + KW - 1) / KW;
    const int64_t RSB = TSP + 256 * (int64_t)J;
    const bool chain = w == CH, owner = w < NB;
    const int Jy = R >> 4, ry = R & 15;
    const bool ycol = owner && w == Jy && c16 == ry;
    const double nu = p.nu ? p.nu[b] : 1.0;
    const double* __restrict__ Ab_ = p.A + b * J;
    const double* __restrict__ Bb_ = p.Bc + b * J;
    const double* gwb = p.gw + b * NW * GWS;
    btab += b * p.tab_draw_stride;                 // per-draw tables (every term's (c, d) per draw): strides 0 for shared tables
    gtab += b * p.gtab_draw_stride;
    if (tid < 64) {
        double a = 0.0, bb = 0.0;
        if (tid < R) {
            const int rm = p.rowmap[tid];
            const int term = rm & 0xfffff;
            a = Ab_[term];
            bb = ((rm >> 30) & 1) ? -Bb_[term] : Bb_[term];
        }
        sh.albe[tid] = double2{a, bb};
        sh.ra[tid] = 0.0; sh.rb[tid] = 0.0; sh.rc[tid] = 0.0; sh.rd[tid] = 0.0;
    }
    if (tid < 8) sh.rs[tid] = 0.0;
    if (tid < 120) {
        int nn = 1;
        while ((nn + 1) * nn / 2 <= tid) ++nn;
        sh.pn[tid] = (unsigned char)nn;
        sh.pj[tid] = (unsigned char)(tid - nn * (nn - 1) / 2);
    }
    __syncthreads();
    const double2 myab = owner ? sh.albe[16 * w + c16] : double2{0.0, 0.0};

    // pair contraction threads: TPT threads per term, each a subset of the 120 pairs
    static_assert(TPT == 8 || TPT == 4, "TPT * J <= 256 threads");
    const bool ethread = tid < TPT * J;
    const bool ewave = 64 * w < TPT * J;           // this wavefront has contraction threads (wave-uniform)
    const int et = tid / TPT, es = tid - et * TPT;
    double acc_ga = 0.0, acc_gb = 0.0;
    [[maybe_unused]] double acc_gc = 0.0, acc_gd = 0.0, acc_c = 0.0, acc_d = 0.0;
    [[maybe_unused]] const double ea = (CD && ethread) ? Ab_[et] : 0.0, eb = (CD && ethread) ? Bb_[et] : 0.0;

    d4 Tb[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) Tb[I] = d4{0.0, 0.0, 0.0, 0.0};
    double acc_al = 0.0, acc_be = 0.0, acc_mu = 0.0, acc_sa = 0.0, acc_nu = 0.0;

    // Where a window's operands come from (round 4).  The block the forward pass left for the window — T | M' | Q | Q (A order) | K, 4864 doubles
    // at three block columns, stored in fragment order — is copied verbatim into LDS by the LDS DMA (1 KB pieces, no registers) and read from
    // there at the point of use: Q at the top of phase A, K in phase B, M' in phase C, T in phase D.
    //   up to three block columns: the whole block one window ahead into one of two buffers, all pieces issued by the chain wavefront after
    //     barrier 2 (it has nothing to do until the next window; the CU's memory path takes ~25 cycles per KB and the issuer stalls on it);
    //   four block columns (every wavefront owns a block column, LDS is short): M' | Q | Q | K (26 pieces) one window ahead into one of two
    //     buffers, T (32 pieces) of the window itself into a single buffer — both issued after barrier 1 by the three wavefronts that wait for
    //     the chain there; T is needed in phase D only (barrier 3 waits for it).
    // Before: ~70 global loads per lane and window — more than the 63 a wavefront can have in flight, so issuing them took a memory latency
    // (3700 cycles of the window's 15 000, tools/block_adjoint_probe.hip) and phase C waited another one for M' and the pair table.
    // What stays a global load: C o v, C o x, sigma2 and the time stamps (one window ahead, in registers), C_K and the d/d(c, d) extras (after
    // phase A) and the pair table (after barrier 1: 41 KB at J = 20, its issue hides behind the chain wavefront's phase B).
    constexpr int DM = NB <= 3 ? 1 : 2;
    constexpr int GWE = DM == 1 ? NB * NB * 256 + 3 * NB * 256 + 256 : 3 * NB * 256 + 256;   // doubles per double-buffered copy
    constexpr int GWT = DM == 1 ? 2 : NB * NB * 256;                                           // the single T buffer (four block columns)
    constexpr int EOFF = DM == 1 ? 0 : OFF_M;                                                  // offset of the double-buffered part inside the block
    static_assert(GWE % 128 == 0 && (DM == 1 || GWT % 128 == 0), "whole DMA pieces");
    __shared__ double gwl0[GWE], gwl1[GWE];   // two arrays, selected at compile time: reads of one never wait for the DMA into the other
    __shared__ double gwt[GWT];
    constexpr int EPT = 15;    // pairs per contraction thread with eight threads per term (up to 32 terms): prefetched; beyond, read in the loop
    double cvc[4], cxc[4];
    double s2w = 0.0;
    [[maybe_unused]] double ttw = 0.0;
    // phase-A operands that stay global loads, one window ahead
    auto fetch_a = [&](int64_t kk, double (&cvc_)[4], double (&cxc_)[4], double& s2w_, double& ttw_) __attribute__((always_inline)) {
        const double* grec = gtab + kk * GS;
        if (owner) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                cvc_[g] = grec[(w * 4 + g) * 64 + lane];
                cxc_[g] = grec[NB * 256 + (w * 4 + g) * 64 + lane];
            }
        }
        if (chain) {
            const int64_t n = kk * KW + c16;
            s2w_ = p.S2 ? (n < N ? p.S2[b * N + n] : 0.0) : grec[2 * NB * 256 + 16 * NB + c16];   // per-draw series: the shifted log-flux models
            if constexpr (CD) ttw_ = grec[OFF_TM + c16];   // the window's time stamps -> sh.tt in phase B (a window ahead: a load of the window itself
                                                           // would sit behind the DMA pieces in the in-order counter and make phase B wait for them)
        }
    };
    dma_pieces_asm(gwb + (NW - 1) * GWS + EOFF, gwl0, GWE / 128, w, 4, lane);
    fetch_a(NW - 1, cvc, cxc, s2w, ttw);
    PIORAN_ASTAMP_DECL
    // one window; BC: which of the two LDS buffers holds its block (alternates from window to window)
    auto window = [&](int64_t k, auto BC) __attribute__((always_inline)) {
        constexpr int BI = decltype(BC)::value;
        const double* cur = BI ? gwl1 : gwl0;
        double* nxt = BI ? gwl0 : gwl1;
        auto ldE = [&](int off) __attribute__((always_inline)) -> double { return cur[off - EOFF]; };                      // M' | Q | Q | K
        auto ldT = [&](int off) __attribute__((always_inline)) -> double { if constexpr (DM == 1) return cur[off]; else return gwt[off]; };
        PIORAN_ASTAMP(0);
        const int par = (int)(k & 1);
        const double* gwk = gwb + k * GWS;
        const double* grec = gtab + k * GS;
        const int64_t n0 = k * KW;
        PIORAN_BLK_BARRIER_DMA();   // B0: this window's block has landed (every wavefront's pieces); everybody is done with the other buffer (and with T)
        PIORAN_ASTAMP(9);
        // ---- this window's own operands (global) ----
        // (no default values on anything loaded under a branch: `x = 0; if (owner) x = load` is a register write the compiler orders after the
        //  loads in flight — s_waitcnt vmcnt(0) at the join, 2000 .. 3000 cycles per window in the middle of the issue)
        double ckc, ckr[NB][4];
        double2 ev[EPT];
        [[maybe_unused]] double hv[4], hx[4], tn[4], tbw, tew;
        const double2* Ewin = reinterpret_cast<const double2*>(btab + k * RSB + TSP) + (ethread ? et : 0) * 128;
        auto issue_window_loads = [&]() __attribute__((always_inline)) {
            if (owner) {
                if constexpr (CD) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        hv[g] = grec[OFF_H + (w * 4 + g) * 64 + lane];
                        hx[g] = grec[OFF_H + NB * 256 + (w * 4 + g) * 64 + lane];
                        tn[g] = grec[OFF_TM + 4 * g + q];
                    }
                    tbw = grec[OFF_TM + 16];
                    tew = grec[OFF_TM + 17];
                }
                ckc = grec[2 * NB * 256 + 16 * w + c16];
#pragma unroll
                for (int I = 0; I < NB; ++I)
#pragma unroll
                    for (int g = 0; g < 4; ++g) ckr[I][g] = grec[2 * NB * 256 + 16 * I + 4 * g + q];
            }
        };
        // pair table of the window: fifteen entries per contraction thread are in flight at a time — all of a thread's pairs with eight
        // threads per term (up to 32 terms); with four (more terms: DRWCelerite-20 has 40) the second fifteen are fetched into the same
        // registers while phase C runs and contracted in phase D.  Idle threads of a wavefront with contraction threads load clamped addresses.
        auto issue_pair_table = [&]() __attribute__((always_inline)) {
            if (ewave) {
#pragma unroll
                for (int i = 0; i < EPT; ++i) {
                    const int pp = es + i * TPT;
                    ev[i] = Ewin[pp < 120 ? pp : 119];
                }
            }
        };
        PIORAN_ASTAMP(10);
        // ---- A: X-' = 2 Q'T-, the partial Q' T- Q, U~' ---------------------------------------------------------------
        double xb[4], uw[4];
        if (owner) {
            d4 qt = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int I = 0; I < NB; ++I)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    qt = __builtin_amdgcn_mfma_f64_16x16x4f64(ldE(OFF_QF + (I * 4 + ks) * 64 + lane), Tb[I][ks], qt, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const double qwv = ldE(OFF_Q + (w * 4 + g) * 64 + lane);
                xb[g] = 2.0 * qt[g];
                if (ycol) {
                    xb[g] -= qwv;
                    if (n0 + 4 * g + q < N) {
                        acc_mu -= xb[g];
                        if (p.g_y) p.g_y[b * N + n0 + 4 * g + q] = xb[g];      // dL/dy_n (v_y = y_n - mu)
                    }
                    sh.qy[4 * g + q] = qwv;
                }
                uw[g] = fma(myab.x, cvc[g], myab.y * cxc[g]);
                sh.tileU[par][w][(4 * g + q) * 18 + c16] = uw[g];
                sh.scr[w][(4 * g + q) * 18 + c16] = qt[g];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            double qtT[4], qfw[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                qtT[ks] = sh.scr[w][c16 * 18 + 4 * ks + q];
                qfw[ks] = ldE(OFF_QF + (w * 4 + ks) * 64 + lane);
            }
            d4 pw = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) pw = __builtin_amdgcn_mfma_f64_16x16x4f64(qfw[ks], qtT[ks], pw, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) sh.P[w][g * 64 + lane] = pw[g];
        }
        PIORAN_ASTAMP(1);
        issue_window_loads();
        // (no initialisers: see above)
        double ncvc[4], ncxc[4], ns2w;
        [[maybe_unused]] double nttw;
        const double kdiag = s2w;
        if (k > 0) fetch_a(k - 1, ncvc, ncxc, ns2w, nttw);
        PIORAN_ASTAMP(2);
        PIORAN_BLK_BARRIER();   // B1
        issue_pair_table();
        if constexpr (DM == 2) {     // wavefronts 0 .. 2 wait for wavefront 3 (the chain) here
            if (w < 3) {
                dma_pieces_asm(gwk, gwt, GWT / 128, w, 3, lane);                                         // T of this window (phase D)
                if (k > 0) dma_pieces_asm(gwb + (k - 1) * GWS + EOFF, nxt, GWE / 128, w, 3, lane);       // M' | Q | Q | K of the next one
            }
        }
        PIORAN_ASTAMP(3);
        // ---- B: S- (chain) -------------------------------------------------------------------------------------------------
        if (chain) {
            if constexpr (CD) {
                if (lane < 16) sh.tt[par][lane] = ttw;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                double ps = sh.P[0][g * 64 + lane];
#pragma unroll
                for (int I = 1; I < NB; ++I) ps += sh.P[I][g * 64 + lane];
                const double sb = -0.5 * ldE(OFF_K + g * 64 + lane) - ps + 0.5 * sh.qy[4 * g + q] * sh.qy[c16];
                sh.Srm[(4 * g + q) * 16 + c16] = sb;
                if (4 * g + q == c16 && n0 + c16 < N) {
                    acc_sa += sb;
                    acc_nu = fma(sb, kdiag, acc_nu);
                    if (p.g_s2) p.g_s2[b * N + n0 + c16] = nu * sb;            // dL/dsigma2_n (the diagonal holds nu sigma2_n)
                }
            }
        }
        PIORAN_ASTAMP(4);
        PIORAN_BLK_BARRIER();   // B2
        if constexpr (DM == 1) {   // the next window's block: from the chain wavefront, which has nothing to do until barrier 0 of the next window
            if (chain && k > 0) dma_pieces_asm(gwb + (k - 1) * GWS, nxt, GWE / 128, 0, 1, lane);
        }
        PIORAN_ASTAMP(5);
        // ---- C: M-' = -cK o X-' - S- U~';  S- M' ------------------------------------------------------------------------
        double mbw[4], mw[4];
        d4 sm = {0.0, 0.0, 0.0, 0.0};
        if (owner) {
#pragma unroll
            for (int g = 0; g < 4; ++g) mw[g] = ldE(OFF_M + (w * 4 + g) * 64 + lane);
            double sA[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) sA[ks] = 0.5 * (sh.Srm[c16 * 16 + 4 * ks + q] + sh.Srm[(4 * ks + q) * 16 + c16]);   // symmetrised
            d4 su = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                su = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[ks], uw[ks], su, 0, 0, 0);
                sm = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[ks], mw[ks], sm, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                mbw[g] = fma(-ckc, xb[g], -su[g]);
                sh.tileM[par][w][(4 * g + q) * 18 + c16] = mbw[g];
            }
        }
        auto contract = [&](int i0) __attribute__((always_inline)) {   // d/da_t, d/db_t (, d/dc_t, d/dd_t): 2 sum over pairs of S-_jn E_t,p (tau ...)
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                const int pp = es + (i0 + i) * TPT;
                if (pp < 120) {
                    const int nn = sh.pn[pp], jj = sh.pj[pp];
                    const double sv = sh.Srm[nn * 16 + jj] + sh.Srm[jj * 16 + nn];
                    acc_ga = fma(sv, ev[i].x, acc_ga);
                    acc_gb = fma(sv, ev[i].y, acc_gb);
                    if constexpr (CD) {
                        const double st = sv * (sh.tt[par][nn] - sh.tt[par][jj]);
                        acc_gc = fma(-st, fma(ea, ev[i].x, eb * ev[i].y), acc_gc);
                        acc_gd = fma(st, fma(eb, ev[i].x, -ea * ev[i].y), acc_gd);
                    }
                }
            }
        };
        PIORAN_ASTAMP(11);
        if (ethread) {
            contract(0);
            if (TPT == 4) {
#pragma unroll
                for (int i = 0; i < EPT; ++i) {
                    const int pp = es + (EPT + i) * TPT;
                    ev[i] = Ewin[pp < 120 ? pp : 119];
                }
            }
        }
        PIORAN_ASTAMP(6);
        if constexpr (DM == 2) { PIORAN_BLK_BARRIER_DMA(); } else { PIORAN_BLK_BARRIER(); }   // B3 (four block columns: T has landed)
        PIORAN_ASTAMP(7);
        // ---- D: U~-' = -S- M' + M-' T;  T- <- (cK cK') o T- + 1/2 (U~ M-' + M- U~') -------------------------------------
        if (owner) {
            auto tkv = [&](int I, int g) __attribute__((always_inline)) -> double { return ldT(((w * NB + I) * 4 + g) * 64 + lane); };
            d4 mt = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int I = 0; I < NB; ++I)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    mt = __builtin_amdgcn_mfma_f64_16x16x4f64(sh.tileM[par][I][c16 * 18 + 4 * ks + q], tkv(I, ks), mt, 0, 0, 0);
            [[maybe_unused]] double ckb = 0.0;     // this lane's share of cK-_r for r = its column (T-', T symmetric): sum over its rows
            if constexpr (CD) {
#pragma unroll
                for (int I = 0; I < NB; ++I)
#pragma unroll
                    for (int g = 0; g < 4; ++g) ckb = fma(Tb[I][g] * ckr[I][g], tkv(I, g), ckb);
                ckb *= 2.0;
            }
#pragma unroll
            for (int I = 0; I < NB; ++I)
#pragma unroll
                for (int g = 0; g < 4; ++g) Tb[I][g] *= ckr[I][g] * ckc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const double ub = mt[g] - sm[g];
                acc_al = fma(ub, cvc[g], acc_al);
                acc_be = fma(ub, cxc[g], acc_be);
                if constexpr (CD) {
                    ckb = fma(-xb[g], mw[g], ckb);
                    acc_c = fma(-ub * uw[g], tn[g] - tbw, acc_c);
                    acc_c = fma(-xb[g] * hv[g], tew - tn[g], acc_c);
                    acc_d = fma(tn[g], fma(ub, fma(myab.x, cxc[g], -myab.y * cvc[g]), xb[g] * hx[g]), acc_d);
                }
            }
            if constexpr (CD) acc_c = fma(-ckb * ckc, tew - tbw, acc_c);
            double hm[4], hu[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { hm[ks] = 0.5 * mbw[ks]; hu[ks] = 0.5 * uw[ks]; }
#pragma unroll
            for (int I = 0; I < NB; ++I)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    Tb[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(sh.tileU[par][I][(4 * ks + q) * 18 + c16], hm[ks], Tb[I], 0, 0, 0);
                    Tb[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(sh.tileM[par][I][(4 * ks + q) * 18 + c16], hu[ks], Tb[I], 0, 0, 0);
                }
        }
        if (ethread && TPT == 4) contract(EPT);
        if (k > 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) { cvc[g] = ncvc[g]; cxc[g] = ncxc[g]; }
            s2w = ns2w;
            if constexpr (CD) ttw = nttw;
        }
        PIORAN_ASTAMP(8);
    };
    for (int64_t k = NW - 1; k >= 0; k -= 2) {
        window(k, ic<0>{});
        if (k >= 1) window(k - 1, ic<1>{});
    }
    PIORAN_ASTAMP_FLUSH
    // ---- reductions: rows -> terms ------------------------------------------------------------------------------------------------
    __syncthreads();
    if (owner) {
        const int row = 16 * w + c16;
        if (row < R) {
            const int rm = p.rowmap[row];
            const int term = rm & 0xfffff;
            atomicAdd(&sh.ra[term], acc_al);
            atomicAdd(&sh.rb[term], ((rm >> 30) & 1) ? -acc_be : acc_be);
            if constexpr (CD) {
                atomicAdd(&sh.rc[term], acc_c);
                atomicAdd(&sh.rd[term], ((rm >> 30) & 1) ? acc_d : -acc_d);
            }
        }
        if (ycol) atomicAdd(&sh.rs[0], acc_mu);
    }
    if (chain) { atomicAdd(&sh.rs[1], acc_sa); atomicAdd(&sh.rs[2], acc_nu); }
    if (ethread) {
        atomicAdd(&sh.ra[et], acc_ga); atomicAdd(&sh.rb[et], acc_gb);
        if constexpr (CD) { atomicAdd(&sh.rc[et], acc_gc); atomicAdd(&sh.rd[et], acc_gd); }
    }
    __syncthreads();
    if (tid < J) {
        grad_a[b * J + tid] = sh.ra[tid] + sh.rs[1];
        grad_b[b * J + tid] = sh.rb[tid];
        if constexpr (CD) {
            if (grad_c) grad_c[b * J + tid] = sh.rc[tid];
            if (grad_d) grad_d[b * J + tid] = sh.rd[tid];
        }
    }
    if (tid == 0) {
        if (grad_mu) grad_mu[b] = sh.rs[0];
        if (grad_nu) grad_nu[b] = sh.rs[2];
    }
}

constexpr size_t kBlockLdsMax = 160 * 1024;
__host__ inline size_t block_lds_bytes(int NB, int J, int emode /*E buffers in LDS: 0 one, 1 two, 2 none*/, int npd = 0 /*per-draw terms*/)
{
    const size_t shared = NB == 1 ? sizeof(BlockSharedT<1>) : NB == 2 ? sizeof(BlockSharedT<2>) : NB == 3 ? sizeof(BlockSharedT<3>) : NB == 4 ? sizeof(BlockSharedT<4>)
                          : NB == 5 ? sizeof(BlockSharedT<5>) : sizeof(BlockSharedT<6>);
    return (size_t)(2 * block_tile_doubles(NB) + (emode == 2 ? 0 : emode == 1 ? 2 : 1) * 256 * J) * sizeof(double) + shared +
           (npd > 0 ? sizeof(BlockPd) + (size_t)2 * npd * 128 * sizeof(double2) : 0);
}

template <int NB, int EM, int PDM = 0>
int launch_block2(const ScanParams& p, const double* btab, hipStream_t stream)
{
    const size_t lds = block_lds_bytes(NB, p.J, EM, PDM ? p.npd_rows / 2 : 0);
    // the attribute belongs to (function, device): one process may drive several devices (pioran_farm_*).  Racing threads at
    // worst set it twice.
    static size_t granted[64] = {};   // per template instance
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return PIORAN_ERR_HIP;
    if (lds > granted[dev]) {
        if (hipFuncSetAttribute((const void*)celerite_block_kernel<NB, EM, PDM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return PIORAN_ERR_HIP;
        granted[dev] = lds;
    }
    hipLaunchKernelGGL((celerite_block_kernel<NB, EM, PDM>), dim3((unsigned)p.B), dim3(NB < 4 ? (PDM == 1 ? 320 : 256) : 512), lds, stream, p, btab);
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}

template <int NB>
int launch_block(const ScanParams& p, const double* btab, hipStream_t stream)
{
    if constexpr (NB >= 5) {   // 64 .. 95 rows (round 4): value only, no per-draw rows; the pair table in LDS where it fits, else from global memory
        if (p.npd_rows > 0) return PIORAN_ERR_UNSUPPORTED;
        if (block_lds_bytes(NB, p.J, 0) <= kBlockLdsMax) return launch_block2<NB, 0>(p, btab, stream);
        if (block_lds_bytes(NB, p.J, 2) <= kBlockLdsMax) return launch_block2<NB, 2>(p, btab, stream);
        return PIORAN_ERR_UNSUPPORTED;
    } else {
    // Where the pair table E lives (tools/sweep_block_emode.py, N = 1e4, late round 3): one LDS buffer is as fast as two or faster at
    // every size measured (SHO-20, 256 draws: 1.92 vs 2.10 ms — the second buffer was worth its LDS in round 2, before the shared block
    // shrank), so two buffers are an option only; above 256 draws what counts is whether TWO workgroups fit a CU — if they do not with E
    // in LDS but do without, E is read from global memory (SHO-20, 512 draws: 2.68 instead of 3.75 ms).
    if (p.npd_rows > 0) {   // per-draw rows
        constexpr int PDH = NB < 4 ? 1 : 2;   // helper wavefront where the workgroup has room for one
        const int npd = p.npd_rows / 2;
        if constexpr (NB < 4) {
            // above 256 draws: two workgroups of four wavefronts per CU (the chain wavefront prepares the per-draw rows, the shared
            // terms' pair table comes from global memory) instead of one of five
            const int pm = p.opt ? p.opt->block_emode : -1;
            if ((pm == 2 || (pm < 0 && p.B > 256)) && 2 * block_lds_bytes(NB, p.J, 2, npd) <= kBlockLdsMax) return launch_block2<NB, 2, 2>(p, btab, stream);
        }
        if (p.B <= 256 && block_lds_bytes(NB, p.J, 1, npd) <= kBlockLdsMax) return launch_block2<NB, 1, PDH>(p, btab, stream);
        if (block_lds_bytes(NB, p.J, 0, npd) <= kBlockLdsMax) return launch_block2<NB, 0, PDH>(p, btab, stream);
        return PIORAN_ERR_UNSUPPORTED;
    }
    const int em = p.opt ? p.opt->block_emode : -1;   // diagnostics: force an E mode
    if (em == 2 && NB < 4) return launch_block2<NB < 4 ? NB : 1, 2>(p, btab, stream);
    if (em == 1 && block_lds_bytes(NB, p.J, 1) <= kBlockLdsMax) return launch_block2<NB, 1>(p, btab, stream);
    if constexpr (NB < 4) {
        if (em < 0 && p.B > 256 && 2 * block_lds_bytes(NB, p.J, 0) > kBlockLdsMax && 2 * block_lds_bytes(NB, p.J, 2) <= kBlockLdsMax)
            return launch_block2<NB, 2>(p, btab, stream);
    }
    if (block_lds_bytes(NB, p.J, 0) <= kBlockLdsMax) return launch_block2<NB, 0>(p, btab, stream);
    return PIORAN_ERR_UNSUPPORTED;
    }
}

template <int NB>
int launch_block_grad(const ScanParams& p, const double* btab, const double* gtab, double* ga, double* gb, double* gnu, double* gmu,
                      double* gc, double* gd, hipStream_t stream)
{
    const size_t lds = block_lds_bytes(NB, p.J, 0);
    if (lds > kBlockLdsMax) return PIORAN_ERR_UNSUPPORTED;
    static size_t granted[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return PIORAN_ERR_HIP;
    if (lds > granted[dev]) {
        if (hipFuncSetAttribute((const void*)celerite_block_kernel<NB, 0, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return PIORAN_ERR_HIP;
        granted[dev] = lds;
    }
    // more than 256 chains: the forward pass with two workgroups per CU (pair table from global memory, as the plain kernel does above 256
    // draws); the reverse pass stays at one (its LDS holds two copies of a window's block)
    bool fwd2 = false;
    if constexpr (NB <= 3) {
        const size_t lds2 = block_lds_bytes(NB, p.J, 2);
        if (p.B > 256 && 2 * lds2 <= kBlockLdsMax && !(p.opt && (p.opt->exp & 16))) {
            static size_t granted2[64] = {};
            if (lds2 > granted2[dev]) {
                if (hipFuncSetAttribute((const void*)celerite_block_kernel<NB, 2, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess)
                    return PIORAN_ERR_HIP;
                granted2[dev] = lds2;
            }
            hipLaunchKernelGGL((celerite_block_kernel<NB, 2, 0, 1>), dim3((unsigned)p.B), dim3(256), lds2, stream, p, btab);
            fwd2 = true;
        }
    }
    if (!fwd2) hipLaunchKernelGGL((celerite_block_kernel<NB, 0, 0, 1>), dim3((unsigned)p.B), dim3(NB < 4 ? 256 : 512), lds, stream, p, btab);
    const bool cd = gc || gd;
    if (p.J <= 32) {
        if (cd) hipLaunchKernelGGL((celerite_block_adjoint_kernel<NB, true, 8>), dim3((unsigned)p.B), dim3(256), 0, stream, p, btab, gtab, ga, gb, gnu, gmu, gc, gd);
        else hipLaunchKernelGGL((celerite_block_adjoint_kernel<NB, false, 8>), dim3((unsigned)p.B), dim3(256), 0, stream, p, btab, gtab, ga, gb, gnu, gmu, gc, gd);
    } else {
        if (cd) hipLaunchKernelGGL((celerite_block_adjoint_kernel<NB, true, 4>), dim3((unsigned)p.B), dim3(256), 0, stream, p, btab, gtab, ga, gb, gnu, gmu, gc, gd);
        else hipLaunchKernelGGL((celerite_block_adjoint_kernel<NB, false, 4>), dim3((unsigned)p.B), dim3(256), 0, stream, p, btab, gtab, ga, gb, gnu, gmu, gc, gd);
    }
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}

// ---- z = K^-1 (y - mu) by block back-substitution (the prediction's first half; round 3) ---------------------------------------
// The windowed factorisation is a block LDL' of the covariance with 16-step blocks: the block of L between a later window v and
// window w is U~_v Phi Q_w' (Phi: the decay between the two windows), Q_w = Sigma_w^-1 X_w as left by the forward pass (ST).  Hence
//   z_w = q_y,w - Q_w h ,      h <- C_K o h + U~_w z_w        (windows backwards; h = sum over later windows of decayed U~ z)
// With h[y row] held at -1 the product Q_w h is -z_w in one go (the y row of U~ is zero and its C_K is one, so it stays -1).
// One wavefront per draw, plain FMAs (a first version ran both products on the matrix cores with the vector replicated over the 16
// columns: 24 products of 64 cycles each per window, 0.88 ms per 625 windows — sixteen times the work for the convenience of the
// layouts): Q_w in A-operand order gives every lane its rows' share of step lane & 15 (12 FMAs, two exchange rounds over the four row
// quarters); for the update every lane owns one row and reads its sixteen steps of U~ (C/D order of the reverse pass's table) and z from LDS.
// Output: gy[b][n] = -z_n (what the reverse mode calls dL/dy: pioran_launch_predict_from_gy takes it from there).
template <int NB>
__global__ void __launch_bounds__(64) celerite_block_backsolve_kernel(const ScanParams p, const double* __restrict__ gtab, double* __restrict__ gy)
{
    constexpr int64_t GWS = block_ws_doubles(NB, 2);
    constexpr int OFF_QF = block_ws_off_qf(NB, 2);
    constexpr int64_t GTS = 4 * (int64_t)NB * 256 + 16 * NB + 16 + 24;
    __shared__ double hs[64], zs[16];
    const int64_t b = blockIdx.x, N = p.N, NW = (N + KW - 1) / KW;
    const int lane = threadIdx.x, q = lane >> 4, c16 = lane & 15;
    const int R = p.R, J = p.J;
    // lane = row for h and its update (rows 0 .. 16 NB - 1); lane = (q, step c16) for the product Q_w h
    const bool rowlane = lane < 16 * NB;
    const int Ir = lane >> 4;                       // block column of this lane's row
    double al = 0.0, be = 0.0;
    if (lane < R) {
        const int rm = p.rowmap[lane];
        const int term = rm & 0xfffff;
        al = p.A[b * J + term];
        be = ((rm >> 30) & 1) ? -p.Bc[b * J + term] : p.Bc[b * J + term];
    }
    double h = lane == R ? -1.0 : 0.0;              // (the y row held at -1: Q_w h is -z_w in one go)
    gtab += b * p.gtab_draw_stride;                // per-draw tables (every term's (c, d) per draw): 0 for the shared table
    const double* gwb = p.gw + b * NW * GWS + OFF_QF + lane;
    // operands of one window: Q in A-operand order ((row 16 I + 4 ks + q, step c16) at [(I 4 + ks) 64 + lane]); C o v, C o x in C/D order
    // ((step 4 g + q', row 16 I + c') at [(I 4 + g) 64 + 16 q' + c']: this lane's row, all sixteen steps); C_K of the row — a window ahead
    struct Ops { double qf[NB][4], cv[16], cx[16], ck; };
    auto fetch = [&](int64_t k, Ops& o) __attribute__((always_inline)) {
        const double* gq = gwb + k * GWS;
        const double* gt = gtab + k * GTS;
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) o.qf[I][ks] = gq[(I * 4 + ks) * 64];
        const int rl = rowlane ? lane : 0;
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) {
            const int off = ((rl >> 4) * 4 + (s_ >> 2)) * 64 + (s_ & 3) * 16 + (rl & 15);
            o.cv[s_] = gt[off];
            o.cx[s_] = gt[NB * 256 + off];
        }
        o.ck = gt[2 * NB * 256 + rl];
    };
    (void)Ir;
    Ops cur, nxt;
    fetch(NW - 1, cur);
    for (int64_t k = NW - 1; k >= 0; --k) {
        fetch(k > 0 ? k - 1 : 0, nxt);
        hs[lane] = rowlane ? h : 0.0;
        __syncthreads();
        double p0 = 0.0, p1 = 0.0;
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            p0 = fma(cur.qf[I][0], hs[16 * I + q], p0);
            p1 = fma(cur.qf[I][1], hs[16 * I + 4 + q], p1);
            p0 = fma(cur.qf[I][2], hs[16 * I + 8 + q], p0);
            p1 = fma(cur.qf[I][3], hs[16 * I + 12 + q], p1);
        }
        double acc = p0 + p1;                       // this lane's rows; the other three quarters of the rows sit 16, 32, 48 lanes away
        acc += __shfl_xor(acc, 16);
        acc += __shfl_xor(acc, 32);                 // = sum_r Q[r][step c16] h[r] = -z[step c16]
        if (q == 0) {
            const int64_t n = k * KW + c16;
            if (n < N) gy[b * N + n] = acc;
            zs[c16] = -acc;
        }
        __syncthreads();
        double d0 = 0.0, d1 = 0.0;
#pragma unroll
        for (int s_ = 0; s_ < 16; s_ += 2) {
            d0 = fma(fma(al, cur.cv[s_], be * cur.cx[s_]), zs[s_], d0);
            d1 = fma(fma(al, cur.cv[s_ + 1], be * cur.cx[s_ + 1]), zs[s_ + 1], d1);
        }
        h = fma(cur.ck, h, d0 + d1);                // h <- C_K o h + U~_w z_w
        cur = nxt;
    }
}

// ---- simulation on the windowed factorisation (sim, src/celerite_solver.jl:515-549; round 3) -----------------------------------------
// y = L D^1/2 q with K = L D L'.  In the block factorisation L = (block L) blockdiag(L_w), Sigma_w = L_w D_w L_w' the window's own LDL'
// (D_w holds the reference's D_n):   xi_w = L_w (D_w^1/2 o q_w)   (every window on its own: block_sim_xi_kernel, sixteen lanes per window,
// forward substitution with the L_w^-1 the chain wavefront left),   y_w = xi_w + U~_w f ,  f <- C_K o f + Q_w' xi_w   (windows forwards;
// f = what the earlier windows contribute, in window-base coordinates: celerite_block_sim_kernel, the mirror image of the back-substitution).
template <int NB>
__global__ void __launch_bounds__(256) block_sim_xi_kernel(const ScanParams p, double* __restrict__ xi)
{
    constexpr int64_t GWS = block_ws_doubles(NB, 3);
    constexpr int OFF_LI = block_ws_off_li(NB, 3);
    const int64_t N = p.N, NW = (N + KW - 1) / KW, b = blockIdx.y;
    const int64_t k = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int i = threadIdx.x & 15;
    const bool live = k < NW;
    const double* li = p.gw + (b * NW + (live ? k : 0)) * GWS + OFF_LI;
    const int64_t n = k * KW + i;
    double x = (live && n < N) ? sqrt(li[i * 18 + i]) * p.noise[b * N + n] : 0.0;
    double row[16];   // (L^-1)[i][j], j < i
#pragma unroll
    for (int j = 0; j < 16; ++j) row[j] = j < i ? li[j * 18 + i] / li[j * 18 + j] : 0.0;
#pragma unroll
    for (int j = 0; j < 15; ++j) {
        const double xj = __shfl(x, j, 16);      // xi_j: final once the columns before j have been taken out
        x = fma(-row[j], xj, x);
    }
    if (live && n < N) xi[b * N + n] = x;
}

template <int NB>
__global__ void __launch_bounds__(64) celerite_block_sim_kernel(const ScanParams p, const double* __restrict__ btab, const double* __restrict__ xi)
{
    constexpr int64_t GWS = block_ws_doubles(NB, 3);
    constexpr int OFF_Q = block_ws_off_q(NB, 3);
    __shared__ double fs[64], xs[16];
    const int64_t b = blockIdx.x, N = p.N, NW = (N + KW - 1) / KW;
    const int lane = threadIdx.x, q = lane >> 4, c16 = lane & 15;
    const int R = p.R, J = p.J;
    const int64_t RSB = block_rec_doubles(NB, J);
    const bool rowlane = lane < 16 * NB;
    // lane = (q, step c16) for y_w = xi_w + U~_w f (U~ in A-operand order: rows 16 I + 4 ks + q); lane = row for f and its update
    double al[NB][4], be[NB][4];
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int row = 16 * I + 4 * ks + q;
            al[I][ks] = be[I][ks] = 0.0;
            if (row < R) {
                const int rm = p.rowmap[row];
                const int term = rm & 0xfffff;
                al[I][ks] = p.A[b * J + term];
                be[I][ks] = ((rm >> 30) & 1) ? -p.Bc[b * J + term] : p.Bc[b * J + term];
            }
        }
    double f = 0.0;
    btab += b * p.tab_draw_stride;                 // per-draw tables: 0 for the shared table
    const double* gwb = p.gw + b * NW * GWS + OFF_Q;
    const double* xib = xi + b * N;
    // Q in C/D order ((row 16 I + c', step 4 g + q') at [(I 4 + g) 64 + 16 q' + c']: this lane's row, all sixteen steps)
    struct Ops { double cv[NB][4], cx[NB][4], qr[16], ck, x; };
    auto fetch = [&](int64_t k, Ops& o) __attribute__((always_inline)) {
        const double* gq = gwb + k * GWS;
        const double* rec = btab + k * RSB;
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                o.cv[I][ks] = rec[(I * 4 + ks) * 64 + lane];
                o.cx[I][ks] = rec[NB * 256 + (I * 4 + ks) * 64 + lane];
            }
        const int rl = rowlane ? lane : 0;
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) o.qr[s_] = gq[((rl >> 4) * 4 + (s_ >> 2)) * 64 + (s_ & 3) * 16 + (rl & 15)];
        o.ck = rec[3 * NB * 256 + rl];
        const int64_t n = k * KW + c16;
        o.x = n < N ? xib[n] : 0.0;
    };
    Ops cur, nxt;
    fetch(0, cur);
    for (int64_t k = 0; k < NW; ++k) {
        fetch(k + 1 < NW ? k + 1 : k, nxt);
        fs[lane] = rowlane ? f : 0.0;
        if (q == 0) xs[c16] = cur.x;
        __syncthreads();
        double p0 = 0.0, p1 = 0.0;
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            p0 = fma(fma(al[I][0], cur.cv[I][0], be[I][0] * cur.cx[I][0]), fs[16 * I + q], p0);
            p1 = fma(fma(al[I][1], cur.cv[I][1], be[I][1] * cur.cx[I][1]), fs[16 * I + 4 + q], p1);
            p0 = fma(fma(al[I][2], cur.cv[I][2], be[I][2] * cur.cx[I][2]), fs[16 * I + 8 + q], p0);
            p1 = fma(fma(al[I][3], cur.cv[I][3], be[I][3] * cur.cx[I][3]), fs[16 * I + 12 + q], p1);
        }
        double acc = p0 + p1;
        acc += __shfl_xor(acc, 16);
        acc += __shfl_xor(acc, 32);                 // (U~_w f)[step c16]
        if (q == 0) {
            const int64_t n = k * KW + c16;
            if (n < N) p.ysim[b * N + n] = cur.x + acc;
        }
        double d0 = 0.0, d1 = 0.0;
#pragma unroll
        for (int s_ = 0; s_ < 16; s_ += 2) {
            d0 = fma(cur.qr[s_], xs[s_], d0);
            d1 = fma(cur.qr[s_ + 1], xs[s_ + 1], d1);
        }
        f = fma(cur.ck, f, d0 + d1);                // f <- C_K o f + Q_w' xi_w
        __syncthreads();
        cur = nxt;
    }
}

template <int NB>
int launch_block_sim(const ScanParams& p, const double* btab, double* xi, hipStream_t stream)
{
    const size_t lds = block_lds_bytes(NB, p.J, 0);
    if (lds > kBlockLdsMax) return PIORAN_ERR_UNSUPPORTED;
    static size_t granted[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return PIORAN_ERR_HIP;
    if (lds > granted[dev]) {
        if (hipFuncSetAttribute((const void*)celerite_block_kernel<NB, 0, 0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return PIORAN_ERR_HIP;
        granted[dev] = lds;
    }
    const int64_t NW = (p.N + KW - 1) / KW;
    hipLaunchKernelGGL((celerite_block_kernel<NB, 0, 0, 3>), dim3((unsigned)p.B), dim3(NB < 4 ? 256 : 512), lds, stream, p, btab);
    hipLaunchKernelGGL((block_sim_xi_kernel<NB>), dim3((unsigned)((NW + 15) / 16), (unsigned)p.B), dim3(256), 0, stream, p, xi);
    hipLaunchKernelGGL((celerite_block_sim_kernel<NB>), dim3((unsigned)p.B), dim3(64), 0, stream, p, btab, xi);
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}

template <int NB>
int launch_block_solve(const ScanParams& p, const double* btab, const double* gtab, double* gy, hipStream_t stream)
{
    const size_t lds = block_lds_bytes(NB, p.J, 0);
    if (lds > kBlockLdsMax) return PIORAN_ERR_UNSUPPORTED;
    static size_t granted[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return PIORAN_ERR_HIP;
    if (lds > granted[dev]) {
        if (hipFuncSetAttribute((const void*)celerite_block_kernel<NB, 0, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return PIORAN_ERR_HIP;
        granted[dev] = lds;
    }
    hipLaunchKernelGGL((celerite_block_kernel<NB, 0, 0, 2>), dim3((unsigned)p.B), dim3(NB < 4 ? 256 : 512), lds, stream, p, btab);
    hipLaunchKernelGGL((celerite_block_backsolve_kernel<NB>), dim3((unsigned)p.B), dim3(64), 0, stream, p, gtab, gy);
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}

}  // namespace

// log L (p.out, p.status) and gy [B][N] = -K^-1 (y - mu): the windowed forward pass with its per-window stores (p.gw:
// pioran_block_store_workspace_doubles(.., 2)) followed by the block back-substitution.  gtab: pioran_launch_block_gtab.
int pioran_launch_block_solve(const ScanParams& p, const double* btab, const double* gtab, double* gy, hipStream_t stream)
{
    if (!btab || !gtab || !gy || !p.gw || p.npd_rows != 0 || p.B < 1 || p.N < 1 || !pioran_block_fits(p.R, p.J)) return PIORAN_ERR_UNSUPPORTED;
    if ((p.Y == nullptr) != (p.S2 == nullptr)) return PIORAN_ERR_ARG;
    switch ((p.R + 1 + 15) / 16) {
        case 1: return launch_block_solve<1>(p, btab, gtab, gy, stream);
        case 2: return launch_block_solve<2>(p, btab, gtab, gy, stream);
        case 3: return launch_block_solve<3>(p, btab, gtab, gy, stream);
        case 4: return launch_block_solve<4>(p, btab, gtab, gy, stream);
    }
    return PIORAN_ERR_UNSUPPORTED;
}

// GP realisations p.ysim [B][N] from the normals p.noise [B][N]: the windowed factorisation of the covariance (data set with y = 0;
// p.gw: pioran_block_store_workspace_doubles(.., 3); xi: B N doubles of scratch), then the two simulation kernels above.
int pioran_launch_block_sim(const ScanParams& p, const double* btab, double* xi, hipStream_t stream)
{
    if (!btab || !xi || !p.gw || !p.noise || !p.ysim || p.npd_rows != 0 || p.B < 1 || p.N < 1 || !pioran_block_fits(p.R, p.J)) return PIORAN_ERR_UNSUPPORTED;
    switch ((p.R + 1 + 15) / 16) {
        case 1: return launch_block_sim<1>(p, btab, xi, stream);
        case 2: return launch_block_sim<2>(p, btab, xi, stream);
        case 3: return launch_block_sim<3>(p, btab, xi, stream);
        case 4: return launch_block_sim<4>(p, btab, xi, stream);
    }
    return PIORAN_ERR_UNSUPPORTED;
}

// ---- windowed reverse mode: host side ----
size_t pioran_block_grad_workspace_doubles(int64_t B, int64_t N, int32_t R)
{
    const int NB = (R + 1 + 15) / 16;
    return (size_t)B * (size_t)((N + KW - 1) / KW) * (size_t)block_grad_ws_doubles(NB);
}
// p.gw of pioran_launch_block_solve (what = 2) / pioran_launch_block_sim (what = 3): the packed per-window stores
size_t pioran_block_store_workspace_doubles(int64_t B, int64_t N, int32_t R, int what)
{
    const int NB = (R + 1 + 15) / 16;
    return (size_t)B * (size_t)((N + KW - 1) / KW) * (size_t)block_ws_doubles(NB, what == 2 ? 2 : 3);
}
size_t pioran_block_gtab_doubles(int64_t N, int32_t R)
{
    const int NB = (R + 1 + 15) / 16;
    return (size_t)((N + KW - 1) / KW) * (size_t)block_gtab_doubles(NB);
}
int pioran_launch_block_gtab(int64_t N, int32_t R, int32_t J, const int32_t* rowmap, const double* t, const double* c, const double* d,
                             const double* s2, double* gtab, hipStream_t stream)
{
    return pioran_launch_block_gtab_batch(N, R, J, 1, rowmap, t, c, d, s2, gtab, 0, stream);
}
// nb tables at once: draw i uses (c, d) + i J and writes gtab + i draw_stride
int pioran_launch_block_gtab_batch(int64_t N, int32_t R, int32_t J, int64_t nb, const int32_t* rowmap, const double* t, const double* C,
                                   const double* D, const double* s2, double* gtab, int64_t draw_stride, hipStream_t stream)
{
    const int NB = (R + 1 + 15) / 16;
    const int64_t NW = (N + KW - 1) / KW;
    if (J < 1 || J > kBlockMaxTerms || NB > 4 || nb < 1 || nb > 65535 || NW > 0x7fffffffLL) return PIORAN_ERR_ARG;
    hipLaunchKernelGGL(block_gtable_window_kernel, dim3((unsigned)NW, (unsigned)nb), dim3(256), 0, stream, N, R, J, NB, rowmap, t, C, D, s2, gtab,
                       (int64_t)J, draw_stride);
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}
// log L (p.out, p.status) and d/d(a_j, b_j) [B][J], d/dnu, d/dmu [B] (either may be nullptr), optionally d/dy_n, d/dsigma2_n (p.g_y, p.g_s2:
// [B][N]) for shared (c, d) without per-draw rows; per-draw series p.Y / p.S2 allowed; p.gw: pioran_block_grad_workspace_doubles;
// btab, gtab: the two tables of this (c, d)
int pioran_launch_block_grad(const ScanParams& p, const double* btab, const double* gtab, double* grad_a, double* grad_b, double* grad_nu,
                             double* grad_mu, double* grad_c, double* grad_d, hipStream_t stream)
{
    if (!btab || !gtab || !p.gw || p.npd_rows != 0 || p.B < 1 || p.N < 1 || !pioran_block_fits(p.R, p.J)) return PIORAN_ERR_UNSUPPORTED;
    if ((p.Y == nullptr) != (p.S2 == nullptr)) return PIORAN_ERR_ARG;
    switch ((p.R + 1 + 15) / 16) {
        case 1: return launch_block_grad<1>(p, btab, gtab, grad_a, grad_b, grad_nu, grad_mu, grad_c, grad_d, stream);
        case 2: return launch_block_grad<2>(p, btab, gtab, grad_a, grad_b, grad_nu, grad_mu, grad_c, grad_d, stream);
        case 3: return launch_block_grad<3>(p, btab, gtab, grad_a, grad_b, grad_nu, grad_mu, grad_c, grad_d, stream);
        case 4: return launch_block_grad<4>(p, btab, gtab, grad_a, grad_b, grad_nu, grad_mu, grad_c, grad_d, stream);
    }
    return PIORAN_ERR_UNSUPPORTED;
}

int pioran_block_supported_rows() { return 63; }

// 0 when (R, J) does not fit the kernel's LDS budget
size_t pioran_block_table_doubles(int64_t N, int32_t R, int32_t J)
{
    const int NB = (R + 1 + 15) / 16;
    return (size_t)((N + KW - 1) / KW) * (size_t)block_rec_doubles(NB, J);
}

int pioran_block_fits(int32_t R, int32_t J)
{
    const int NB = (R + 1 + 15) / 16;
    if (R < 1 || NB > 4 || J < 1 || J > kBlockMaxTerms) return 0;
    return block_lds_bytes(NB, J, 0) <= kBlockLdsMax;
}

// ... the value-only kernel also runs five and six block columns (64 .. 95 rows)
int pioran_block_fits_value(int32_t R, int32_t J)
{
    const int NB = (R + 1 + 15) / 16;
    if (NB <= 4) return pioran_block_fits(R, J);
    if (NB > 6 || J < 1 || J > kBlockMaxTerms) return 0;
    return block_lds_bytes(NB, J, 0) <= kBlockLdsMax || block_lds_bytes(NB, J, 2) <= kBlockLdsMax;
}

// ... with `npd_terms` per-draw terms (their rows last)
int pioran_block_fits_pd(int32_t R, int32_t J, int32_t npd_terms)
{
    const int NB = (R + 1 + 15) / 16;
    if (R < 1 || NB > 4 || J < 1 || J > kBlockMaxTerms || npd_terms < 1 || npd_terms > kBlockMaxPdTerms || R < 2 * npd_terms) return 0;
    return block_lds_bytes(NB, J, 0, npd_terms) <= kBlockLdsMax;
}

namespace {
// (cos, sin)(d_term t_n) of the per-draw terms, [draw][per-draw term][step (npad = 16 windows)][2]: what the block kernel stages per
// window for its per-draw rows.  Same sincos as the shared tables (full-range reduction), evaluated once per (draw, term, step).
__global__ void __launch_bounds__(256) pd_trig_kernel(int64_t N, int64_t npad, int64_t B, int32_t J, int32_t npd,
                                                      const int32_t* __restrict__ pd_terms, const double* __restrict__ t,
                                                      const double* __restrict__ D, double* __restrict__ out)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * npd * npad) return;
    const int64_t n = idx % npad, bi = idx / npad;
    const int i = (int)(bi % npd);
    const int64_t b = bi / npd;
    double si = 0.0, co = 0.0;
    if (n < N) sincos(D[b * J + pd_terms[i]] * t[n], &si, &co);   // :52-53
    reinterpret_cast<double2*>(out)[idx] = double2{co, si};
}
}  // namespace

size_t pioran_block_pd_trig_doubles(int64_t N, int64_t B, int32_t npd_terms)
{
    return (size_t)B * (size_t)npd_terms * (size_t)((N + KW - 1) / KW * KW) * 2;
}

int pioran_launch_block_pd_trig(int64_t N, int64_t B, int32_t J, int32_t npd_terms, const int32_t* pd_terms, const double* t,
                                const double* D, double* out, hipStream_t stream)
{
    const int64_t npad = (N + KW - 1) / KW * KW, total = B * npd_terms * npad;
    if (total <= 0 || (total + 255) / 256 > 0x7fffffffLL) return PIORAN_ERR_ARG;
    hipLaunchKernelGGL(pd_trig_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, N, npad, B, J, npd_terms, pd_terms, t, D, out);
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}

int pioran_launch_block_table(int64_t N, int32_t R, int32_t J, const int32_t* rowmap, const double* t, const double* c,
                              const double* d, const double* y, const double* s2, double* btab, hipStream_t stream)
{
    return pioran_launch_block_table_batch(N, R, J, 1, rowmap, t, c, d, y, s2, btab, 0, stream);
}

// the entry-per-thread table kernel (round 2), kept as the cross-check of the windowed one
int pioran_launch_block_table_reference(int64_t N, int32_t R, int32_t J, const int32_t* rowmap, const double* t, const double* c,
                                        const double* d, const double* y, const double* s2, double* btab, hipStream_t stream)
{
    const int NB = (R + 1 + 15) / 16;
    const int64_t total = (int64_t)pioran_block_table_doubles(N, R, J);
    hipLaunchKernelGGL(block_table_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, N, R, J, NB, rowmap, t, c, d, y,
                       s2, btab);
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}

// nb tables at once: draw i uses (c, d) + i J and writes btab + i draw_stride (draw_stride >= pioran_block_table_doubles)
int pioran_launch_block_table_batch(int64_t N, int32_t R, int32_t J, int64_t nb, const int32_t* rowmap, const double* t, const double* C,
                                    const double* D, const double* y, const double* s2, double* btab, int64_t draw_stride, hipStream_t stream)
{
    const int NB = (R + 1 + 15) / 16;
    const int64_t NW = (N + KW - 1) / KW;
    if (J < 1 || J > kBlockMaxTerms || NB > 6 || nb < 1 || nb > 65535 || NW > 0x7fffffffLL) return PIORAN_ERR_ARG;
    hipLaunchKernelGGL(block_table_window_kernel, dim3((unsigned)NW, (unsigned)nb), dim3(256), 0, stream, N, R, J, NB, rowmap, t, C, D, y, s2,
                       btab, (int64_t)J, draw_stride);
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}

// shared-(c, d) launches; btab from pioran_launch_block_table for the same (N, R, J, rowmap).  Per-draw rows (npd_rows = 2 or 4, the
// LAST rows of the row map) need p.pd_C ([B][J]) and p.pd_trig (pioran_launch_block_pd_trig), p.pd_npad
int pioran_launch_scan_block(const ScanParams& p, const double* btab, hipStream_t stream)
{
    if (!btab || p.B < 1 || p.N < 1) return PIORAN_ERR_UNSUPPORTED;
    if (p.npd_rows != 0) {
        if ((p.npd_rows & 1) || !p.pd_C || !p.pd_trig || p.pd_npad < p.N || !pioran_block_fits_pd(p.R, p.J, p.npd_rows / 2)) return PIORAN_ERR_UNSUPPORTED;
    } else if (!pioran_block_fits_value(p.R, p.J)) {
        return PIORAN_ERR_UNSUPPORTED;
    }
    if ((p.Y == nullptr) != (p.S2 == nullptr)) return PIORAN_ERR_ARG;
    switch ((p.R + 1 + 15) / 16) {
        case 1: return launch_block<1>(p, btab, stream);
        case 2: return launch_block<2>(p, btab, stream);
        case 3: return launch_block<3>(p, btab, stream);
        case 4: return launch_block<4>(p, btab, stream);
        case 5: return launch_block<5>(p, btab, stream);
        case 6: return launch_block<6>(p, btab, stream);
    }
    return PIORAN_ERR_UNSUPPORTED;
}
