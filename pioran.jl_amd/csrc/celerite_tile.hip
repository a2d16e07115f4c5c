// Windowed form of the celerite factorisation for LARGE batches (gfx950, round 5): ONE DRAW PER WAVEFRONT, sixteen time steps
// per window, the O(R^2) work of a window on the fp64 matrix cores.
//
// Same mathematics as celerite_block.hip (init_semi_separable! + the forward half of solve_prec!, src/celerite_solver.jl:12-100,
// 115-142; logl :312-334, re-associated into windows: the header of celerite_block.hip has the algebra), same table
// (window_common.h: one fragment-order record per window, built once per prepared (c, d)).  What differs is the work split.
// celerite_block.hip spreads ONE draw over a workgroup (a block column of T per wavefront, a chain wavefront, three barriers per
// window): latency.  Here a wavefront owns the WHOLE state T of its draw as NB x NB accumulator tiles of v_mfma_f64_16x16x4_f64
// (register g of tile (I, J) = element (16 I + 4 g + (lane >> 4), 16 J + (lane & 15))) and walks the whole window by itself:
//   M'  = U~' T            4 NB^2 matrix instructions (T's registers are the B operand as they stand; T is symmetric and only its lower
//                          tiles are kept: an off-diagonal tile serves twice, as B operand and — transposed for free — as A operand)
//   G   = U~' M            4 NB   (one transposing LDS round trip per block of M')
//   Sigma = A - G ; Sigma = L D L' ; L^-1    on v_fmac_f64_dpp row_newbcast (window_common.h), no LDS inside the factorisation
//   Y^' = L^-1 X'          4 NB
//   T  <- (C_K C_K') o T + Y^ D^-1 Y^'       2 NB (NB + 1)
// No exchange between wavefronts at all: the four wavefronts of a workgroup (four draws) only share the LDS copy of the window's
// record (LDS DMA, two buffers, one barrier per window); two workgroups share a CU up to four block columns (<= 256 registers), so
// every SIMD holds two independent instruction streams — one can run its vector phase (the LDL', the rescaling of T, the pair
// contraction) while the other has the matrix pipe.  Per step and draw at 60 rows: 10 matrix instructions (640 SIMD cycles) + ~45
// vector instructions, against ~1300 SIMD cycles of the register-resident step-by-step scan (celerite_scan.hip), whose every FMA is
// a vector instruction.
// The window's own covariance block A (kappa on the 120 pairs of the window: a contraction of the pair table E with the draw's
// (a, b)) does not depend on the state: tile_pairs_mfma_kernel forms it for every (draw, window) of the launch beforehand — E in
// registers, 32 draws per workgroup — into a workspace of 1 KB per draw and window that the factorisation reads one window ahead
// (first version: the contraction inside the window loop, 40 .. 80 dependent reads of E from L2 per window: a third of the time).
// Restrictions: shared (c, d) without per-draw rows; 1 .. 95 active rows; the series shared or per draw (Y, S2: the shifted log-flux models).
// Everything else stays on the other kernels.
#include "common.h"
#include "window_common.h"

// Diagnostic hooks: compiled out in the product; tools/tile_probe.hip defines them to s_memtime accumulators.
#ifndef PIORAN_TSTAMP
#define PIORAN_TSTAMP(i) __builtin_amdgcn_sched_barrier(0)
#define PIORAN_TSTAMP_DECL
#define PIORAN_TSTAMP_FLUSH
#endif
#ifndef PIORAN_ASTAMP2
#define PIORAN_ASTAMP2(i) __builtin_amdgcn_sched_barrier(0)
#define PIORAN_ASTAMP2_DECL
#define PIORAN_ASTAMP2_FLUSH
#endif

namespace {

#ifndef PIORAN_TILE_FWD_NOWAIT
#define PIORAN_TILE_FWD_NOWAIT 1      // no counter wait between a wavefront's LDS writes and its own reads of them (the LDS serves a wavefront in order): -0.3 %, same box
#endif
#if PIORAN_TILE_FWD_NOWAIT
#define PIORAN_TILE_FWD_ORDER() asm volatile("" ::: "memory")
#else
#define PIORAN_TILE_FWD_ORDER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif
#ifndef PIORAN_TILE_ADJ_RAGFIX
#define PIORAN_TILE_ADJ_RAGFIX 1   // the same in the reverse kernel: 4096 chains of SHO-20 45.60 -> 45.33 ms, same box
#endif
#ifndef PIORAN_TILE_RAGFIX4
#define PIORAN_TILE_RAGFIX4 1     // ... at four block columns and more as a fix-up after the loop with the values loaded again: DRWCelerite-20 -1.5 %, same box
#endif
#ifndef PIORAN_TILE_QUADALL
#define PIORAN_TILE_QUADALL 1     // a quadratic-form accumulator per block column, the y row's picked after the loop: -0.8 %, same box
#endif
#ifndef PIORAN_TILE_RAGFIX
#define PIORAN_TILE_RAGFIX 1      // the ragged window's mask once per window under its wave-uniform test: -0.9 %, same box
#endif
#ifndef PIORAN_TILE_YSFIX
#define PIORAN_TILE_YSFIX 1       // the per-draw series as a fix-up of X' under its wave-uniform test instead of a select per block column and step: -0.7 %, same box
#endif
#ifndef PIORAN_TILE_BIG_VHA
#define PIORAN_TILE_BIG_VHA 1      // five and six block columns (one wavefront per SIMD, 512 registers): both preloads, SHO-40 38.0 -> 32.4 ms per 4096 draws, same box
#endif
#ifndef PIORAN_TILE_BIG_CKP
#define PIORAN_TILE_BIG_CKP 1
#endif
#ifndef PIORAN_TILE_VHA      // compile-time switches of measured choices (tools/ab_variant_lib.py builds the other side for a same-box A/B)
#define PIORAN_TILE_VHA 1
#endif
constexpr int kTileMaxTerms = 64;
constexpr int kTileWaves = 4;      // wavefronts (= draws) per workgroup
// ... of the reverse mode: T_k and T- of a draw are 45 KB (53 with d/d(c, d)) of LDS at four block columns — three draws fit a CU, not four
// Between a wavefront's LDS writes and its own reads of the same addresses no counter wait is needed — the LDS executes a wavefront's instructions in order —
// only the compiler has to keep the order:
#define PIORAN_LDS_ORDER() asm volatile("" ::: "memory")
template <int NB, bool CD = false>
constexpr int tile_adj_waves() { return NB <= 3 ? 4 : (CD ? 2 : 3); }

typedef unsigned int tile_u32x2 __attribute__((ext_vector_type(2)));
// 8 bytes through a buffer resource: per-lane byte offset in a VGPR (constant over the kernel), everything that moves (window, block, register
// of the fragment) in the scalar offset — no vector address arithmetic in the window loop
__device__ __forceinline__ double tile_bload(__amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    const tile_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
    return __hiloint2double((int)v.y, (int)v.x);
}

template <int NB>
struct TileWave {                  // LDS of one wavefront
    double scr[16 * 18];           // in turn: a block of M' for the transposing read-back; Sigma [j][n]; D_k (L^-1)_ik at [k * 18 + i]
    double2 albe[16 * NB];         // per row: u = al v + be x (:59-63)
    double ck[16 * NB + 16];       // the window's C_K per row, then sigma2 per step (staged from the record: read again and again by the update)
    double ys[16];                 // per-draw series (p.Y: the shifted log-flux models, docs/src/ultranest.md:199-205): y_n of the window's steps
    double up[NB > 1 ? NB * (NB - 1) / 2 : 1][16 * 18];   // the strictly lower tiles of T once more, [row][column, stride 18]: read back transposed
                                                          // they are the upper tiles as B operands
};

// The contraction is a GEMM — out[draw][(window, pair)] = sum over (t, cos | sin) of coef[draw][(t, cos | sin)] E[(t, cos | sin)][(window, pair)],
// 4096 x 2 J x 80 000 at the bench shape — on v_mfma_f64_16x16x4_f64: a wavefront owns 16 draws (their coefficients as A operands in 2 JQ registers for
// the whole launch) and walks kPairWin windows x 8 tiles of 16 pairs; per tile JQ 16-byte table reads per lane (cos and sin of one term: both halves feed a
// matrix instruction), 2 JQ matrix instructions, one 32-byte store per lane (128 contiguous bytes per draw).  JQ = ceil(J / 4), compile-time.
// (Vector forms of round 5, removed in round 6: 0.93 ms per 4096 draws at 20 terms with the table in registers, 1.97 ms at 40 terms with the coefficients in LDS.)
constexpr int kPairWin = 5;
template <int JQ>
__global__ void __launch_bounds__(256) tile_pairs_mfma_kernel(const ScanParams p, const double* __restrict__ btab, int64_t rsb, int64_t tsp, double* __restrict__ out)
{
    const int J = p.J;
    const int64_t NW = (p.N + KW - 1) / KW;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
    const int64_t b0 = ((int64_t)blockIdx.y * 4 + w) * 16;
    if (b0 >= p.B) return;                         // (no workgroup-level synchronisation in this kernel)
    const int64_t draw = b0 + li < p.B ? b0 + li : p.B - 1;
    // The PAIRS are the rows of the product (table entries as A operands), the draws its columns (coefficients as B operands), and row r of a tile is
    // pair 4 (r & 3) + (r >> 2): result register g of lane (lk, li) is then pair 4 lk + g of draw li — four ADJACENT pairs, one 32-byte store per
    // lane and tile (up to round 5 the draws were the rows: four 8-byte stores per lane; the kernel is bound by its 2.6 GB of stores).
    double aop[JQ], bop[JQ];
    int toff[JQ];
#pragma unroll
    for (int q = 0; q < JQ; ++q) {
        const int t = 4 * q + lk;
        aop[q] = t < J ? p.A[draw * J + t] : 0.0;
        bop[q] = t < J ? p.Bc[draw * J + t] : 0.0;
        toff[q] = (t < J ? t : J - 1) * 128 + 4 * (li & 3) + (li >> 2);
    }
    const bool live = b0 + li < p.B;
    double* orow = out + (b0 + li) * NW * 128 + 4 * lk;
    const int64_t k0 = (int64_t)blockIdx.x * kPairWin, k1 = k0 + kPairWin < NW ? k0 + kPairWin : NW;
    for (int64_t k = k0; k < k1; ++k) {
        const double2* E = reinterpret_cast<const double2*>(btab + k * rsb + tsp);
#pragma unroll 2
        for (int pt = 0; pt < 8; ++pt) {
            double2 e[JQ];
#pragma unroll
            for (int q = 0; q < JQ; ++q) e[q] = E[toff[q] + 16 * pt];
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < JQ; ++q) {
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(e[q].x, aop[q], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(e[q].y, bop[q], acc, 0, 0, 0);
            }
            if (live) *reinterpret_cast<d4*>(orow + k * 128 + 16 * pt) = acc;
        }
    }
}

template <typename... Args>
static inline bool launch_pairs_mfma(const ScanParams& p, hipStream_t stream, Args... args)
{
    const int64_t NW = (p.N + KW - 1) / KW;
    if ((p.B + 63) / 64 > 65535 || p.J > kTileMaxTerms) return false;       // (more than 4 M draws per launch / 64 terms: the callers chunk / refuse)
    const dim3 gr((unsigned)((NW + kPairWin - 1) / kPairWin), (unsigned)((p.B + 63) / 64));
    const int jq = (p.J + 3) / 4;
    if (jq <= 5) hipLaunchKernelGGL(tile_pairs_mfma_kernel<5>, gr, dim3(256), 0, stream, p, args...);
    else if (jq <= 10) hipLaunchKernelGGL(tile_pairs_mfma_kernel<10>, gr, dim3(256), 0, stream, p, args...);
    else if (jq <= 16) hipLaunchKernelGGL(tile_pairs_mfma_kernel<16>, gr, dim3(256), 0, stream, p, args...);
    else return false;
    return true;
}

// ST (gradient, round 5): the forward pass of the one-draw-per-wavefront reverse mode — it leaves the lower tiles of T at the START of every window
// in p.gw ([draw][window][tile][lane][register], NB (NB + 1) / 2 x 2 KB per window) and nothing else: the reverse kernel
// (celerite_tile_adjoint_kernel) recomputes M', Sigma, the LDL' and Q' from it.  The value is bit-identical to the plain kernel's.
template <int NB, bool ST = false>
__global__ void __launch_bounds__(64 * kTileWaves, NB <= 4 ? 2 : 1) celerite_tile_kernel(const ScanParams p, const double* __restrict__ btab,
                                                                                          const double* __restrict__ pairs)
{
    constexpr int TS = 3 * NB * 256 + 16 * NB + 16, TSP = (TS + 127) & ~127;
    extern __shared__ double lds_[];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, q = lane >> 4, c16 = lane & 15;
    const int64_t b = (int64_t)blockIdx.x * kTileWaves + w;
    if (b >= p.B) return;          // (no workgroup-level synchronisation anywhere below: a wavefront is on its own)
    const int64_t N = p.N;
    const int J = p.J, R = p.R;
    const int64_t NW = (N + KW - 1) / KW;
    const int64_t RSB = TSP + 256 * (int64_t)J;
    TileWave<NB>& sw = reinterpret_cast<TileWave<NB>*>(lds_)[w];
    // the table (at most 2 GB: capi.hip ensure_btab) and this draw's slice of the pre-pass workspace as buffer resources
    const __amdgpu_buffer_rsrc_t rs_tab = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(btab), 0, 0x7ffffffc, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_pw = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(pairs + b * NW * 128), 0, 0x7ffffffc, 0x00020000);
    const int lane8 = lane * 8;
    const int rsb8 = (int)(RSB * 8);
    const int Jy = R >> 4, ry = R & 15;            // block column / lane column of the y row
    const double mu = p.mu ? p.mu[b] : 0.0;
    const double nu = p.nu ? p.nu[b] : 1.0;
    const bool has_nu = p.nu != nullptr;
    const double* __restrict__ Ab_ = p.A + b * J;
    const double* __restrict__ Bb_ = p.Bc + b * J;
    double suma = 0.0;  // :21
    for (int j = 0; j < J; ++j) suma += Ab_[j];

    for (int r = lane; r < 16 * NB; r += 64) {
        double a = 0.0, bb = 0.0;
        if (r < R) {
            const int rm = p.rowmap[r];
            const int term = rm & 0xfffff;
            a = Ab_[term];
            bb = ((rm >> 30) & 1) ? -Bb_[term] : Bb_[term];
        }
        sw.albe[r] = double2{a, bb};
    }
    constexpr int NU = NB * (NB - 1) / 2;
    for (int i = lane; i < (NU > 0 ? NU : 1) * 16 * 18; i += 64) (&sw.up[0][0])[i] = 0.0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    constexpr int NT = NB * (NB + 1) / 2;          // T is symmetric: the tiles (I, Jc), I >= Jc, live in registers
    auto tix = [](int I, int Jc) constexpr { return I * (I + 1) / 2 + Jc; };
    auto uix = [](int I, int Jc) constexpr { return I * (I - 1) / 2 + Jc; };   // I > Jc: slot of the tile's LDS copy
    d4 T[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) T[i] = d4{0.0, 0.0, 0.0, 0.0};
    double Uf[NB][4];

    // U~ of window k, A-operand order: (row 16 I + 4 ks + q, step c16), from the record's C o v and C o x (coalesced 512-byte reads), one
    // row block at a time: the reads of block I + 1 are on their way while block column I of T is updated
    double cvn[4], cxn[4];
    auto fetch_u = [&](int64_t k, int I) __attribute__((always_inline)) {
        const int so = (int)k * rsb8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            cvn[ks] = tile_bload(rs_tab, lane8, so + (I * 4 + ks) * 512);
            cxn[ks] = tile_bload(rs_tab, lane8, so + (NB * 256 + (I * 4 + ks) * 64) * 8);
        }
    };
    auto form_u = [&](int I) __attribute__((always_inline)) {
        double2 cf[4];          // the four (al, be) reads together, one wait (left to the compiler: read, wait, two FMAs, four times in a row)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) cf[ks] = sw.albe[16 * I + 4 * ks + q];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) Uf[I][ks] = fma(cf[ks].x, cvn[ks], cf[ks].y * cxn[ks]);
    };
    // C_K and sigma2 of window k: fetched a window ahead, staged in LDS at the start of the window
    constexpr int NCK = (16 * NB + 16 + 63) / 64;
    double ckpre[NCK];
    auto fetch_ck = [&](int64_t k) __attribute__((always_inline)) {
        const int so = (int)k * rsb8 + 3 * NB * 256 * 8;
#pragma unroll
        for (int i = 0; i < NCK; ++i) ckpre[i] = tile_bload(rs_tab, lane8, so + 512 * i);   // (past 16 NB + 16 doubles: padding / the pair table, not used)
    };
    // A of window k, C/D order (row 4 g + q, column c16): the off-diagonal entries from the workspace of tile_pairs_mfma_kernel (pair p =
    // nn (nn - 1) / 2 + jj, jj < nn), fetched one window ahead; the diagonal sum(a) + nu sigma2_n here (:92)
    int pidx[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int r = 4 * g + q, hi = r > c16 ? r : c16, lo = r > c16 ? c16 : r;
        pidx[g] = 8 * (r == c16 ? 127 : hi * (hi - 1) / 2 + lo);      // byte offset (127: a padding entry of the record, never used)
    }
    const int gd = (c16 - q) >> 2;                               // the register that holds this lane's diagonal entry, if (c16 - q) % 4 == 0
    const bool on_diag = ((c16 - q) & 3) == 0;
    double apre[4];
    auto fetch_A = [&](int64_t k) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) apre[g] = tile_bload(rs_pw, pidx[g], (int)k * 1024);
    };
    // per-draw series (y, sigma2) [B][N]: lanes 0 .. 15 fetch the window's sixteen steps a window ahead (clamped past the end: those steps are masked)
    const bool has_series = p.Y != nullptr;
    double ypre = 0.0, spre = 0.0;
    auto fetch_series = [&](int64_t k) __attribute__((always_inline)) {
        if (has_series) {
            int64_t n = k * KW + c16;
            n = n < N ? n : N - 1;
            ypre = p.Y[b * N + n];
            spre = p.S2[b * N + n];
        }
    };
    fetch_A(0);
    fetch_ck(0);
    fetch_series(0);
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        fetch_u(0, I);
        form_u(I);
    }
    // the y row's column of X': v = y_n - mu in the lane column ry of block Jy (z_n = y_n - u'f, :141); elsewhere mu_sel = 0
    double mu_sel[NB];
#pragma unroll
    for (int Jc = 0; Jc < NB; ++Jc) mu_sel[Jc] = (Jc == Jy && c16 == ry) ? mu : 0.0;
    const int64_t k_ragged = (N % KW) ? NW - 1 : NW;   // the window whose steps past N are padding (none if N is a multiple of 16)
    double quad = 0.0;                 // meaningful in the y-row lanes
#if PIORAN_TILE_QUADALL
    double quadb[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) quadb[I] = 0.0;
#endif
    double Pm = 1.0;                   // per lane (step c16 of every window): running product of |D| (sign of D_1 kept: :126)
    int Pe = 0;
    bool nonpd = false;

    PIORAN_TSTAMP_DECL
    for (int64_t k = 0; k < NW; ++k) {
        PIORAN_TSTAMP(0);
        const int wso = (int)k * rsb8;
        const bool more = k + 1 < NW;
#pragma unroll
        for (int i = 0; i < NCK; ++i)
            if (lane + 64 * i < 16 * NB + 16) sw.ck[lane + 64 * i] = ckpre[i];
        if (has_series) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane < 16) { sw.ys[lane] = ypre; sw.ck[16 * NB + lane] = spre; }
        }
        if (more) { fetch_ck(k + 1); fetch_series(k + 1); }
        if constexpr (ST) {
            d4* gtk = reinterpret_cast<d4*>(p.gw + ((b * NW + k) * NT) * 256);      // [tile][lane][register]: 32 bytes per lane, two 16-byte stores
#pragma unroll
            for (int i = 0; i < NT; ++i) gtk[i * 64 + lane] = T[i];
        }
        // (C_K / C) o v of the window: up to three block columns all of it now, behind the matrix instructions of M' (round 6: fetched a block ahead inside the
        // G phase before, its wait stood right behind the issue); more block columns: a block ahead as before (registers)
        constexpr bool VHA = PIORAN_TILE_VHA && (NB <= 3 || (PIORAN_TILE_BIG_VHA && NB >= 5));
        [[maybe_unused]] double vha[VHA ? NB : 1][4];
        if constexpr (VHA) {
#pragma unroll
            for (int I = 0; I < NB; ++I)
#pragma unroll
                for (int g = 0; g < 4; ++g) vha[I][g] = tile_bload(rs_tab, lane8, wso + (2 * NB * 256 + (I * 4 + g) * 64) * 8);
#if PIORAN_TILE_RAGFIX
            // the padded steps of the last, ragged window: V^' - mu = 0 there.  Once, under the wave-uniform test, on the loaded values (as a select inside the
            // block-column loop: four v_cndmask per block column and step in EVERY window, 48 of the ~520 vector instructions)
            if (k == k_ragged) {
#pragma unroll
                for (int I = 0; I < NB; ++I)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (k * KW + 4 * g + q >= N) vha[I][g] = mu_sel[I];
            }
#endif
        }
        // ---- M' = U~' T: the lower tiles from registers, the upper ones as transposed reads of their LDS copies -------------------
        d4 x[NB];
#pragma unroll
        for (int Jc = 0; Jc < NB; ++Jc) {
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int I = 0; I < Jc; ++I) {
                double bt[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) bt[ks] = sw.up[uix(Jc, I)][c16 * 18 + 4 * ks + q];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Uf[I][ks], bt[ks], acc, 0, 0, 0);
            }
#pragma unroll
            for (int I = Jc; I < NB; ++I)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Uf[I][ks], T[tix(I, Jc)][ks], acc, 0, 0, 0);
            x[Jc] = acc;
        }
        PIORAN_TSTAMP(1);
        // ---- G = U~' M: each block of M' transposed through LDS into the B operand; X' = V^' - C_K o M' behind it (the record's
        //      (C_K / C) o v is fetched a block ahead) ------------------------------------------------------------------------
        d4 G = {0.0, 0.0, 0.0, 0.0};
        double vh[4];
        if constexpr (!VHA) {
#pragma unroll
            for (int g = 0; g < 4; ++g) vh[g] = tile_bload(rs_tab, lane8, wso + (2 * NB * 256 + g * 64) * 8);
        }
        double ysv[4] = {0.0, 0.0, 0.0, 0.0};     // per-draw series: read here, under the wave-uniform test (inside the block-column loop the compiler turns the
        if (has_series) {                         // test into a select and reads sw.ys for every block column, a wait each — also when there is no series)
#pragma unroll
            for (int g = 0; g < 4; ++g) ysv[g] = sw.ys[4 * g + q];
        }
#pragma unroll
        for (int Jc = 0; Jc < NB; ++Jc) {
#pragma unroll
            for (int g = 0; g < 4; ++g) sw.scr[(4 * g + q) * 18 + c16] = x[Jc][g];
            PIORAN_TILE_FWD_ORDER();
            double mb[4];   // M [row 16 Jc + 4 ks + q][step c16]
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) mb[ks] = sw.scr[c16 * 18 + 4 * ks + q];
            const double ckc = sw.ck[16 * Jc + c16];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) G = __builtin_amdgcn_mfma_f64_16x16x4f64(Uf[Jc][ks], mb[ks], G, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                double v;
                if constexpr (VHA) v = vha[Jc][g]; else v = vh[g];
#if !PIORAN_TILE_YSFIX
                if (has_series && Jc == Jy && c16 == ry) v = ysv[g];     // (the table's y row holds the shared series)
#endif
                v -= mu_sel[Jc];
#if PIORAN_TILE_RAGFIX && !PIORAN_TILE_RAGFIX4
                if constexpr (!VHA)      // (four block columns and more: (C_K / C) o v arrives a block ahead; a forced branch per block there costs more than the selects — 16.7 -> 21.9 ms)
#endif
#if !PIORAN_TILE_RAGFIX4
                if (k == k_ragged && k * KW + 4 * g + q >= N) v = 0.0;
#endif
                x[Jc][g] = fma(-ckc, x[Jc][g], v);
                asm volatile("" : "+v"(x[Jc][g]));   // formed HERE: left to itself the compiler sinks these FMAs below the LDL' and keeps (C_K / C) o v live across it
            }
            if constexpr (!VHA) {
                if (Jc + 1 < NB) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) vh[g] = tile_bload(rs_tab, lane8, wso + (2 * NB * 256 + ((Jc + 1) * 4 + g) * 64) * 8);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#if PIORAN_TILE_RAGFIX4
        // four block columns and more: the padded steps of the last, ragged window took V^' - mu like the others; taken back here, once, with the table's values
        // loaded again (a load cannot be speculated: this stays a branch)
        if constexpr (!VHA) {
            if (k == k_ragged) {
#pragma unroll
                for (int Jc = 0; Jc < NB; ++Jc)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const double vt = tile_bload(rs_tab, lane8, wso + (2 * NB * 256 + (Jc * 4 + g) * 64) * 8);
                        if (k * KW + 4 * g + q >= N) x[Jc][g] -= vt - mu_sel[Jc];
                    }
            }
        }
#endif
#if PIORAN_TILE_YSFIX
        // per-draw series: X' = V^' - C_K o M' took the table's (shared) series in the y row; the draw's own replaces it here, under the wave-uniform test — as a
        // select inside the block-column loop it cost 24 v_cndmask per window whether there is a series or not
        if (has_series) {
            static_for<0, NB>([&](auto Jcc) __attribute__((always_inline)) {
                constexpr int Jc = decltype(Jcc)::value;
                if (Jc == Jy) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        double vt;
                        if constexpr (VHA) vt = vha[Jc][g]; else vt = tile_bload(rs_tab, lane8, wso + (2 * NB * 256 + (Jc * 4 + g) * 64) * 8);
                        const bool pad = k == k_ragged && k * KW + 4 * g + q >= N;
                        if (c16 == ry && !pad) x[Jc][g] += ysv[g] - vt;
                    }
                }
            });
        }
#endif
        PIORAN_TSTAMP(2);
        PIORAN_TSTAMP(3);
        // ---- Sigma = A - G, Sigma = L D L', L^-1 ----------------------------------------------------------------------------
        {
            const double s2n = sw.ck[16 * NB + c16];
            const double dg = k * KW + c16 < N ? suma + (has_nu ? nu * s2n : s2n) : 1.0;   // :92; padded steps of the last window: D = 1
#pragma unroll
            for (int g = 0; g < 4; ++g) sw.scr[(4 * g + q) * 16 + c16] = ((on_diag && g == gd) ? dg : apre[g]) - G[g];
        }
        if (k + 1 < NW) fetch_A(k + 1);
        PIORAN_TILE_FWD_ORDER();
        double m[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) m[j] = sw.scr[j * 16 + c16];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PIORAN_TSTAMP(4);
        __builtin_amdgcn_s_setprio(2);     // the dependent chain of the window: ahead of the SIMD's other wavefront's matrix work (+1 .. 2 %)
        double mult = ldl_first_mult(m, c16);
        static_for<0, 16>([&](auto Pc) __attribute__((always_inline)) { ldl_step<decltype(Pc)::value>(m, mult, c16); });
        __builtin_amdgcn_s_setprio(0);
        PIORAN_TSTAMP(5);
        if (q == 0) {   // lane n holds column n of L^-1 (scaled by D_n) in m[j], j > n, D_n in m[n]; the rest of m is left-over Sigma (masked below)
            double2* dst = reinterpret_cast<double2*>(sw.scr + c16 * 18);
#pragma unroll
            for (int j = 0; j < 16; j += 2) dst[j / 2] = double2{m[j], m[j + 1]};
        }
        PIORAN_TILE_FWD_ORDER();
        double li[4], idv[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kk = 4 * ks + q;                             // L^-1 [i = c16][k = kk]: unit lower triangular
            const double lv = sw.scr[kk * 18 + c16];
            idv[ks] = recip_f64(sw.scr[kk * 18 + kk]);             // D_n sits on the diagonal
            li[ks] = kk < c16 ? lv * idv[ks] : (kk == c16 ? 1.0 : 0.0);   // the column arrives scaled by D_kk
        }
        {   // log-determinant bookkeeping: this lane follows step c16 of every window
            const double dj = sw.scr[c16 * 18 + c16];
            nonpd |= !(dj > 0.0);
            Pm *= (k == 0 && c16 == 0) ? dj : fabs(dj);    // log(D[1]) :126, log(abs(D[n])) :140
            int ex;
            Pm = frexp(Pm, &ex);
            Pe += ex;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PIORAN_TSTAMP(6);
        // ---- Y^' = L^-1 X' ------------------------------------------------------------------------------------------------
        if (more) fetch_u(k + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        d4 yt[NB];
#pragma unroll
        for (int Jc = 0; Jc < NB; ++Jc) {
            d4 a = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a = __builtin_amdgcn_mfma_f64_16x16x4f64(li[ks], x[Jc][ks], a, 0, 0, 0);
            yt[Jc] = a;
        }
        PIORAN_TSTAMP(7);
        // C_K of the tiles' rows and columns: up to three block columns read once, in front of the tiles (round 6: left at its uses, every tile starts with an
        // LDS read and a wait for it; SHO-20 10.18 -> 9.97 ms per 4096 draws).  With four block columns the 20 values cost spilled registers (DRWCelerite-20
        // 15.99 -> 16.33 ms): read per tile there, as before.
        constexpr bool CKP = NB <= 3 || (PIORAN_TILE_BIG_CKP && NB >= 5);
        [[maybe_unused]] double ckrow[CKP ? NB : 1][4], ckcol[CKP ? NB : 1];
        if constexpr (CKP) {
#pragma unroll
            for (int I = 0; I < NB; ++I) {
#pragma unroll
                for (int g = 0; g < 4; ++g) ckrow[I][g] = sw.ck[16 * I + 4 * g + q];
                ckcol[I] = sw.ck[16 * I + c16];
            }
        }
        auto ck_row = [&](int I, int g) __attribute__((always_inline)) -> double { if constexpr (CKP) return ckrow[I][g]; else return sw.ck[16 * I + 4 * g + q]; };
        // ---- T <- (C_K C_K') o T + Y^ D^-1 Y^', lower tiles, one block column at a time; the off-diagonal ones are copied to LDS for the
        //      next window's M'; U~ of the next window is formed on the way ------------------------------------------------------
#pragma unroll
        for (int Jc = 0; Jc < NB; ++Jc) {
            if (more) {
                form_u(Jc);
                if (Jc + 1 < NB) fetch_u(k + 1, Jc + 1);
            }
            double ysc[4];
            double ckc;
            if constexpr (CKP) ckc = ckcol[Jc]; else ckc = sw.ck[16 * Jc + c16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                ysc[g] = yt[Jc][g] * idv[g];
#if PIORAN_TILE_QUADALL
                quadb[Jc] = fma(yt[Jc][g], ysc[g], quadb[Jc]);         // z_n^2 / D_n (== y'K^-1 y, :333) of EVERY block column; the y row's is picked after the loop
#else                                                                  // (picked here — `if (Jc == Jy)` — it is an FMA and two v_cndmask per block column and step)
                if (Jc == Jy) quad = fma(yt[Jc][g], ysc[g], quad);     // z_n^2 / D_n (== y'K^-1 y, :333), in the y-row lanes
#endif
            }
#pragma unroll
            for (int I = Jc; I < NB; ++I) {
#pragma unroll
                for (int g = 0; g < 4; ++g) T[tix(I, Jc)][g] *= ck_row(I, g) * ckc;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) T[tix(I, Jc)] = __builtin_amdgcn_mfma_f64_16x16x4f64(yt[I][ks], ysc[ks], T[tix(I, Jc)], 0, 0, 0);
                if (Jc < I) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) sw.up[uix(I, Jc)][(4 * g + q) * 18 + c16] = T[tix(I, Jc)][g];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        PIORAN_TSTAMP(8);
        PIORAN_TILE_FWD_ORDER();
    }

    PIORAN_TSTAMP_FLUSH
    // ---- result ------------------------------------------------------------------------------------------------------------
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if PIORAN_TILE_QUADALL
#pragma unroll
    for (int I = 0; I < NB; ++I)
        if (I == Jy) quad = quadb[I];
#endif
    if (c16 == ry) sw.scr[q] = quad;
    if (q == 0) sw.scr[16 + c16] = log(Pm) + (double)Pe * 0.6931471805599453094;
    const bool any_nonpd = __builtin_amdgcn_ballot_w64(nonpd) != 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) {
        const double qs = (sw.scr[0] + sw.scr[1]) + (sw.scr[2] + sw.scr[3]);
        double logdet = 0.0;
        for (int j = 0; j < 16; ++j) logdet += sw.scr[16 + j];
        const double res = -0.5 * logdet - 0.5 * (double)N * 1.8378770664093454836 - 0.5 * qs;
        p.out[b] = res;
        if (p.status) p.status[b] = !isfinite(res) ? 2 : (any_nonpd ? 1 : 0);
    }
}

// ---- reverse mode, one draw per wavefront (round 5) --------------------------------------------------------------------------------------
// Gradient of log L with respect to (a_j, b_j, mu, nu) through the windowed form, walking the windows backwards.  Same adjoint algebra as
// celerite_block_adjoint_kernel (celerite_block.hip; numpy prototype tools/block_adjoint_proto.py, checked against the complex-step oracle), with
// K = Sigma^-1, Q' = K X' (primes: steps x rows):
//   X-' = 2 Q' T-  (- Q'[:, y] in the y column);   S- = -1/2 K - Q' T- Q + 1/2 q_y q_y';   M-' = -cK o X-' - S- U~'
//   U~-' = (M-' - S- U~') T   (= -S- M' + M-' T: ONE product with T instead of two);   T- <- (cK cK') o T- + 1/2 (U~ M-' + M- U~')
//   d/dal_r += sum_n U~-'[n][r] (C v)[n][r], d/dbe_r likewise with x;  d/dmu -= sum_n X-'[n][y];  d/dsum(a) += tr S-;  d/dnu += sum_n S-_nn sigma2_n;
//   d/da_t += 2 sum_pairs S-_jn E_t,p.cos, d/db_t += 2 sum_pairs S-_jn E_t,p.sin   (tile_pairs_grad_kernel, from the symmetrised S- this kernel leaves
//   where the pre-pass left A)
// What differs from the small-batch reverse kernel is what the forward pass keeps: there T, M', Q' (twice) and K per window — 41 KB per window and chain
// each way, 210 GB for 4096 chains at N = 1e4 — here ONLY the lower tiles of T (12 KB at three block columns); M', Sigma, the LDL', Q' and K are
// recomputed from it (76 matrix instructions + the factorisation), then the adjoint's products run on tiles (144): ~2.6 forward windows per reverse window.
// T_k is read twice as B operand (M' and U~-'): its lower tiles as fragments straight from the workspace (the second time from L2), its upper
// tiles as transposed reads of LDS copies; T- lives like T in the forward kernel (lower tiles in registers, transposed LDS copies).
template <int NB, bool CD = false>
struct TileAdjWave {
    double scr[16 * 18];           // transposing scratch (blocks of M', Q', X-' ... in turn); D_k (L^-1)_ik
    double srm[16 * 16];           // S- row-major
    double2 albe[16 * NB];
    double ck[16 * NB + 16];       // C_K per row, sigma2 per step
    double qy[16];
    double red[16 * NB][4];        // end of the kernel: per-row sums over the four step quarters
    double tk[NB * (NB + 1) / 2][16 * 18];                 // the lower tiles of T_k, [row][column]: read as they stand (tiles on and below the
                                                           // diagonal) and transposed (above it) — registers hold T_k only on its way here
    double upB[NB > 1 ? NB * (NB - 1) / 2 : 1][16 * 18];   // strictly lower tiles of T-
    double mw[CD ? NB : 1][CD ? 256 : 1];                  // d/d(c, d): M' of the window, C/D order [register][lane] (x goes on to hold X', then Q')
    double accd[CD ? 2 * NB : 1][CD ? 64 : 1];             // ... its per-lane sums (d/dc rows | d/dd rows of block column Jc) and the window's time stamps t_n, t_b, t_e:
    double tm[CD ? 24 : 1];                                //     in LDS, not in registers (24 live registers less: 175 -> 90 spilled, 4096 chains of SHO-20 84 -> 69 ms)
    double scrb[(NB <= 3 && !CD) ? NB : 1][(NB <= 3 && !CD) ? 16 * 18 : 1];   // one transposing scratch per block column where LDS has room (tile_adj_batched): the
                                                           // window's 15 LDS round trips (write a block, wait, read it transposed, wait) become 5
    double acab[2 * NB][64];                               // per-lane sums of d/dal | d/dbe of block column Jc, the same move for both instantiations: 43 -> 1 spilled
                                                           // registers without d/d(c, d), 90 -> 4 with (53.7 -> 49.9 ms, 69 -> 57 ms per 4096 chains of SHO-20)
};

// CD (round 6): also the ROW part of d/d(c_t, d_t) with (c, d) shared by the chains (the formulas of celerite_block_adjoint_kernel<.., CD>:
//   d/dc_r -= sum_n U~-'[n][r] U~'[n][r] (t_n - t_b) + sum_n X-'[n][r] V^'[n][r] (t_e - t_n) + cK-_r cK_r (t_e - t_b),  cK-_r = 2 sum_j T-_jr cK_j T_jr - sum_n X-'[n][r] M'[n][r]
//   d/dd_r += s_r sum_n t_n (U~-'[n][r] (al_r C x - be_r C v)[n][r] + X-'[n][r] ((C_K / C) x)[n][r]),  s_r = -1 (cos row), +1 (sin row));
// the pair part (d/dc E = -tau E, d/dd (E.cos, E.sin) = tau (-E.sin, E.cos)) is the post-pass's second product (tile_pairs_grad_kernel<true>).
template <int NB, bool CD>
constexpr bool tile_adj_batched() { return NB <= 3 && !CD; }
template <int NB, bool CD = false>
__global__ void __launch_bounds__((64 * tile_adj_waves<NB, CD>()), 1) celerite_tile_adjoint_kernel(const ScanParams p, const double* __restrict__ btab,
                                                                                                  const double* __restrict__ gtab, double* __restrict__ pairs,
                                                                                                  double* __restrict__ grad_a, double* __restrict__ grad_b,
                                                                                                  double* __restrict__ grad_nu, double* __restrict__ grad_mu,
                                                                                                  double* __restrict__ grad_c, double* __restrict__ grad_d)
{
    constexpr int TS = 3 * NB * 256 + 16 * NB + 16, TSP = (TS + 127) & ~127;
    constexpr int64_t GS = 4 * (int64_t)NB * 256 + 16 * NB + 16 + 24;     // block_gtab_doubles (celerite_block.hip): C o v | C o x (C/D order) | C_K | sigma2 | ...
    constexpr int NT = NB * (NB + 1) / 2, NU = NB * (NB - 1) / 2;
    constexpr bool BT = tile_adj_batched<NB, CD>();
    extern __shared__ double lds_[];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, q = lane >> 4, c16 = lane & 15;
    const int64_t b = (int64_t)blockIdx.x * tile_adj_waves<NB, CD>() + w;
    if (b >= p.B) return;
    const int64_t N = p.N;
    const int J = p.J, R = p.R;
    const int64_t NW = (N + KW - 1) / KW;
    const int64_t RSB = TSP + 256 * (int64_t)J;
    TileAdjWave<NB, CD>& sw = reinterpret_cast<TileAdjWave<NB, CD>*>(lds_)[w];
    const __amdgpu_buffer_rsrc_t rs_tab = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(btab), 0, 0x7ffffffc, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_gt = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(gtab), 0, 0x7ffffffc, 0x00020000);
    double* const pw = pairs + b * NW * 128;
    const double* const gtb = p.gw + b * NW * NT * 256;       // T_k of this draw
    const int lane8 = lane * 8;
    const int rsb8 = (int)(RSB * 8), gs8 = (int)(GS * 8);
    const int Jy = R >> 4, ry = R & 15;
    const double mu = p.mu ? p.mu[b] : 0.0;
    const double nu = p.nu ? p.nu[b] : 1.0;
    const bool has_nu = p.nu != nullptr;
    const double* __restrict__ Ab_ = p.A + b * J;
    const double* __restrict__ Bb_ = p.Bc + b * J;
    double suma = 0.0;
    for (int j = 0; j < J; ++j) suma += Ab_[j];
    for (int r = lane; r < 16 * NB; r += 64) {
        double a = 0.0, bb = 0.0;
        if (r < R) {
            const int rm = p.rowmap[r];
            const int term = rm & 0xfffff;
            a = Ab_[term];
            bb = ((rm >> 30) & 1) ? -Bb_[term] : Bb_[term];
        }
        sw.albe[r] = double2{a, bb};
    }
    for (int i = lane; i < (NU > 0 ? NU : 1) * 16 * 18; i += 64) (&sw.upB[0][0])[i] = 0.0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    double2 myab[NB];          // (al, be) of the row this lane's column stands for in block w
    double ymask[NB], mu_sel[NB];
#pragma unroll
    for (int Jc = 0; Jc < NB; ++Jc) {
        myab[Jc] = sw.albe[16 * Jc + c16];
        ymask[Jc] = (Jc == Jy && c16 == ry) ? 1.0 : 0.0;
        mu_sel[Jc] = ymask[Jc] * mu;
    }
    auto tix = [](int I, int Jc) constexpr { return I * (I + 1) / 2 + Jc; };
    auto uix = [](int I, int Jc) constexpr { return I * (I - 1) / 2 + Jc; };
    d4 Tb[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) Tb[i] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int Jc = 0; Jc < 2 * NB; ++Jc) sw.acab[Jc][lane] = 0.0;
    double acc_mu = 0.0, acc_sa = 0.0, acc_nu = 0.0;
    if constexpr (CD) {
#pragma unroll
        for (int Jc = 0; Jc < 2 * NB; ++Jc) sw.accd[Jc][lane] = 0.0;
    }
    // per-lane sums in LDS: ds_add_f64 without a return value — nothing to wait for
    // (with d/d(c, d) the plain read - add - write stays: there the atomic form costs 19 more spilled registers, 57 -> 60 ms per 4096 chains of SHO-20)
    auto lds_add = [](double* slot, double v) __attribute__((always_inline)) {
        if constexpr (CD) *slot += v;
        else (void)__hip_atomic_fetch_add(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    };
    [[maybe_unused]] auto add_c = [&](int Jc, double v) __attribute__((always_inline)) { lds_add(&sw.accd[Jc][lane], v); };
    [[maybe_unused]] auto add_d = [&](int Jc, double v) __attribute__((always_inline)) { lds_add(&sw.accd[NB + Jc][lane], v); };
    [[maybe_unused]] constexpr int OFF_H = 2 * NB * 256 + 16 * NB + 16, OFF_TM = OFF_H + 2 * NB * 256;     // gtab: (C_K / C) o v | (C_K / C) o x | t_n x 16, t_b, t_e
    int pidx[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int r = 4 * g + q, hi = r > c16 ? r : c16, lo = r > c16 ? c16 : r;
        pidx[g] = r == c16 ? 127 : hi * (hi - 1) / 2 + lo;
    }
    const int gd = (c16 - q) >> 2;
    const bool on_diag = ((c16 - q) & 3) == 0;
    int pnn[2], pjj[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int pq = lane + 64 * h;
        int n_ = 1;
        while ((n_ + 1) * n_ / 2 <= pq) ++n_;
        pnn[h] = n_ < 16 ? n_ : 15;
        pjj[h] = n_ < 16 ? pq - n_ * (n_ - 1) / 2 : 0;
    }
    const int64_t k_ragged = (N % KW) ? NW - 1 : NW;
    // Global reads of a window and when they are issued (one wavefront per SIMD, 512 registers: no operand is a load inside a chain of matrix
    // instructions — the first version had T_k's fragments there, at two wavefronts per SIMD with 58 spilled registers: 48.8 ms per 2048 chains):
    //   T_k (lower tiles, two 16-byte loads each), C o v / C o x in A-operand order, C_K / sigma2, the pre-pass's pair block: for window k - 1 right after
    //   the last use of T_k in window k (the update of T- and the head of the next window cover their latency);
    //   (C_K / C) o v at the head of the window (used after M'), C o v / C o x in C/D order before phase A (used in C and D).
    // T_k itself sits in LDS for the window (lower tiles, read as they stand or transposed): registers hold it only on its way there.
    constexpr int NCK = (16 * NB + 16 + 63) / 64;
    constexpr int NWARM = (NT * 2048 + 8191) / 8192;
    constexpr bool PFT = NB <= 3;       // T_k a window ahead in registers; four block columns: 80 registers the kernel does not have (253 spilled: a window
                                        // took four times the cycles of three block columns) — there T_k is loaded at the head of its own window, in two halves
    struct WinIn {
        d4 T[PFT ? NT : 1];
        double cva[PFT ? NB : 1][4], cxa[PFT ? NB : 1][4];      // C o v, C o x, A-operand order (btab)
        double ckp[NCK];
        double ap[4];
        float warm[PFT ? 1 : NWARM];        // four block columns: one word of every 128-byte line of the next T_k, so that the head's loads find it in L2
    };
    auto fetch_win = [&](int64_t kk, WinIn& wi) __attribute__((always_inline)) {
        const int wso = (int)kk * rsb8;
        const d4* tk = reinterpret_cast<const d4*>(gtb + kk * NT * 256);
        if constexpr (PFT) {
#pragma unroll
            for (int i = 0; i < NT; ++i) wi.T[i] = tk[i * 64 + lane];
        } else {
            const float* tw = reinterpret_cast<const float*>(tk);
#pragma unroll
            for (int j = 0; j < NWARM; ++j) {
                const int line = j * 64 + lane;
                wi.warm[j] = tw[(line < NT * 16 ? line : NT * 16 - 1) * 32];
            }
        }
        if constexpr (PFT) {
#pragma unroll
            for (int I = 0; I < NB; ++I)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    wi.cva[I][ks] = tile_bload(rs_tab, lane8, wso + (I * 4 + ks) * 512);
                    wi.cxa[I][ks] = tile_bload(rs_tab, lane8, wso + (NB * 256 + (I * 4 + ks) * 64) * 8);
                }
        }
#pragma unroll
        for (int i = 0; i < NCK; ++i) wi.ckp[i] = tile_bload(rs_tab, lane8, wso + 3 * NB * 256 * 8 + 512 * i);
#pragma unroll
        for (int g = 0; g < 4; ++g) wi.ap[g] = pw[kk * 128 + pidx[g]];
    };
    WinIn cur;
    fetch_win(NW - 1, cur);
    PIORAN_ASTAMP2_DECL

    for (int64_t k = NW - 1; k >= 0; --k) {
        PIORAN_ASTAMP2(0);
        const int wso = (int)k * rsb8, gso = (int)k * gs8;
        [[maybe_unused]] d4 tnow[PFT ? 1 : NT];      // four block columns: T_k of this window, first in the queue (warmed into L2 a window ago)
        [[maybe_unused]] double cvh[PFT ? 1 : NB][4], cxh[PFT ? 1 : NB][4];
        if constexpr (!PFT) {
#pragma unroll
            for (int j = 0; j < NWARM; ++j) asm volatile("" ::"v"(cur.warm[j]));
#pragma unroll
            for (int I = 0; I < NB; ++I)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    cvh[I][ks] = tile_bload(rs_tab, lane8, wso + (I * 4 + ks) * 512);
                    cxh[I][ks] = tile_bload(rs_tab, lane8, wso + (NB * 256 + (I * 4 + ks) * 64) * 8);
                }
            const d4* tkg = reinterpret_cast<const d4*>(gtb + k * NT * 256);
#pragma unroll
            for (int i = 0; i < NT; ++i) tnow[i] = tkg[i * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);      // (all of them in flight together: left to itself the compiler issues them pair by pair next to their uses,
                                                    //  sixteen memory round trips in a row — 14 000 of a window's 57 000 cycles)
        }
        double vhs[NB][4];
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int g = 0; g < 4; ++g) vhs[I][g] = tile_bload(rs_tab, lane8, wso + (2 * NB * 256 + (I * 4 + g) * 64) * 8);
#if PIORAN_TILE_ADJ_RAGFIX
        if (k == k_ragged) {        // the padded steps of the last, ragged window: V^' - mu = 0 there — once, on the loaded values (celerite_tile_kernel, PIORAN_TILE_RAGFIX)
#pragma unroll
            for (int I = 0; I < NB; ++I)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (k * KW + 4 * g + q >= N) vhs[I][g] = mu_sel[I];
        }
#endif
        // ---- this window's inputs ---------------------------------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < NCK; ++i)
            if (lane + 64 * i < 16 * NB + 16) sw.ck[lane + 64 * i] = cur.ckp[i];
        double apre[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) apre[g] = cur.ap[g];
        double Uf[NB][4];
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const double2 cf = sw.albe[16 * I + 4 * ks + q];
                if constexpr (PFT) Uf[I][ks] = fma(cf.x, cur.cva[I][ks], cf.y * cur.cxa[I][ks]);
                else Uf[I][ks] = fma(cf.x, cvh[I][ks], cf.y * cxh[I][ks]);
            }
        // the lower tiles of T_k -> LDS
        if constexpr (PFT) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) sw.tk[i][(4 * g + q) * 18 + c16] = cur.T[i][g];
        } else {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) sw.tk[i][(4 * g + q) * 18 + c16] = tnow[i][g];
        }
        PIORAN_LDS_ORDER();
        __builtin_amdgcn_sched_barrier(0);
        PIORAN_ASTAMP2(1);
        // ---- forward window again: M' = U~' T_k ----------------------------------------------------------------------------------
        d4 x[NB];
#pragma unroll
        for (int Jc = 0; Jc < NB; ++Jc) {
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int I = 0; I < Jc; ++I) {
                double bt[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) bt[ks] = sw.tk[tix(Jc, I)][c16 * 18 + 4 * ks + q];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Uf[I][ks], bt[ks], acc, 0, 0, 0);
            }
#pragma unroll
            for (int I = Jc; I < NB; ++I) {
                double bt[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) bt[ks] = sw.tk[tix(I, Jc)][(4 * ks + q) * 18 + c16];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Uf[I][ks], bt[ks], acc, 0, 0, 0);
            }
            x[Jc] = acc;
        }
        __builtin_amdgcn_sched_barrier(0);
        PIORAN_ASTAMP2(2);
        // ---- G = U~' M, X' = V^' - C_K o M' ----------------------------------------------------------------------------------------
        d4 G = {0.0, 0.0, 0.0, 0.0};
        if constexpr (BT) {
            double mb[NB][4], ckc[NB];
#pragma unroll
            for (int Jc = 0; Jc < NB; ++Jc)
#pragma unroll
                for (int g = 0; g < 4; ++g) sw.scrb[Jc][(4 * g + q) * 18 + c16] = x[Jc][g];
            PIORAN_LDS_ORDER();
#pragma unroll
            for (int Jc = 0; Jc < NB; ++Jc) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) mb[Jc][ks] = sw.scrb[Jc][c16 * 18 + 4 * ks + q];
                ckc[Jc] = sw.ck[16 * Jc + c16];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int Jc = 0; Jc < NB; ++Jc)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) G = __builtin_amdgcn_mfma_f64_16x16x4f64(Uf[Jc][ks], mb[Jc][ks], G, 0, 0, 0);
#pragma unroll
            for (int Jc = 0; Jc < NB; ++Jc)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    double v = vhs[Jc][g] - mu_sel[Jc];
#if !PIORAN_TILE_ADJ_RAGFIX
                    if (k == k_ragged && k * KW + 4 * g + q >= N) v = 0.0;
#endif
                    x[Jc][g] = fma(-ckc[Jc], x[Jc][g], v);
                    asm volatile("" : "+v"(x[Jc][g]));
                }
        } else {
#pragma unroll
        for (int Jc = 0; Jc < NB; ++Jc) {
#pragma unroll
            for (int g = 0; g < 4; ++g) sw.scr[(4 * g + q) * 18 + c16] = x[Jc][g];
            if constexpr (CD) {
#pragma unroll
                for (int g = 0; g < 4; ++g) sw.mw[Jc][g * 64 + lane] = x[Jc][g];
            }
            PIORAN_LDS_ORDER();
            double mb[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) mb[ks] = sw.scr[c16 * 18 + 4 * ks + q];
            const double ckc = sw.ck[16 * Jc + c16];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) G = __builtin_amdgcn_mfma_f64_16x16x4f64(Uf[Jc][ks], mb[ks], G, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                double v = vhs[Jc][g] - mu_sel[Jc];
#if !PIORAN_TILE_ADJ_RAGFIX
                if (k == k_ragged && k * KW + 4 * g + q >= N) v = 0.0;
#endif
                x[Jc][g] = fma(-ckc, x[Jc][g], v);
                asm volatile("" : "+v"(x[Jc][g]));
            }
        }
        }
        __builtin_amdgcn_sched_barrier(0);
        PIORAN_ASTAMP2(3);
        // ---- Sigma = A - G, LDL', L^-1 ---------------------------------------------------------------------------------------------
        const double s2n = sw.ck[16 * NB + c16];
        const bool live = k * KW + c16 < N;
        {
            const double dg = live ? suma + (has_nu ? nu * s2n : s2n) : 1.0;
#pragma unroll
            for (int g = 0; g < 4; ++g) sw.scr[(4 * g + q) * 16 + c16] = ((on_diag && g == gd) ? dg : apre[g]) - G[g];
        }
        PIORAN_LDS_ORDER();
        double m[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) m[j] = sw.scr[j * 16 + c16];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PIORAN_ASTAMP2(4);
        double mult = ldl_first_mult(m, c16);
        static_for<0, 16>([&](auto Pc) __attribute__((always_inline)) { ldl_step<decltype(Pc)::value>(m, mult, c16); });
        PIORAN_ASTAMP2(5);
        if (q == 0) {
            double2* dst = reinterpret_cast<double2*>(sw.scr + c16 * 18);
#pragma unroll
            for (int j = 0; j < 16; j += 2) dst[j / 2] = double2{m[j], m[j + 1]};
        }
        PIORAN_LDS_ORDER();
        // operands: L^-1 (A operand of Y^' = L^-1 X'), L^-T D^-1 (A operand of Q' = L^-T D^-1 Y^'), and K = L^-T D^-1 L^-1
        double li[4], lt[4], la[4], lb[4];
        {
            const double idm = recip_f64(sw.scr[c16 * 18 + c16]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int kk = 4 * ks + q;
                const double idk = recip_f64(sw.scr[kk * 18 + kk]);
                const double lv = sw.scr[kk * 18 + c16];          // D_kk (L^-1)[c16][kk], c16 > kk
                const double lu = sw.scr[c16 * 18 + kk];          // D_c16 (L^-1)[kk][c16], kk > c16
                li[ks] = kk < c16 ? lv * idk : (kk == c16 ? 1.0 : 0.0);
                lb[ks] = kk > c16 ? lu * idm : (kk == c16 ? 1.0 : 0.0);      // (L^-1)[kk][c16]
                la[ks] = lb[ks] * idk;
                lt[ks] = la[ks];                                               // (L^-T D^-1)[c16][kk] = (L^-1)[kk][c16] / D_kk
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        d4 Kv = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) Kv = __builtin_amdgcn_mfma_f64_16x16x4f64(la[ks], lb[ks], Kv, 0, 0, 0);
        PIORAN_ASTAMP2(6);
        // ---- Q' = Sigma^-1 X' ---------------------------------------------------------------------------------------------------------
        if constexpr (BT) {      // the block columns' chains side by side: a product waits for the one that feeds its B operand, not for its accumulator
            d4 yt[NB], qv[NB];
#pragma unroll
            for (int Jc = 0; Jc < NB; ++Jc) { yt[Jc] = d4{0.0, 0.0, 0.0, 0.0}; qv[Jc] = d4{0.0, 0.0, 0.0, 0.0}; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int Jc = 0; Jc < NB; ++Jc) yt[Jc] = __builtin_amdgcn_mfma_f64_16x16x4f64(li[ks], x[Jc][ks], yt[Jc], 0, 0, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int Jc = 0; Jc < NB; ++Jc) qv[Jc] = __builtin_amdgcn_mfma_f64_16x16x4f64(lt[ks], yt[Jc][ks], qv[Jc], 0, 0, 0);
#pragma unroll
            for (int Jc = 0; Jc < NB; ++Jc) x[Jc] = qv[Jc];
        } else {
#pragma unroll
            for (int Jc = 0; Jc < NB; ++Jc) {
                d4 yt = {0.0, 0.0, 0.0, 0.0}, qv = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) yt = __builtin_amdgcn_mfma_f64_16x16x4f64(li[ks], x[Jc][ks], yt, 0, 0, 0);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) qv = __builtin_amdgcn_mfma_f64_16x16x4f64(lt[ks], yt[ks], qv, 0, 0, 0);
                x[Jc] = qv;                                            // from here on x holds Q'
            }
        }
        if (c16 == ry) {                                            // q_y: the y column of Q' (block Jy), by step
            static_for<0, NB>([&](auto Ic) __attribute__((always_inline)) {
                constexpr int I = decltype(Ic)::value;
                if (I == Jy) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) sw.qy[4 * g + q] = x[I][g];
                }
            });
        }
        __builtin_amdgcn_sched_barrier(0);
        PIORAN_ASTAMP2(7);
        double cvs[NB][4], cxs[NB][4];      // C o v, C o x, C/D order (gtab): on their way during phases A and B
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                cvs[I][g] = tile_bload(rs_gt, lane8, gso + ((I * 4 + g) * 64) * 8);
                cxs[I][g] = tile_bload(rs_gt, lane8, gso + (NB * 256 + (I * 4 + g) * 64) * 8);
            }
        __builtin_amdgcn_sched_barrier(0);
        // ---- A: Q in A-operand order; X-' = 2 Q' T- (- q_y in the y column); P = Q' T- Q -----------------------------------------------
        double qf[NB][4];
        d4 xb[NB];
        d4 P = {0.0, 0.0, 0.0, 0.0};
        if constexpr (BT) {
#pragma unroll
            for (int I = 0; I < NB; ++I)
#pragma unroll
                for (int g = 0; g < 4; ++g) sw.scrb[I][(4 * g + q) * 18 + c16] = x[I][g];
            PIORAN_LDS_ORDER();
#pragma unroll
            for (int I = 0; I < NB; ++I)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) qf[I][ks] = sw.scrb[I][c16 * 18 + 4 * ks + q];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            d4 qt[NB];
#pragma unroll
            for (int Jc = 0; Jc < NB; ++Jc) {
                qt[Jc] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int I = 0; I < Jc; ++I) {
                    double bt[4];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) bt[ks] = sw.upB[uix(Jc, I)][c16 * 18 + 4 * ks + q];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) qt[Jc] = __builtin_amdgcn_mfma_f64_16x16x4f64(qf[I][ks], bt[ks], qt[Jc], 0, 0, 0);
                }
#pragma unroll
                for (int I = Jc; I < NB; ++I)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) qt[Jc] = __builtin_amdgcn_mfma_f64_16x16x4f64(qf[I][ks], Tb[tix(I, Jc)][ks], qt[Jc], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) sw.scrb[Jc][(4 * g + q) * 18 + c16] = qt[Jc][g];
            }
            PIORAN_LDS_ORDER();
            double qtT[NB][4];
#pragma unroll
            for (int Jc = 0; Jc < NB; ++Jc)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) qtT[Jc][ks] = sw.scrb[Jc][c16 * 18 + 4 * ks + q];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int Jc = 0; Jc < NB; ++Jc) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) P = __builtin_amdgcn_mfma_f64_16x16x4f64(qf[Jc][ks], qtT[Jc][ks], P, 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    xb[Jc][g] = fma(-ymask[Jc], x[Jc][g], 2.0 * qt[Jc][g]);
                    acc_mu = fma(-ymask[Jc], xb[Jc][g], acc_mu);      // (padded steps: Q' = 0 there, nothing to mask)
                }
            }
        } else {
#pragma unroll
        for (int I = 0; I < NB; ++I) {
#pragma unroll
            for (int g = 0; g < 4; ++g) sw.scr[(4 * g + q) * 18 + c16] = x[I][g];
            PIORAN_LDS_ORDER();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qf[I][ks] = sw.scr[c16 * 18 + 4 * ks + q];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int Jc = 0; Jc < NB; ++Jc) {
            d4 qt = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int I = 0; I < Jc; ++I) {
                double bt[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) bt[ks] = sw.upB[uix(Jc, I)][c16 * 18 + 4 * ks + q];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) qt = __builtin_amdgcn_mfma_f64_16x16x4f64(qf[I][ks], bt[ks], qt, 0, 0, 0);
            }
#pragma unroll
            for (int I = Jc; I < NB; ++I)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) qt = __builtin_amdgcn_mfma_f64_16x16x4f64(qf[I][ks], Tb[tix(I, Jc)][ks], qt, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) sw.scr[(4 * g + q) * 18 + c16] = qt[g];
            PIORAN_LDS_ORDER();
            double qtT[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qtT[ks] = sw.scr[c16 * 18 + 4 * ks + q];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) P = __builtin_amdgcn_mfma_f64_16x16x4f64(qf[Jc][ks], qtT[ks], P, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                xb[Jc][g] = fma(-ymask[Jc], x[Jc][g], 2.0 * qt[g]);
                acc_mu = fma(-ymask[Jc], xb[Jc][g], acc_mu);      // (padded steps: Q' = 0 there, nothing to mask)
            }
        }
        }
        __builtin_amdgcn_sched_barrier(0);
        PIORAN_ASTAMP2(8);
        // ---- B: S- = -1/2 K - P + 1/2 q_y q_y' -------------------------------------------------------------------------------------------
        {
            const double qyc = sw.qy[c16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const double sb = fma(0.5 * sw.qy[4 * g + q], qyc, fma(-0.5, Kv[g], -P[g]));
                sw.srm[(4 * g + q) * 16 + c16] = sb;
                if (on_diag && g == gd && live) {
                    acc_sa += sb;
                    acc_nu = fma(sb, s2n, acc_nu);
                }
            }
        }
        PIORAN_LDS_ORDER();
        double sA[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) sA[ks] = 0.5 * (sw.srm[c16 * 16 + 4 * ks + q] + sw.srm[(4 * ks + q) * 16 + c16]);   // symmetrised
        // the pairs' S-_nj + S-_jn for the pair-table contraction of the post-pass: where the pre-pass left A (read above)
        {
            const double sv0 = sw.srm[pnn[0] * 16 + pjj[0]] + sw.srm[pjj[0] * 16 + pnn[0]];
            const double sv1 = sw.srm[pnn[1] * 16 + pjj[1]] + sw.srm[pjj[1] * 16 + pnn[1]];
            pw[k * 128 + lane] = sv0;
            pw[k * 128 + 64 + lane] = lane < 120 - 64 ? sv1 : 0.0;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        PIORAN_ASTAMP2(9);
        // ---- C: U~' (C/D order), S- U~', W' = M-' - S- U~' = -cK o X-' - 2 S- U~', M-' ------------------------------------------------------
        d4 uw[NB], mbk[NB];
#pragma unroll
        for (int Jc = 0; Jc < NB; ++Jc)
#pragma unroll
            for (int g = 0; g < 4; ++g) uw[Jc][g] = fma(myab[Jc].x, cvs[Jc][g], myab[Jc].y * cxs[Jc][g]);
        double wa[NB][4];          // W' in A-operand order
        // d/d(c, d), round 6: the three groups of terms sit where their factor is consumed anyway — X-' here, U~-' in phase D, T- o T_k in the update — so
        // that no operand's life grows (all of them in phase D: 334 spilled registers, the reverse kernel twice as slow)
        if constexpr (CD) {
            if (lane < 18) sw.tm[lane] = tile_bload(rs_gt, lane8, gso + OFF_TM * 8);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        if constexpr (BT) {      // (never with d/d(c, d): tile_adj_batched)
            d4 su[NB];
#pragma unroll
            for (int Jc = 0; Jc < NB; ++Jc) su[Jc] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int Jc = 0; Jc < NB; ++Jc) su[Jc] = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[ks], uw[Jc][ks], su[Jc], 0, 0, 0);
#pragma unroll
            for (int Jc = 0; Jc < NB; ++Jc) {
                const double ckc = sw.ck[16 * Jc + c16];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const double cx_ = -ckc * xb[Jc][g];
                    mbk[Jc][g] = cx_ - su[Jc][g];
                    sw.scrb[Jc][(4 * g + q) * 18 + c16] = cx_ - 2.0 * su[Jc][g];
                }
            }
            PIORAN_LDS_ORDER();
#pragma unroll
            for (int Jc = 0; Jc < NB; ++Jc)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) wa[Jc][ks] = sw.scrb[Jc][c16 * 18 + 4 * ks + q];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
#pragma unroll
        for (int Jc = 0; Jc < NB; ++Jc) {
            [[maybe_unused]] double hvw[4], hxw[4];
            if constexpr (CD) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    hvw[g] = tile_bload(rs_gt, lane8, gso + (OFF_H + (Jc * 4 + g) * 64) * 8);
                    hxw[g] = tile_bload(rs_gt, lane8, gso + (OFF_H + NB * 256 + (Jc * 4 + g) * 64) * 8);
                }
            }
            d4 su = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) su = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[ks], uw[Jc][ks], su, 0, 0, 0);
            const double ckc = sw.ck[16 * Jc + c16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const double cx_ = -ckc * xb[Jc][g];
                mbk[Jc][g] = cx_ - su[g];
                sw.scr[(4 * g + q) * 18 + c16] = cx_ - 2.0 * su[g];
            }
            PIORAN_LDS_ORDER();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wa[Jc][ks] = sw.scr[c16 * 18 + 4 * ks + q];
            if constexpr (CD) {             // the X-' terms: -sum X-' V^' (t_e - t_n), + sum t_n X-' ((C_K / C) x), and cK-_r's  - sum X-' M'  times  -cK_r (t_e - t_b)
                double xm = 0.0, pc = 0.0, pd = 0.0;
                const double tbw = sw.tm[16], tew = sw.tm[17];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const double tn = sw.tm[4 * g + q];
                    xm = fma(xb[Jc][g], sw.mw[Jc][g * 64 + lane], xm);
                    pc = fma(-xb[Jc][g] * hvw[g], tew - tn, pc);
                    pd = fma(tn * xb[Jc][g], hxw[g], pd);
                }
                add_c(Jc, fma(xm * ckc, tew - tbw, pc));
                add_d(Jc, pd);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        }
        __builtin_amdgcn_sched_barrier(0);
        PIORAN_ASTAMP2(10);
        // ---- D: U~-' = W' T_k -> d/dal, d/dbe;  T- <- (cK cK') o T- + 1/2 (U~ M-' + M- U~') --------------------------------------------------
#pragma unroll
        for (int Jc = 0; Jc < NB; ++Jc) {
            d4 ub = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int I = 0; I < Jc; ++I) {
                double bt[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) bt[ks] = sw.tk[tix(Jc, I)][c16 * 18 + 4 * ks + q];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) ub = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[I][ks], bt[ks], ub, 0, 0, 0);
            }
#pragma unroll
            for (int I = Jc; I < NB; ++I) {
                double bt[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) bt[ks] = sw.tk[tix(I, Jc)][(4 * ks + q) * 18 + c16];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) ub = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[I][ks], bt[ks], ub, 0, 0, 0);
            }
            {
                double pa = 0.0, pb = 0.0;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    pa = fma(ub[g], cvs[Jc][g], pa);
                    pb = fma(ub[g], cxs[Jc][g], pb);
                }
                lds_add(&sw.acab[Jc][lane], pa);
                lds_add(&sw.acab[NB + Jc][lane], pb);
            }
            if constexpr (CD) {             // the U~-' terms of d/d(c, d) (the X-' terms: phase C; the T- o T_k term: the update below)
                double pc = 0.0, pd = 0.0;
                const double tbw = sw.tm[16];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const double tn = sw.tm[4 * g + q];
                    pc = fma(-ub[g] * uw[Jc][g], tn - tbw, pc);
                    pd = fma(tn * ub[g], fma(myab[Jc].x, cxs[Jc][g], -myab[Jc].y * cvs[Jc][g]), pd);
                }
                add_c(Jc, pc);
                add_d(Jc, pd);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        PIORAN_ASTAMP2(11);
        if (k > 0) fetch_win(k - 1, cur);          // T_k has had its last use: window k - 1's inputs, straight into the same registers
        __builtin_amdgcn_sched_barrier(0);
        // C_K of the tiles' rows and columns read once, in front of the tiles (a read and a wait per tile before).  Not with d/d(c, d): 20 more registers
        // there are 20 .. 30 more spilled ones (57 -> 59 ms per 4096 chains of SHO-20).  (Also measured, round 6: all rescalings, then the matrix instructions of
        // all tiles back to back, then the LDS copies — no faster, 45.9 -> 47.1 ms at three block columns.)
        [[maybe_unused]] double ckrow[CD ? 1 : NB][4], ckcol[CD ? 1 : NB];
        if constexpr (!CD) {
#pragma unroll
            for (int I = 0; I < NB; ++I) {
#pragma unroll
                for (int g = 0; g < 4; ++g) ckrow[I][g] = sw.ck[16 * I + 4 * g + q];
                ckcol[I] = sw.ck[16 * I + c16];
            }
        }
        auto ck_row = [&](int I, int g) __attribute__((always_inline)) -> double { if constexpr (CD) return sw.ck[16 * I + 4 * g + q]; else return ckrow[I][g]; };
        auto ck_col = [&](int I) __attribute__((always_inline)) -> double { if constexpr (CD) return sw.ck[16 * I + c16]; else return ckcol[I]; };
#pragma unroll
        for (int Jc = 0; Jc < NB; ++Jc) {
            const double ckc = ck_col(Jc);
            double hm[4], hu[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { hm[ks] = 0.5 * mbk[Jc][ks]; hu[ks] = 0.5 * uw[Jc][ks]; }
#pragma unroll
            for (int I = Jc; I < NB; ++I) {
                if constexpr (CD) {
                    // cK-_r's  2 sum_j T-_jr cK_j T_jr  times  -cK_r (t_e - t_b), tile by tile, T- BEFORE this window's rescaling: this tile's entries for
                    // the columns of block Jc (this lane: column c16, rows 4 g + q of block I), and — tiles below the diagonal — their mirror images for the
                    // columns of block I, read transposed from the LDS copies (T- from the previous window's update, T_k from this window's head)
                    double s1 = 0.0;
#pragma unroll
                    for (int g = 0; g < 4; ++g) s1 = fma(Tb[tix(I, Jc)][g] * ck_row(I, g), sw.tk[tix(I, Jc)][(4 * g + q) * 18 + c16], s1);
                    const double tspan = sw.tm[17] - sw.tm[16];
                    add_c(Jc, -2.0 * s1 * ckc * tspan);
                    if (Jc < I) {
                        double s2 = 0.0;
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            s2 = fma(sw.upB[uix(I, Jc)][c16 * 18 + 4 * g + q] * ck_row(Jc, g), sw.tk[tix(I, Jc)][c16 * 18 + 4 * g + q], s2);
                        add_c(I, -2.0 * s2 * ck_col(I) * tspan);
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) Tb[tix(I, Jc)][g] *= ck_row(I, g) * ckc;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    Tb[tix(I, Jc)] = __builtin_amdgcn_mfma_f64_16x16x4f64(uw[I][ks], hm[ks], Tb[tix(I, Jc)], 0, 0, 0);
                    Tb[tix(I, Jc)] = __builtin_amdgcn_mfma_f64_16x16x4f64(mbk[I][ks], hu[ks], Tb[tix(I, Jc)], 0, 0, 0);
                }
                if (Jc < I) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) sw.upB[uix(I, Jc)][(4 * g + q) * 18 + c16] = Tb[tix(I, Jc)][g];
                }
            }
        }
        PIORAN_LDS_ORDER();
        PIORAN_ASTAMP2(12);
    }
    PIORAN_ASTAMP2_FLUSH

    // ---- reductions: steps -> rows -> terms --------------------------------------------------------------------------------------------------
#pragma unroll
    for (int Jc = 0; Jc < NB; ++Jc) {
        sw.red[16 * Jc + c16][q] = sw.acab[Jc][lane];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    double row_al = 0.0, row_be = 0.0;     // lane r < 16 NB (two rounds when NB > 4 is not reached here: 16 NB <= 64)
    if (lane < 16 * NB) row_al = (sw.red[lane][0] + sw.red[lane][1]) + (sw.red[lane][2] + sw.red[lane][3]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int Jc = 0; Jc < NB; ++Jc) sw.red[16 * Jc + c16][q] = sw.acab[NB + Jc][lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane < 16 * NB) row_be = (sw.red[lane][0] + sw.red[lane][1]) + (sw.red[lane][2] + sw.red[lane][3]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // scalars: sum over the wavefront (every lane holds a share)
    double s_mu = acc_mu, s_sa = acc_sa, s_nu = acc_nu;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s_mu += __shfl_xor(s_mu, off);
        s_sa += __shfl_xor(s_sa, off);
        s_nu += __shfl_xor(s_nu, off);
    }
    [[maybe_unused]] double row_c = 0.0, row_d = 0.0;
    if constexpr (CD) {
#pragma unroll
        for (int Jc = 0; Jc < NB; ++Jc) sw.red[16 * Jc + c16][q] = sw.accd[Jc][lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < 16 * NB) row_c = (sw.red[lane][0] + sw.red[lane][1]) + (sw.red[lane][2] + sw.red[lane][3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int Jc = 0; Jc < NB; ++Jc) sw.red[16 * Jc + c16][q] = sw.accd[NB + Jc][lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < 16 * NB) row_d = (sw.red[lane][0] + sw.red[lane][1]) + (sw.red[lane][2] + sw.red[lane][3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // rows -> terms through LDS (scr as ra | rb | rc | rd, at most 64 terms: 16 x 18 doubles hold 4 x 64)
    double* ra = sw.scr;
    double* rb = sw.scr + 64;
    for (int t = lane; t < 256; t += 64) sw.scr[t] = 0.0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane < R) {
        const int rm = p.rowmap[lane];
        const int term = rm & 0xfffff;
        atomicAdd(&ra[term], row_al);
        atomicAdd(&rb[term], ((rm >> 30) & 1) ? -row_be : row_be);
        if constexpr (CD) {
            atomicAdd(&sw.scr[128 + term], row_c);
            atomicAdd(&sw.scr[192 + term], ((rm >> 30) & 1) ? row_d : -row_d);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int t = lane; t < J; t += 64) {
        grad_a[b * J + t] = ra[t] + s_sa;
        grad_b[b * J + t] = rb[t];
        if constexpr (CD) {
            grad_c[b * J + t] = sw.scr[128 + t];
            grad_d[b * J + t] = sw.scr[192 + t];
        }
    }
    if (lane == 0) {
        if (grad_mu) grad_mu[b] = s_mu;
        if (grad_nu) grad_nu[b] = s_nu;
    }
}

// d/da_t += sum over windows and pairs of (S-_nj + S-_jn) E_t,p.cos, d/db_t likewise with sin: the pair-table contraction of the reverse pass, from
// the symmetrised S- that celerite_tile_adjoint_kernel left in the pre-pass workspace.  It is a GEMM — out[draw][(t, cos | sin)] = sum over (window,
// pair) of S-[draw][(window, pair)] E[(window, pair)][(t, cos | sin)], 4096 x 2 J x 80 000 at the bench shape, 26 GFLOP — and runs on the matrix cores:
// a workgroup owns 16 draws (the A operand's rows) and its eight wavefronts split the windows; per 16 pairs a lane loads four consecutive values of
// its draw (A operand of four MFMA steps: the k index of step s in lane group q is pair 16 g + 4 q + s on both operands) and, per 16 output columns
// (eight terms x (cos, sin)), four consecutive table entries; the eight partial tiles are summed through LDS in a fixed order.
// (Vector forms measured before, per 4096 chains at N = 1e4, 20 terms: thread = (draw, term) with the table in LDS 6.8 .. 8.0 ms — one 16-byte LDS read
//  per two FMAs; lanes = pairs with the table from L2 14.5 ms; lanes = pairs with the table in LDS and 40 accumulators per lane ~5 ms, two draws per
//  wavefront 247 spilled registers.)
constexpr int kPairGradTiles = 6;      // 16-column tiles of (term, cos | sin): 2 J <= 94 columns at the 47 rows of three block columns
// CD (round 6): the pair part of d/d(c_t, d_t) with shared (c, d) — d/dc E_t,p = -tau_p E_t,p, d/dd (E.cos, E.sin) = tau_p (-E.sin, E.cos), tau_p = t_n - t_j of
// the pair — is the same product with the A operand scaled by tau_p:  G_t = sum (S-_nj + S-_jn) tau_p E_t,p  (cos | sin), then per chain
//   d/dc_t -= a_t G_t.cos + b_t G_t.sin,   d/dd_t += b_t G_t.cos - a_t G_t.sin;
// a second set of accumulators, twice the matrix instructions.  The window's 16 time stamps sit in LDS per wavefront.
template <bool CD>
__global__ void __launch_bounds__(512) tile_pairs_grad_kernel(const ScanParams p, const double* __restrict__ btab, int64_t rsb, int64_t tsp,
                                                              const double* __restrict__ pairs, double* __restrict__ grad_a, double* __restrict__ grad_b,
                                                              double* __restrict__ grad_c, double* __restrict__ grad_d)
{
    __shared__ double red[8][256];
    __shared__ double tw[8][16];
    __shared__ double fin[256];
    const int J = p.J, nct = (2 * J + 15) / 16;
    const int64_t NW = (p.N + KW - 1) / KW;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
    const int64_t b0 = (int64_t)blockIdx.x * 16;
    const int64_t draw = b0 + i < p.B ? b0 + i : p.B - 1;       // (rows past the batch: a live draw's values, never written)
    d4 acc[kPairGradTiles];
    [[maybe_unused]] d4 acc2[CD ? kPairGradTiles : 1];
#pragma unroll
    for (int ct = 0; ct < kPairGradTiles; ++ct) acc[ct] = d4{0.0, 0.0, 0.0, 0.0};
    if constexpr (CD) {
#pragma unroll
        for (int ct = 0; ct < kPairGradTiles; ++ct) acc2[ct] = d4{0.0, 0.0, 0.0, 0.0};
    }
    int toff[kPairGradTiles];
#pragma unroll
    for (int ct = 0; ct < kPairGradTiles; ++ct) {               // (columns past the last term: the last term's entries, never written)
        const int t = 8 * ct + (i >> 1);
        toff[ct] = (t < J ? t : J - 1) * 128 + 4 * q;
    }
    // the steps (n > j) of this lane's pairs 16 g + 4 q + s, packed n | j << 4 (pair p = n (n - 1) / 2 + j; pairs 120 .. 127: n = j = 0, tau = 0)
    [[maybe_unused]] unsigned char nj[8][4];
    if constexpr (CD) {
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                const int pp = 16 * g + 4 * q + s_;
                int n_ = 1;
                while ((n_ + 1) * n_ / 2 <= pp) ++n_;
                nj[g][s_] = pp < 120 ? (unsigned char)(n_ | ((pp - n_ * (n_ - 1) / 2) << 4)) : (unsigned char)0;
            }
    }
    const bool sine = i & 1;
    for (int64_t k = w; k < NW; k += 8) {
        const double* sv = pairs + (draw * NW + k) * 128 + 4 * q;
        const double2* E = reinterpret_cast<const double2*>(btab + k * rsb + tsp);
        if constexpr (CD) {
            if (lane < 16) { const int64_t n = k * KW + lane; tw[w][lane] = p.t[n < p.N ? n : p.N - 1]; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (one wavefront writes and reads its own row: LDS serves a wavefront in order)
        }
#pragma unroll 2
        for (int g = 0; g < 8; ++g) {                           // (pairs 120 .. 127: zeros from the reverse kernel, zeros in the table)
            const d4 a = *reinterpret_cast<const d4*>(sv + 16 * g);
            [[maybe_unused]] d4 at;
            if constexpr (CD) {
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) at[s_] = a[s_] * (tw[w][nj[g][s_] & 15] - tw[w][nj[g][s_] >> 4]);
            }
#pragma unroll
            for (int ct = 0; ct < kPairGradTiles; ++ct) {
                if (ct < nct) {
                    const double2* e = E + toff[ct] + 16 * g;
                    const double2 e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3];
                    const double b0_ = sine ? e0.y : e0.x, b1_ = sine ? e1.y : e1.x, b2_ = sine ? e2.y : e2.x, b3_ = sine ? e3.y : e3.x;
                    acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b0_, acc[ct], 0, 0, 0);
                    acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b1_, acc[ct], 0, 0, 0);
                    acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b2_, acc[ct], 0, 0, 0);
                    acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b3_, acc[ct], 0, 0, 0);
                    if constexpr (CD) {
                        acc2[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(at[0], b0_, acc2[ct], 0, 0, 0);
                        acc2[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(at[1], b1_, acc2[ct], 0, 0, 0);
                        acc2[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(at[2], b2_, acc2[ct], 0, 0, 0);
                        acc2[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(at[3], b3_, acc2[ct], 0, 0, 0);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int ct = 0; ct < kPairGradTiles; ++ct) {
        if (ct < nct) {
#pragma unroll
            for (int g = 0; g < 4; ++g) red[w][g * 64 + lane] = acc[ct][g];
            __syncthreads();
            if (threadIdx.x < 256) {
                double sum = 0.0;
#pragma unroll
                for (int ww = 0; ww < 8; ++ww) sum += red[ww][threadIdx.x];
                const int gg = threadIdx.x >> 6, ll = threadIdx.x & 63;      // C/D order: register gg of lane ll = (row 4 gg + (ll >> 4), column ll & 15)
                const int64_t b = b0 + 4 * gg + (ll >> 4);
                const int col = 16 * ct + (ll & 15), t = col >> 1;
                if (b < p.B && t < J) { double* o = (col & 1) ? grad_b : grad_a; o[b * J + t] += sum; }
            }
            __syncthreads();
            if constexpr (CD) {
#pragma unroll
                for (int g = 0; g < 4; ++g) red[w][g * 64 + lane] = acc2[ct][g];
                __syncthreads();
                if (threadIdx.x < 256) {
                    double sum = 0.0;
#pragma unroll
                    for (int ww = 0; ww < 8; ++ww) sum += red[ww][threadIdx.x];
                    fin[threadIdx.x] = sum;
                }
                __syncthreads();
                if (threadIdx.x < 256 && !(threadIdx.x & 1)) {                // the (cos, sin) columns of a term sit in neighbouring lanes
                    const int gg = threadIdx.x >> 6, ll = threadIdx.x & 63;
                    const int64_t b = b0 + 4 * gg + (ll >> 4);
                    const int t = (16 * ct + (ll & 15)) >> 1;
                    if (b < p.B && t < J) {
                        const double gco = fin[threadIdx.x], gsi = fin[threadIdx.x + 1], a_ = p.A[b * J + t], b_ = p.Bc[b * J + t];
                        grad_c[b * J + t] -= fma(a_, gco, b_ * gsi);
                        grad_d[b * J + t] += fma(b_, gco, -a_ * gsi);
                    }
                }
                __syncthreads();
            }
        }
    }
}

template <int NB>
constexpr size_t tile_lds_bytes() { return kTileWaves * sizeof(TileWave<NB>); }

template <int NB>
int launch_tile(const ScanParams& p, const double* btab, double* pairs, hipStream_t stream)
{
    constexpr size_t lds = tile_lds_bytes<NB>();
    static_assert(lds <= 160 * 1024, "one workgroup must fit a CU");
    static bool granted[64] = {};   // (function, device): one process may drive several devices (pioran_farm_*); racing threads at worst set it twice
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return PIORAN_ERR_HIP;
    if (!granted[dev]) {
        if (hipFuncSetAttribute((const void*)celerite_tile_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return PIORAN_ERR_HIP;
        granted[dev] = true;
    }
    const int64_t groups = (p.B + kTileWaves - 1) / kTileWaves;
    if (groups > 0x7fffffffLL) return PIORAN_ERR_UNSUPPORTED;
    const int64_t NW = (p.N + KW - 1) / KW;
    if (NW > 0x7fffffffLL) return PIORAN_ERR_UNSUPPORTED;
    if (!launch_pairs_mfma(p, stream, btab, (int64_t)block_rec_doubles(NB, p.J), (int64_t)block_tile_doubles(NB), pairs)) return PIORAN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((celerite_tile_kernel<NB>), dim3((unsigned)groups), dim3(64 * kTileWaves), lds, stream, p, btab, (const double*)pairs);
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}

template <int NB>
constexpr size_t tile_adj_lds_bytes() { return tile_adj_waves<NB>() * sizeof(TileAdjWave<NB>); }

template <int NB>
int launch_tile_grad(const ScanParams& p, const double* btab, const double* gtab, double* pairs, double* grad_a, double* grad_b, double* grad_nu,
                     double* grad_mu, double* grad_c, double* grad_d, hipStream_t stream)
{
    constexpr size_t lds_f = tile_lds_bytes<NB>(), lds_r = tile_adj_lds_bytes<NB>(), lds_rc = tile_adj_waves<NB, true>() * sizeof(TileAdjWave<NB, true>);
    constexpr int AW = tile_adj_waves<NB>(), AWC = tile_adj_waves<NB, true>();
    static_assert(lds_f <= 160 * 1024 && lds_r <= 160 * 1024 && lds_rc <= 160 * 1024, "one workgroup must fit a CU");
    static bool granted[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return PIORAN_ERR_HIP;
    const bool cd = grad_c && grad_d;
    if ((grad_c != nullptr) != (grad_d != nullptr)) return PIORAN_ERR_ARG;
    if (!granted[dev]) {
        if (hipFuncSetAttribute((const void*)celerite_tile_kernel<NB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f) != hipSuccess) return PIORAN_ERR_HIP;
        if (hipFuncSetAttribute((const void*)celerite_tile_adjoint_kernel<NB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r) != hipSuccess) return PIORAN_ERR_HIP;
        if (hipFuncSetAttribute((const void*)celerite_tile_adjoint_kernel<NB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rc) != hipSuccess) return PIORAN_ERR_HIP;
        granted[dev] = true;
    }
    const int64_t groups = (p.B + kTileWaves - 1) / kTileWaves;
    const int64_t NW = (p.N + KW - 1) / KW;
    if (groups > 0x7fffffffLL || NW > 0x7fffffffLL) return PIORAN_ERR_UNSUPPORTED;
    const int64_t rsb = block_rec_doubles(NB, p.J), tsp = block_tile_doubles(NB);
    if (!launch_pairs_mfma(p, stream, btab, rsb, tsp, pairs)) return PIORAN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((celerite_tile_kernel<NB, true>), dim3((unsigned)groups), dim3(64 * kTileWaves), lds_f, stream, p, btab, (const double*)pairs);
    const unsigned agroups = (unsigned)((p.B + AW - 1) / AW);
    if (cd) {
        hipLaunchKernelGGL((celerite_tile_adjoint_kernel<NB, true>), dim3((unsigned)((p.B + AWC - 1) / AWC)), dim3(64 * AWC), lds_rc, stream, p, btab, gtab, pairs, grad_a, grad_b,
                           grad_nu, grad_mu, grad_c, grad_d);
        hipLaunchKernelGGL(tile_pairs_grad_kernel<true>, dim3((unsigned)((p.B + 15) / 16)), dim3(512), 0, stream, p, btab, rsb, tsp, (const double*)pairs, grad_a, grad_b,
                           grad_c, grad_d);
    } else {
        hipLaunchKernelGGL((celerite_tile_adjoint_kernel<NB, false>), dim3(agroups), dim3(64 * AW), lds_r, stream, p, btab, gtab, pairs, grad_a, grad_b,
                           grad_nu, grad_mu, (double*)nullptr, (double*)nullptr);
        hipLaunchKernelGGL(tile_pairs_grad_kernel<false>, dim3((unsigned)((p.B + 15) / 16)), dim3(512), 0, stream, p, btab, rsb, tsp, (const double*)pairs, grad_a, grad_b,
                           (double*)nullptr, (double*)nullptr);
    }
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}

}  // namespace

int pioran_tile_supported_rows() { return 95; }
int pioran_tile_grad_supported_rows() { return 63; }   // four block columns (DRWCelerite-20 is 60 rows; there three draws per workgroup: tile_adj_waves)

// doubles of the state workspace of the reverse mode: the lower tiles of T at the start of every window, per draw
size_t pioran_tile_grad_workspace_doubles(int64_t B, int64_t N, int32_t R)
{
    const int NB = (R + 1 + 15) / 16;
    return (size_t)B * (size_t)((N + KW - 1) / KW) * (size_t)(NB * (NB + 1) / 2) * 256;
}

// log L and its gradient with respect to (a, b, mu, nu) and — grad_c, grad_d both given — the shared (c, d), shared series; p.gw: pioran_tile_grad_workspace_doubles, pairs:
// pioran_tile_workspace_doubles; btab / gtab: the tables of pioran_launch_block_table / pioran_launch_block_gtab for the same (N, R, J, rowmap)
int pioran_launch_tile_grad(const ScanParams& p, const double* btab, const double* gtab, double* pairs, double* grad_a, double* grad_b, double* grad_nu,
                            double* grad_mu, double* grad_c, double* grad_d, hipStream_t stream)
{
    if (!btab || !gtab || !pairs || !p.gw || !grad_a || !grad_b || p.B < 1 || p.N < 1 || p.npd_rows != 0 || p.Y || p.S2 || p.J > kTileMaxTerms ||
        p.R < 1 || p.R > pioran_tile_grad_supported_rows())
        return PIORAN_ERR_UNSUPPORTED;
    switch ((p.R + 1 + 15) / 16) {
        case 1: return launch_tile_grad<1>(p, btab, gtab, pairs, grad_a, grad_b, grad_nu, grad_mu, grad_c, grad_d, stream);
        case 2: return launch_tile_grad<2>(p, btab, gtab, pairs, grad_a, grad_b, grad_nu, grad_mu, grad_c, grad_d, stream);
        case 3: return launch_tile_grad<3>(p, btab, gtab, pairs, grad_a, grad_b, grad_nu, grad_mu, grad_c, grad_d, stream);
        case 4: return launch_tile_grad<4>(p, btab, gtab, pairs, grad_a, grad_b, grad_nu, grad_mu, grad_c, grad_d, stream);
    }
    return PIORAN_ERR_UNSUPPORTED;
}

int pioran_tile_fits(int32_t R, int32_t J)
{
    const int NB = (R + 1 + 15) / 16;
    return R >= 1 && NB <= 6 && J >= 1 && J <= kTileMaxTerms;
}

// draws one full pass of the kernel holds on `cus` compute units
int64_t pioran_tile_pass_draws(int32_t R, int cus)
{
    const int NB = (R + 1 + 15) / 16;
    return (int64_t)cus * kTileWaves * (NB <= 4 ? 2 : 1);
}

// shared-(c, d) launches without per-draw rows; btab from pioran_launch_block_table for the same (N, R, J, rowmap);
// work: pioran_tile_workspace_doubles(B, N) doubles
// doubles of workspace a launch of B draws needs (the window's own covariance block per draw and window)
size_t pioran_tile_workspace_doubles(int64_t B, int64_t N) { return (size_t)B * (size_t)((N + KW - 1) / KW) * 128; }

int pioran_launch_scan_tile(const ScanParams& p, const double* btab, double* work, hipStream_t stream)
{
    if (!work) return PIORAN_ERR_ARG;
    if (!btab || p.B < 1 || p.N < 1 || p.npd_rows != 0 || !pioran_tile_fits(p.R, p.J)) return PIORAN_ERR_UNSUPPORTED;
    if ((p.Y == nullptr) != (p.S2 == nullptr)) return PIORAN_ERR_ARG;
    switch ((p.R + 1 + 15) / 16) {
        case 1: return launch_tile<1>(p, btab, work, stream);
        case 2: return launch_tile<2>(p, btab, work, stream);
        case 3: return launch_tile<3>(p, btab, work, stream);
        case 4: return launch_tile<4>(p, btab, work, stream);
        case 5: return launch_tile<5>(p, btab, work, stream);
        case 6: return launch_tile<6>(p, btab, work, stream);
    }
    return PIORAN_ERR_UNSUPPORTED;
}
