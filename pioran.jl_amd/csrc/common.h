// Shared declarations of the pioran-hip native library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PIORAN_OK 0
#define PIORAN_ERR_ARG (-1)
#define PIORAN_ERR_HIP (-2)
#define PIORAN_ERR_ALLOC (-3)
#define PIORAN_ERR_UNSUPPORTED (-4)

// layout of the shared per-step table ("trig/phi table"):
//   record n (n <= N, the last one a readable copy of N-1) = [ v_r | x_r | phi_r | y_n sigma2_n ], r < R + 2:
//   cos row (v, x) = (cos, sin)(d_j t_n), sin row (sin, cos); phi = exp(-c_j (t_n - t_{n-1}));
//   row R is the inert padding row (v, x, phi) = (1, 0, 1), row R+1 the y row (0, 0, 1).
// Diagnostic switches of a context (none is needed in production).  Read ONCE from the environment when the context is
// created (PIORAN_SCAN_CONFIG, PIORAN_NO_WIDE, PIORAN_NO_BLOCK, PIORAN_NO_PAIRED, PIORAN_NO_MIXED, PIORAN_FORCE_FALLBACK) and changed
// afterwards only through pioran_ctx_set_option — the launch path never calls getenv.
// dense path (dense.hip): diagnostics / tuning of the factorisation schedule, carried per context (-1 = the defaults of dense.hip)
struct DenseOptions {
    int quad_threshold = -1;        // trailing tiles per side above which a batched launch takes its steps in fours
    int pair_tiles = -1;            // dense_step_kernel: tiles of the trailing matrix above which the bulk goes in pairs of panels (-1: the measured default)
    int half_tile_limit = -1;       // dense_step_kernel: tiles per launch up to which a tile is split over two wavefronts (-1: default)
    int batch_pair_threshold = -1;  // ... in pairs
    int old_chain = 0;              // 1: one matrix on the panel / update chain of rounds 1-3 (A/B and cross-check of dense_step_kernel);
                                    // 2 / 3 / 4: timing experiments of dense_step_kernel's roles (tools/dense_roles.py)
    int no_halves = 0;              // 1: dense_step_kernel's bulk never splits a tile over two wavefronts
    int no_pairs = 0;               // 1: dense_step_kernel's bulk one panel per launch from the start (no paired phase)
};
struct ScanOptions {
    char scan_config[48];   // "" automatic; "wide" / "block": that small-batch kernel for any batch size; else a throughput configuration's name
    bool no_wide, no_paired, no_mixed, force_fallback, no_block;
    bool force_tp, no_tp;       // celerite_tp.hip (time-parallel evaluation of a handful of draws): force (scan_config "tp") / forbid
    int tp_segments = 0;        // ... its segment count (0 = automatic)
    int tp_scan = -1;           // ... its boundary phase: 1 the scan over the segments' elements (tp_combine_kernel, round 6), 0 the sequential walk, -1 automatic
    int tp_check = 0;           // ... what the scan's check goes by: 0 (default) the state distance on the scale of the innovation variance; tools: 1 the state discrepancy relative to its largest entry, 3 an estimate of log L's relative error from the distance
    bool tp_unchecked = false;  // ... no check, no repair: the family's own values whatever they are (tools, tests of the family's arithmetic; with scan_config "tp")
    bool tp_walk_repair = false; // ... draws whose scan fails its check go through the family's own boundary walk instead of the serial-chain kernel
    int tp_scan_waves = 0;      // ... 4: four wavefronts per combination also at 33 .. 48 rows (default there: eight)
    int tp_scan_lean = 0;       // ... 1: the scan's combinations with operands from global memory (tp_combine_lean_kernel) also below 49 rows (tests)
    double tp_scan_tol = 0.0;   // ... the scan's acceptance threshold (largest relative discrepancy of a boundary state; 0 = the default, celerite_tp.hip kTpScanTol; negative: every draw is repaired)
    bool force_tile, no_tile;   // celerite_tile.hip (windowed form, one draw per wavefront; default from 49 rows on above the small-batch range): force / forbid
    bool no_split;        // never send the remainder of a multi-pass batch to the windowed kernel on the second stream (capi.hip split_dispatch)
    bool win3, no_win3;   // throughput layouts with two / three rows per lane: force / forbid the three-step form (celerite_scan.hip;
                          // default: on for two rows per lane from 13 source lanes on, large batches)
    bool win2, no_win2;   // throughput layouts: force / forbid the two-step form of the recurrence (celerite_scan.hip)
    int dense_streams = 0;   // pioran_dense_nll_batch: factorisations per batched launch (0 = default 32; at most 64)
    DenseOptions dense;
    long long workspace_limit_mb = 0;   // absolute cap on a call's NEW chunked workspace, MiB (0 = default 16384; capi.hip ws_allow)
    int gsum = -1;        // throughput layouts, two-step form: row sums with fewer exchange rounds (group_sum's GS); -1 = automatic
    int block_emode = -1; // windowed kernel, diagnostics: where the pair table E lives (0 one LDS buffer, 1 two, 2 global memory); -1 automatic
    bool btab_reference;  // windowed kernel: build its table with the entry-per-thread kernel of round 2 (cross-check of the windowed table kernel)
    int exp;              // diagnostics: experiment selector of the kernel under study (0 in the product; tools/ only)
    bool wide2, no_wide2; // latency layout: force / forbid the lean form (celerite_wide2_kernel; default from 48 rows on)
};

struct ScanParams {
    const ScanOptions* opt;   // host-side only (the launch dispatch); nullptr = defaults
    int64_t N;            // time stamps
    int32_t J;            // celerite terms
    int32_t R;            // active rows (<= 2J; structurally-zero sin rows of d=b=0 terms dropped)
    int32_t standard_rows;  // 1: R = 2J and row 2j / 2j+1 are the cos / sin rows of term j (no dropped rows);
                            // 2: n_complex two-row terms first, then one-row (real) terms only — DRWCelerite (src/psd.jl:264-275)
    int32_t n_complex;      // leading two-row terms when standard_rows == 2
    int64_t B;            // batch (independent draws)
    const double* tab;    // shared table [N+1][3(R+2)+2], or nullptr when (c,d) are per draw
    const int32_t* rowmap;  // [R]: term (bits 0-19) | per-draw row index (20-28) | per-draw flag (29) | sin row (30)
    const double* t;      // [N]   (used only when tab == nullptr)
    const double* y;      // [N]   shared series (mean NOT subtracted), or nullptr if Y given
    const double* s2;     // [N]   shared measurement variances
    const double* Y;      // [B][N] per-draw series or nullptr
    const double* S2;     // [B][N] per-draw variances or nullptr
    const double* A;      // [B][J]
    const double* Bc;     // [B][J]
    const double* C;      // [J] or [B][J] (per-draw path)
    const double* D;      // [J] or [B][J]
    const double* mu;     // [B] or nullptr
    const double* nu;     // [B] or nullptr
    double* out;          // [B]
    int32_t* status;      // [B] or nullptr
    double* scratch;      // fallback kernel: [B][R*R + 4R]
    // step record stride of `tab` in doubles: 3(R+2)+2, plus B*npd_rows*3 in mixed mode, where the rows of the few
    // terms with per-draw (c, d) read (v, x, phi) from a per-draw block appended to every step record
    int64_t rec_stride;
    int64_t tab_draw_stride;   // celerite_wide2_kernel only: draw b reads the table at tab + b * tab_draw_stride (0: one shared table)
    int32_t npd_rows;     // per-draw rows (2 per per-draw term), 0 if none
    // celerite_block.hip with per-draw rows (the last npd_rows rows of the row map): the per-draw c [B][J] and the compact table of
    // (cos, sin)(d t_n) [B][npd terms][pd_npad][2] (pioran_launch_block_pd_trig)
    int64_t gtab_draw_stride;   // celerite_block_adjoint_kernel: doubles between per-draw reverse-pass tables (0: one shared table)
    double* gw;           // celerite_block.hip, gradient: workspace [B][windows][block_grad_ws_doubles] the forward pass leaves for the reverse pass
    const double* pd_C;
    const double* pd_trig;
    int64_t pd_npad;
    // celerite_wide.hip only.  Factor store (pioran_launch_scan_wide_store): W [B][N][R] (the reference's V after
    // init_semi_separable!, src/celerite_solver.jl:95-97), D [B][N], forward-solved z [B][N] (:141).
    double* st_w;
    double* st_d;
    double* st_z;
    // Simulation (pioran_launch_scan_wide_sim): standard-normal draws q [B][N] in, GP realisations [B][N] out
    // (sim, src/celerite_solver.jl:515-549); y / Y are not read.
    const double* noise;
    double* ysim;
    // Gradient (pioran_launch_scan_wide_grad).  The forward pass (MODE 3) stores v - q of ALL 16 RPL row slots in st_w
    // [B][N][16 RPL], D_n in st_d [B][N] and S_n — lane layout [256][RPL*RPL rounded up to even] — only at the CHECKPOINTS
    // n = k * ckpt_every (st_ck [B][nck][256][SP]).  The reverse pass walks the segments between checkpoints from the last to
    // the first: celerite_replay_kernel rebuilds S_n of one segment from its checkpoint and the stored (v - q, D) into
    // st_s [B][ckpt_every][256][SP], celerite_adjoint_kernel then runs steps seg_hi .. seg_lo backwards, its adjoint state
    // parked in st_state [B][NSTATE][256] between launches.  Outputs: row adjoints of (al, be) and the row accumulators of
    // dL/dd and dL/dc [B][16 RPL] each, the scalars (dL/dsum(a), dL/dnu, dL/dmu) [B][4] and, optionally, dL/dy_n, dL/dsigma2_n.
    double* st_s;
    double* st_ck;
    double* st_state;
    int32_t exp;          // diagnostics (ScanOptions::exp, copied by the launcher: kernels never dereference `opt`, a host pointer)
    int32_t ckpt_every;   // K
    int32_t seg_first;    // 1: the first launch of the reverse pass (adjoint state starts at zero)
    int64_t seg_n0;       // checkpoint step of the segment: st_s slot j holds S_{seg_n0 + 1 + j}
    int64_t seg_hi;       // steps seg_hi, seg_hi - 1, ..., seg_lo are processed by this adjoint launch
    int64_t seg_lo;
    double* g_al;
    double* g_be;
    double* g_d;
    double* g_c;
    double* g_scal;
    double* g_y;
    double* g_s2;
    // celerite_block_kernel as the repair pass behind the time-parallel scan (round 6): a workgroup runs only if only_if[draw] > only_if_tol
    // (only_if = nullptr: every draw) — the draws whose scan failed its check are evaluated again on the serial chain, into the same out / status
    const double* only_if;
    double only_if_tol;
};

// celerite_scan.hip
int pioran_launch_scan(const ScanParams& p, hipStream_t stream);
int64_t pioran_scan_pass_draws(const ScanParams& p, int* waves_per_simd);   // draws one full pass of the selected throughput kernel holds (0: unknown)
int pioran_scan_supported_rows();
int pioran_scan_supported_rows_shared();
const char* pioran_scan_config_name(int R);
// celerite_wide.hip: latency layout for small batches (one draw per workgroup); shared-table launches only
int pioran_launch_scan_wide(const ScanParams& p, hipStream_t stream);
int pioran_launch_scan_wide_store(const ScanParams& p, hipStream_t stream);   // log L + (W, D, z) to HBM
int pioran_launch_scan_wide_sim(const ScanParams& p, hipStream_t stream);     // y = L D^(1/2) q
// log L and its gradient with respect to (a_j, b_j) [B][J], nu, mu (reverse mode through the recurrence)
size_t pioran_grad_workspace_doubles(int64_t B, int64_t N, int32_t R);
int pioran_launch_scan_wide_grad(ScanParams p, double* work, double* grad_a, double* grad_b, double* grad_c, double* grad_d,
                                 double* grad_nu, double* grad_mu, hipStream_t stream, hipStream_t aux, hipEvent_t* ev /*[5]*/);
int pioran_wide_supported_rows();
int pioran_wide_supported_rows_grad();    // step-by-step reverse mode (143)
int pioran_wide_supported_rows_modes();   // store / simulate modes of the latency kernels (143)
int pioran_predict_supported_rows();      // step-by-step prediction (143: lean latency kernel's factor store + celerite_predict.hip)
int64_t pioran_wide_max_batch();
// celerite_block.hip: windowed form (16 steps per window on the matrix cores), one draw per workgroup; shared (c, d) without
// per-draw rows; its own table (fragment order), built once per prepared (c, d)
int pioran_block_supported_rows();
int pioran_block_fits(int32_t R, int32_t J);   // rows and terms within the kernel's LDS budget
int pioran_block_fits_value(int32_t R, int32_t J);   // the value-only kernel: also five and six block columns (64 .. 95 rows)
int pioran_block_fits_pd(int32_t R, int32_t J, int32_t npd_terms);   // ... with per-draw terms (at most two)
size_t pioran_block_pd_trig_doubles(int64_t N, int64_t B, int32_t npd_terms);
int pioran_launch_block_pd_trig(int64_t N, int64_t B, int32_t J, int32_t npd_terms, const int32_t* pd_terms /*device*/, const double* t,
                                const double* D /*[B][J]*/, double* out, hipStream_t stream);
size_t pioran_block_table_doubles(int64_t N, int32_t R, int32_t J);
int pioran_launch_block_table(int64_t N, int32_t R, int32_t J, const int32_t* rowmap, const double* t, const double* c,
                              const double* d, const double* y, const double* s2, double* btab, hipStream_t stream);
int pioran_launch_scan_block(const ScanParams& p, const double* btab, hipStream_t stream);   // p.tab_draw_stride: doubles between per-draw tables (0: one shared table)
int pioran_launch_block_table_batch(int64_t N, int32_t R, int32_t J, int64_t nb, const int32_t* rowmap, const double* t, const double* C /*[nb][J]*/,
                                    const double* D, const double* y, const double* s2, double* btab, int64_t draw_stride, hipStream_t stream);
// windowed reverse mode (gradient w.r.t. a, b, mu, nu): forward pass with stores + adjoint kernel
size_t pioran_block_grad_workspace_doubles(int64_t B, int64_t N, int32_t R);
size_t pioran_block_store_workspace_doubles(int64_t B, int64_t N, int32_t R, int what);   // prediction (what = 2) / simulation (3): packed stores
size_t pioran_block_gtab_doubles(int64_t N, int32_t R);
int pioran_launch_block_gtab(int64_t N, int32_t R, int32_t J, const int32_t* rowmap, const double* t, const double* c, const double* d,
                             const double* s2, double* gtab, hipStream_t stream);
int pioran_launch_block_gtab_batch(int64_t N, int32_t R, int32_t J, int64_t nb, const int32_t* rowmap, const double* t, const double* C /*[nb][J]*/,
                                   const double* D, const double* s2, double* gtab, int64_t draw_stride, hipStream_t stream);
int pioran_launch_block_grad(const ScanParams& p, const double* btab, const double* gtab, double* grad_a, double* grad_b, double* grad_nu,
                             double* grad_mu, double* grad_c /*nullptr: not wanted*/, double* grad_d, hipStream_t stream);
// celerite_tile.hip: the windowed form for LARGE batches, one draw per wavefront (round 5); same table as celerite_block.hip
int pioran_tile_supported_rows();
int pioran_tile_fits(int32_t R, int32_t J);
int64_t pioran_tile_pass_draws(int32_t R, int cus);
size_t pioran_tile_workspace_doubles(int64_t B, int64_t N);
int pioran_launch_scan_tile(const ScanParams& p, const double* btab, double* work, hipStream_t stream);
// ... its reverse mode: log L and d/d(a, b, mu, nu), one draw per wavefront (1 .. 63 rows; shared (c, d) and series)
int pioran_tile_grad_supported_rows();
// celerite_tp.hip: time-parallel evaluation for a handful of draws (state-space form, associative filter elements; round 5)
int pioran_tp_supported_rows();
int pioran_tp_padded_rows(int rows);
size_t pioran_tp_workspace_doubles(int64_t B, int64_t N, int RP, int nseg);
// scan: 0 boundary walk, unchecked (tools); 1 scan + verification launch + walk for the draws that fail (tools); 2 scan, checked by the filter, the caller repairs the draws whose
// discrepancy (pioran_tp_disc) exceeds the threshold; 4 boundary walk, checked and repaired the same way
int pioran_launch_tp(const ScanParams& p, int RP, int nseg, int64_t L, const int32_t* row_term, const int32_t* row_kind, double* work, hipStream_t stream, int scan = 0);
int pioran_tp_scan_rows(int RP);
// where pioran_launch_tp(.., scan != 0) leaves the scan's largest discrepancy per draw ([B], inside `work`), and the threshold it is held against
const double* pioran_tp_disc(const double* work, int64_t B, int64_t N, int RP, int nseg);
double pioran_tp_scan_tol(const ScanOptions* opt);
size_t pioran_tile_grad_workspace_doubles(int64_t B, int64_t N, int32_t R);
int pioran_launch_tile_grad(const ScanParams& p, const double* btab, const double* gtab, double* pairs, double* grad_a, double* grad_b, double* grad_nu,
                            double* grad_mu, double* grad_c, double* grad_d, hipStream_t stream);
int pioran_launch_block_table_reference(int64_t N, int32_t R, int32_t J, const int32_t* rowmap, const double* t, const double* c,
                                        const double* d, const double* y, const double* s2, double* btab, hipStream_t stream);
// celerite_predict.hip: posterior mean at new times (pred, src/celerite_solver.jl:363-483)
size_t pioran_predict_workspace_doubles(int64_t B, int64_t N, int32_t R);
int pioran_launch_predict(ScanParams p, double* work, const double* t, int64_t M, const double* tau, double* mean_out,
                          hipStream_t stream);
// celerite_fallback.hip
int pioran_launch_scan_fallback(const ScanParams& p, hipStream_t stream);
size_t pioran_fallback_scratch_doubles(int R);
// table.hip
size_t pioran_table_doubles(int64_t N, int32_t R);
int pioran_launch_table(int64_t N, int32_t R, const int32_t* rowmap, const double* t, const double* c,
                        const double* d, const double* y, const double* s2, double* tab, int64_t rec_stride,
                        hipStream_t stream);
int pioran_launch_table_batch(int64_t N, int32_t R, int32_t J, int64_t nb, const int32_t* rowmap, const double* t, const double* C,
                              const double* D, const double* y, const double* s2, double* tab, int64_t rec_stride,
                              int64_t tab_draw_stride, hipStream_t stream);
int pioran_launch_pd_table(int64_t N, int64_t B, int32_t J, int32_t npd_terms, const int32_t* pd_terms /*device*/,
                           const double* t, const double* C /*[B][J]*/, const double* D, double* tab, int64_t rec_stride,
                           int64_t rs_shared, hipStream_t stream);
int pioran_launch_shift_transform(int64_t N, int64_t B, const double* y, const double* s2, const double* shift,
                                  double* Y, double* S2, hipStream_t stream);
int pioran_launch_shift_grad(int64_t N, int64_t B, const double* y, const double* s2, const double* shift, const double* gY,
                             const double* gS, double* gshift, hipStream_t stream);
// approx.hip
#ifdef __cplusplus
#include <vector>
int pioran_approx_setup_host(int64_t J, int basis, double f_min, double f_max, double S_low, double S_high,
                             std::vector<double>& sp, std::vector<double>& LU, std::vector<int32_t>& piv,
                             std::vector<double>& c, std::vector<double>& d, std::vector<int32_t>& real_term);
#endif
int pioran_launch_approx(int64_t B, int model, int P, int J, int basis, int integrated, double f_min, double f_max,
                         const double* sp, const double* LU, const int32_t* piv, const double* theta, const double* norm,
                         int n_qpo, const double* qpo /*[B][n_qpo][3]: S0, f0, Q*/, double* A, double* Bc,
                         double* Cq /*[B][Jt]: per-draw c of the feature terms*/, double* Dq, hipStream_t stream);
// table.hip (diagnostics)
int pioran_launch_fma_stream(int blocks, int iters, double* scratch, double* flop, hipStream_t stream);
// dense.hip
// doubles behind a slab: 1024 (the four 16 x 16 inverses of the current diagonal block) + 4 x 4096 (dense_step_kernel's tile snapshots)
#define PIORAN_DENSE_WS (1024 + 4 * 4096 + 1024)   // inverse copies | tile snapshots | flags of the persistent-chain prototype
void pioran_dense_dims(int64_t N, int64_t* Mp, int64_t* ld);
int pioran_dense_nll_device(int64_t N, int32_t J, const double* a, const double* b, const double* c,
                            const double* d, const double* t, const double* y, const double* s2,
                            double* K /*ld*Mp + PIORAN_DENSE_WS*/, hipEvent_t* phase_ev /*nullptr or [3]*/, double* out, int32_t* info,
                            int sorted, hipStream_t stream, double mu = 0.0 /*subtracted from y*/, double nu = 1.0 /*scales s2*/,
                            const DenseOptions* dopt = nullptr);
int pioran_dense_predict_cov_device(int64_t N, int64_t M, int32_t J, const double* a, const double* b, const double* c,
                                    const double* d, const double* te, const double* s2e, double* K, int32_t* info,
                                    const double* y, double* mean, hipStream_t stream);
int pioran_dense_build_device(int64_t N, int32_t J, const double* a, const double* b, const double* c,
                              const double* d, const double* t, const double* y, const double* s2,
                              double* K, int sorted, hipStream_t stream);
