/*
 * pioran_hip.h — C ABI of libpioran_hip.so: the MI355X (gfx950) implementation of the ScalableGP
 * log-likelihood hot path of mlefkir/Pioran.jl (v1.2.0).
 *
 * This is the drop-in boundary.  The reference is pure Julia and has no FFI on this path; the seam
 * it would bind with `ccall` is the narrow waist
 *     logl(a, b, c, d, τ, y, σ2)                      src/celerite_solver.jl:312-334
 * reached from
 *     Distributions.logpdf(f::FiniteScalableGP, Y)    src/scalable_GP.jl:162-166
 *       -> log_likelihood(cov, τ, y, σ2; solver)      src/celerite_solver.jl:262-294
 * and, for the dense solver,
 *     log_likelihood_direct(cov, t, y, σ²)            src/direct_solver.jl:6-21.
 * INTEGRATION.md shows the Julia-side binding (julia/PioranHIP.jl in the package directory).
 *
 * Conventions
 *   - plain C, all floating point is IEEE fp64, all sizes int64_t, no torch / C++ types;
 *   - "host" pointers are ordinary process memory, "device" pointers are HBM of the ctx's GPU;
 *   - matrices over the batch are "J x B column-major" = Julia `Matrix{Float64}(J, B)` = C `[B][J]`:
 *     draw b's coefficients are contiguous at  A + b*J ;  series over the batch likewise `[B][N]`;
 *   - every entry point returns 0 (PIORAN_OK) or a negative error code; nothing unwinds across the
 *     ABI.  Numerical failures of single draws are reported per draw in `status` (below);
 *   - calls on one ctx must not overlap in time (one ctx per host thread / Julia task); different
 *     ctx objects are independent.  No global mutable state, no HIP call at library load; the
 *     environment is read once, when a context is created (diagnostic switches, pioran_ctx_set_option).
 *   - a data set keeps the state declared by pioran_dataset_prepare (the "(c, d) + table" the asynchronous
 *     *_dev entries use) SEPARATE from the scratch state of the host-pointer entries: pioran_celerite_logl_batch,
 *     _shift, pioran_logpdf_batch_theta, _predict, _logl_grad prepare their own tables and never change what
 *     pioran_dataset_prepare declared, so host-pointer and *_dev calls can be mixed freely on one data set.
 *
 * Accuracy (which kernel family evaluates a draw depends on rows, batch size and per-draw inputs; the calling thread's last family:
 *   pioran_celerite_config_name(-1)).  With ratio = nu min(sigma2) / sum(a) ~ 1 / cond(K) of a draw:
 *   - ratio >= 1e-8: every family is within 4.5e-9 (relative) of the exact value of its fp64 inputs, <= 3e-11 from ratio 1e-5 on
 *     (every stored chain of the reference sits there); families agree to 3e-10 over prior draws of the bench model;
 *   - ratio < 1e-8: the reference's own recurrence in fp64 is up to 7.8e-9 from the exact value; the step-by-step kernels follow it
 *     (6 .. 8e-9), the windowed kernels ("tile", "block": large and small batches from 5 rows on) reach 2 .. 3.2e-8 on single draws,
 *     the time-parallel family ("tp") 1e-10 — for a draw that passes its check: a draw of a long series whose boundary scan fails the check (0.3 .. 2 %
 *     of the prior draws of the SHO models, 6 .. 15 % of DRWCelerite's; option "tp_scan_tol") is evaluated again by the windowed kernel and has ITS
 *     accuracy, under the same name "tp".  Two families can therefore differ by up to 4e-8 on such a draw; no single stage of the
 *     windowed form removes that within 1 % of its time (profiles/r06_window_precision_by_stage.txt).  A caller that needs one
 *     family for every batch size pins it: options "no_tile", "no_block", "no_tp", "scan_config".
 *   - draws flagged in `status` (below): the families agree only to ~1e-6 (which D_n crosses zero within rounding differs).
 *
 * status[b]:  0 ok;  1 some D_n <= 0 (matrix not positive definite — the reference silently uses
 *             log(abs(D_n)) for n >= 2, src/celerite_solver.jl:140, and so does out[b]);
 *             2 non-finite result (the reference would throw DomainError from log(D_1 < 0), :126).
 */
#ifndef PIORAN_HIP_H
#define PIORAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PIORAN_OK 0
#define PIORAN_ERR_ARG (-1)         /* null pointer, non-positive size, inconsistent arguments */
#define PIORAN_ERR_HIP (-2)         /* a HIP runtime call failed; see pioran_last_hip_error */
#define PIORAN_ERR_ALLOC (-3)       /* device or host allocation failed */
#define PIORAN_ERR_UNSUPPORTED (-4) /* rank / size outside what the kernels handle */

typedef struct pioran_ctx pioran_ctx; /* one GPU + one stream + scratch; not thread-safe */
typedef struct pioran_ds pioran_ds;   /* a time series resident in HBM + its cached shared table */

const char* pioran_strerror(int code);
const char* pioran_last_hip_error(const pioran_ctx* ctx);
/* ABI version of this header (bumped on any signature change). */
int pioran_abi_version(void);   /* currently 7 */

/* ---- context ------------------------------------------------------------------------------- */
/* Creates a context on GPU `device` with its own non-blocking stream. */
int pioran_ctx_create(int device, pioran_ctx** out);
/* Same, but work is enqueued on the caller's hipStream_t (e.g. torch's current stream). */
int pioran_ctx_create_on_stream(int device, void* hip_stream, pioran_ctx** out);
int pioran_ctx_destroy(pioran_ctx* ctx);
int pioran_ctx_synchronize(pioran_ctx* ctx);
/* Releases every scratch buffer the context has grown (staging, gradient / prediction workspaces, the dense slab); they are
 * re-allocated on demand.  Data sets and their tables are not touched. */
int pioran_ctx_trim(pioran_ctx* ctx);
/* Diagnostic switches (none is needed in production; tests and tuning runs use them to pin a code path):
 *   "scan_config"     name of a throughput configuration of the scan, "wide" = latency layout or "block" = windowed kernel for
 *                     any batch; NULL/"" = automatic
 *   "no_wide" / "no_block" / "no_paired" / "no_mixed" / "force_fallback"   value "1" disables the latency layout / the windowed
 *                     small-batch kernel / the column-paired variants / the mixed shared+per-draw table / sends everything through
 *                     the any-rank kernel; NULL, "" or "0" = off.
 *   "win2" / "no_win2"  force / forbid the two-step form of the throughput layouts (default: on up to four rows per lane, i.e. R <= 63)
 *   "win3" / "no_win3"  force / forbid the three-step form (two or three rows per lane; default: on for R = 24 .. 31 in large batches)
 *   "wide2" / "no_wide2"  force / forbid the lean form of the latency layout (default: from 48 rows on; the only form for 96 .. 143 rows);
 *                     also selects the lean / round-1 kernels of the step-by-step gradient, prediction and simulation (64 .. 95 rows)
 *   "no_split"        a batch that is not a whole number of passes runs as one launch (default: the remainder goes to the windowed kernel)
 *   "workspace_limit_mb"  cap on the per-call workspaces (default 16384)
 *   "no_tile"         value "1": never the windowed form with one draw per wavefront (celerite_tile.hip); "scan_config" = "tile" forces it for
 *                     every launch it can take
 *   "no_tp"           value "1": never the time-parallel evaluation (celerite_tp.hip: segments of the series on different CUs, for a handful of
 *                     draws of a long series).  Automatic choice, round 6 (its boundary phase as a scan over the segments' elements): one or two draws with
 *                     3-4 / 5-16 / 24 / 32 / 40-48 / 56-64 state rows from 2048 / 1024 / 1536 / 2048 / 3072 / 4096 steps on, 3 .. 32 draws where a model of its
 *                     time promises 15 % against the serial chains; with the boundary walk of round 5: up to 8 draws with up to 4 / 8 / 12 / 16 state rows
 *                     from 1024 / 2048 / 4096 / 6144 steps on, up to 64 draws at up to 4 rows from 4096, up to 8 draws with up to 24 / 32 / 40 / 48 / 64 rows from
 *                     5120 / 6144 / 8192 / 8192 / 12288;
 *                     "scan_config" = "tp" forces it wherever it applies (shared (c, d), up to 64 state rows, up to 64 draws);
 *                     "tp_segments" its segment count (0 / NULL = automatic);
 *                     "tp_scan" -1 (default) automatic, 0 the boundary walk, 1 the scan wherever the rows allow (3 .. 64);
 *                     "tp_scan_tol" the scan's acceptance threshold: the filter measures how far the scan's boundary states are from the sequential ones (on the scale of the
 *                     innovation variance) and a draw whose largest distance exceeds it is evaluated again on the serial chain — 0.3 .. 2 % of the
 *                     prior draws of the SHO models, 6 .. 15 % of DRWCelerite's (0 / NULL = 1e-3; negative: every draw — tests); the boundary walk is checked the
 *                     same way; "tp_unchecked" = "1" (with "scan_config" = "tp"): the family's own arithmetic, no check, no repair (tests, tools); "tp_walk_repair" = "1": by the family's own
 *                     boundary walk instead; "tp_scan_lean" = "1", "tp_scan_waves" = "4": the forms of the combination kernel that are the default only at
 *                     49 .. 64 rows / up to 16 rows (tests, tools)
 *   "dense_old_chain" 0 one launch per block column (default), 1 the panel / update chain of rounds 1-3 (2 .. 8: timing experiments, only in
 *                     builds with -DPIORAN_EXPERIMENTS; PIORAN_ERR_ARG otherwise);
 *                     "dense_no_pairs", "dense_no_halves", "dense_pair_tiles", "dense_half_tile_limit", "dense_quad_threshold", "dense_batch_pair_threshold", "dense_streams": schedule knobs
 *   "block_emode", "gsum", "exp"   tuning / experiment selectors of single kernels (tools/ only; "exp" can make results meaningless)
 * Initial values come from the environment variables PIORAN_SCAN_CONFIG, PIORAN_NO_WIDE, PIORAN_NO_BLOCK, PIORAN_NO_TILE, PIORAN_NO_TP, PIORAN_NO_PAIRED,
 * PIORAN_NO_MIXED, PIORAN_FORCE_FALLBACK, PIORAN_WIN2, PIORAN_NO_WIN2, PIORAN_GSUM, PIORAN_WIDE2, read once when the context is created. */
int pioran_ctx_set_option(pioran_ctx* ctx, const char* key, const char* value);
/* hipEvent-based timing on the ctx stream: record slot i (0..11), elapsed between two slots. */
int pioran_ctx_event_record(pioran_ctx* ctx, int slot);
int pioran_ctx_event_elapsed_ms(pioran_ctx* ctx, int slot_start, int slot_stop, float* ms);

/* ---- data set ------------------------------------------------------------------------------ */
/* Uploads (t, y, sigma2), each of length N (host pointers), once.  y is the raw series: the mean
 * is subtracted per draw (mu), as logpdf does (src/scalable_GP.jl:164). */
int pioran_dataset_create(pioran_ctx* ctx, int64_t N, const double* t, const double* y,
                          const double* sigma2, pioran_ds** out);
int pioran_dataset_destroy(pioran_ds* ds);
/* Declares the (c_j, d_j) shared by every draw of the following *_dev batches and builds the
 * cos/sin/exp table for them (src/celerite_solver.jl:52-54, once instead of per draw).
 * c, d: host, length J.  real_term (host, length J, may be NULL): non-zero marks a term whose
 * b_j = 0 and d_j = 0 for EVERY draw (Exp / DRW terms, src/Exp.jl:29-33, src/psd.jl:270-273); its
 * identically-zero sin row is dropped, which leaves results bit-identical.  (The value 2 — a term whose (c, d)
 * differ per draw — is accepted for table-building purposes but the *_dev entries then return PIORAN_ERR_ARG: per-draw
 * terms go through the host-pointer entry with cd_shared = 0, which builds the mixed table itself.)
 * The declared state persists until the next pioran_dataset_prepare on this data set; no other entry changes it.
 * Device memory: the table takes (6 J + 8)(N + 1) * 8 bytes (10 MB at N = 1e4, J = 20); the first small batch (at most 512
 * draws; 768 with 32 .. 47 rows, 1024 with 36 .. 47 rows; 5 .. 63 rows — and up to 256 draws with 64 .. 95 rows; fewer than five rows: series of 16384 and more stamps, and the scalar call from N = 2048 on) builds a second table for the windowed small-batch kernel,
 * ~(30 J + 300) N bytes (37 MB there), and falls back to the other kernels if that does not fit (2 GB cap).
 * Entries that chunk their draws (per-draw (c, d) tables: 37 MB per draw; prediction: 2.6 GB, simulation: 1.4 GB, gradient: 6 GB per 256 draws at
 * N = 1e4, J = 20) take at most half of the free device memory and never more than the context option "workspace_limit_mb"
 * (default 16384); the buffers stay with the context until pioran_ctx_trim. */
int pioran_dataset_prepare(pioran_ds* ds, int64_t J, const double* c, const double* d,
                           const int32_t* real_term);

/* ---- celerite solver ------------------------------------------------------------------------ */
/* Scalar drop-in for logl(a,b,c,d,τ,y,σ2) (src/celerite_solver.jl:312-334): host vectors in, one
 * double out.  y has the mean already subtracted, as in the reference.  status may be NULL.
 * The series handle (and the tables of (c, d)) is kept while t is unchanged; small calls issue no copy commands for the coefficients
 * and the result (the kernels read / write the context's pinned host memory; late round 4: 39 -> 29 us at N = 32). */
int pioran_celerite_logl(pioran_ctx* ctx, int64_t N, int64_t J, const double* a, const double* b,
                         const double* c, const double* d, const double* t, const double* y,
                         const double* sigma2, double* out, int32_t* status);

/* B independent log-likelihoods on one data set; host pointers; blocks until `out` is filled.
 *   A, Bc      : [B][J]
 *   C, Dd      : [J] when cd_shared != 0, else [B][J] (per-draw decay/frequency: QPO, CARMA, ...)
 *   mu         : [B] or NULL (0):   y_n - mu_b                      (ConstMean, scalable_GP.jl:164)
 *   nu         : [B] or NULL (1):   sigma2_n * nu_b                 (the models' ν, README.md:44)
 *   Y, S2      : [B][N] or NULL: per-draw series / variances that REPLACE the data set's y / sigma2
 *                (models with a sampled shift, docs/src/ultranest.md:199-205); mu, nu still apply
 *   out        : [B]     status: [B] or NULL */
int pioran_celerite_logl_batch(pioran_ds* ds, int64_t B, int64_t J, const double* A,
                               const double* Bc, const double* C, const double* Dd, int cd_shared,
                               const double* mu, const double* nu, const double* Y,
                               const double* S2, double* out, int32_t* status);

/* Same computation with every array already in HBM (device pointers) and (c, d) declared by
 * pioran_dataset_prepare.  Asynchronous: enqueues on the ctx stream and returns; use
 * pioran_ctx_synchronize (or the stream) before reading out/status.  dmu, dnu, dY, dS2, dstatus
 * may be NULL. */
int pioran_celerite_logl_batch_dev(pioran_ds* ds, int64_t B, const double* dA, const double* dBc,
                                   const double* dmu, const double* dnu, const double* dY,
                                   const double* dS2, double* dout, int32_t* dstatus);
/* Per-draw (c, d) variant of the above: dC, dDd are [B][J] device arrays, no shared table. */
int pioran_celerite_logl_batch_dev_cd(pioran_ds* ds, int64_t B, int64_t J, const double* dA,
                                      const double* dBc, const double* dC, const double* dDd,
                                      const double* dmu, const double* dnu, const double* dY,
                                      const double* dS2, double* dout, int32_t* dstatus);
/* Per-draw data transform on the device — the reference's production models sample a shift c_b and fit the
 * log-flux (docs/src/ultranest.md:199-205, examples/ultranest/single_pl.jl:65-86, benchmark/benchmarks.jl:58-59):
 *     y_b      = log(y - shift_b)
 *     sigma2_b = sigma2 / (y - shift_b)^2            (then mu_b, nu_b exactly as above)
 * The data set holds the RAW flux y and sigma2 = yerr^2.  Only shift[B] crosses the boundary; the [B][N]
 * transformed series are built in HBM by a transform kernel.  y - shift_b <= 0 gives status 2 (NaN), where the
 * reference throws DomainError from log.  Host-pointer (blocking) and device-pointer (asynchronous) forms. */
int pioran_celerite_logl_batch_shift(pioran_ds* ds, int64_t B, int64_t J, const double* A, const double* Bc,
                                     const double* C, const double* Dd, int cd_shared, const double* mu,
                                     const double* nu, const double* shift, double* out, int32_t* status);
int pioran_celerite_logl_batch_shift_dev(pioran_ds* ds, int64_t B, const double* dA, const double* dBc,
                                         const double* dmu, const double* dnu, const double* dshift,
                                         double* dout, int32_t* dstatus);
/* theta -> log L in one call: `approx` (src/psd.jl:214-289) runs on the device in front of the scan, so only the
 * sampled parameters cross the boundary.  Continuum PSD models:
 *   model 0  SingleBendingPowerLaw(alpha1, f1, alpha2)                  P = 3 parameters per draw
 *   model 1  DoubleBendingPowerLaw(alpha1, f1, alpha2, f2, alpha3)      P = 5
 * basis 0 "SHO" (J = n_components terms), 1 "DRWCelerite" (J = 2 n_components terms);
 * theta [B][P], norm [B] (the `norm` argument of approx), is_integrated_power / S_low / S_high as in approx;
 * mu, nu, shift: [B] or NULL, as in pioran_celerite_logl_batch[_shift].
 * PSD features (`continuum + QPO(S0, f0, Q) + ...`, src/psd.jl:15-27, 228-241, 254-261): n_qpo (0..8) Lorentzians per draw,
 * qpo [B][n_qpo][3] = (S0, f0, Q) (NULL when n_qpo = 0); each becomes one more celerite term whose (c, d) differ per draw —
 * the scan then runs in the mixed mode (shared table for the continuum, per-draw rows for the features).
 * Host pointers, blocking.  Also returns, when A_out / Bc_out are non-NULL ([B][J + n_qpo] host), the coefficients
 * approx produced (feature terms last, as the reference concatenates them). */
int pioran_logpdf_batch_theta(pioran_ds* ds, int64_t B, int model, int64_t n_components, int basis,
                              int is_integrated_power, double f_min, double f_max, double S_low, double S_high,
                              const double* theta, const double* norm, const double* mu, const double* nu,
                              const double* shift, int64_t n_qpo, const double* qpo, double* out, int32_t* status,
                              double* A_out, double* Bc_out);
/* ---- posterior mean and simulation (the callers either side of the likelihood) -----------
 * predict (src/celerite_solver.jl:348-361 -> pred :363-483; mean(::PosteriorGP, tau) of src/scalable_GP.jl:64-72,90-91):
 *     mean_out[b][m] = mu_b + sum_n z_n k_b(|tau_m - t_n|),   z = K_b^-1 (y - mu_b),  K_b = kernel_b + diag(nu_b sigma2)
 * for B draws of (a, b); (c, d) [J] shared (cd_shared != 0) or [B][J] per draw (see the last paragraph).  tau: M evaluation
 * times, any order (the reference wants them sorted).
 * status (may be NULL) as in pioran_celerite_logl_batch.  Host pointers, blocking.
 * Up to 63 rows (from one row on since late round 4; six before) and shared (c, d) both calls run on the windowed factorisation (celerite_block.hip, round 3): z by a block
 * back-substitution, the two running vectors of `pred` in 128-step segments, the tau-only factors once per call — 6.1 ms per 256
 * draws x 1e4 times at N = 1e4, J = 20 (18.5 ms before); the simulation applies L window by window (5.0 ms per 256 draws, 8.7 before).
 * Workspace: the factor as the consumer reads it (6 KB per 16-step window and draw at three block columns; 8.5 KB for the simulation — up to
 * late round 4 both used the reverse mode's 41 KB layout) + 2 N R doubles per draw for the running vectors (2.6 GB for 256 draws at N = 1e4,
 * R = 40, 8.3 GB before; the chunk of draws shrinks to what is free).  Other shapes: the step-by-step
 * kernels — the factor stored by the latency kernels (the lean one from 64 rows on, round 4): posterior mean and simulation up to 143 rows (the reference
 * benchmark grid's j = 64 is 128); beyond: PIORAN_ERR_UNSUPPORTED before any workspace is taken.  pioran_celerite_config_name(-1)
 * tells which ran ("block (windowed prediction)" / "wide (step-by-step prediction)", likewise "... simulation").
 * cd_shared == 0 with several draws: where 2 J rows fit the windowed kernel all draws of a chunk go through every kernel in one launch, each
 * with its own tables (64 draws: 7.5 ms predict, 4.6 ms simulate at N = M = 1e4, J = 20; draw by draw 910 / 439 ms). */
int pioran_celerite_predict(pioran_ds* ds, int64_t B, int64_t J, const double* A, const double* Bc, const double* C,
                            const double* Dd, int cd_shared, const double* mu, const double* nu, int64_t M, const double* tau,
                            double* mean_out, int32_t* status);
/* log L and its gradient (what ForwardDiff obtains through the generic `logl`,
 * src/celerite_solver.jl:316, test/test_likelihood.jl:55-60) by reverse mode through the recurrence, for B draws:
 *   grad_a, grad_b [B][J] = dlogL/da_j, dlogL/db_j;
 *   grad_c, grad_d [B][J] (may be NULL) = dlogL/dc_j, dlogL/dd_j — what QPO features (src/psd.jl:15-27), CARMA kernels
 *       (src/CARMA.jl:98-143) and free Celerite terms need; per draw also when (c, d) are shared (sum over b for a common parameter);
 *   grad_nu, grad_mu [B] (may be NULL);
 *   grad_y, grad_sigma2 [B][N] (may be NULL) = dlogL/dy_n and dlogL/dsigma2_n of the series in the data set — what a
 *       model that transforms the data per draw (the sampled shift of docs/src/ultranest.md:199-205) chains through.
 * C, Dd: [J] when cd_shared != 0, else [B][J] (windowed reverse mode: every draw its own tables, all draws in one launch — 16 chains
 *   8 ms at N = 1e4, J = 20; shapes past 63 rows: each draw evaluated as its own one-draw batch).
 * Two reverse modes.  Up to 63 rows (from one row on since late round 4: 16.4 -> 4.4 ms at N = 1e4; six before) the WINDOWED reverse mode runs (celerite_block.hip, round 3): the windowed forward pass leaves T,
 * M', Sigma^-1 X' and Sigma^-1 of every 16-step window (38 KB per window at J = 20: 24 MB per draw at N = 1e4; chains are
 * processed 512 at a time — 12 GB —, fewer when the context option "workspace_limit_mb" or the free memory say so) and the adjoint kernel
 * walks the windows backwards with six GEMM stages each — value + gradient 5.4 ms for one chain, 5.8 ms for 256 (6.1 .. 6.3 ms with
 * grad_c / grad_d).  With 64 .. 143 rows (and as a cross-check: context option "no_block") the STEP-BY-STEP reverse mode runs
 * (celerite_wide.hip: forward pass with checkpoints, replayed segments, lean adjoint kernel since round 4): 40 .. 45 ms at 64 .. 95 rows,
 * 62 .. 157 ms at 96 .. 143 (the reference benchmark grid's j = 64 is 128 rows) at N = 1e4.  More than 143 rows: PIORAN_ERR_UNSUPPORTED.
 * Many chains (round 5): with more than 512 chains, 17 .. 63 rows (round 6: four block columns), shared (c, d), grad_y = grad_sigma2 = NULL (value and
 * d/d(a, b, mu, nu) — what the approx-based models' samplers ask for: (c, d) shared by the chains are fixed by the spectral grid — and, round 6, d/d(c, d) of
 * the shared (c, d): grad_c and grad_d both or neither) the reverse mode
 * with ONE DRAW PER WAVEFRONT runs (celerite_tile.hip): its forward pass keeps only the lower tiles of T per window (7.5 MB per chain at N = 1e4,
 * chunks of 2048 chains under the default workspace limit) and the reverse kernel recomputes the rest — 4096 chains of SHO-20 45.7 ms (56.0 with d/d(c, d);
 * 88 / 100 ms on the small-batch kernels), of DRWCelerite-20 (60 rows) 125 / 154 ms (166 / 175).  Context option "no_tile" keeps every chain count on the small-batch kernels, "scan_config" = "tile" forces the new
 * family from one chain on; config name "tile (windowed gradient, one draw per wavefront)".
 * Memory (step-by-step mode): the forward pass keeps the R x R state only at checkpoints (every ~2 sqrt(N) steps) and the reverse pass replays one
 * segment at a time: ~15 MB of workspace per draw at N = 1e4, J = 20 (two replayed segments + checkpoints 11 MB, stored m / D 4 MB: pioran_grad_workspace_doubles;
 * 55 MB at 80 rows, 105 MB at 128); draws are processed in chunks sized to the free memory.
 * The workspace stays in the context for the next call; pioran_ctx_trim releases it. */
int pioran_celerite_logl_grad(pioran_ds* ds, int64_t B, int64_t J, const double* A, const double* Bc, const double* C,
                              const double* Dd, int cd_shared, const double* mu, const double* nu, double* out,
                              int32_t* status, double* grad_a, double* grad_b, double* grad_c, double* grad_d,
                              double* grad_nu, double* grad_mu, double* grad_y, double* grad_sigma2);
/* The same for the shifted log-flux models (docs/src/turing.md:205-230: c ~ LogUniform(...), y = log.(y .- c),
 * sigma2 = nu sigma.^2 ./ (y .- c).^2): the data set holds the raw flux and yerr^2, the transform runs on the device as in
 * pioran_celerite_logl_batch_shift, and grad_shift [B] = dlogL/dc_b comes out of the series gradients by the chain rule. */
int pioran_celerite_logl_grad_shift(pioran_ds* ds, int64_t B, int64_t J, const double* A, const double* Bc, const double* C,
                                    const double* Dd, int cd_shared, const double* mu, const double* nu, const double* shift,
                                    double* out, int32_t* status, double* grad_a, double* grad_b, double* grad_c,
                                    double* grad_d, double* grad_nu, double* grad_mu, double* grad_shift);
/* simulate (src/celerite_solver.jl:497-513 -> sim :515-549; rand(f(t, sigma2)) of src/scalable_GP.jl:137-146):
 * realisations y_b = L_b D_b^(1/2) q_b of the GP with kernel (a_b, b_b, c, d) + diag(sigma2) at the times t, from
 * caller-supplied standard-normal draws q [B][N] (the reference draws them with its rng, :528).  y_out [B][N].
 * C, Dd: [J] (cd_shared != 0) or [B][J]. */
int pioran_celerite_simulate(pioran_ctx* ctx, int64_t N, int64_t B, int64_t J, const double* A, const double* Bc,
                             const double* C, const double* Dd, int cd_shared, const double* t, const double* sigma2,
                             const double* q, double* y_out);
/* Name of the kernel configuration a large batch with R active rows (all terms with both rows when R is even) runs on;
 * R == 0: the configuration the calling thread's last throughput-layout launch actually ran on; R < 0: the kernel FAMILY of the
 * calling thread's last launch — "tile" (windowed form, one draw per wavefront: large batches from 49 rows on and batch sizes between the
 * passes of the throughput layouts), "block" (windowed kernel), "block+pd" (with per-draw rows), "block (per-draw tables)", "wide"
 * (latency layout), "scan" (throughput layouts), "fallback", "block (windowed gradient[, per-draw tables])",
 * "wide (step-by-step gradient)" (diagnostics). */
const char* pioran_celerite_config_name(int64_t R);
/* Diagnostics (no GPU needed): the automatic choice between the windowed form with one draw per wavefront (1, "tile") and the other families (0)
 * for a shared-table batch of B draws with R active rows; `pass` = draws per pass of the step-by-step layout of these rows (0 = unknown:
 * returns -1 where the choice depends on it), no_split != 0: option "no_split".  The thresholds come from one sweep (profiles/r05_tile_batch_sweep.txt);
 * tests/test_host.py holds this function against that file, tools/retune_thresholds.py prints both side by side. */
int pioran_tile_choice(int32_t R, int64_t B, int64_t pass, int no_split);
/* Diagnostics: the FP64 FMA rate (TFLOP/s) the device sustains right now with `waves_per_simd` (1 .. 8) wavefronts on every SIMD — about
 * `ms` milliseconds of a pure stream of independent v_fma_f64, event-timed on the context's stream.  The measured ceiling of any FP64
 * vector kernel on this box at that occupancy (the 78.6 TFLOP/s vendor figure assumes one FMA per SIMD every 4 cycles at 2.4 GHz).
 * Side effects: blocks until the stream has drained; may re-allocate the context's scratch buffer (no call may be in flight on another
 * thread of the same context); uses the context's own event slots, never the caller's 0 .. 11. */
int pioran_ctx_fp64_probe(pioran_ctx* ctx, int waves_per_simd, double ms, double* tflops);

/* ---- in-process farm over several GPUs ---------------------------------------------------------------------------
 * For hosts without a process-per-GPU launcher (a single Julia process driving the 8 GPUs of a node): one context and
 * one resident copy of the data set per listed device, one host thread per device per call.  Draws are independent, so
 * the batch is cut into contiguous shards (the first B % ngpu devices get one more draw) and every device writes its
 * slice of out/status directly — there is no collective.  `devices` may list a device more than once.
 * Arguments of pioran_farm_logl_batch are those of pioran_celerite_logl_batch[_shift] (shift may be NULL);
 * pioran_farm_logl_batch_series takes per-draw series Y, S2 [B][N] instead (the models with a sampled mean FUNCTION, e.g. the
 * sinusoid of examples/ultranest/single_pl_periodicity.jl:115: CustomMean makes y - mean(t) a per-draw series): every device
 * receives the rows of its shard. */
typedef struct pioran_farm pioran_farm;
int pioran_farm_create(int ngpu, const int* devices, int64_t N, const double* t, const double* y,
                       const double* sigma2, pioran_farm** out);
int pioran_farm_destroy(pioran_farm* farm);
int pioran_farm_size(const pioran_farm* farm);
int pioran_farm_logl_batch(pioran_farm* farm, int64_t B, int64_t J, const double* A, const double* Bc,
                           const double* C, const double* Dd, int cd_shared, const double* mu, const double* nu,
                           const double* shift, double* out, int32_t* status);
int pioran_farm_logl_batch_series(pioran_farm* farm, int64_t B, int64_t J, const double* A, const double* Bc,
                                  const double* C, const double* Dd, int cd_shared, const double* mu, const double* nu,
                                  const double* Y, const double* S2, double* out, int32_t* status);

/* ---- dense solver ---------------------------------------------------------------------------- */
/* log_likelihood_direct (src/direct_solver.jl:6-21): builds K_ik = sum_j k_j(|t_i - t_k|) +
 * diag(sigma2) in HBM, Cholesky-factorises it and returns the POSITIVE negative-log-likelihood,
 * like the reference (callers negate, test/test_likelihood.jl:54).  info (may be NULL): 0, or the
 * 1-based index of the first non-positive pivot (LAPACK convention; out is NaN then — the reference
 * throws PosDefException). */
int pioran_dense_nll(pioran_ctx* ctx, int64_t N, int64_t J, const double* a, const double* b,
                     const double* c, const double* d, const double* t, const double* y,
                     const double* sigma2, double* out, int32_t* info);
/* B dense evaluations on one data set (the reference has no batch dimension: one log_likelihood_direct per call).  A, Bc
 * [B][J]; C, Dd [J] (cd_shared != 0) or [B][J]; mu, nu [B] or NULL as in pioran_celerite_logl_batch: y - mu_b, nu_b sigma2.
 * out[b] = +NLL of draw b, info[b] (may be NULL) as above.  The factorisations are independent: up to 32 (fewer when memory is short:
 * one slab of about N^2 doubles each) go through every kernel of the chain in ONE launch (the matrix index is a grid dimension), which is
 * what fills the matrix cores at N of a few thousand — one factorisation alone is a chain of N/64 latency-bound steps (N = 4096, J = 40:
 * 1.65 ms alone, 0.49 ms each in a batch). */
int pioran_dense_nll_batch(pioran_ctx* ctx, int64_t N, int64_t J, int64_t B, const double* A, const double* Bc,
                           const double* C, const double* Dd, int cd_shared, const double* t, const double* y,
                           const double* sigma2, const double* mu, const double* nu, double* out, int32_t* info);
/* Same call with event timing of its phases on the ctx stream: phase_ms[0] covariance build, [1] factorisation (all panel
 * and trailing-update launches), [2] finish kernel — what bench.py reports as the dense path's MFMA utilisation. */
int pioran_dense_nll_timed(pioran_ctx* ctx, int64_t N, int64_t J, const double* a, const double* b,
                           const double* c, const double* d, const double* t, const double* y,
                           const double* sigma2, double* out, int32_t* info, float* phase_ms);
/* predict_cov (src/direct_solver.jl:28-69; cov / std / rand of a PosteriorGP, src/scalable_GP.jl:73-104):
 *     cov_out = K(tau,tau) - K(tau,t) (K(t,t) + diag(sigma2))^-1 K(t,tau),   M x M, symmetric,
 * computed as the Schur complement the blocked MFMA Cholesky of the augmented (N+M) x (N+M) matrix leaves when it stops
 * after the data columns.  info as in pioran_dense_nll (cov_out is NaN when K(t,t) + diag(sigma2) is not PD). */
int pioran_dense_predict_cov(pioran_ctx* ctx, int64_t N, int64_t J, const double* a, const double* b,
                             const double* c, const double* d, const double* t, const double* sigma2, int64_t M,
                             const double* tau, double* cov_out, int32_t* info);
/* predict_direct(cov, tau, t, y, sigma2[, with_covariance]) (src/direct_solver.jl:75-119): the dense posterior mean
 * K(tau,t) (K(t,t) + diag(sigma2))^-1 y from the same partial factorisation (y rides as one more row, as in
 * pioran_dense_nll), and optionally (cov_out != NULL) the covariance of pioran_dense_predict_cov. */
int pioran_dense_predict(pioran_ctx* ctx, int64_t N, int64_t J, const double* a, const double* b,
                         const double* c, const double* d, const double* t, const double* y, const double* sigma2,
                         int64_t M, const double* tau, double* mean_out, double* cov_out, int32_t* info);
/* Covariance build alone (K as N x N column-major host array) — kappa of src/acvf.jl:138-140. */
int pioran_dense_covariance(pioran_ctx* ctx, int64_t N, int64_t J, const double* a, const double* b,
                            const double* c, const double* d, const double* t, const double* sigma2,
                            double* K_out);

#ifdef __cplusplus
}
#endif
#endif /* PIORAN_HIP_H */
