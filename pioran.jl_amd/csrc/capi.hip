// C ABI of libpioran_hip.so — see include/pioran_hip.h for the contract and the reference lines
// each entry point replaces.
#include "../../include/pioran_hip.h"
#include "common.h"

// (celerite_predict.hip; declared here: the windowed prediction was added after the PMC profiles of common.h's kernels were taken)
size_t pioran_predict_q_workspace_doubles(int64_t B, int64_t N, int32_t R);
int pioran_dense_nll_device_batch(int64_t nbatch, int64_t N, int32_t J, const double* a, const double* b, const double* c, const double* d,
                                  int64_t cd_stride, const double* t, const double* y, const double* s2, double* K, int64_t slab,
                                  const double* mu, const double* nu, double* out, int32_t* info, int sorted, hipStream_t stream,
                                  const DenseOptions* dopt);   // dense.hip
int pioran_launch_block_sim(const ScanParams& p, const double* btab, double* xi, hipStream_t stream);   // celerite_block.hip
int pioran_launch_block_solve(const ScanParams& p, const double* btab, const double* gtab, double* gy, hipStream_t stream);   // celerite_block.hip
size_t pioran_predict_tau_workspace_doubles(int64_t M, int32_t R, int64_t ntab);
int pioran_launch_predict_from_gy(ScanParams p, double* work, double* tau_work, const double* t, int64_t M, const double* tau, double* mean_out,
                                  hipStream_t stream, int cd_per_draw, int tau_sorted);


#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <new>
#include <string>
#include <thread>
#include <vector>

struct pioran_ctx {
    int device = 0;
    int ncu = 0;                    // compute units of the device (read once, at creation: the dispatch functions must not call into the runtime per launch)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev[16] = {};
    std::string last_err;
    ScanOptions opt{};   // diagnostic switches: environment at creation, then pioran_ctx_set_option
    // pinned staging of the host-pointer entries: pageable hipMemcpyAsync is staged by the runtime with a hidden
    // synchronisation per call (~0.4 ms each); copies out of / into this buffer are true asynchronous DMA.  A bump
    // allocator, reset at every stream synchronisation; results land here and are handed to the caller after the sync.
    char* pin = nullptr;
    size_t pin_cap = 0, pin_off = 0;
    struct Pending { void* host; const void* pinned; size_t bytes; };
    std::vector<Pending> pending;
    // batched dense solver: independent factorisations on their own streams, one slab each (lazy)
    static constexpr int kDenseStreams = 16;
    hipStream_t dstream[kDenseStreams] = {};
    hipEvent_t dev_[kDenseStreams + 1] = {};
    // second stream + events of the gradient's reverse pass (replay of one segment overlaps the adjoint of the next); lazy
    hipStream_t aux = nullptr;
    hipEvent_t gev[5] = {};
    // growable device staging for the host-pointer entry points
    struct Buf {
        void* p = nullptr;
        size_t cap = 0;
    };
    Buf bA, bB, bC, bD, bmu, bnu, bY, bS2, bout, bst, bscratch, bK, bwork, bshift, bgtab, bq, bpair, btp, btprow;
    // scalar entry point: the last series' time stamps stay resident (samplers call logl with the same t)
    pioran_ds* scalar_ds = nullptr;
    std::vector<double> scalar_t;
};

// Prepared shared-(c, d) state: table, row map, device copies of (c, d).  A data set holds TWO of them: `user` is what
// pioran_dataset_prepare declared for the asynchronous *_dev entries and is changed by nothing else; `host` is the
// scratch state of the host-pointer entries (logl_batch, mixed mode, theta, predict, grad), which prepare on their own.
struct PrepState {
    int32_t J = 0, R = 0;
    std::vector<double> c_host, d_host;
    std::vector<int32_t> real_host;
    double* tab = nullptr;
    size_t tab_cap = 0;
    int32_t* rowmap = nullptr;
    size_t rowmap_cap = 0;
    double *dc = nullptr, *dd = nullptr;
    size_t dcd_cap = 0;
    bool prepared = false;
    // mixed mode (term kind 2): indices of the per-draw terms, on the device
    int32_t npd_terms = 0;
    int32_t* dpd_terms = nullptr;
    // row layout class for the scan's configuration choice (ScanParams::standard_rows / n_complex)
    int32_t row_layout = 0, n_complex = 0;
    // table of the windowed kernel (celerite_block.hip), built on first use for the prepared (c, d)
    double* btab = nullptr;
    size_t btab_cap = 0;
    bool btab_ready = false;
};

struct pioran_ds {
    pioran_ctx* ctx = nullptr;
    int64_t N = 0;
    double *t = nullptr, *y = nullptr, *s2 = nullptr;  // device
    PrepState user, host;
};

namespace {

#define HIPCHK(ctx, expr)                                                                   \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            (ctx)->last_err = std::string(#expr) + ": " + hipGetErrorString(e_);            \
            return PIORAN_ERR_HIP;                                                          \
        }                                                                                   \
    } while (0)

// How much NEW device memory a call may take for its chunked workspaces (per-draw tables, factor stores, gradient / prediction
// workspaces): half of what is free, but never more than the context's absolute budget (option "workspace_limit_mb", default 16 GiB —
// the 256-draw chunks of every entry fit: per-draw windowed tables 9.5 GB, prediction 2.6 GB, gradient 6 GB at N = 1e4, J = 20), so
// that a co-resident allocator (torch's caching allocator, a second context) is not starved on a 288 GB device.  The buffers stay in
// the context until pioran_ctx_trim.
static size_t ws_allow(const pioran_ctx* ctx, size_t free_b)
{
    const size_t cap = (size_t)(ctx->opt.workspace_limit_mb > 0 ? ctx->opt.workspace_limit_mb : 16384) << 20;
    return free_b / 2 < cap ? free_b / 2 : cap;
}

// May `b` hold `bytes` under the workspace budget?  The runtime is asked for the free memory only when the buffer would have to GROW: the
// asynchronous *_dev entries run this on every launch, and a launch that fits what is already allocated must not block on the host.
static bool ws_fits(pioran_ctx* ctx, const pioran_ctx::Buf& b, size_t bytes)
{
    if (bytes <= b.cap) return true;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return false; }
    return bytes <= ws_allow(ctx, free_b) + b.cap;
}

int ensure(pioran_ctx* ctx, pioran_ctx::Buf& b, size_t bytes)
{
    if (bytes <= b.cap) return PIORAN_OK;
    if (b.p) HIPCHK(ctx, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + (bytes < (size_t(1) << 28) ? bytes / 4 : 0) + 256;   // growth slack for small buffers only
    if (hipMalloc(&b.p, want) != hipSuccess) {
        (void)hipGetLastError();   // the failure is reported through the return code: do not leave it sticky for the callers'
                                   // shrink-and-retry loops (a later launch wrapper would read it as its own error)
        b.p = nullptr;
        ctx->last_err = "hipMalloc failed";
        return PIORAN_ERR_ALLOC;
    }
    b.cap = want;
    return PIORAN_OK;
}

// stream synchronisation + delivery of the results staged in pinned memory + reset of the staging allocator
int ctx_sync(pioran_ctx* ctx)
{
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        // nothing is delivered after a failed synchronisation, and nothing may be delivered LATER either: the caller's
        // buffers are only valid for the duration of the call that queued them
        ctx->pending.clear();
        ctx->pin_off = 0;
        ctx->last_err = std::string("hipStreamSynchronize: ") + hipGetErrorString(e);
        return PIORAN_ERR_HIP;
    }
    for (const auto& q : ctx->pending) std::memcpy(q.host, q.pinned, q.bytes);
    ctx->pending.clear();
    ctx->pin_off = 0;
    return PIORAN_OK;
}

// Every host-pointer entry opens with one of these: result deliveries queued by download() refer to memory the caller
// owns only until the entry returns, so an entry that leaves early (any error between download() and the final SYNC)
// must not leave them behind for the next successful ctx_sync of some later call.  On the normal path the queue is
// already empty when the guard runs.
struct PendingGuard {
    pioran_ctx* ctx;
    explicit PendingGuard(pioran_ctx* c) : ctx(c) {}
    ~PendingGuard()
    {
        if (ctx && !ctx->pending.empty()) {
            (void)hipStreamSynchronize(ctx->stream);   // the staged copies may still be in flight: the pinned slots are reused
            ctx->pending.clear();
            ctx->pin_off = 0;
        }
    }
};
#define SYNC(ctx)                      \
    do {                               \
        int rc_sync_ = ctx_sync(ctx);  \
        if (rc_sync_) return rc_sync_; \
    } while (0)

constexpr size_t kPinMaxRequest = size_t(32) << 20;   // larger transfers go straight from / to the caller's memory

// bytes of pinned staging, 256-byte aligned; nullptr when the request is too large for staging (or pinning fails)
void* pin_reserve(pioran_ctx* ctx, size_t bytes)
{
    if (bytes > kPinMaxRequest) return nullptr;
    const size_t need = (bytes + 255) & ~size_t(255);
    if (ctx->pin_off + need > ctx->pin_cap) {
        if (ctx_sync(ctx) != PIORAN_OK) return nullptr;           // nothing in flight uses the old buffer any more
        if (need > ctx->pin_cap) {
            if (ctx->pin) (void)hipHostFree(ctx->pin);
            ctx->pin = nullptr;
            ctx->pin_cap = 0;
            size_t want = need * 4 < (size_t(8) << 20) ? (size_t(8) << 20) : need * 4;
            if (hipHostMalloc((void**)&ctx->pin, want, hipHostMallocDefault) != hipSuccess) { ctx->pin = nullptr; return nullptr; }
            ctx->pin_cap = want;
        }
    }
    void* p = ctx->pin + ctx->pin_off;
    ctx->pin_off += need;
    return p;
}

int upload(pioran_ctx* ctx, pioran_ctx::Buf& b, const void* host, size_t bytes)
{
    int rc = ensure(ctx, b, bytes);
    if (rc) return rc;
    const void* src = host;
    if (void* st = pin_reserve(ctx, bytes)) {
        std::memcpy(st, host, bytes);
        src = st;
    }
    HIPCHK(ctx, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return PIORAN_OK;
}

// device -> caller: through pinned staging (delivered by the next ctx_sync) where it fits, else directly
int download(pioran_ctx* ctx, void* host, const void* dev, size_t bytes)
{
    if (void* st = pin_reserve(ctx, bytes)) {
        HIPCHK(ctx, hipMemcpyAsync(st, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        ctx->pending.push_back({host, st, bytes});
    } else {
        HIPCHK(ctx, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    return PIORAN_OK;
}

// rows kept for a term list.  kind[j]: 0 = shared (c, d): cos + sin row from the shared table;
//   1 = "real" term (b = d = 0 for every draw): cos row only; 2 = per-draw (c, d): cos + sin row from the per-draw table.
// Encoding: term (bits 0-19) | per-draw row index (20-28) | per-draw flag (29) | sin row (30).
std::vector<int32_t> build_rowmap(int64_t J, const int32_t* kind)
{
    // shared rows first, the rows of the per-draw terms LAST (in term order, cos row then sin row): every kernel reads the row
    // map, none assumes an order, and the windowed kernel relies on the per-draw rows being the last ones (celerite_block.hip)
    std::vector<int32_t> rm;
    rm.reserve(2 * J);
    for (int64_t j = 0; j < J; ++j) {
        const int32_t k = kind ? kind[j] : 0;
        if (k == 2) continue;
        rm.push_back((int32_t)j);
        if (k != 1) rm.push_back((int32_t)j | (1 << 30));
    }
    int32_t npd = 0;
    for (int64_t j = 0; j < J; ++j) {
        if (!kind || kind[j] != 2) continue;
        rm.push_back((int32_t)j | ((2 * npd) << 20) | (1 << 29));
        rm.push_back((int32_t)j | ((2 * npd + 1) << 20) | (1 << 29) | (1 << 30));
        ++npd;
    }
    return rm;
}

int set_rowmap(pioran_ds* ds, PrepState& s, const std::vector<int32_t>& rm)
{
    pioran_ctx* ctx = ds->ctx;
    if (rm.size() > s.rowmap_cap) {
        if (s.rowmap) HIPCHK(ctx, hipFree(s.rowmap));
        s.rowmap = nullptr;
        if (hipMalloc((void**)&s.rowmap, rm.size() * sizeof(int32_t)) != hipSuccess) return PIORAN_ERR_ALLOC;
        s.rowmap_cap = rm.size();
    }
    HIPCHK(ctx, hipMemcpyAsync(s.rowmap, rm.data(), rm.size() * sizeof(int32_t), hipMemcpyHostToDevice,
                               ctx->stream));
    // the host vector is about to go out of scope in the callers: finish the copy first
    SYNC(ctx);
    s.R = (int32_t)rm.size();
    return PIORAN_OK;
}

thread_local const char* g_last_kernel = "none";   // which kernel family the calling thread's last launch ran on (diagnostics)

// Register-resident scan: small shared-table batches take the latency layout (celerite_wide.hip, one draw per
// workgroup), everything else the throughput layouts (celerite_scan.hip).  PIORAN_SCAN_CONFIG=wide forces the former
// for any batch size, any other value names a throughput configuration; PIORAN_NO_WIDE=1 disables the former.
int scan_dispatch(const ScanParams& p, hipStream_t stream)
{
    static const ScanOptions kDefaults{};
    const ScanOptions& o = p.opt ? *p.opt : kDefaults;
    const char* cfg = o.scan_config[0] ? o.scan_config : nullptr;
    const bool force_wide = cfg && !std::strcmp(cfg, "wide");
    // (below 16 rows the per-step exchange of the latency layout costs more than the whole step of a throughput layout)
    const bool auto_wide = !cfg && p.B <= pioran_wide_max_batch() && p.R >= 16 && !o.no_wide;
    // 80..95 rows: the throughput layouts do not hold S in registers any more, the latency layout still does
    // (exactly 80 rows with a shared table: the throughput layout holds them with y as a vector — large batches go there: 54 k
    //  instead of 43 k evaluations per second at B = 1024 .. 4096, N = 1e4; at 512 draws the latency layout is still ahead,
    //  39 k vs 27 k: tools/sweep_r80.py)
    const bool y80 = p.R == pioran_scan_supported_rows_shared() && p.tab && p.npd_rows == 0 && !o.no_win2 && (p.B > 768 || o.no_wide);
    const bool only_wide = p.R > pioran_scan_supported_rows() && !o.no_wide && !y80;
    if (p.tab && p.R <= pioran_wide_supported_rows() && (force_wide || auto_wide || only_wide)) {
        g_last_kernel = "wide";
        return pioran_launch_scan_wide(p, stream);
    }
    if (p.R > pioran_scan_supported_rows() && !y80) return PIORAN_ERR_UNSUPPORTED;
    g_last_kernel = "scan";
    return pioran_launch_scan(p, stream);
}

// the windowed kernel's own table of a prepared (c, d): built on first use (PIORAN_ERR_UNSUPPORTED: too long a series / no memory)
int ensure_btab(pioran_ds* ds, PrepState& s)
{
    pioran_ctx* ctx = ds->ctx;
    if (s.btab_ready) return PIORAN_OK;
    // 60 KB per 16 time stamps at J = 20: very long series (or a device short of memory) stay on the other kernels
    const size_t need = pioran_block_table_doubles(ds->N, s.R, s.J);
    if (need * sizeof(double) > (size_t(2) << 30)) return PIORAN_ERR_UNSUPPORTED;
    if (need > s.btab_cap) {
        if (s.btab) HIPCHK(ctx, hipFree(s.btab));
        s.btab = nullptr;
        s.btab_cap = 0;
        if (hipMalloc((void**)&s.btab, need * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError();
            return PIORAN_ERR_UNSUPPORTED;
        }
        s.btab_cap = need;
    }
    int rc = ctx->opt.btab_reference
                 ? pioran_launch_block_table_reference(ds->N, s.R, s.J, s.rowmap, ds->t, s.dc, s.dd, ds->y, ds->s2, s.btab, ctx->stream)
                 : pioran_launch_block_table(ds->N, s.R, s.J, s.rowmap, ds->t, s.dc, s.dd, ds->y, ds->s2, s.btab, ctx->stream);
    if (rc) return rc;
    s.btab_ready = true;
    return PIORAN_OK;
}

// Small shared-table batches without per-draw rows: the windowed kernel (celerite_block.hip), which needs its own table.
// Returns PIORAN_ERR_UNSUPPORTED when the launch is not one of those (the caller goes on to the other kernels).
int block_dispatch(pioran_ds* ds, const ScanParams& p)
{
    pioran_ctx* ctx = ds->ctx;
    const ScanOptions& o = ctx->opt;
    const char* cfg = o.scan_config[0] ? o.scan_config : nullptr;
    const bool force = cfg && !std::strcmp(cfg, "block");
    // measured on N = 1e4 (tools/sweep_block.py, tools/sweep_midbatch.py, tools/sweep_block_emode.py): faster than both other kernels
    // up to 512 draws from 6 rows on.  Late round 3: with the pair table read from global memory two workgroups share a CU at three
    // block columns, which moves the crossover up — R = 32 .. 35: 768 draws 4.9 vs 5.4 ms; R = 36 .. 47: 1024 draws 5.3 .. 5.8 vs
    // 5.7 .. 7.5 ms on the throughput shapes.  With four block columns (48 rows and more) the table stays in LDS and 512 draws is the
    // limit (DRWCelerite-20 at 768 draws: 7.8 vs 7.1 ms on the throughput shape, which got faster this round).
    // Round 4: five and six block columns (64 .. 95 rows; value only, one workgroup per CU: up to 256 draws).
    // Late round 4 (tools/scalar_small_j.py, profiles/r04_few_rows.txt): five rows 1.84 -> 1.48 ms at N = 1e4; four and fewer rows stay on the
    // throughput layout (1.42 against 1.47 ms) except for long series — its 20-double step records outgrow the L2 (N = 65536: 13.4 against
    // 9.5 ms up to 256 draws, 11.2 at 512) — and for the scalar call, whose series arrive as per-draw (y, sigma2): N = 8192 1.36 -> 1.25 ms.
    const bool few_rows = p.R < 5 && ((p.N >= 16384 && p.B <= 512) || (p.Y && p.B == 1 && p.N >= 2048));
    const bool automatic = !cfg && !o.no_block &&
                           (p.R < 5 ? few_rows
                            : p.R > pioran_block_supported_rows()
                                ? p.B <= 256
                                : (p.B <= 512 || (p.B <= 768 && p.R >= 32 && p.R <= 47) || (p.B <= 1024 && p.R >= 36 && p.R <= 47)));
    if (!(force || automatic) || !p.tab || p.npd_rows != 0 || !pioran_block_fits_value(p.R, p.J)) return PIORAN_ERR_UNSUPPORTED;
    PrepState* s = p.tab == ds->user.tab ? &ds->user : (p.tab == ds->host.tab ? &ds->host : nullptr);
    if (!s || !s->prepared || s->npd_terms != 0 || p.rec_stride != 3 * (int64_t)(s->R + 2) + 2) return PIORAN_ERR_UNSUPPORTED;
    int rc = ensure_btab(ds, *s);
    if (rc) return rc;
    g_last_kernel = "block";
    return pioran_launch_scan_block(p, s->btab, ctx->stream);
}

// Batches that are not a whole number of passes (round 4).  A throughput launch is a sequence of PASSES — every SIMD of the chip holding as
// many wavefronts as the kernel's registers allow (SHO-20: 2 x 1024 wavefronts x 2 draws = 4096 draws) — and a wavefront walks the whole
// series whatever its pass carries: 4200 draws cost two passes' time less what the scheduler backfills (16.9 against 11.9 ms for 4096).
// Small remainders are what the windowed kernel (celerite_block.hip) is fast at, and its workgroups fit BESIDE a resident scan wavefront
// (205 + 250 registers per SIMD lane pair): so the remainder goes to that kernel on the context's second stream, launched first, while the
// whole passes run on the main stream.  The two launches write disjoint slices of out / status; the main stream waits for the second
// one's event, so the call is stream-ordered like any other.
static ScanParams slice_draws(const ScanParams& p, int64_t off, int64_t n)
{
    ScanParams q = p;
    q.B = n;
    q.A = p.A + off * p.J; q.Bc = p.Bc + off * p.J;
    if (p.mu) q.mu = p.mu + off;
    if (p.nu) q.nu = p.nu + off;
    if (p.Y) q.Y = p.Y + off * p.N;
    if (p.S2) q.S2 = p.S2 + off * p.N;
    q.out = p.out + off;
    if (p.status) q.status = p.status + off;
    return q;
}

// A handful of draws of a long series: the time-parallel evaluation (celerite_tp.hip, round 5) — segments of the series on different CUs instead
// of one serial chain per draw.  scan_config = "tp" forces it wherever it applies (shared (c, d), at most 64 state rows, at most 64 draws).
int tp_dispatch(pioran_ds* ds, const ScanParams& p)
{
    pioran_ctx* ctx = ds->ctx;
    const ScanOptions& o = ctx->opt;
    if (o.no_tp || (!o.force_tp && (o.scan_config[0] || o.force_tile)) || !p.tab || p.npd_rows != 0 || p.B > 64) return PIORAN_ERR_UNSUPPORTED;
    PrepState* s = p.tab == ds->user.tab ? &ds->user : (p.tab == ds->host.tab ? &ds->host : nullptr);
    if (!s || !s->prepared || s->npd_terms != 0 || !s->dc || !s->dd) return PIORAN_ERR_UNSUPPORTED;
    // state rows: the two-row terms first (pairs on even / odd lanes), then the one-row terms, padded to an even count
    const int J = s->J;
    std::vector<int32_t> rows;
    rows.reserve(256);
    std::vector<int32_t> term, kind;
    for (int j = 0; j < J; ++j)
        if (!s->real_host[j]) { term.push_back(j); kind.push_back(0); term.push_back(j); kind.push_back(1); }
    for (int j = 0; j < J; ++j)
        if (s->real_host[j]) { term.push_back(j); kind.push_back(2); }
    if ((int)term.size() > pioran_tp_supported_rows() || p.N < 64) return PIORAN_ERR_UNSUPPORTED;
    // The boundary phase as a scan over the segments' elements (tp_combine_kernel, round 6: ceil(log2 nseg) launches of one workgroup per (draw, target)
    // instead of nseg - 1 dependent boundary steps) — up to two draws (nseg targets per draw and level want a CU each), 5 .. 64 state rows (padded to
    // a multiple of 8 for it) — moves every crossover (tools/tp_scan_sweep.py, profiles/r06_time_parallel_scan.txt; one scalar call, PCIe included):
    // 8 / 16 rows from 1024 steps on (N = 1024: 0.146 / 0.172 against 0.159 / 0.211 ms on the serial chain; N = 8192: 0.23 / 0.27 against 1.08 / 1.50),
    // 24 rows from 1536 (0.254 against 0.306), 32 from 2048 (0.32 against 0.42), 40 / 48 from 3072 (0.50 / 0.58 against 0.60 / 0.76; N = 1e4:
    // 0.64 / 0.76 against 1.87 / 2.40; N = 65536: 1.19 / 1.35 against 12.0 / 16.4).
    const int nrows = (int)term.size();
    // (three and four rows — the reference grid's j = 2 — padded to eight: N = 8192 0.19 ms against 0.28 on the one-thread boundary walk; from 2048 steps on)
    const bool scan_rows = (nrows > 4 || (nrows > 2 && p.B <= 2 && (p.N >= 2048 || o.tp_scan > 0))) && nrows <= 64;
    // Three to 32 draws (tools/tp_scan_batch_sweep.py, section 8 of the profile): the combinations of one level want a CU slot each — a CU holds kc = 4 / 2 / 1
    // workgroups of tp_combine_kernel at up to 8 / up to 32 / more rows (its LDS) — so the segment count is the largest power of two with B nseg <= 256 kc
    // (SHO-20, N = 1e4, 4 / 8 draws: 64 / 32 segments 0.76 / 1.06 ms against 1.50 / 1.53 on the walk and 1.85 on the serial chains; 128 segments 1.08 / 2.0).
    const int RPs = (nrows + 7) & ~7, kc = RPs <= 8 ? 4 : (RPs <= 32 ? 2 : 1);
    const bool scan_ok = o.tp_scan != 0 && scan_rows && (o.tp_scan > 0 || p.B <= 2 || (p.B <= 32 && 16 * p.B <= 256 * kc));      // the scan is possible
    bool scan = scan_ok;                                                                                                                      // ... and chosen (below)
    int scan_cap = 256;
    if (scan_ok && p.B > 2) { scan_cap = 16; while (2 * scan_cap * p.B <= 256 * kc && scan_cap < 256) scan_cap *= 2; }
    const int RPw = pioran_tp_padded_rows(nrows);        // rows as the boundary walk pads them (RPs: as the scan does)
    // measured (tools/ab_tp.py sweep, profiles/r05_time_parallel_gpu.txt): with up to 8 draws it beats the serial-chain kernels from 1024 steps on at
    // up to 4 state rows (N = 8192: one SHO term 0.17 against 1.16 ms, two 0.27 against 1.15; there also at 64 draws from 4096 steps on: 0.90
    // against 1.16 ms), from 2048 steps at up to 8 rows (four terms, N = 8192: 0.47 against 1.21), from 4096 at up to 12, from 6144 at up to 16
    // (eight terms: 0.96 against 1.47 ms); with more rows the boundary solves (R^3 each, one after the other) eat the gain (20 terms, N = 1e4:
    // 2.6 against 1.83 ms).
    if (!o.force_tp) {
        const bool few = RPw <= 4 && ((p.B <= 8 && p.N >= 1024) || p.N >= 4096);
        const bool mid = RPw > 4 && p.B <= 8 && p.N >= (RPw <= 8 ? 2048 : (RPw <= 12 ? 4096 : 6144)) && RPw <= 16;
        // 17 .. 64 state rows: the boundary solves cost 14 .. 47 us each (four wavefronts, products and rank-4 updates on the matrix cores, four pivots per barrier), and
        // the gain comes with the length of the series (its time grows like sqrt(N), the serial chain's like N): SHO-12 (24 rows) N = 8192 / 1e4 /
        // 65536 0.84 / 0.93 / 2.4 against 1.45 / 1.77 / 11.6 ms; SHO-20 (40 rows) N = 8192 / 1e4 / 65536 1.39 / 1.54 / 3.9 against 1.50 / 1.83 / 11.9;
        // SHO-24 (48 rows; three block columns on the serial chain) N = 1e4 1.80 against 2.47
        // 49 .. 64 state rows: DRWCelerite-20 (60 rows; four block columns on the serial chain) N = 1e4 2.71 against 2.61 (not chosen), N = 16384 / 65536
        // 3.5 / 7.0 against 4.3 / 19.1 ms; SHO-32 (64 rows; FIVE block columns on the serial chain) N = 8192 / 65536 2.45 / 7.0 against 3.7 / 34.4 ms
        const int64_t nwide = p.R + 1 > 64 ? 6144 : 12288;
        const int64_t nmin12 = RPw <= 24 ? 4096 : (RPw <= 32 ? 5120 : (RPw <= 40 ? 8192 : (RPw <= 48 ? 6144 : nwide)));
        const int64_t nmin8 = RPw <= 24 ? 5120 : (RPw <= 32 ? 6144 : (RPw <= 40 ? 8192 : (RPw <= 48 ? 8192 : nwide)));
        const bool many = RPw > 16 && ((p.B <= 2 && p.N >= nmin12) || (p.B <= 8 && p.N >= nmin8));
        // (49 .. 64 rows, tp_combine_lean_kernel: 56 / 60 rows from 4096 steps on — 0.89 / 1.02 against 1.03 / 1.08 ms; N = 1e4: 1.10 / 1.21 against 2.46 / 2.58;
        //  64 rows, five block columns on the serial chain, from 2048 — 0.89 against 0.98; N = 1e4: 1.21 against 4.6)
        const bool scanned = scan_ok && p.N >= (nrows <= 4 ? 2048 : RPs <= 16 ? 1024 : (RPs <= 24 ? 1536 : (RPs <= 32 ? 2048 : (RPs <= 48 ? 3072 : (p.R + 1 > 64 ? 2048 : 4096)))));
        // three and more draws on the scan: a model of its time (records + two phases of N / nseg steps + one combination per level and the check, in us)
        // against the serial chain's time per step (measured at N = 1e4, resident inputs), taken when it promises 15 % off (up to 8 rows, where the model is
        // optimistic at 32 draws: a quarter) — profiles/r06_time_parallel_scan.txt section 8 has the sweep this was held against at N = 2048 / 4096 / 1e4
        bool scanned_b = false;
        if (scan_ok && p.B > 2) {
            const int RP = RPs;
            const double tau = RP <= 16 ? 0.7 + RP / 8.0 : 1.0 + RP / 32.0, tc = 8.0 + (double)RP * RP / 50.0;
            int lv = 0;
            for (int c = scan_cap; c > 1; c >>= 1) ++lv;
            // (17 .. 32 rows: two combinations and eight phase wavefronts share a CU at the cap — measured 1.5 x the steps' time there)
            const double load = RP > 16 && RP <= 32 ? (double)p.B * scan_cap * 4.0 / 1024.0 : 1.0, rp = 1.0 + 0.5 * (load > 1.0 ? load - 1.0 : 0.0);
            double t_scan = 35.0 + rp * tau * (double)p.N / scan_cap + (lv + 1) * tc;
            const double s_chain = RP <= 8 ? 0.127 : (RP <= 24 ? 0.178 : (RP <= 40 ? 0.19 : (RP <= 48 ? 0.24 : (p.R + 1 > 64 ? 0.46 : 0.25))));
            // ~2 % of the prior draws of the SHO models and ~7 % of the models with one-row terms (DRWCelerite) fail the check (profiles/r06_time_parallel_scan.txt
            // section 11), and one failing draw sends the launch through the serial chain as well — its expected share
            t_scan += (1.0 - std::pow(nrows != 2 * J ? 0.93 : 0.98, (double)p.B)) * s_chain * (double)p.N;
            scanned_b = (int64_t)scan_cap * 16 <= p.N && t_scan < (RP <= 8 ? 0.75 : 0.85) * s_chain * (double)p.N;
        }
        // the scan where ITS rule says so (o.tp_scan > 0: wherever possible); else the boundary walk where its rules say so — a batch of 8 draws of SHO-20 that the
        // walk's rule admits is better off there (1.53 ms) than on the scan with its expected repair (0.98 + 57 % x 1.86)
        if (o.tp_scan < 0) scan = p.B <= 2 ? scanned : scanned_b;
        if (!scan && !few && !mid && !many) return PIORAN_ERR_UNSUPPORTED;
    }
    const int RP = scan ? RPs : RPw;
    while ((int)term.size() < RP) { term.push_back(0); kind.push_back(3); }
    // segments: phases 1 + 3 cost tau ~ 0.7 + R / 8 us per step with one wavefront per segment (up to 16 rows), ~ 1 + R / 32 with four; phase 2 t2 per
    // boundary as below (measured at 2 .. 48 rows, tools/ab_tp.py): N / nseg tau + nseg t2 is least at sqrt(tau N / t2)
    int nseg = o.tp_segments;
    if (nseg <= 0 && scan) {
        // N / nseg tau + ceil(log2 nseg) t_c, t_c = one combination (measured, tools/ab_tp.py: see profiles/r06_time_parallel_scan.txt): powers of two
        const double tau = RP <= 16 ? 0.7 + RP / 8.0 : 1.0 + RP / 32.0, tc = 8.0 + (double)RP * RP / 50.0;
        double best = 1e300;
        for (int cand = 8, lv = 3; cand <= scan_cap; cand *= 2, ++lv) {
            const double est = tau * (double)p.N / cand + lv * tc;
            if (est < best && (int64_t)cand * 16 <= p.N) { best = est; nseg = cand; }
        }
        if (nseg <= 0) nseg = 1;
    }
    if (nseg <= 0) {
        const double tau = RP <= 16 ? 0.7 + RP / 8.0 : 1.0 + RP / 32.0, t2 = RP == 2 ? 0.6 : (RP == 4 ? 2.0 : (RP <= 16 ? 1.3 + RP * RP / 21.0 : 5.0 + (double)RP * RP / 80.0));   // (2 / 4 rows: one thread per draw; up to 16: one wavefront, in registers;
                                                                                 //  above: four wavefronts, products on the matrix cores, four pivots per barrier)
        nseg = (int)std::lround(std::sqrt(tau * (double)p.N / t2));
    }
    if (nseg < 1) nseg = 1;
    if (nseg > (scan ? 256 : 128)) nseg = scan ? 256 : 128;
    if (scan && p.B > 2 && o.tp_segments <= 0) nseg = scan_cap;
    if ((int64_t)nseg * 16 > p.N) nseg = (int)(p.N / 16);
    const int64_t L = (p.N + nseg - 1) / nseg;
    nseg = (int)((p.N + L - 1) / L);
    rows = term;
    rows.insert(rows.end(), kind.begin(), kind.end());
    int rc = upload(ctx, ctx->btprow, rows.data(), rows.size() * sizeof(int32_t));
    if (rc) return rc;
    // (B N (6 RP + 5) doubles of records + 133 KB per (draw, segment): 25 GB for eight draws of 64 rows at N = 1e6 — under the same budget as every
    //  other chunked workspace; over it the serial-chain kernels take the call)
    if (!ws_fits(ctx, ctx->btp, pioran_tp_workspace_doubles(p.B, p.N, RP, nseg) * sizeof(double))) return PIORAN_ERR_UNSUPPORTED;
    rc = ensure(ctx, ctx->btp, pioran_tp_workspace_doubles(p.B, p.N, RP, nseg) * sizeof(double));
    if (rc) return rc == PIORAN_ERR_ALLOC ? PIORAN_ERR_UNSUPPORTED : rc;
    ScanParams q = p;
    q.C = s->dc; q.D = s->dd; q.J = J; q.opt = &ctx->opt;
    g_last_kernel = "tp";
    const int32_t* dr = (const int32_t*)ctx->btprow.p;
    // A draw whose boundary states fail the filter's check (tp_filter_kernel: 1 .. 3 % of the prior draws of the SHO models, 6 .. 8 % of the DRWCelerite models; the scan
    // or the walk ALONE is wrong by more than 1e-8 on a few per thousand of the latter — tools/tp_scan_accept.py, tp_walk_accuracy.py) is evaluated again by the
    // serial-chain windowed kernel (celerite_block_kernel with ScanParams::only_if: its workgroups leave at once for every draw that passed).
    bool repair = !o.tp_walk_repair && !o.tp_unchecked && !o.no_block && pioran_block_fits_value(p.R, p.J) &&
                  p.rec_stride == 3 * (int64_t)(s->R + 2) + 2 && (p.Y == nullptr) == (p.S2 == nullptr);
    if (repair) {
        rc = ensure_btab(ds, *s);
        if (rc == PIORAN_ERR_UNSUPPORTED) repair = false;
        else if (rc) return rc;
    }
    // (no repair pass available — the windowed kernel's table does not fit, "no_block" — and the walk-repair mode not asked for: the boundary walk instead of the
    //  scan; the walk-repair mode's own check, a state discrepancy relative to the state's largest entry, lets bad draws through: tools/tp_scan_metrics.py)
    // Late round 6: the boundary WALK is checked and repaired the same way (mode 4) — a long segment's element is no better conditioned than a composite of the scan:
    // on prior draws of DRWCelerite-10 the walk alone is off by up to 8e-7 where the serial chain holds 4e-10 (tools/tp_walk_accuracy.py).  Without a repair pass the
    // family is not an AUTOMATIC choice any more; forced (scan_config "tp"; options tp_unchecked / tp_walk_repair: tools) it runs unchecked as in round 5.
    if (!repair && !o.force_tp) return PIORAN_ERR_UNSUPPORTED;
    const int mode = repair ? (scan ? 2 : 4) : (scan && o.tp_walk_repair ? 1 : 0);
    if (scan && mode == 0 && nseg > 128) return PIORAN_ERR_UNSUPPORTED;      // (the segment count was chosen for the scan; the walk's kernels take up to 128)
    rc = pioran_launch_tp(q, RP, nseg, L, dr, dr + RP, (double*)ctx->btp.p, ctx->stream, mode);
    if (rc || !repair) return rc;
    ScanParams qr = p;
    qr.only_if = pioran_tp_disc((const double*)ctx->btp.p, p.B, p.N, RP, nseg);
    qr.only_if_tol = pioran_tp_scan_tol(&ctx->opt);
    rc = pioran_launch_scan_block(qr, s->btab, ctx->stream);
    if (rc == PIORAN_ERR_UNSUPPORTED) {
        ctx->last_err = "time-parallel scan: the serial-chain repair pass refused a launch its own conditions admit";
        return PIORAN_ERR_HIP;
    }
    return rc;
}

// The automatic choice between the windowed form with one draw per wavefront ("tile", 1) and the rest (0: step-by-step throughput layouts / the
// small-batch windowed kernel) for a shared-table batch of B draws with R active rows — a PURE function of its arguments, exported so that
// tests/test_host.py can hold it against the committed sweep (profiles/r05_tile_batch_sweep.txt: the choice must be within 5 % of the faster
// family on every measured line) and tools/retune_thresholds.py can print where it is not.  pass = draws per pass of the step-by-step
// layout that would take the batch (pioran_scan_pass_draws; 0 = not known: -1 is returned where the ladder needs it).
static int tile_choice(int32_t R, int64_t B, int64_t pass, int no_split)
{
    if (R >= 49) return B > (R > pioran_block_supported_rows() ? 256 : 512) ? 1 : 0;
    if (R < 17 || B <= 512) return 0;
    if (B <= 1024) return 1;
    if (R < 33) return B <= 2048 ? 1 : 0;   // one round of this kernel's workgroups (2048 draws): SHO-12 1536 / 2048 draws 3.5 / 3.6 against 4.3 / 4.4 ms,
                                          // SHO-16 level (5.3 / 5.4 against 5.4); beyond, the step-by-step layouts' pass (8192 / 4096 draws) is ahead
    if (R >= 39 && R <= 47) return 1;   // three block columns cost the same for 33 .. 47 rows, the step-by-step layouts ~R^2: from 39 rows on this
                                        // kernel is ahead on whole passes too (SHO-20, 4096 draws: 10.7 against 11.4 .. 12.0 ms)
    if (pass <= 0) return -1;
    const int64_t r = B % pass;
    // a remainder of up to one round of the small-batch kernel rides beside the scan (split_dispatch) where that kernel takes these rows
    const bool split = B > pass && r > 0 && r <= (R <= 47 ? 512 : 256) && !no_split;
    return r > 0 && 4 * r <= 3 * pass && !split ? 1 : 0;
}

// Large shared-table batches: the windowed form with one draw per wavefront (celerite_tile.hip, round 5).  Same table as the windowed
// kernel for small batches.  scan_config = "tile" forces it for any batch size.
int tile_dispatch(pioran_ds* ds, const ScanParams& p)
{
    pioran_ctx* ctx = ds->ctx;
    const ScanOptions& o = ctx->opt;
    const bool force = o.force_tile;
    // measured on N = 1e4 (tools/ab_tile.py, profiles/r05_tile_batch_sweep.txt): from 49 rows on (four block columns and more) it beats the
    // step-by-step layouts at every batch size above the small-batch windowed kernel's range — DRWCelerite-20 (60 rows) 1024 draws 6.3
    // against 7.8 ms, 4096 draws 17.5 against 25.1 ms; SHO-40 (80 rows) 512 draws 10.0 against 13.1 ms, 4096 draws 41.3 against 75.3 ms.
    // Up to 48 rows the throughput layouts (two draws per wavefront) are level with it (SHO-20: 11.2 against 11.4 ms) and stay the default.
    // Up to 38 rows (and at 48) the throughput layouts (two and four draws per wavefront) are level with it or ahead on whole passes (SHO-24, 4096
    // draws: 16.5 against 16.9 ms), well ahead of it below 33 rows (SHO-16, 4096 draws: 5.8 against 10.7 ms) — but their time is a staircase of passes
    // (SHO-20: 4096 draws), and this kernel's steps are a quarter of that (1024 draws: one workgroup per CU): it takes what falls between
    // (profiles/r05_tile_batch_sweep.txt: SHO-20 1024 draws 3.96 against 5.31 ms, 3072 draws 9.3 against 11.0, 5000 draws 14.8 against 16.7;
    // SHO-12 / SHO-16 / SHO-24 at 1024 draws 2.9 / 3.9 / 6.2 against 4.2 / 5.6 / 8.2 ms).
    bool automatic = !o.scan_config[0] && !o.no_tile && !o.no_block && p.tab && p.npd_rows == 0;
    if (automatic) {
        int choice = tile_choice(p.R, p.B, 0, o.no_split ? 1 : 0);
        if (choice < 0) choice = tile_choice(p.R, p.B, pioran_scan_pass_draws(p, nullptr), o.no_split ? 1 : 0);   // (the occupancy query only where the ladder needs it)
        automatic = choice == 1;
    }
    if (!(force || automatic) || !p.tab || p.npd_rows != 0 || !pioran_tile_fits(p.R, p.J)) return PIORAN_ERR_UNSUPPORTED;
    PrepState* s = p.tab == ds->user.tab ? &ds->user : (p.tab == ds->host.tab ? &ds->host : nullptr);
    if (!s || !s->prepared || s->npd_terms != 0 || p.rec_stride != 3 * (int64_t)(s->R + 2) + 2) return PIORAN_ERR_UNSUPPORTED;
    int rc = ensure_btab(ds, *s);
    if (rc) return rc;
    // workspace: 1 KB per draw and window (the windows' own covariance blocks); large batches in chunks of whole passes
    const int64_t pass = pioran_tile_pass_draws(p.R, ctx->ncu);
    int64_t chunk = p.B;
    while (chunk > pass && !ws_fits(ctx, ctx->bpair, pioran_tile_workspace_doubles(chunk, p.N) * sizeof(double)))
        chunk = ((chunk / 2 + pass - 1) / pass) * pass;
    if (!ws_fits(ctx, ctx->bpair, pioran_tile_workspace_doubles(chunk, p.N) * sizeof(double))) return PIORAN_ERR_UNSUPPORTED;
    rc = ensure(ctx, ctx->bpair, pioran_tile_workspace_doubles(chunk, p.N) * sizeof(double));
    if (rc) return rc == PIORAN_ERR_ALLOC ? PIORAN_ERR_UNSUPPORTED : rc;
    g_last_kernel = "tile";
    for (int64_t off = 0; off < p.B; off += chunk) {
        const ScanParams qc = slice_draws(p, off, p.B - off < chunk ? p.B - off : chunk);
        rc = pioran_launch_scan_tile(qc, s->btab, (double*)ctx->bpair.p, ctx->stream);
        if (rc) return rc;
    }
    return PIORAN_OK;
}

static int split_dispatch(pioran_ds* ds, const ScanParams& p)
{
    pioran_ctx* ctx = ds->ctx;
    const ScanOptions& o = ctx->opt;
    if (o.no_split || o.scan_config[0] || o.no_block || o.force_fallback || !p.tab || p.npd_rows != 0 || p.R < 6 || p.R > 95 ||
        !pioran_block_fits_value(p.R, p.J))
        return PIORAN_ERR_UNSUPPORTED;
    PrepState* s = p.tab == ds->user.tab ? &ds->user : (p.tab == ds->host.tab ? &ds->host : nullptr);
    if (!s || !s->prepared || s->npd_terms != 0 || p.rec_stride != 3 * (int64_t)(s->R + 2) + 2) return PIORAN_ERR_UNSUPPORTED;
    int wps = 0;
    const int64_t pass = pioran_scan_pass_draws(p, &wps);
    if (pass < 1024) return PIORAN_ERR_UNSUPPORTED;
    (void)wps;
    // ONE round of the windowed kernel's workgroups: 512 draws (two workgroups per CU) up to three block columns, 256 with four
    // (tools/sweep_batch_sizes.py, profiles/r04_batch_sizes.txt: SHO-20 4200 draws 16.8 -> 13.1 ms, 4608 16.8 -> 14.8; DRWCelerite-20 4200
    // 32.5 -> 28.6.  A second round no longer hides behind the scan — SHO-20 5000 draws: 19.7 against 16.8 ms in one launch — and neither
    // does sending what exceeds HALF a pass: 2500 draws 11.6 against 10.6 ms; both were measured and are not done.)
    const int64_t rem_max = p.R <= 47 ? 512 : 256;   // (64 .. 95 rows, five / six block columns: a pass of the scan is 1024 .. 2048 draws there)
    int64_t main_n = 0;
    const int64_t k = p.B / pass, r = p.B - k * pass;
    if (k >= 1 && r > 0 && r <= rem_max) main_n = k * pass;
    if (main_n <= 0 || main_n >= p.B) return PIORAN_ERR_UNSUPPORTED;
    // everything that can refuse is asked BEFORE the second stream gets work: the whole passes must be a launch the scan takes
    {
        const ScanParams probe = slice_draws(p, 0, main_n);
        const bool y80 = probe.R == pioran_scan_supported_rows_shared() && !o.no_win2 && (probe.B > 768 || o.no_wide);
        if (probe.R > pioran_scan_supported_rows() && !y80) return PIORAN_ERR_UNSUPPORTED;
    }
    int rc = ensure_btab(ds, *s);
    if (rc) return rc;
    if (!ctx->aux) {
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking));
        for (auto& e : ctx->gev) HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    // fork: the second stream sees everything the main stream has queued so far (inputs, tables)
    HIPCHK(ctx, hipEventRecord(ctx->gev[3], ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->aux, ctx->gev[3], 0));
    ScanParams qr = slice_draws(p, main_n, p.B - main_n);
    rc = pioran_launch_scan_block(qr, s->btab, ctx->aux);
    if (rc) { if (rc == PIORAN_ERR_HIP) ctx->last_err = "block kernel launch failed"; return rc; }
    HIPCHK(ctx, hipEventRecord(ctx->gev[4], ctx->aux));
    ScanParams qm = slice_draws(p, 0, main_n);
    rc = scan_dispatch(qm, ctx->stream);
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->gev[4], 0));   // join — on every path: the second stream's kernel writes the caller's buffers
    if (rc) { if (rc == PIORAN_ERR_HIP) ctx->last_err = "scan kernel launch failed"; return rc == PIORAN_ERR_UNSUPPORTED ? PIORAN_ERR_HIP : rc; }
    g_last_kernel = "scan + block (remainder)";
    return PIORAN_OK;
}

int launch(pioran_ds* ds, ScanParams& p)
{
    pioran_ctx* ctx = ds->ctx;
    p.opt = &ctx->opt;
    if (!ctx->opt.force_fallback) {
        int rc = tp_dispatch(ds, p);
        if (rc != PIORAN_ERR_UNSUPPORTED) {
            if (rc == PIORAN_ERR_HIP) ctx->last_err = "time-parallel kernel launch failed";
            return rc;
        }
        rc = tile_dispatch(ds, p);
        if (rc != PIORAN_ERR_UNSUPPORTED) {
            if (rc == PIORAN_ERR_HIP) ctx->last_err = "tile kernel launch failed";
            return rc;
        }
        rc = block_dispatch(ds, p);
        if (rc != PIORAN_ERR_UNSUPPORTED) {
            if (rc == PIORAN_ERR_HIP) ctx->last_err = "block kernel launch failed";
            return rc;
        }
        rc = split_dispatch(ds, p);
        if (rc != PIORAN_ERR_UNSUPPORTED) return rc;
    }
    if (p.R <= pioran_wide_supported_rows() && !ctx->opt.force_fallback) {
        int rc = scan_dispatch(p, ctx->stream);
        if (rc != PIORAN_ERR_UNSUPPORTED) {
            if (rc == PIORAN_ERR_HIP) ctx->last_err = "scan kernel launch failed";
            return rc;
        }
    }
    const int64_t chunk = p.B < 1024 ? p.B : 1024;
    int rc = ensure(ctx, ctx->bscratch, (size_t)chunk * pioran_fallback_scratch_doubles(p.R) * sizeof(double));
    if (rc) return rc;
    p.scratch = (double*)ctx->bscratch.p;
    g_last_kernel = "fallback";
    rc = pioran_launch_scan_fallback(p, ctx->stream);
    if (rc == PIORAN_ERR_HIP) ctx->last_err = "fallback kernel launch failed";
    return rc;
}

}  // namespace

int pioran_tile_choice(int32_t R, int64_t B, int64_t pass, int no_split) { return tile_choice(R, B, pass, no_split); }


extern "C" {

const char* pioran_strerror(int code)
{
    switch (code) {
        case PIORAN_OK: return "ok";
        case PIORAN_ERR_ARG: return "invalid argument";
        case PIORAN_ERR_HIP: return "HIP runtime error";
        case PIORAN_ERR_ALLOC: return "allocation failed";
        case PIORAN_ERR_UNSUPPORTED: return "unsupported size";
        default: return "unknown error";
    }
}

const char* pioran_last_hip_error(const pioran_ctx* ctx) { return ctx ? ctx->last_err.c_str() : ""; }

int pioran_abi_version(void) { return 7; }

// The FP64 FMA rate the device sustains now, at `waves_per_simd` (1 .. 8) wavefronts per SIMD on every SIMD: ~`ms` milliseconds of a pure
// v_fma_f64 stream, event-timed on the context's stream (table.hip).  Diagnostics: bench.py's frac_of_measured_fma_ceiling.
int pioran_ctx_fp64_probe(pioran_ctx* ctx, int waves_per_simd, double ms, double* tflops)
{
    if (!ctx || !tflops || waves_per_simd < 1 || waves_per_simd > 8 || !(ms > 0.0) || ms > 1000.0) return PIORAN_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int blocks = ctx->ncu * waves_per_simd;
    int rc = ensure(ctx, ctx->bscratch, (size_t)blocks * 256 * sizeof(double));
    if (rc) return rc;
    // 64 FMAs per trip at ~4.6 issue cycles each and `waves_per_simd` wavefronts sharing the SIMD, ~2 GHz
    int iters = (int)(ms * 1e-3 * 2.0e9 / (64.0 * 4.6 * waves_per_simd));
    if (iters < 64) iters = 64;
    double flop = 0.0;
    if ((rc = pioran_launch_fma_stream(blocks, 64, (double*)ctx->bscratch.p, nullptr, ctx->stream))) return rc;   // warm
    // (the context's internal event slots: 0 .. 11 are the caller's, pioran_ctx_event_record)
    HIPCHK(ctx, hipEventRecord(ctx->ev[14], ctx->stream));
    if ((rc = pioran_launch_fma_stream(blocks, iters, (double*)ctx->bscratch.p, &flop, ctx->stream))) return rc;
    HIPCHK(ctx, hipEventRecord(ctx->ev[15], ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev[15]));
    float t = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&t, ctx->ev[14], ctx->ev[15]));
    *tflops = t > 0.f ? flop / (t * 1e-3) / 1e12 : 0.0;
    return PIORAN_OK;
}

int pioran_ctx_set_option(pioran_ctx* ctx, const char* key, const char* value)
{
    if (!ctx || !key) return PIORAN_ERR_ARG;
    const bool on = value && value[0] && std::strcmp(value, "0") != 0;
    ScanOptions& o = ctx->opt;
    if (!std::strcmp(key, "scan_config")) {
        if (value && std::strlen(value) >= sizeof(o.scan_config)) return PIORAN_ERR_ARG;
        std::memset(o.scan_config, 0, sizeof(o.scan_config));
        // "tile": celerite_tile.hip for every launch it can take (any batch size); the launches it cannot take stay automatic
        o.force_tile = value && !std::strcmp(value, "tile");
        o.force_tp = value && !std::strcmp(value, "tp");
        if (value && !o.force_tile && !o.force_tp) std::strcpy(o.scan_config, value);
    } else if (!std::strcmp(key, "no_tp")) o.no_tp = on; else if (!std::strcmp(key, "tp_segments")) o.tp_segments = (value && value[0]) ? std::atoi(value) : 0;
    else if (!std::strcmp(key, "tp_scan")) o.tp_scan = (value && value[0]) ? std::atoi(value) : -1;
    else if (!std::strcmp(key, "tp_walk_repair")) o.tp_walk_repair = on;
    else if (!std::strcmp(key, "tp_unchecked")) o.tp_unchecked = on;
    else if (!std::strcmp(key, "tp_check")) o.tp_check = (value && value[0]) ? std::atoi(value) : 0;
    else if (!std::strcmp(key, "tp_scan_waves")) o.tp_scan_waves = (value && value[0]) ? std::atoi(value) : 0;
    else if (!std::strcmp(key, "tp_scan_lean")) o.tp_scan_lean = (value && value[0]) ? std::atoi(value) : 0;
    else if (!std::strcmp(key, "tp_scan_tol")) o.tp_scan_tol = (value && value[0]) ? std::atof(value) : 0.0;
    else if (!std::strcmp(key, "no_tile")) o.no_tile = on; else if (!std::strcmp(key, "no_wide")) o.no_wide = on;
    else if (!std::strcmp(key, "no_paired")) o.no_paired = on;
    else if (!std::strcmp(key, "no_mixed")) o.no_mixed = on;
    else if (!std::strcmp(key, "force_fallback")) o.force_fallback = on;
    else if (!std::strcmp(key, "no_block")) o.no_block = on;
    else if (!std::strcmp(key, "no_split")) o.no_split = on;
    else if (!std::strcmp(key, "win2")) o.win2 = on;
    else if (!std::strcmp(key, "no_win2")) o.no_win2 = on;
    else if (!std::strcmp(key, "win3")) o.win3 = on;
    else if (!std::strcmp(key, "no_win3")) o.no_win3 = on;
    else if (!std::strcmp(key, "btab_reference")) o.btab_reference = on;
    else if (!std::strcmp(key, "block_emode")) o.block_emode = (value && value[0]) ? std::atoi(value) : -1;
    else if (!std::strcmp(key, "dense_quad_threshold")) o.dense.quad_threshold = (value && value[0]) ? std::atoi(value) : -1;
    else if (!std::strcmp(key, "dense_pair_tiles")) o.dense.pair_tiles = (value && value[0]) ? std::atoi(value) : -1;
    else if (!std::strcmp(key, "dense_half_tile_limit")) o.dense.half_tile_limit = (value && value[0]) ? std::atoi(value) : -1;
    else if (!std::strcmp(key, "dense_batch_pair_threshold")) o.dense.batch_pair_threshold = (value && value[0]) ? std::atoi(value) : -1;
    else if (!std::strcmp(key, "dense_old_chain")) {
        const int v = (value && value[0]) ? std::atoi(value) : 0;
#ifndef PIORAN_EXPERIMENTS
        if (v < 0 || v > 1) return PIORAN_ERR_ARG;     // 2 .. 8 (timing experiments, garbage results) exist in experiment builds only
#endif
        o.dense.old_chain = v;
    }
    else if (!std::strcmp(key, "dense_no_pairs")) o.dense.no_pairs = on ? 1 : 0;
    else if (!std::strcmp(key, "dense_no_halves")) o.dense.no_halves = on ? 1 : 0;
    else if (!std::strcmp(key, "workspace_limit_mb")) o.workspace_limit_mb = (value && value[0]) ? std::atoll(value) : 0;
    else if (!std::strcmp(key, "dense_streams")) o.dense_streams = (value && value[0]) ? std::atoi(value) : 0;
    else if (!std::strcmp(key, "gsum")) o.gsum = (value && value[0]) ? std::atoi(value) : -1;
    else if (!std::strcmp(key, "exp")) o.exp = (value && value[0]) ? std::atoi(value) : 0;
    else if (!std::strcmp(key, "wide2")) o.wide2 = on;
    else if (!std::strcmp(key, "no_wide2")) o.no_wide2 = on;
    else return PIORAN_ERR_ARG;
    return PIORAN_OK;
}

static int ctx_create_impl(int device, void* stream, bool own, pioran_ctx** out)
{
    if (!out) return PIORAN_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return PIORAN_ERR_HIP;
    pioran_ctx* ctx = new (std::nothrow) pioran_ctx;
    if (!ctx) return PIORAN_ERR_ALLOC;
    ctx->device = device;
    // the only place the environment is read
    pioran_ctx_set_option(ctx, "scan_config", std::getenv("PIORAN_SCAN_CONFIG"));
    pioran_ctx_set_option(ctx, "no_wide", std::getenv("PIORAN_NO_WIDE"));
    pioran_ctx_set_option(ctx, "no_paired", std::getenv("PIORAN_NO_PAIRED"));
    pioran_ctx_set_option(ctx, "no_mixed", std::getenv("PIORAN_NO_MIXED"));
    pioran_ctx_set_option(ctx, "force_fallback", std::getenv("PIORAN_FORCE_FALLBACK"));
    pioran_ctx_set_option(ctx, "no_block", std::getenv("PIORAN_NO_BLOCK"));
    pioran_ctx_set_option(ctx, "no_tile", std::getenv("PIORAN_NO_TILE"));
    pioran_ctx_set_option(ctx, "no_tp", std::getenv("PIORAN_NO_TP"));
    pioran_ctx_set_option(ctx, "win2", std::getenv("PIORAN_WIN2"));
    pioran_ctx_set_option(ctx, "no_win2", std::getenv("PIORAN_NO_WIN2"));
    pioran_ctx_set_option(ctx, "gsum", std::getenv("PIORAN_GSUM"));
    pioran_ctx_set_option(ctx, "wide2", std::getenv("PIORAN_WIDE2"));
#ifdef PIORAN_EXPERIMENTS
    pioran_ctx_set_option(ctx, "exp", std::getenv("PIORAN_EXP"));     // experiment builds only: the product library never reads it
#endif
    pioran_ctx_set_option(ctx, "no_wide2", std::getenv("PIORAN_NO_WIDE2"));
    if (hipSetDevice(device) != hipSuccess) { delete ctx; return PIORAN_ERR_HIP; }
    if (hipDeviceGetAttribute(&ctx->ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ctx->ncu < 1) { delete ctx; return PIORAN_ERR_HIP; }
    if (own) {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return PIORAN_ERR_HIP; }
        ctx->own_stream = true;
    } else {
        ctx->stream = (hipStream_t)stream;
    }
    for (auto& e : ctx->ev)
        if (hipEventCreate(&e) != hipSuccess) { delete ctx; return PIORAN_ERR_HIP; }
    *out = ctx;
    return PIORAN_OK;
}

int pioran_ctx_create(int device, pioran_ctx** out) { return ctx_create_impl(device, nullptr, true, out); }

int pioran_ctx_create_on_stream(int device, void* hip_stream, pioran_ctx** out)
{
    return ctx_create_impl(device, hip_stream, false, out);
}

int pioran_ctx_destroy(pioran_ctx* ctx)
{
    if (!ctx) return PIORAN_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->scalar_ds) pioran_dataset_destroy(ctx->scalar_ds);
    ctx->scalar_ds = nullptr;
    pioran_ctx::Buf* bufs[] = {&ctx->bA, &ctx->bB, &ctx->bC, &ctx->bD, &ctx->bmu, &ctx->bnu, &ctx->bY,
                               &ctx->bS2, &ctx->bout, &ctx->bst, &ctx->bscratch, &ctx->bK, &ctx->bwork, &ctx->bshift, &ctx->bgtab, &ctx->bq, &ctx->bpair, &ctx->btp, &ctx->btprow};
    for (auto* b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (auto& e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : ctx->gev)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : ctx->dev_)
        if (e) (void)hipEventDestroy(e);
    for (auto& st : ctx->dstream)
        if (st) (void)hipStreamDestroy(st);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    if (ctx->aux) (void)hipStreamDestroy(ctx->aux);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return PIORAN_OK;
}

int pioran_ctx_trim(pioran_ctx* ctx)
{
    if (!ctx) return PIORAN_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    SYNC(ctx);
    pioran_ctx::Buf* bufs[] = {&ctx->bA, &ctx->bB, &ctx->bC, &ctx->bD, &ctx->bmu, &ctx->bnu, &ctx->bY,
                               &ctx->bS2, &ctx->bout, &ctx->bst, &ctx->bscratch, &ctx->bK, &ctx->bwork, &ctx->bshift, &ctx->bgtab, &ctx->bq, &ctx->bpair, &ctx->btp, &ctx->btprow};
    for (auto* b : bufs) {
        if (b->p) (void)hipFree(b->p);
        b->p = nullptr;
        b->cap = 0;
    }
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    ctx->pin = nullptr;
    ctx->pin_cap = ctx->pin_off = 0;
    return PIORAN_OK;
}

int pioran_ctx_synchronize(pioran_ctx* ctx)
{
    if (!ctx) return PIORAN_ERR_ARG;
    SYNC(ctx);
    return PIORAN_OK;
}

int pioran_ctx_event_record(pioran_ctx* ctx, int slot)
{
    if (!ctx || slot < 0 || slot >= 12) return PIORAN_ERR_ARG;   // 12..15: internal
    HIPCHK(ctx, hipEventRecord(ctx->ev[slot], ctx->stream));
    return PIORAN_OK;
}

int pioran_ctx_event_elapsed_ms(pioran_ctx* ctx, int a, int b, float* ms)
{
    if (!ctx || !ms || a < 0 || a >= 12 || b < 0 || b >= 12) return PIORAN_ERR_ARG;
    HIPCHK(ctx, hipEventSynchronize(ctx->ev[b]));
    HIPCHK(ctx, hipEventElapsedTime(ms, ctx->ev[a], ctx->ev[b]));
    return PIORAN_OK;
}

int pioran_dataset_create(pioran_ctx* ctx, int64_t N, const double* t, const double* y, const double* sigma2,
                          pioran_ds** out)
{
    if (!ctx || !out || N < 1 || !t || !y || !sigma2) return PIORAN_ERR_ARG;
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    pioran_ds* ds = new (std::nothrow) pioran_ds;
    if (!ds) return PIORAN_ERR_ALLOC;
    ds->ctx = ctx;
    ds->N = N;
    const size_t bytes = (size_t)N * sizeof(double);
    double* base = nullptr;
    if (hipMalloc((void**)&base, 3 * bytes) != hipSuccess) { delete ds; return PIORAN_ERR_ALLOC; }
    ds->t = base;
    ds->y = base + N;
    ds->s2 = base + 2 * N;
    hipError_t e = hipMemcpyAsync(ds->t, t, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ds->y, y, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ds->s2, sigma2, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        ctx->last_err = hipGetErrorString(e);
        (void)hipFree(base);
        delete ds;
        return PIORAN_ERR_HIP;
    }
    *out = ds;
    return PIORAN_OK;
}

int pioran_dataset_destroy(pioran_ds* ds)
{
    if (!ds) return PIORAN_ERR_ARG;
    (void)hipSetDevice(ds->ctx->device);
    (void)hipStreamSynchronize(ds->ctx->stream);
    if (ds->t) (void)hipFree(ds->t);
    for (PrepState* s : {&ds->user, &ds->host}) {
        if (s->tab) (void)hipFree(s->tab);
        if (s->btab) (void)hipFree(s->btab);
        if (s->rowmap) (void)hipFree(s->rowmap);
        if (s->dc) (void)hipFree(s->dc);
        if (s->dpd_terms) (void)hipFree(s->dpd_terms);
    }
    delete ds;
    return PIORAN_OK;
}

static int prepare_state(pioran_ds* ds, PrepState& s, int64_t J, const double* c, const double* d, const int32_t* real_term)
{
    if (!ds || J < 1 || J > (1 << 20) || !c || !d) return PIORAN_ERR_ARG;
    pioran_ctx* ctx = ds->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::vector<int32_t> real(J, 0);
    std::vector<double> cc(c, c + J), dv(d, d + J);
    std::vector<int32_t> pdlist;
    for (int64_t j = 0; j < J; ++j) {
        real[j] = real_term ? (real_term[j] == 2 ? 2 : (real_term[j] ? 1 : 0)) : 0;
        if (real[j] == 1 && d[j] != 0.0) return PIORAN_ERR_ARG;  // a real term must have d = 0
        if (real[j] == 2) { cc[j] = 0.0; dv[j] = 0.0; pdlist.push_back((int32_t)j); }   // (c, d) come per draw
    }
    if (pdlist.size() > 255 || J > 0xfffff) return PIORAN_ERR_UNSUPPORTED;
    c = cc.data();
    d = dv.data();
    const bool same = s.prepared && s.J == J && !std::memcmp(s.c_host.data(), c, J * sizeof(double)) &&
                      !std::memcmp(s.d_host.data(), d, J * sizeof(double)) && s.real_host == real;
    if (same) return PIORAN_OK;
    s.prepared = false;
    s.btab_ready = false;
    if ((size_t)J > s.dcd_cap) {
        if (s.dc) HIPCHK(ctx, hipFree(s.dc));
        s.dc = nullptr;
        if (hipMalloc((void**)&s.dc, 2 * (size_t)J * sizeof(double)) != hipSuccess) return PIORAN_ERR_ALLOC;
        s.dcd_cap = (size_t)J;
    }
    s.dd = s.dc + J;
    s.c_host.assign(c, c + J);
    s.d_host.assign(d, d + J);
    HIPCHK(ctx, hipMemcpyAsync(s.dc, s.c_host.data(), J * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(s.dd, s.d_host.data(), J * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    int rc = set_rowmap(ds, s, build_rowmap(J, real.data()));   // sets s.R
    if (rc) return rc;
    s.npd_terms = (int32_t)pdlist.size();
    if (!pdlist.empty()) {
        if (!s.dpd_terms && hipMalloc((void**)&s.dpd_terms, 256 * sizeof(int32_t)) != hipSuccess) return PIORAN_ERR_ALLOC;
        HIPCHK(ctx, hipMemcpyAsync(s.dpd_terms, pdlist.data(), pdlist.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        SYNC(ctx);
    }
    // the table is per ROW (v, x, phi) + (y_n, sigma2_n) per step, see table.hip
    const size_t need = pioran_table_doubles(ds->N, s.R);
    if (need > s.tab_cap) {
        if (s.tab) HIPCHK(ctx, hipFree(s.tab));
        s.tab = nullptr;
        s.tab_cap = 0;
        if (hipMalloc((void**)&s.tab, need * sizeof(double)) != hipSuccess) return PIORAN_ERR_ALLOC;
        s.tab_cap = need;
    }
    rc = pioran_launch_table(ds->N, s.R, s.rowmap, ds->t, s.dc, s.dd, ds->y, ds->s2, s.tab,
                             3 * (int64_t)(s.R + 2) + 2, ctx->stream);
    if (rc) return rc;
    s.J = (int32_t)J;
    s.real_host = real;
    {   // 1: every term has both rows; 2: two-row terms first, then one-row terms only; 0: anything else
        int64_t nc = 0;
        while (nc < J && real[nc] == 0) ++nc;
        bool rest_real = true;
        for (int64_t j = nc; j < J; ++j) rest_real = rest_real && real[j] == 1;
        s.row_layout = nc == J ? 1 : ((rest_real && pdlist.empty()) ? 2 : 0);
        s.n_complex = (int32_t)nc;
    }
    s.prepared = true;
    return PIORAN_OK;
}

int pioran_dataset_prepare(pioran_ds* ds, int64_t J, const double* c, const double* d, const int32_t* real_term)
{
    if (!ds) return PIORAN_ERR_ARG;
    return prepare_state(ds, ds->user, J, c, d, real_term);
}

static int batch_dev_impl(pioran_ds* ds, const PrepState& s, int64_t B, const double* dA, const double* dBc, const double* dmu,
                          const double* dnu, const double* dY, const double* dS2, double* dout, int32_t* dstatus)
{
    if (!ds || B < 1 || !dA || !dBc || !dout) return PIORAN_ERR_ARG;
    if (!s.prepared || s.npd_terms > 0) return PIORAN_ERR_ARG;   // per-draw terms need the mixed-mode host entry
    if ((dY == nullptr) != (dS2 == nullptr)) return PIORAN_ERR_ARG;
    pioran_ctx* ctx = ds->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ScanParams p{};
    p.N = ds->N; p.J = s.J; p.R = s.R; p.B = B;
    p.standard_rows = s.row_layout; p.n_complex = s.n_complex;
    p.rec_stride = 3 * (int64_t)(s.R + 2) + 2;
    p.tab = s.tab; p.rowmap = s.rowmap; p.t = ds->t; p.y = ds->y; p.s2 = ds->s2;
    p.Y = dY; p.S2 = dS2; p.A = dA; p.Bc = dBc; p.C = s.dc; p.D = s.dd;
    p.mu = dmu; p.nu = dnu; p.out = dout; p.status = dstatus;
    return launch(ds, p);
}

int pioran_celerite_logl_batch_dev(pioran_ds* ds, int64_t B, const double* dA, const double* dBc, const double* dmu,
                                   const double* dnu, const double* dY, const double* dS2, double* dout,
                                   int32_t* dstatus)
{
    if (!ds) return PIORAN_ERR_ARG;
    return batch_dev_impl(ds, ds->user, B, dA, dBc, dmu, dnu, dY, dS2, dout, dstatus);
}

int pioran_celerite_logl_batch_dev_cd(pioran_ds* ds, int64_t B, int64_t J, const double* dA, const double* dBc,
                                      const double* dC, const double* dDd, const double* dmu, const double* dnu,
                                      const double* dY, const double* dS2, double* dout, int32_t* dstatus)
{
    if (!ds || B < 1 || J < 1 || !dA || !dBc || !dC || !dDd || !dout) return PIORAN_ERR_ARG;
    if ((dY == nullptr) != (dS2 == nullptr)) return PIORAN_ERR_ARG;
    pioran_ctx* ctx = ds->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // per-draw (c, d): full 2J rows, no table; the prepared shared state is left untouched
    std::vector<int32_t> rm = build_rowmap(J, nullptr);
    int32_t* drm = nullptr;
    int rc = ensure(ctx, ctx->bwork, rm.size() * sizeof(int32_t));
    if (rc) return rc;
    drm = (int32_t*)ctx->bwork.p;
    HIPCHK(ctx, hipMemcpyAsync(drm, rm.data(), rm.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    SYNC(ctx);
    if ((int64_t)rm.size() > pioran_scan_supported_rows() && (int64_t)rm.size() <= pioran_wide_supported_rows() && !ctx->opt.force_fallback) {
        // More rows than the throughput layouts hold (which evaluate per-draw transcendentals in the kernel): every draw gets its
        // OWN table, built for a chunk of draws at a time, and the lean latency kernel walks it (one draw per workgroup) — the
        // reference benchmark's j = 64 with the reference's call pattern (one random (a, b, c, d) per call,
        // benchmark/benchmarks.jl:74-91): 16 k instead of 0.8 k evaluations per second (any-rank kernel, S in HBM).
        const int32_t R = (int32_t)rm.size();
        const int64_t rec = 3 * (int64_t)(R + 2) + 2, tdoubles = (int64_t)pioran_table_doubles(ds->N, R);
        int64_t chunk = B < 256 ? B : 256;
        {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
                while (chunk > 1 && (size_t)chunk * (size_t)tdoubles * sizeof(double) > ws_allow(ctx, free_b) + ctx->bscratch.cap) chunk /= 2;
        }
        while ((rc = ensure(ctx, ctx->bscratch, (size_t)chunk * (size_t)tdoubles * sizeof(double))) == PIORAN_ERR_ALLOC && chunk > 1) chunk /= 2;
        if (rc) return rc;
        for (int64_t b0 = 0; b0 < B; b0 += chunk) {
            const int64_t nb = B - b0 < chunk ? B - b0 : chunk;
            rc = pioran_launch_table_batch(ds->N, R, (int32_t)J, nb, drm, ds->t, dC + b0 * J, dDd + b0 * J, ds->y, ds->s2,
                                           (double*)ctx->bscratch.p, rec, tdoubles, ctx->stream);
            if (rc) return rc;
            ScanParams q{};
            q.opt = &ctx->opt;
            q.N = ds->N; q.J = (int32_t)J; q.R = R; q.B = nb; q.standard_rows = 1;
            q.rec_stride = rec; q.tab_draw_stride = tdoubles;
            q.tab = (const double*)ctx->bscratch.p; q.rowmap = drm; q.t = ds->t; q.y = ds->y; q.s2 = ds->s2;
            q.Y = dY ? dY + b0 * ds->N : nullptr; q.S2 = dS2 ? dS2 + b0 * ds->N : nullptr;
            q.A = dA + b0 * J; q.Bc = dBc + b0 * J; q.C = dC + b0 * J; q.D = dDd + b0 * J;
            q.mu = dmu ? dmu + b0 : nullptr; q.nu = dnu ? dnu + b0 : nullptr;
            q.out = dout + b0; q.status = dstatus ? dstatus + b0 : nullptr;
            g_last_kernel = "wide (per-draw tables)";
            rc = pioran_launch_scan_wide(q, ctx->stream);
            if (rc) { if (rc == PIORAN_ERR_HIP) ctx->last_err = "per-draw-table latency kernel launch failed"; return rc; }
        }
        return PIORAN_OK;
    }
    {
        // Small batches, up to 63 rows: every draw gets its own table of the WINDOWED kernel (celerite_block.hip; one workgroup per
        // (window, draw) builds it: ~8 us per table at N = 1e4, J = 20) instead of evaluating 3 J transcendentals per step and draw inside
        // the throughput layout (14.7 ms per launch at N = 1e4, J = 20 whatever the batch): free Celerite / CARMA terms under a sampler
        // (src/CARMA.jl:98-143).  tools/bench_per_draw_small.py.
        const ScanOptions& o = ctx->opt;
        const int32_t R = (int32_t)rm.size();
        const bool automatic = !o.scan_config[0] && !o.no_block && B <= 768 && R >= 6;
        const bool force = !std::strcmp(o.scan_config, "block");
        if ((automatic || force) && !o.force_fallback && pioran_block_fits(R, (int32_t)J)) {
            const int64_t tdoubles = (int64_t)pioran_block_table_doubles(ds->N, R, (int32_t)J);
            int64_t chunk = B < 256 ? B : 256;
            {
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
                    while (chunk > 1 && (size_t)chunk * (size_t)tdoubles * sizeof(double) > ws_allow(ctx, free_b) + ctx->bscratch.cap) chunk /= 2;
            }
            while ((rc = ensure(ctx, ctx->bscratch, (size_t)chunk * (size_t)tdoubles * sizeof(double))) == PIORAN_ERR_ALLOC && chunk > 1) chunk /= 2;
            if (rc) return rc;
            for (int64_t b0 = 0; b0 < B; b0 += chunk) {
                const int64_t nb = B - b0 < chunk ? B - b0 : chunk;
                rc = pioran_launch_block_table_batch(ds->N, R, (int32_t)J, nb, drm, ds->t, dC + b0 * J, dDd + b0 * J, ds->y, ds->s2,
                                                     (double*)ctx->bscratch.p, tdoubles, ctx->stream);
                if (rc) return rc;
                ScanParams q{};
                q.opt = &ctx->opt;
                q.N = ds->N; q.J = (int32_t)J; q.R = R; q.B = nb; q.standard_rows = 1;
                q.rec_stride = 3 * (int64_t)(R + 2) + 2; q.tab_draw_stride = tdoubles;
                q.rowmap = drm; q.t = ds->t; q.y = ds->y; q.s2 = ds->s2;
                q.Y = dY ? dY + b0 * ds->N : nullptr; q.S2 = dS2 ? dS2 + b0 * ds->N : nullptr;
                q.A = dA + b0 * J; q.Bc = dBc + b0 * J; q.C = dC + b0 * J; q.D = dDd + b0 * J;
                q.mu = dmu ? dmu + b0 : nullptr; q.nu = dnu ? dnu + b0 : nullptr;
                q.out = dout + b0; q.status = dstatus ? dstatus + b0 : nullptr;
                g_last_kernel = "block (per-draw tables)";
                rc = pioran_launch_scan_block(q, (const double*)ctx->bscratch.p, ctx->stream);
                if (rc) { if (rc == PIORAN_ERR_HIP) ctx->last_err = "windowed kernel (per-draw tables) launch failed"; return rc; }
            }
            return PIORAN_OK;
        }
    }
    ScanParams p{};
    p.N = ds->N; p.J = (int32_t)J; p.R = (int32_t)rm.size(); p.B = B;
    p.standard_rows = 1;
    p.rec_stride = 0;
    p.tab = nullptr; p.rowmap = drm; p.t = ds->t; p.y = ds->y; p.s2 = ds->s2;
    p.Y = dY; p.S2 = dS2; p.A = dA; p.Bc = dBc; p.C = dC; p.D = dDd;
    p.mu = dmu; p.nu = dnu; p.out = dout; p.status = dstatus;
    return launch(ds, p);
}

// Shared (c, d): terms whose sin row is identically zero for the whole batch (d_j = 0 and b_j = 0 for all draws) keep
// only their cos row; builds (or reuses) the shared table.
static int prepare_shared(pioran_ds* ds, int64_t B, int64_t J, const double* Bc, const double* C, const double* Dd)
{
    std::vector<int32_t> real(J, 0);
    for (int64_t j = 0; j < J; ++j) {
        if (Dd[j] != 0.0) continue;
        bool allzero = true;
        for (int64_t b = 0; b < B && allzero; ++b) allzero = (Bc[b * J + j] == 0.0);
        real[j] = allzero;
    }
    return prepare_state(ds, ds->host, J, C, Dd, real.data());
}

// Mixed mode (host-pointer entry, cd_shared == 0): when only a few terms really differ between draws (QPO features on
// top of an approx continuum, src/psd.jl:254-261), the shared terms keep using the shared table and only the per-draw
// terms get a per-draw table, built by a pre-pass kernel for chunks of draws (32-bit buffer offsets).
// Returns 1 if it handled the batch, 0 if the caller should take the generic per-draw path, < 0 on error.
// Mixed-mode core: kind[J] (0 shared two-row, 1 shared real, 2 per-draw) and the shared values (C0, D0: [J], entries of
// per-draw terms ignored) declare the layout; `fetch(b0, nb, ptrs)` hands the DEVICE pointers of one chunk of draws
// (A, Bc, C, D as [nb][J]; mu, nu [nb] or nullptr; Y, S2 [nb][N] or nullptr).  Results go to the caller's host arrays.
// Returns 1 if it handled the batch, 0 if this layout is not worth / not able to run mixed (caller takes the generic path).
struct MixedChunk { const double *A, *Bc, *C, *D, *mu, *nu, *Y, *S2; };
static int mixed_core(pioran_ds* ds, int64_t B, int64_t J, const std::vector<int32_t>& kind, const double* C0, const double* D0,
                      const std::function<int(int64_t, int64_t, MixedChunk&)>& fetch, double* out, int32_t* status,
                      bool must_run = false)
{
    pioran_ctx* ctx = ds->ctx;
    PrepState& s = ds->host;
    int64_t npd = 0, rows = 0;
    for (int64_t j = 0; j < J; ++j) { npd += kind[j] == 2; rows += kind[j] == 1 ? 1 : 2; }
    // all shared is handled by the caller; many per-draw terms: the generic per-draw path is as good.  must_run: the caller has
    // no generic path to fall back to (theta entry: the continuum's (c, d) exist only as a shared table), so the two
    // "not worth it" cuts — a performance heuristic, not a kernel constraint — are skipped
    // (the windowed kernel takes small batches with one or two per-draw terms whatever the share of per-draw terms: see below)
    bool blk_ok = false;
    {
        const ScanOptions& o = ctx->opt;
        const char* cfg = o.scan_config[0] ? o.scan_config : nullptr;
        const bool force = cfg && !std::strcmp(cfg, "block");
        // (fewer than six rows, late round 4, tools/per_draw_few_rows.py: the generic per-draw path took 7.5 ms for 16 draws of ONE term at N = 1e4 —
        //  1.7 ms here; 768 draws 8.2 -> 4.2 ms)
        const bool automatic = !cfg && !o.no_block && (rows >= 6 ? B <= 512 : B <= 768);
        blk_ok = (force || automatic) && !o.force_fallback && npd >= 1 && pioran_block_fits_pd((int32_t)rows, (int32_t)J, (int32_t)npd);
    }
    if (npd == 0 || npd > 8 || (!must_run && !blk_ok && npd * 2 > J)) return 0;
    if (rows > pioran_scan_supported_rows()) return 0;
    const int64_t rs_shared = 3 * (rows + 2) + 2;               // shared part of a step record (doubles)
    // combined table: (N+1) records of rs_shared + chunk * 2 npd * 3 doubles, addressed with 32-bit byte offsets
    int64_t chunk = ((int64_t)0x7fff0000 / ((ds->N + 1) * 8) - rs_shared) / (6 * npd);
    chunk = chunk > B ? B : (chunk >= 16 ? chunk & ~(int64_t)15 : chunk);
    if (chunk < 1 || (!must_run && !blk_ok && chunk < 16)) return 0;
    int rc;
    if ((rc = prepare_state(ds, s, J, C0, D0, kind.data()))) return rc;
    // Small batches with one or two per-draw terms (a QPO feature on an approx continuum at a few hundred live points,
    // src/psd.jl:254-261): the windowed kernel with per-draw rows (celerite_block.hip; round 3) — same automatic range as for shared
    // batches (block_dispatch).  Needs the kernel's own table of the shared rows and the per-draw (cos, sin)(d t_n).
    {
        if (blk_ok && (rc = ensure_btab(ds, s)) != PIORAN_ERR_UNSUPPORTED) {
            if (rc) return rc;
            const int64_t cb = B < 4096 ? B : 4096;
            const size_t trig_bytes = pioran_block_pd_trig_doubles(ds->N, cb, s.npd_terms) * sizeof(double);
            if ((rc = ensure(ctx, ctx->bscratch, trig_bytes))) return rc;
            if ((rc = ensure(ctx, ctx->bout, cb * sizeof(double)))) return rc;
            if ((rc = ensure(ctx, ctx->bst, cb * sizeof(int32_t)))) return rc;
            for (int64_t b0 = 0; b0 < B; b0 += cb) {
                const int64_t nb = B - b0 < cb ? B - b0 : cb;
                MixedChunk m{};
                if ((rc = fetch(b0, nb, m))) return rc;
                if ((rc = pioran_launch_block_pd_trig(ds->N, nb, (int32_t)J, s.npd_terms, s.dpd_terms, ds->t, m.D, (double*)ctx->bscratch.p, ctx->stream)))
                    return rc;
                ScanParams p{};
                p.N = ds->N; p.J = s.J; p.R = s.R; p.B = nb;
                p.standard_rows = s.row_layout; p.n_complex = s.n_complex;
                p.rec_stride = 3 * (int64_t)(s.R + 2) + 2;
                p.tab = s.tab; p.rowmap = s.rowmap; p.t = ds->t; p.y = ds->y; p.s2 = ds->s2;
                p.Y = m.Y; p.S2 = m.S2; p.A = m.A; p.Bc = m.Bc; p.C = s.dc; p.D = s.dd;
                p.mu = m.mu; p.nu = m.nu;
                p.out = (double*)ctx->bout.p; p.status = (int32_t*)ctx->bst.p;
                p.npd_rows = 2 * s.npd_terms;
                p.pd_C = m.C; p.pd_trig = (const double*)ctx->bscratch.p; p.pd_npad = (ds->N + 15) / 16 * 16;
                p.opt = &ctx->opt;
                g_last_kernel = "block+pd";
                rc = pioran_launch_scan_block(p, s.btab, ctx->stream);
                if (rc) { ctx->last_err = "windowed kernel (per-draw rows) launch failed"; return rc; }
                if ((rc = download(ctx, out + b0, ctx->bout.p, nb * sizeof(double)))) return rc;
                if (status) if ((rc = download(ctx, status + b0, ctx->bst.p, nb * sizeof(int32_t)))) return rc;
                SYNC(ctx);
            }
            return 1;
        }
    }
    const int64_t rec_stride = rs_shared + chunk * 6 * npd;
    if ((rc = ensure(ctx, ctx->bscratch, (size_t)(ds->N + 1) * (size_t)rec_stride * sizeof(double)))) return rc;
    double* ctab = (double*)ctx->bscratch.p;
    // shared rows into the combined layout (same kernel as the plain table, wider record stride)
    if ((rc = pioran_launch_table(ds->N, s.R, s.rowmap, ds->t, s.dc, s.dd, ds->y, ds->s2, ctab, rec_stride, ctx->stream)))
        return rc;
    if ((rc = ensure(ctx, ctx->bout, chunk * sizeof(double)))) return rc;
    if ((rc = ensure(ctx, ctx->bst, chunk * sizeof(int32_t)))) return rc;
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        const int64_t nb = B - b0 < chunk ? B - b0 : chunk;
        MixedChunk m{};
        if ((rc = fetch(b0, nb, m))) return rc;
        rc = pioran_launch_pd_table(ds->N, nb, (int32_t)J, s.npd_terms, s.dpd_terms, ds->t, m.C, m.D, ctab, rec_stride, rs_shared,
                                    ctx->stream);
        if (rc) return rc;
        ScanParams p{};
        p.N = ds->N; p.J = s.J; p.R = s.R; p.B = nb;
        p.standard_rows = s.row_layout; p.n_complex = s.n_complex;
        p.rec_stride = rec_stride;
        p.tab = ctab; p.rowmap = s.rowmap; p.t = ds->t; p.y = ds->y; p.s2 = ds->s2;
        p.Y = m.Y; p.S2 = m.S2; p.A = m.A; p.Bc = m.Bc; p.C = s.dc; p.D = s.dd;
        p.mu = m.mu; p.nu = m.nu;
        p.out = (double*)ctx->bout.p; p.status = (int32_t*)ctx->bst.p;
        p.npd_rows = 2 * s.npd_terms;
        p.opt = &ctx->opt;
        rc = scan_dispatch(p, ctx->stream);
        if (rc) { ctx->last_err = "mixed-mode scan launch failed"; return rc; }
        if ((rc = download(ctx, out + b0, ctx->bout.p, nb * sizeof(double)))) return rc;
        if (status) if ((rc = download(ctx, status + b0, ctx->bst.p, nb * sizeof(int32_t)))) return rc;
        SYNC(ctx);
    }
    return 1;
}

static int batch_host_mixed(pioran_ds* ds, int64_t B, int64_t J, const double* A, const double* Bc, const double* C,
                            const double* Dd, const double* mu, const double* nu, const double* Y, const double* S2,
                            bool series_on_device, double* out, int32_t* status)
{
    pioran_ctx* ctx = ds->ctx;
    if (ctx->opt.no_mixed) return 0;
    std::vector<int32_t> kind(J, 0);
    for (int64_t j = 0; j < J; ++j) {
        bool shared = true;
        for (int64_t b = 1; b < B && shared; ++b) shared = C[b * J + j] == C[j] && Dd[b * J + j] == Dd[j];
        if (!shared) { kind[j] = 2; continue; }
        if (Dd[j] == 0.0) {
            bool allzero = true;
            for (int64_t b = 0; b < B && allzero; ++b) allzero = Bc[b * J + j] == 0.0;
            kind[j] = allzero ? 1 : 0;
        }
    }
    auto fetch = [&](int64_t b0, int64_t nb, MixedChunk& m) -> int {
        int rc;
        const size_t bj = (size_t)nb * (size_t)J * sizeof(double);
        if ((rc = upload(ctx, ctx->bA, A + b0 * J, bj))) return rc;
        if ((rc = upload(ctx, ctx->bB, Bc + b0 * J, bj))) return rc;
        if ((rc = upload(ctx, ctx->bC, C + b0 * J, bj))) return rc;
        if ((rc = upload(ctx, ctx->bD, Dd + b0 * J, bj))) return rc;
        if (mu && (rc = upload(ctx, ctx->bmu, mu + b0, nb * sizeof(double)))) return rc;
        if (nu && (rc = upload(ctx, ctx->bnu, nu + b0, nb * sizeof(double)))) return rc;
        m.A = (const double*)ctx->bA.p; m.Bc = (const double*)ctx->bB.p; m.C = (const double*)ctx->bC.p; m.D = (const double*)ctx->bD.p;
        m.mu = mu ? (const double*)ctx->bmu.p : nullptr; m.nu = nu ? (const double*)ctx->bnu.p : nullptr;
        if (Y) {
            const size_t bn = (size_t)nb * (size_t)ds->N * sizeof(double);
            if ((rc = upload(ctx, ctx->bY, Y + b0 * ds->N, bn))) return rc;
            if ((rc = upload(ctx, ctx->bS2, S2 + b0 * ds->N, bn))) return rc;
            m.Y = (const double*)ctx->bY.p; m.S2 = (const double*)ctx->bS2.p;
        } else if (series_on_device) {
            m.Y = (const double*)ctx->bY.p + b0 * ds->N; m.S2 = (const double*)ctx->bS2.p + b0 * ds->N;
        }
        return PIORAN_OK;
    };
    return mixed_core(ds, B, J, kind, C, Dd, fetch, out, status);   // row 0 of C, Dd: the shared values
}

// host-pointer batch; series_on_device: ctx->bY / ctx->bS2 already hold the per-draw series (shift transform)
static int batch_host_impl(pioran_ds* ds, int64_t B, int64_t J, const double* A, const double* Bc, const double* C,
                           const double* Dd, int cd_shared, const double* mu, const double* nu, const double* Y,
                           const double* S2, bool series_on_device, double* out, int32_t* status)
{
    if (!ds || B < 1 || J < 1 || !A || !Bc || !C || !Dd || !out) return PIORAN_ERR_ARG;
    if ((Y == nullptr) != (S2 == nullptr)) return PIORAN_ERR_ARG;
    pioran_ctx* ctx = ds->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PendingGuard pending_guard(ctx);
    const size_t bj = (size_t)B * (size_t)J * sizeof(double);
    const size_t bn = (size_t)B * (size_t)ds->N * sizeof(double);
    int rc;
    if (!cd_shared) {   // one draw, or the same (c, d) in every draw: that IS the shared case (table built once, windowed kernel for small batches)
        bool same = true;
        for (int64_t b = 1; b < B && same; ++b)
            same = !std::memcmp(C + b * J, C, (size_t)J * sizeof(double)) && !std::memcmp(Dd + b * J, Dd, (size_t)J * sizeof(double));
        if (same) cd_shared = 1;
    }
    if (!cd_shared && B > 1) {
        rc = batch_host_mixed(ds, B, J, A, Bc, C, Dd, mu, nu, Y, S2, series_on_device, out, status);
        if (rc < 0) return rc;
        if (rc == 1) return PIORAN_OK;
    }
    if (cd_shared && (rc = prepare_shared(ds, B, J, Bc, C, Dd))) return rc;
    // Small calls with shared (c, d) — the scalar drop-in, a few walkers (late round 4): the kernels read the coefficients straight from
    // pinned host memory (each value once) and write log L / status straight into it, so the call issues no copy command for them; a
    // per-draw series, which the kernels stream step by step, still goes to HBM, (y | sigma2) in ONE copy.  Scalar call, N = 32: 44 -> 3x us.
    const size_t zc_in = 2 * bj + (mu ? B * sizeof(double) : 0) + (nu ? B * sizeof(double) : 0);
    const size_t zc_all = zc_in + B * sizeof(double) + ((B * sizeof(int32_t) + 7) & ~size_t(7));
    const size_t zc_pad = (zc_all + 255) & ~size_t(255);
    const bool y_staged = Y && zc_pad + 2 * bn <= kPinMaxRequest;      // (ONE reservation: a second one could drain or move the staging area)
    char* zc = (cd_shared && zc_all <= (size_t(16) << 10) && !(ctx->opt.exp & 64)) ? (char*)pin_reserve(ctx, zc_pad + (y_staged ? 2 * bn : 0)) : nullptr;
    if (zc) {
        const double *dA = (const double*)zc, *dB = dA + B * J, *dmu = nullptr, *dnu = nullptr;
        char* q = zc + 2 * bj;
        std::memcpy(zc, A, bj); std::memcpy(zc + bj, Bc, bj);
        if (mu) { std::memcpy(q, mu, B * sizeof(double)); dmu = (const double*)q; q += B * sizeof(double); }
        if (nu) { std::memcpy(q, nu, B * sizeof(double)); dnu = (const double*)q; q += B * sizeof(double); }
        double* dout = (double*)q; q += B * sizeof(double);
        int32_t* dst = (int32_t*)q;
        const double *dY = nullptr, *dS2 = nullptr;
        if (Y) {
            if ((rc = ensure(ctx, ctx->bY, 2 * bn))) return rc;
            if (y_staged) {
                char* sy = zc + zc_pad;
                std::memcpy(sy, Y, bn); std::memcpy(sy + bn, S2, bn);
                HIPCHK(ctx, hipMemcpyAsync(ctx->bY.p, sy, 2 * bn, hipMemcpyHostToDevice, ctx->stream));
            } else {
                HIPCHK(ctx, hipMemcpyAsync(ctx->bY.p, Y, bn, hipMemcpyHostToDevice, ctx->stream));
                HIPCHK(ctx, hipMemcpyAsync((char*)ctx->bY.p + bn, S2, bn, hipMemcpyHostToDevice, ctx->stream));
            }
            dY = (const double*)ctx->bY.p; dS2 = dY + (size_t)B * (size_t)ds->N;
        } else if (series_on_device) {
            dY = (const double*)ctx->bY.p; dS2 = (const double*)ctx->bS2.p;
        }
        if ((rc = batch_dev_impl(ds, ds->host, B, dA, dB, dmu, dnu, dY, dS2, dout, dst))) return rc;
        ctx->pending.push_back({out, dout, B * sizeof(double)});
        if (status) ctx->pending.push_back({status, dst, B * sizeof(int32_t)});
        SYNC(ctx);
        return PIORAN_OK;
    }
    if ((rc = upload(ctx, ctx->bA, A, bj))) return rc;
    if ((rc = upload(ctx, ctx->bB, Bc, bj))) return rc;
    if (!cd_shared) {
        if ((rc = upload(ctx, ctx->bC, C, bj))) return rc;
        if ((rc = upload(ctx, ctx->bD, Dd, bj))) return rc;
    }
    if (mu && (rc = upload(ctx, ctx->bmu, mu, B * sizeof(double)))) return rc;
    if (nu && (rc = upload(ctx, ctx->bnu, nu, B * sizeof(double)))) return rc;
    if (Y) {
        if ((rc = upload(ctx, ctx->bY, Y, bn))) return rc;
        if ((rc = upload(ctx, ctx->bS2, S2, bn))) return rc;
    }
    if ((rc = ensure(ctx, ctx->bout, B * sizeof(double)))) return rc;
    if ((rc = ensure(ctx, ctx->bst, B * sizeof(int32_t)))) return rc;
    const double* dmu = mu ? (const double*)ctx->bmu.p : nullptr;
    const double* dnu = nu ? (const double*)ctx->bnu.p : nullptr;
    const double* dY = (Y || series_on_device) ? (const double*)ctx->bY.p : nullptr;
    const double* dS2 = (Y || series_on_device) ? (const double*)ctx->bS2.p : nullptr;
    if (cd_shared)
        rc = batch_dev_impl(ds, ds->host, B, (const double*)ctx->bA.p, (const double*)ctx->bB.p, dmu, dnu, dY,
                                            dS2, (double*)ctx->bout.p, (int32_t*)ctx->bst.p);
    else
        rc = pioran_celerite_logl_batch_dev_cd(ds, B, J, (const double*)ctx->bA.p, (const double*)ctx->bB.p,
                                               (const double*)ctx->bC.p, (const double*)ctx->bD.p, dmu, dnu, dY, dS2,
                                               (double*)ctx->bout.p, (int32_t*)ctx->bst.p);
    if (rc) return rc;
    if ((rc = download(ctx, out, ctx->bout.p, B * sizeof(double)))) return rc;
    if (status)
        if ((rc = download(ctx, status, ctx->bst.p, B * sizeof(int32_t)))) return rc;
    SYNC(ctx);
    return PIORAN_OK;
}

int pioran_celerite_logl_batch(pioran_ds* ds, int64_t B, int64_t J, const double* A, const double* Bc,
                               const double* C, const double* Dd, int cd_shared, const double* mu, const double* nu,
                               const double* Y, const double* S2, double* out, int32_t* status)
{
    return batch_host_impl(ds, B, J, A, Bc, C, Dd, cd_shared, mu, nu, Y, S2, false, out, status);
}

static int batch_shift_dev_impl(pioran_ds* ds, const PrepState& s, int64_t B, const double* dA, const double* dBc, const double* dmu,
                                const double* dnu, const double* dshift, double* dout, int32_t* dstatus)
{
    if (!ds || B < 1 || !dA || !dBc || !dshift || !dout) return PIORAN_ERR_ARG;
    if (!s.prepared) return PIORAN_ERR_ARG;
    pioran_ctx* ctx = ds->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t bn = (size_t)B * (size_t)ds->N * sizeof(double);
    int rc;
    if ((rc = ensure(ctx, ctx->bY, bn))) return rc;
    if ((rc = ensure(ctx, ctx->bS2, bn))) return rc;
    rc = pioran_launch_shift_transform(ds->N, B, ds->y, ds->s2, dshift, (double*)ctx->bY.p, (double*)ctx->bS2.p, ctx->stream);
    if (rc) return rc;
    return batch_dev_impl(ds, s, B, dA, dBc, dmu, dnu, (const double*)ctx->bY.p, (const double*)ctx->bS2.p,
                                          dout, dstatus);
}

int pioran_celerite_logl_batch_shift_dev(pioran_ds* ds, int64_t B, const double* dA, const double* dBc, const double* dmu,
                                         const double* dnu, const double* dshift, double* dout, int32_t* dstatus)
{
    if (!ds) return PIORAN_ERR_ARG;
    return batch_shift_dev_impl(ds, ds->user, B, dA, dBc, dmu, dnu, dshift, dout, dstatus);
}

int pioran_celerite_logl_batch_shift(pioran_ds* ds, int64_t B, int64_t J, const double* A, const double* Bc,
                                     const double* C, const double* Dd, int cd_shared, const double* mu, const double* nu,
                                     const double* shift, double* out, int32_t* status)
{
    if (!ds || B < 1 || J < 1 || !A || !Bc || !C || !Dd || !shift || !out) return PIORAN_ERR_ARG;
    pioran_ctx* ctx = ds->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    const size_t bn = (size_t)B * (size_t)ds->N * sizeof(double);
    if ((rc = ensure(ctx, ctx->bY, bn))) return rc;
    if ((rc = ensure(ctx, ctx->bS2, bn))) return rc;
    if ((rc = upload(ctx, ctx->bshift, shift, B * sizeof(double)))) return rc;
    rc = pioran_launch_shift_transform(ds->N, B, ds->y, ds->s2, (const double*)ctx->bshift.p, (double*)ctx->bY.p,
                                       (double*)ctx->bS2.p, ctx->stream);
    if (rc) return rc;
    return batch_host_impl(ds, B, J, A, Bc, C, Dd, cd_shared, mu, nu, nullptr, nullptr, /*series_on_device=*/true, out,
                           status);
}

int pioran_logpdf_batch_theta(pioran_ds* ds, int64_t B, int model, int64_t n_components, int basis, int is_integrated_power,
                              double f_min, double f_max, double S_low, double S_high, const double* theta,
                              const double* norm, const double* mu, const double* nu, const double* shift, int64_t n_qpo,
                              const double* qpo, double* out, int32_t* status, double* A_out, double* Bc_out)
{
    if (!ds || B < 1 || !theta || !norm || !out || model < 0 || model > 1 || basis < 0 || basis > 1) return PIORAN_ERR_ARG;
    if (n_qpo < 0 || n_qpo > 8 || (n_qpo > 0 && !qpo)) return PIORAN_ERR_ARG;
    pioran_ctx* ctx = ds->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PendingGuard pending_guard(ctx);
    const int P = model == 0 ? 3 : 5;
    const int64_t J = n_components, Jc = basis == 0 ? J : 2 * J, Jt = Jc + n_qpo;
    std::vector<double> sp, LU, c, d;
    std::vector<int32_t> piv, real;
    int rc = pioran_approx_setup_host(J, basis, f_min, f_max, S_low, S_high, sp, LU, piv, c, d, real);
    if (rc) return rc;
    // staging: [sp J | LU J*J | piv (as int32, J)] in bwork; theta in bC / norm in bD without QPO features (both free when
    // (c, d) are shared); with features bC / bD receive the per-draw (c, d) of the feature terms and theta, norm, qpo ride in bwork
    const size_t nd = (size_t)J + (size_t)J * J + (size_t)J;
    const size_t nextra = n_qpo ? (size_t)B * (P + 1 + 3 * n_qpo) : 0;
    if ((rc = ensure(ctx, ctx->bwork, (nd + nextra) * sizeof(double)))) return rc;
    double* dsp = (double*)ctx->bwork.p;
    double* dLU = dsp + J;
    int32_t* dpiv = (int32_t*)(dLU + J * J);
    HIPCHK(ctx, hipMemcpyAsync(dsp, sp.data(), J * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dLU, LU.data(), J * J * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dpiv, piv.data(), J * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    const size_t bj = (size_t)B * (size_t)Jt * sizeof(double);
    if ((rc = ensure(ctx, ctx->bA, bj))) return rc;
    if ((rc = ensure(ctx, ctx->bB, bj))) return rc;
    const double *dtheta, *dnorm, *dqpo = nullptr;
    double *dCq = nullptr, *dDq = nullptr;
    if (n_qpo) {
        double* ex = dsp + nd;
        HIPCHK(ctx, hipMemcpyAsync(ex, theta, (size_t)B * P * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(ex + (size_t)B * P, norm, (size_t)B * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(ex + (size_t)B * (P + 1), qpo, (size_t)B * 3 * n_qpo * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        dtheta = ex; dnorm = ex + (size_t)B * P; dqpo = ex + (size_t)B * (P + 1);
        if ((rc = ensure(ctx, ctx->bC, bj))) return rc;
        if ((rc = ensure(ctx, ctx->bD, bj))) return rc;
        dCq = (double*)ctx->bC.p; dDq = (double*)ctx->bD.p;
    } else {
        if ((rc = upload(ctx, ctx->bC, theta, (size_t)B * P * sizeof(double)))) return rc;
        if ((rc = upload(ctx, ctx->bD, norm, (size_t)B * sizeof(double)))) return rc;
        dtheta = (const double*)ctx->bC.p; dnorm = (const double*)ctx->bD.p;
    }
    rc = pioran_launch_approx(B, model, P, (int)J, basis, is_integrated_power, f_min, f_max, dsp, dLU, dpiv, dtheta, dnorm,
                              (int)n_qpo, dqpo, (double*)ctx->bA.p, (double*)ctx->bB.p, dCq, dDq, ctx->stream);
    if (rc) return rc;
    SYNC(ctx);   // sp/LU/piv host vectors go out of scope below
    if (mu && (rc = upload(ctx, ctx->bmu, mu, B * sizeof(double)))) return rc;
    if (nu && (rc = upload(ctx, ctx->bnu, nu, B * sizeof(double)))) return rc;
    const double* dmu = mu ? (const double*)ctx->bmu.p : nullptr;
    const double* dnu = nu ? (const double*)ctx->bnu.p : nullptr;
    const size_t bn = (size_t)B * (size_t)ds->N * sizeof(double);
    if (shift) {
        if ((rc = upload(ctx, ctx->bshift, shift, B * sizeof(double)))) return rc;
        if (n_qpo) {   // transformed series of the whole batch, chunks of the mixed core read their slices
            if ((rc = ensure(ctx, ctx->bY, bn))) return rc;
            if ((rc = ensure(ctx, ctx->bS2, bn))) return rc;
            if ((rc = pioran_launch_shift_transform(ds->N, B, ds->y, ds->s2, (const double*)ctx->bshift.p, (double*)ctx->bY.p,
                                                    (double*)ctx->bS2.p, ctx->stream))) return rc;
        }
    }
    if (n_qpo) {
        // continuum terms from the shared table, the feature terms (per-draw c, d) from the per-draw block: mixed mode
        std::vector<int32_t> kind((size_t)Jt, 0);
        std::vector<double> c0((size_t)Jt, 0.0), d0((size_t)Jt, 0.0);
        for (int64_t j = 0; j < Jc; ++j) { kind[j] = real[j] ? 1 : 0; c0[j] = c[j]; d0[j] = d[j]; }
        for (int64_t q = 0; q < n_qpo; ++q) kind[Jc + q] = 2;
        auto fetch = [&](int64_t b0, int64_t nb, MixedChunk& m) -> int {
            (void)nb;
            m.A = (const double*)ctx->bA.p + b0 * Jt; m.Bc = (const double*)ctx->bB.p + b0 * Jt;
            m.C = dCq + b0 * Jt; m.D = dDq + b0 * Jt;
            m.mu = dmu ? dmu + b0 : nullptr; m.nu = dnu ? dnu + b0 : nullptr;
            if (shift) { m.Y = (const double*)ctx->bY.p + b0 * ds->N; m.S2 = (const double*)ctx->bS2.p + b0 * ds->N; }
            return PIORAN_OK;
        };
        rc = mixed_core(ds, B, Jt, kind, c0.data(), d0.data(), fetch, out, status, /*must_run=*/true);
        if (rc < 0) return rc;
        if (rc == 0) return PIORAN_ERR_UNSUPPORTED;   // too many rows for the register-resident kernels
    } else {
        if ((rc = prepare_state(ds, ds->host, Jc, c.data(), d.data(), real.data()))) return rc;
        if ((rc = ensure(ctx, ctx->bout, B * sizeof(double)))) return rc;
        if ((rc = ensure(ctx, ctx->bst, B * sizeof(int32_t)))) return rc;
        if (shift)
            rc = batch_shift_dev_impl(ds, ds->host, B, (const double*)ctx->bA.p, (const double*)ctx->bB.p, dmu, dnu,
                                      (const double*)ctx->bshift.p, (double*)ctx->bout.p, (int32_t*)ctx->bst.p);
        else
            rc = batch_dev_impl(ds, ds->host, B, (const double*)ctx->bA.p, (const double*)ctx->bB.p, dmu, dnu, nullptr,
                                nullptr, (double*)ctx->bout.p, (int32_t*)ctx->bst.p);
        if (rc) return rc;
        if ((rc = download(ctx, out, ctx->bout.p, B * sizeof(double)))) return rc;
        if (status) if ((rc = download(ctx, status, ctx->bst.p, B * sizeof(int32_t)))) return rc;
    }
    if (A_out) if ((rc = download(ctx, A_out, ctx->bA.p, bj))) return rc;
    if (Bc_out) if ((rc = download(ctx, Bc_out, ctx->bB.p, bj))) return rc;
    SYNC(ctx);
    return PIORAN_OK;
}

int pioran_celerite_logl(pioran_ctx* ctx, int64_t N, int64_t J, const double* a, const double* b, const double* c,
                         const double* d, const double* t, const double* y, const double* sigma2, double* out,
                         int32_t* status)
{
    if (!ctx || N < 1 || J < 1 || !a || !b || !c || !d || !t || !y || !sigma2 || !out) return PIORAN_ERR_ARG;
    // A sampler calls logl thousands of times with the same t (and usually the same c, d) and a fresh y - mu,
    // sigma2 * nu: keep the series handle (and through it the cos/sin/exp table) while t is unchanged, and pass
    // y, sigma2 as this call's per-draw series.
    int rc;
    const bool same_t = ctx->scalar_ds && (int64_t)ctx->scalar_t.size() == N &&
                        !std::memcmp(ctx->scalar_t.data(), t, (size_t)N * sizeof(double));
    if (!same_t) {
        if (ctx->scalar_ds) pioran_dataset_destroy(ctx->scalar_ds);
        ctx->scalar_ds = nullptr;
        ctx->scalar_t.clear();
        if ((rc = pioran_dataset_create(ctx, N, t, y, sigma2, &ctx->scalar_ds))) return rc;
        ctx->scalar_t.assign(t, t + N);
    }
    return pioran_celerite_logl_batch(ctx->scalar_ds, 1, J, a, b, c, d, 1, nullptr, nullptr, y, sigma2, out, status);
}

static int is_sorted(const double* t, int64_t N);   // (below, with the dense solver)

// ---- posterior mean / simulation --------------------------------------------------------------------
// shared (c, d) only; draws are processed in chunks of at most 256 (one workgroup per draw, factor kept in HBM)
static int predict_shared(pioran_ds* ds, int64_t B, int64_t J, const double* A, const double* Bc, const double* C,
                          const double* Dd, const double* mu, const double* nu, int64_t M, const double* tau,
                          double* mean_out, int32_t* status)
{
    if (!ds || B < 1 || J < 1 || M < 1 || !A || !Bc || !C || !Dd || !tau || !mean_out) return PIORAN_ERR_ARG;
    pioran_ctx* ctx = ds->ctx;
    PrepState& s = ds->host;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PendingGuard pending_guard(ctx);
    int rc;
    if ((rc = prepare_shared(ds, B, J, Bc, C, Dd))) return rc;
    if (s.R > pioran_predict_supported_rows() || s.npd_terms) return PIORAN_ERR_UNSUPPORTED;   // before any upload / workspace
    int64_t chunk = B < 256 ? B : 256;
    // Windowed path (round 3): z = K^-1 (y - mu) from the windowed factorisation and a block back-substitution (celerite_block.hip),
    // then the two Q recurrences segment-parallel (celerite_predict.hip) — no step-by-step factor, no per-step wave reduction
    bool windowed = !ctx->opt.no_block && !ctx->opt.force_fallback && !ctx->opt.scan_config[0] && s.R <= 63 &&
                    pioran_block_fits(s.R, s.J);
    if (windowed) {
        rc = ensure_btab(ds, s);
        if (rc == PIORAN_ERR_UNSUPPORTED) windowed = false;
        else if (rc) return rc;
    }
    if (windowed) {
        size_t free_b = 0, total_b = 0;
        auto need = [&](int64_t nb) {
            return (pioran_block_store_workspace_doubles(nb, ds->N, s.R, 2) + pioran_predict_q_workspace_doubles(nb, ds->N, s.R)) * sizeof(double);
        };
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            while (chunk > 1 && need(chunk) > ws_allow(ctx, free_b) + ctx->bwork.cap + ctx->bscratch.cap) chunk /= 2;
        rc = ensure(ctx, ctx->bwork, pioran_block_store_workspace_doubles(chunk, ds->N, s.R, 2) * sizeof(double));
        if (!rc) rc = ensure(ctx, ctx->bscratch, pioran_predict_q_workspace_doubles(chunk, ds->N, s.R) * sizeof(double));
        if (!rc) rc = ensure(ctx, ctx->bgtab, pioran_block_gtab_doubles(ds->N, s.R) * sizeof(double));
        if (!rc) rc = ensure(ctx, ctx->bK, pioran_predict_tau_workspace_doubles(M, s.R, 1) * sizeof(double));
        if (rc == PIORAN_ERR_ALLOC) { windowed = false; chunk = B < 256 ? B : 256; }
        else if (rc) return rc;
        if (windowed && (rc = pioran_launch_block_gtab(ds->N, s.R, s.J, s.rowmap, ds->t, s.dc, s.dd, ds->s2, (double*)ctx->bgtab.p, ctx->stream)))
            return rc;
    }
    if (!windowed && (rc = ensure(ctx, ctx->bwork, pioran_predict_workspace_doubles(chunk, ds->N, s.R) * sizeof(double)))) return rc;
    if ((rc = upload(ctx, ctx->bshift, tau, (size_t)M * sizeof(double)))) return rc;          // tau
    if ((rc = ensure(ctx, ctx->bY, (size_t)chunk * (size_t)M * sizeof(double)))) return rc;   // mean [chunk][M]
    if ((rc = ensure(ctx, ctx->bout, chunk * sizeof(double)))) return rc;
    if ((rc = ensure(ctx, ctx->bst, chunk * sizeof(int32_t)))) return rc;
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        const int64_t nb = B - b0 < chunk ? B - b0 : chunk;
        if ((rc = upload(ctx, ctx->bA, A + b0 * J, (size_t)nb * J * sizeof(double)))) return rc;
        if ((rc = upload(ctx, ctx->bB, Bc + b0 * J, (size_t)nb * J * sizeof(double)))) return rc;
        if (mu && (rc = upload(ctx, ctx->bmu, mu + b0, nb * sizeof(double)))) return rc;
        if (nu && (rc = upload(ctx, ctx->bnu, nu + b0, nb * sizeof(double)))) return rc;
        ScanParams p{};
        p.N = ds->N; p.J = s.J; p.R = s.R; p.B = nb;
        p.standard_rows = s.row_layout; p.n_complex = s.n_complex;
        p.rec_stride = 3 * (int64_t)(s.R + 2) + 2;
        p.tab = s.tab; p.rowmap = s.rowmap; p.t = ds->t; p.y = ds->y; p.s2 = ds->s2;
        p.A = (const double*)ctx->bA.p; p.Bc = (const double*)ctx->bB.p; p.C = s.dc; p.D = s.dd;
        p.mu = mu ? (const double*)ctx->bmu.p : nullptr; p.nu = nu ? (const double*)ctx->bnu.p : nullptr;
        p.out = (double*)ctx->bout.p; p.status = (int32_t*)ctx->bst.p;
        p.opt = &ctx->opt;
        if (windowed) {
            p.gw = (double*)ctx->bwork.p;
            g_last_kernel = "block (windowed prediction)";
            // -z = -K^-1 (y - mu) [nb][N] into the head of the Q workspace
            rc = pioran_launch_block_solve(p, s.btab, (const double*)ctx->bgtab.p, (double*)ctx->bscratch.p, ctx->stream);
            if (!rc) rc = pioran_launch_predict_from_gy(p, (double*)ctx->bscratch.p, (double*)ctx->bK.p, ds->t, M, (const double*)ctx->bshift.p, (double*)ctx->bY.p,
                                                        ctx->stream, 0, is_sorted(tau, M));
        } else {
            g_last_kernel = "wide (step-by-step prediction)";
            rc = pioran_launch_predict(p, (double*)ctx->bwork.p, ds->t, M, (const double*)ctx->bshift.p, (double*)ctx->bY.p,
                                       ctx->stream);
        }
        if (rc) { ctx->last_err = "prediction launch failed"; return rc; }
        if ((rc = download(ctx, mean_out + b0 * M, ctx->bY.p, (size_t)nb * M * sizeof(double)))) return rc;
        if (status) if ((rc = download(ctx, status + b0, ctx->bst.p, nb * sizeof(int32_t)))) return rc;
        SYNC(ctx);
    }
    return PIORAN_OK;
}

// Per-draw (c, d) in every term (posterior draws of QPO / CARMA / free Celerite models), several draws: every draw its own windowed-kernel
// tables, all draws of a chunk in one launch of every kernel (as logl_grad_perdraw_windowed below).  PIORAN_ERR_UNSUPPORTED when the
// shape does not fit the windowed kernel (the caller then goes draw by draw).
static int predict_perdraw_windowed(pioran_ds* ds, int64_t B, int64_t J, const double* A, const double* Bc, const double* C, const double* Dd,
                                    const double* mu, const double* nu, int64_t M, const double* tau, double* mean_out, int32_t* status)
{
    pioran_ctx* ctx = ds->ctx;
    PrepState& s = ds->host;
    if (ctx->opt.no_block || ctx->opt.force_fallback || ctx->opt.scan_config[0] || !pioran_block_fits((int32_t)(2 * J), (int32_t)J))
        return PIORAN_ERR_UNSUPPORTED;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PendingGuard pending_guard(ctx);
    int rc;
    if ((rc = prepare_state(ds, s, J, C, Dd, nullptr))) return rc;    // row map with both rows of every term
    const int64_t N = ds->N;
    const int64_t bt = (int64_t)pioran_block_table_doubles(N, s.R, s.J), gt = (int64_t)pioran_block_gtab_doubles(N, s.R);
    auto per_chunk = [&](int64_t nb) {
        return ((size_t)nb * ((size_t)bt + (size_t)gt) + pioran_block_store_workspace_doubles(nb, N, s.R, 2) + pioran_predict_q_workspace_doubles(nb, N, s.R) +
                pioran_predict_tau_workspace_doubles(M, s.R, nb) + (size_t)nb * (size_t)M) * sizeof(double);
    };
    int64_t chunk = B < 256 ? B : 256;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            while (chunk > 1 && per_chunk(chunk) > ws_allow(ctx, free_b) + ctx->bwork.cap + ctx->bscratch.cap + ctx->bgtab.cap + ctx->bK.cap + ctx->bq.cap) chunk /= 2;
    }
    for (;;) {
        rc = ensure(ctx, ctx->bwork, pioran_block_store_workspace_doubles(chunk, N, s.R, 2) * sizeof(double));
        if (!rc) rc = ensure(ctx, ctx->bscratch, (size_t)chunk * (size_t)bt * sizeof(double));
        if (!rc) rc = ensure(ctx, ctx->bgtab, (size_t)chunk * (size_t)gt * sizeof(double));
        if (!rc) rc = ensure(ctx, ctx->bq, pioran_predict_q_workspace_doubles(chunk, N, s.R) * sizeof(double));
        if (!rc) rc = ensure(ctx, ctx->bK, pioran_predict_tau_workspace_doubles(M, s.R, chunk) * sizeof(double));
        if (!rc) rc = ensure(ctx, ctx->bY, (size_t)chunk * (size_t)M * sizeof(double));
        if (rc != PIORAN_ERR_ALLOC || chunk == 1) break;
        chunk /= 2;
    }
    if (rc) return rc;
    if ((rc = upload(ctx, ctx->bshift, tau, (size_t)M * sizeof(double)))) return rc;
    if ((rc = ensure(ctx, ctx->bout, chunk * sizeof(double)))) return rc;
    if ((rc = ensure(ctx, ctx->bst, chunk * sizeof(int32_t)))) return rc;
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        const int64_t nb = B - b0 < chunk ? B - b0 : chunk;
        const size_t nbj = (size_t)nb * J * sizeof(double);
        if ((rc = upload(ctx, ctx->bA, A + b0 * J, nbj))) return rc;
        if ((rc = upload(ctx, ctx->bB, Bc + b0 * J, nbj))) return rc;
        if ((rc = upload(ctx, ctx->bC, C + b0 * J, nbj))) return rc;
        if ((rc = upload(ctx, ctx->bD, Dd + b0 * J, nbj))) return rc;
        if (mu && (rc = upload(ctx, ctx->bmu, mu + b0, nb * sizeof(double)))) return rc;
        if (nu && (rc = upload(ctx, ctx->bnu, nu + b0, nb * sizeof(double)))) return rc;
        double* btab = (double*)ctx->bscratch.p; double* gtab = (double*)ctx->bgtab.p;
        if ((rc = pioran_launch_block_table_batch(N, s.R, s.J, nb, s.rowmap, ds->t, (const double*)ctx->bC.p, (const double*)ctx->bD.p, ds->y, ds->s2,
                                                  btab, bt, ctx->stream))) return rc;
        if ((rc = pioran_launch_block_gtab_batch(N, s.R, s.J, nb, s.rowmap, ds->t, (const double*)ctx->bC.p, (const double*)ctx->bD.p, ds->s2, gtab,
                                                 gt, ctx->stream))) return rc;
        ScanParams p{};
        p.opt = &ctx->opt;
        p.N = N; p.J = s.J; p.R = s.R; p.B = nb; p.standard_rows = 1;
        p.rec_stride = 3 * (int64_t)(s.R + 2) + 2;
        p.tab_draw_stride = bt; p.gtab_draw_stride = gt;
        p.rowmap = s.rowmap; p.t = ds->t; p.y = ds->y; p.s2 = ds->s2;
        p.A = (const double*)ctx->bA.p; p.Bc = (const double*)ctx->bB.p; p.C = (const double*)ctx->bC.p; p.D = (const double*)ctx->bD.p;
        p.mu = mu ? (const double*)ctx->bmu.p : nullptr; p.nu = nu ? (const double*)ctx->bnu.p : nullptr;
        p.out = (double*)ctx->bout.p; p.status = (int32_t*)ctx->bst.p;
        p.gw = (double*)ctx->bwork.p;
        g_last_kernel = "block (windowed prediction, per-draw tables)";
        rc = pioran_launch_block_solve(p, btab, gtab, (double*)ctx->bq.p, ctx->stream);
        if (!rc) rc = pioran_launch_predict_from_gy(p, (double*)ctx->bq.p, (double*)ctx->bK.p, ds->t, M, (const double*)ctx->bshift.p,
                                                    (double*)ctx->bY.p, ctx->stream, 1, is_sorted(tau, M));
        if (rc) { ctx->last_err = "windowed prediction launch failed"; return rc; }
        if ((rc = download(ctx, mean_out + b0 * M, ctx->bY.p, (size_t)nb * M * sizeof(double)))) return rc;
        if (status) if ((rc = download(ctx, status + b0, ctx->bst.p, nb * sizeof(int32_t)))) return rc;
        SYNC(ctx);
    }
    return PIORAN_OK;
}

int pioran_celerite_predict(pioran_ds* ds, int64_t B, int64_t J, const double* A, const double* Bc, const double* C,
                            const double* Dd, int cd_shared, const double* mu, const double* nu, int64_t M, const double* tau,
                            double* mean_out, int32_t* status)
{
    if (!ds || B < 1 || J < 1 || M < 1 || !A || !Bc || !C || !Dd || !tau || !mean_out) return PIORAN_ERR_ARG;
    if (cd_shared || B == 1) return predict_shared(ds, B, J, A, Bc, C, Dd, mu, nu, M, tau, mean_out, status);
    {
        const int rc = predict_perdraw_windowed(ds, B, J, A, Bc, C, Dd, mu, nu, M, tau, mean_out, status);
        if (rc != PIORAN_ERR_UNSUPPORTED) return rc;
    }
    for (int64_t b = 0; b < B; ++b) {   // per-draw (c, d): every draw is its own one-draw batch with its own table
        const int rc = predict_shared(ds, 1, J, A + b * J, Bc + b * J, C + b * J, Dd + b * J, mu ? mu + b : nullptr, nu ? nu + b : nullptr,
                                      M, tau, mean_out + b * M, status ? status + b : nullptr);
        if (rc) return rc;
    }
    return PIORAN_OK;
}

// shift / grad_shift != nullptr: the shifted log-flux models (the data set holds raw flux and yerr^2); grad_y / grad_sigma2
// then refer to the TRANSFORMED series of each draw.  Shared (c, d) [J] only; per-draw (c, d) are looped over by the callers below.
static int logl_grad_shared(pioran_ds* ds, int64_t B, int64_t J, const double* A, const double* Bc, const double* C,
                            const double* Dd, const double* mu, const double* nu, const double* shift, double* out,
                            int32_t* status, double* grad_a, double* grad_b, double* grad_c, double* grad_d, double* grad_nu,
                            double* grad_mu, double* grad_y, double* grad_sigma2, double* grad_shift)
{
    pioran_ctx* ctx = ds->ctx;
    PrepState& s = ds->host;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PendingGuard pending_guard(ctx);
    int rc;
    if ((rc = prepare_shared(ds, B, J, Bc, C, Dd))) return rc;
    if (s.R > pioran_wide_supported_rows_grad() || s.npd_terms) return PIORAN_ERR_UNSUPPORTED;
    // Windowed reverse mode (celerite_block.hip, round 3) whenever the rows fit the windowed kernel, with or without d/d(c, d):
    // 6.3 ms (7.0 with d/d(c, d)) instead of 25 at N = 1e4, J = 20 (series gradients and the shifted log-flux models included).
    bool windowed = !ctx->opt.no_block && !ctx->opt.force_fallback && !ctx->opt.scan_config[0] &&
                    pioran_block_fits(s.R, s.J);
    if (windowed) {
        rc = ensure_btab(ds, s);
        if (rc == PIORAN_ERR_UNSUPPORTED) windowed = false;
        else if (rc) return rc;
    }
    // Many chains (round 5): the one-draw-per-wavefront reverse mode (celerite_tile.hip) — value and d/d(a, b, mu, nu), since round 6 also d/d(c, d) of the
    // SHARED (c, d) (both or neither), shared series.  Its
    // forward pass keeps the lower tiles of T per window (12 KB at three block columns) and the reverse kernel recomputes the rest, where the
    // small-batch kernels keep 41 KB per window and chain and hold one chain per CU.  scan_config = "tile" forces it for any chain count.
    // From 513 chains on (measured, SHO-20 / SHO-12 at N = 1e4: 512 chains 15.1 / 9.3 ms against the small-batch kernels' 11.0 / 9.2; 640 chains 15.2 /
    // 9.4 against 16.6 / 14.0; 2048 chains 27 / 16 against 44 / 36).
    const bool tilegrad = windowed && (grad_c != nullptr) == (grad_d != nullptr) && !grad_y && !grad_sigma2 && !shift && s.R <= pioran_tile_grad_supported_rows() &&
                          (ctx->opt.force_tile || (!ctx->opt.no_tile && B > 512 && s.R >= 17));
    // (48 .. 63 rows — DRWCelerite-20 is 60: three draws per workgroup there (two with d/d(c, d)); 4096 chains take 126 ms (163 with d/d(c, d)) against 166 (175)
    //  in 512-chain launches of the small-batch kernels: tools/ab_tile_grad_nb4.py, profiles/r06_tile_grad_four_block_columns.txt.  Until the reverse kernel
    //  stopped spilling at four block columns — T_k and the window's U operands loaded at the head of their own window instead of a window ahead — it was 177 (208).)
    auto ws_doubles = [&](int64_t nb) {
        return tilegrad ? pioran_tile_grad_workspace_doubles(nb, ds->N, s.R)
                        : (windowed ? pioran_block_grad_workspace_doubles(nb, ds->N, s.R) : pioran_grad_workspace_doubles(nb, ds->N, s.R));
    };
    // Workspace per draw: (m, D) of every step + S at the checkpoints + two replayed segments (celerite_wide.hip): ~15 MB at
    // N = 1e4, R = 40.  The chunk is bounded by half of the memory that is free right now (plus what this buffer already
    // holds) and halved again if the allocation still fails.
    int64_t chunk = B < 1024 ? B : 1024;
    // windowed: 512 chains per launch pair (the forward pass then runs two workgroups per CU, the reverse pass two rounds of one)
    if (windowed && chunk > 512) chunk = 512;
    if (tilegrad) chunk = B < 4096 ? B : 4096;      // whole passes of 2048 (1024) chains: 7.7 GB of T per 1024 chains at N = 1e4, three block columns
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            // (tile: the limit holds for the buffer as a whole — "limit + what it holds" let the second call grow a 16 GB buffer to 31 GB)
            const size_t allowed = tilegrad ? ws_allow(ctx, free_b + ctx->bwork.cap) : ws_allow(ctx, free_b) + ctx->bwork.cap;
            while (chunk > 1 && ws_doubles(chunk) * sizeof(double) > allowed) chunk = tilegrad && chunk > 1024 ? chunk - 1024 : chunk / 2;
        }
    }
    if (tilegrad && (rc = ensure(ctx, ctx->bpair, pioran_tile_workspace_doubles(chunk, ds->N) * sizeof(double)))) return rc;
    while ((rc = ensure(ctx, ctx->bwork, ws_doubles(chunk) * sizeof(double))) == PIORAN_ERR_ALLOC && chunk > 1)
        chunk /= 2;
    if (rc) return rc;
    double* gtab = nullptr;
    if (windowed) {   // the reverse pass's table (C o v, C o x in C/D order, C_K, sigma2): 13 KB per window, rebuilt per call (10 us)
        if ((rc = ensure(ctx, ctx->bgtab, pioran_block_gtab_doubles(ds->N, s.R) * sizeof(double)))) return rc;
        gtab = (double*)ctx->bgtab.p;
        if ((rc = pioran_launch_block_gtab(ds->N, s.R, s.J, s.rowmap, ds->t, s.dc, s.dd, ds->s2, gtab, ctx->stream))) return rc;
    }
    const size_t cj = (size_t)chunk * (size_t)J * sizeof(double), cn = (size_t)chunk * (size_t)ds->N * sizeof(double);
    if ((rc = ensure(ctx, ctx->bC, 4 * cj))) return rc;              // grad_a | grad_b | grad_c | grad_d
    if ((rc = ensure(ctx, ctx->bD, 2 * chunk * sizeof(double)))) return rc;   // grad_nu | grad_mu
    const bool want_series = grad_y || grad_sigma2 || shift;   // the shift's chain rule needs both series gradients
    if (want_series && (rc = ensure(ctx, ctx->bY, cn))) return rc;
    if (want_series && (rc = ensure(ctx, ctx->bS2, cn))) return rc;
    if (shift && (rc = ensure(ctx, ctx->bscratch, 2 * cn))) return rc;           // transformed Y | S2 of the chunk
    if (shift && (rc = ensure(ctx, ctx->bshift, 2 * chunk * sizeof(double)))) return rc;   // shift | grad_shift
    if ((rc = ensure(ctx, ctx->bout, chunk * sizeof(double)))) return rc;
    if ((rc = ensure(ctx, ctx->bst, chunk * sizeof(int32_t)))) return rc;
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        const int64_t nb = B - b0 < chunk ? B - b0 : chunk;
        if ((rc = upload(ctx, ctx->bA, A + b0 * J, (size_t)nb * J * sizeof(double)))) return rc;
        if ((rc = upload(ctx, ctx->bB, Bc + b0 * J, (size_t)nb * J * sizeof(double)))) return rc;
        if (mu && (rc = upload(ctx, ctx->bmu, mu + b0, nb * sizeof(double)))) return rc;
        if (nu && (rc = upload(ctx, ctx->bnu, nu + b0, nb * sizeof(double)))) return rc;
        ScanParams p{};
        p.N = ds->N; p.J = s.J; p.R = s.R; p.B = nb;
        p.standard_rows = s.row_layout; p.n_complex = s.n_complex;
        p.rec_stride = 3 * (int64_t)(s.R + 2) + 2;
        p.tab = s.tab; p.rowmap = s.rowmap; p.t = ds->t; p.y = ds->y; p.s2 = ds->s2;
        p.A = (const double*)ctx->bA.p; p.Bc = (const double*)ctx->bB.p; p.C = s.dc; p.D = s.dd;
        p.mu = mu ? (const double*)ctx->bmu.p : nullptr; p.nu = nu ? (const double*)ctx->bnu.p : nullptr;
        p.out = (double*)ctx->bout.p; p.status = (int32_t*)ctx->bst.p;
        p.g_y = want_series ? (double*)ctx->bY.p : nullptr;
        p.g_s2 = want_series ? (double*)ctx->bS2.p : nullptr;
        double* dshift = (double*)ctx->bshift.p;
        if (shift) {
            HIPCHK(ctx, hipMemcpyAsync(dshift, shift + b0, nb * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            double* dYt = (double*)ctx->bscratch.p; double* dSt = dYt + (size_t)chunk * (size_t)ds->N;
            if ((rc = pioran_launch_shift_transform(ds->N, nb, ds->y, ds->s2, dshift, dYt, dSt, ctx->stream))) return rc;
            p.Y = dYt; p.S2 = dSt;
        }
        double* dga = (double*)ctx->bC.p; double* dgb = dga + (size_t)chunk * J;
        double* dgc = dgb + (size_t)chunk * J; double* dgd = dgc + (size_t)chunk * J;
        double* dgn = (double*)ctx->bD.p; double* dgm = dgn + chunk;
        if (!ctx->aux) {
            HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking));
            for (auto& e : ctx->gev) HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        p.opt = &ctx->opt;
        if (tilegrad) {
            p.gw = (double*)ctx->bwork.p;
            g_last_kernel = "tile (windowed gradient, one draw per wavefront)";
            rc = pioran_launch_tile_grad(p, s.btab, gtab, (double*)ctx->bpair.p, dga, dgb, dgn, dgm, grad_c ? dgc : nullptr, grad_d ? dgd : nullptr, ctx->stream);
        } else if (windowed) {
            p.gw = (double*)ctx->bwork.p;
            g_last_kernel = "block (windowed gradient)";
            rc = pioran_launch_block_grad(p, s.btab, gtab, dga, dgb, dgn, dgm, grad_c ? dgc : nullptr, grad_d ? dgd : nullptr, ctx->stream);
        } else {
            g_last_kernel = "wide (step-by-step gradient)";
            rc = pioran_launch_scan_wide_grad(p, (double*)ctx->bwork.p, dga, dgb, grad_c ? dgc : nullptr, grad_d ? dgd : nullptr, dgn, dgm,
                                              ctx->stream, ctx->aux, ctx->gev);
        }
        if (rc) { ctx->last_err = "gradient launch failed"; return rc; }
        if (shift) {
            rc = pioran_launch_shift_grad(ds->N, nb, ds->y, ds->s2, dshift, p.g_y, p.g_s2, dshift + chunk, ctx->stream);
            if (rc) return rc;
            if ((rc = download(ctx, grad_shift + b0, dshift + chunk, nb * sizeof(double)))) return rc;
        }
        const size_t nbj = (size_t)nb * J * sizeof(double);
        if ((rc = download(ctx, out + b0, ctx->bout.p, nb * sizeof(double)))) return rc;
        if (status) if ((rc = download(ctx, status + b0, ctx->bst.p, nb * sizeof(int32_t)))) return rc;
        if ((rc = download(ctx, grad_a + b0 * J, dga, nbj))) return rc;
        if ((rc = download(ctx, grad_b + b0 * J, dgb, nbj))) return rc;
        if (grad_c) if ((rc = download(ctx, grad_c + b0 * J, dgc, nbj))) return rc;
        if (grad_d) if ((rc = download(ctx, grad_d + b0 * J, dgd, nbj))) return rc;
        if (grad_nu) if ((rc = download(ctx, grad_nu + b0, dgn, nb * sizeof(double)))) return rc;
        if (grad_mu) if ((rc = download(ctx, grad_mu + b0, dgm, nb * sizeof(double)))) return rc;
        if (grad_y) if ((rc = download(ctx, grad_y + b0 * ds->N, ctx->bY.p, (size_t)nb * ds->N * sizeof(double)))) return rc;
        if (grad_sigma2) if ((rc = download(ctx, grad_sigma2 + b0 * ds->N, ctx->bS2.p, (size_t)nb * ds->N * sizeof(double)))) return rc;
        // Chunks follow each other on the stream without a host synchronisation in between (round 4): every transfer is stream-ordered
        // and staged through pinned memory, which drains itself when it fills (pin_reserve); only the large series gradients, which go
        // straight to the caller's pageable memory, are waited for per chunk.  (No change in time: 4096 chains are 16 launches of 7.5 ms, 122 ms.)
        if (want_series) SYNC(ctx);
    }
    SYNC(ctx);
    return PIORAN_OK;
}

// Per-draw (c, d) in every term, several chains (CARMA kernels, QPO features, free Celerite sums under NUTS): the windowed reverse
// mode with one pair of tables per draw (the forward table of pioran_launch_block_table_batch and the reverse pass's), all chains in
// one launch — 16 chains in the time of one instead of 16 one-draw calls.  Returns PIORAN_ERR_UNSUPPORTED when the shape does not
// fit the windowed kernel (the caller then evaluates draw by draw).
static int logl_grad_perdraw_windowed(pioran_ds* ds, int64_t B, int64_t J, const double* A, const double* Bc, const double* C,
                                      const double* Dd, const double* mu, const double* nu, double* out, int32_t* status,
                                      double* grad_a, double* grad_b, double* grad_c, double* grad_d, double* grad_nu, double* grad_mu,
                                      double* grad_y, double* grad_sigma2)
{
    pioran_ctx* ctx = ds->ctx;
    PrepState& s = ds->host;
    if (ctx->opt.no_block || ctx->opt.force_fallback || ctx->opt.scan_config[0] || !pioran_block_fits((int32_t)(2 * J), (int32_t)J))
        return PIORAN_ERR_UNSUPPORTED;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PendingGuard pending_guard(ctx);
    int rc;
    if ((rc = prepare_state(ds, s, J, C, Dd, nullptr))) return rc;    // row map with both rows of every term (tables of draw 0: unused)
    const int64_t N = ds->N;
    const int64_t bt = (int64_t)pioran_block_table_doubles(N, s.R, s.J), gt = (int64_t)pioran_block_gtab_doubles(N, s.R);
    const size_t per_draw = ((size_t)bt + (size_t)gt + pioran_block_grad_workspace_doubles(1, N, s.R)) * sizeof(double);
    int64_t chunk = B < 256 ? B : 256;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            while (chunk > 1 && (size_t)chunk * per_draw > ws_allow(ctx, free_b) + ctx->bwork.cap + ctx->bscratch.cap + ctx->bgtab.cap) chunk /= 2;
    }
    for (;;) {
        rc = ensure(ctx, ctx->bwork, pioran_block_grad_workspace_doubles(chunk, N, s.R) * sizeof(double));
        if (!rc) rc = ensure(ctx, ctx->bscratch, (size_t)chunk * (size_t)bt * sizeof(double));
        if (!rc) rc = ensure(ctx, ctx->bgtab, (size_t)chunk * (size_t)gt * sizeof(double));
        if (rc != PIORAN_ERR_ALLOC || chunk == 1) break;
        chunk /= 2;
    }
    if (rc) return rc;
    const size_t cj = (size_t)chunk * (size_t)J * sizeof(double), cn = (size_t)chunk * (size_t)N * sizeof(double);
    if ((rc = ensure(ctx, ctx->bK, 4 * cj))) return rc;                   // grad_a | grad_b | grad_c | grad_d of the chunk
    if ((rc = ensure(ctx, ctx->bshift, 2 * chunk * sizeof(double)))) return rc;   // grad_nu | grad_mu
    const bool want_series = grad_y || grad_sigma2;
    if (want_series && (rc = ensure(ctx, ctx->bY, cn))) return rc;
    if (want_series && (rc = ensure(ctx, ctx->bS2, cn))) return rc;
    if ((rc = ensure(ctx, ctx->bout, chunk * sizeof(double)))) return rc;
    if ((rc = ensure(ctx, ctx->bst, chunk * sizeof(int32_t)))) return rc;
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        const int64_t nb = B - b0 < chunk ? B - b0 : chunk;
        const size_t nbj = (size_t)nb * J * sizeof(double);
        if ((rc = upload(ctx, ctx->bA, A + b0 * J, nbj))) return rc;
        if ((rc = upload(ctx, ctx->bB, Bc + b0 * J, nbj))) return rc;
        if ((rc = upload(ctx, ctx->bC, C + b0 * J, nbj))) return rc;
        if ((rc = upload(ctx, ctx->bD, Dd + b0 * J, nbj))) return rc;
        if (mu && (rc = upload(ctx, ctx->bmu, mu + b0, nb * sizeof(double)))) return rc;
        if (nu && (rc = upload(ctx, ctx->bnu, nu + b0, nb * sizeof(double)))) return rc;
        double* btab = (double*)ctx->bscratch.p; double* gtab = (double*)ctx->bgtab.p;
        if ((rc = pioran_launch_block_table_batch(N, s.R, s.J, nb, s.rowmap, ds->t, (const double*)ctx->bC.p, (const double*)ctx->bD.p, ds->y, ds->s2,
                                                  btab, bt, ctx->stream))) return rc;
        if ((rc = pioran_launch_block_gtab_batch(N, s.R, s.J, nb, s.rowmap, ds->t, (const double*)ctx->bC.p, (const double*)ctx->bD.p, ds->s2, gtab,
                                                 gt, ctx->stream))) return rc;
        ScanParams p{};
        p.opt = &ctx->opt;
        p.N = N; p.J = s.J; p.R = s.R; p.B = nb; p.standard_rows = 1;
        p.rec_stride = 3 * (int64_t)(s.R + 2) + 2;
        p.tab_draw_stride = bt; p.gtab_draw_stride = gt;
        p.rowmap = s.rowmap; p.t = ds->t; p.y = ds->y; p.s2 = ds->s2;
        p.A = (const double*)ctx->bA.p; p.Bc = (const double*)ctx->bB.p; p.C = (const double*)ctx->bC.p; p.D = (const double*)ctx->bD.p;
        p.mu = mu ? (const double*)ctx->bmu.p : nullptr; p.nu = nu ? (const double*)ctx->bnu.p : nullptr;
        p.out = (double*)ctx->bout.p; p.status = (int32_t*)ctx->bst.p;
        p.g_y = want_series ? (double*)ctx->bY.p : nullptr;
        p.g_s2 = want_series ? (double*)ctx->bS2.p : nullptr;
        p.gw = (double*)ctx->bwork.p;
        double* dga = (double*)ctx->bK.p; double* dgb = dga + (size_t)chunk * J;
        double* dgc = dgb + (size_t)chunk * J; double* dgd = dgc + (size_t)chunk * J;
        double* dgn = (double*)ctx->bshift.p; double* dgm = dgn + chunk;
        g_last_kernel = "block (windowed gradient, per-draw tables)";
        rc = pioran_launch_block_grad(p, btab, gtab, dga, dgb, dgn, dgm, grad_c ? dgc : nullptr, grad_d ? dgd : nullptr, ctx->stream);
        if (rc) { ctx->last_err = "windowed gradient launch failed"; return rc; }
        if ((rc = download(ctx, out + b0, ctx->bout.p, nb * sizeof(double)))) return rc;
        if (status) if ((rc = download(ctx, status + b0, ctx->bst.p, nb * sizeof(int32_t)))) return rc;
        if ((rc = download(ctx, grad_a + b0 * J, dga, nbj))) return rc;
        if ((rc = download(ctx, grad_b + b0 * J, dgb, nbj))) return rc;
        if (grad_c) if ((rc = download(ctx, grad_c + b0 * J, dgc, nbj))) return rc;
        if (grad_d) if ((rc = download(ctx, grad_d + b0 * J, dgd, nbj))) return rc;
        if (grad_nu) if ((rc = download(ctx, grad_nu + b0, dgn, nb * sizeof(double)))) return rc;
        if (grad_mu) if ((rc = download(ctx, grad_mu + b0, dgm, nb * sizeof(double)))) return rc;
        if (grad_y) if ((rc = download(ctx, grad_y + b0 * N, ctx->bY.p, (size_t)nb * N * sizeof(double)))) return rc;
        if (grad_sigma2) if ((rc = download(ctx, grad_sigma2 + b0 * N, ctx->bS2.p, (size_t)nb * N * sizeof(double)))) return rc;
        SYNC(ctx);
    }
    return PIORAN_OK;
}

static int logl_grad_impl(pioran_ds* ds, int64_t B, int64_t J, const double* A, const double* Bc, const double* C,
                          const double* Dd, int cd_shared, const double* mu, const double* nu, const double* shift, double* out,
                          int32_t* status, double* grad_a, double* grad_b, double* grad_c, double* grad_d, double* grad_nu,
                          double* grad_mu, double* grad_y, double* grad_sigma2, double* grad_shift)
{
    if (!ds || B < 1 || J < 1 || !A || !Bc || !C || !Dd || !out || !grad_a || !grad_b) return PIORAN_ERR_ARG;
    if ((shift == nullptr) != (grad_shift == nullptr)) return PIORAN_ERR_ARG;
    if (cd_shared || B == 1)
        return logl_grad_shared(ds, B, J, A, Bc, C, Dd, mu, nu, shift, out, status, grad_a, grad_b, grad_c, grad_d, grad_nu, grad_mu,
                                grad_y, grad_sigma2, grad_shift);
    // per-draw (c, d) — CARMA, QPO features, free Celerite sums under NUTS (a handful of chains): all chains in one launch of the
    // windowed reverse mode where the shape fits it, else every draw is its own one-draw batch with its own table
    if (!shift) {
        const int rc = logl_grad_perdraw_windowed(ds, B, J, A, Bc, C, Dd, mu, nu, out, status, grad_a, grad_b, grad_c, grad_d, grad_nu, grad_mu,
                                                  grad_y, grad_sigma2);
        if (rc != PIORAN_ERR_UNSUPPORTED) return rc;
    }
    const int64_t N = ds->N;
    for (int64_t b = 0; b < B; ++b) {
        const int rc = logl_grad_shared(ds, 1, J, A + b * J, Bc + b * J, C + b * J, Dd + b * J, mu ? mu + b : nullptr,
                                        nu ? nu + b : nullptr, shift ? shift + b : nullptr, out + b, status ? status + b : nullptr,
                                        grad_a + b * J, grad_b + b * J, grad_c ? grad_c + b * J : nullptr,
                                        grad_d ? grad_d + b * J : nullptr, grad_nu ? grad_nu + b : nullptr,
                                        grad_mu ? grad_mu + b : nullptr, grad_y ? grad_y + b * N : nullptr,
                                        grad_sigma2 ? grad_sigma2 + b * N : nullptr, grad_shift ? grad_shift + b : nullptr);
        if (rc) return rc;
    }
    return PIORAN_OK;
}

int pioran_celerite_logl_grad(pioran_ds* ds, int64_t B, int64_t J, const double* A, const double* Bc, const double* C,
                              const double* Dd, int cd_shared, const double* mu, const double* nu, double* out, int32_t* status,
                              double* grad_a, double* grad_b, double* grad_c, double* grad_d, double* grad_nu, double* grad_mu,
                              double* grad_y, double* grad_sigma2)
{
    return logl_grad_impl(ds, B, J, A, Bc, C, Dd, cd_shared, mu, nu, nullptr, out, status, grad_a, grad_b, grad_c, grad_d, grad_nu,
                          grad_mu, grad_y, grad_sigma2, nullptr);
}

int pioran_celerite_logl_grad_shift(pioran_ds* ds, int64_t B, int64_t J, const double* A, const double* Bc, const double* C,
                                    const double* Dd, int cd_shared, const double* mu, const double* nu, const double* shift,
                                    double* out, int32_t* status, double* grad_a, double* grad_b, double* grad_c, double* grad_d,
                                    double* grad_nu, double* grad_mu, double* grad_shift)
{
    if (!shift || !grad_shift) return PIORAN_ERR_ARG;
    return logl_grad_impl(ds, B, J, A, Bc, C, Dd, cd_shared, mu, nu, shift, out, status, grad_a, grad_b, grad_c, grad_d, grad_nu,
                          grad_mu, nullptr, nullptr, grad_shift);
}

static int simulate_shared(pioran_ctx* ctx, int64_t N, int64_t B, int64_t J, const double* A, const double* Bc,
                           const double* C, const double* Dd, const double* t, const double* sigma2, const double* q,
                           double* y_out)
{
    if (!ctx || N < 1 || B < 1 || J < 1 || !A || !Bc || !C || !Dd || !t || !sigma2 || !q || !y_out) return PIORAN_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    pioran_ds* ds = nullptr;
    std::vector<double> zeros((size_t)N, 0.0);
    int rc = pioran_dataset_create(ctx, N, t, zeros.data(), sigma2, &ds);
    if (rc) return rc;
    auto done = [&](int code) { pioran_dataset_destroy(ds); return code; };
    if ((rc = prepare_shared(ds, B, J, Bc, C, Dd))) return done(rc);
    PrepState& s = ds->host;
    if (s.R > pioran_wide_supported_rows_modes() || s.npd_terms) return done(PIORAN_ERR_UNSUPPORTED);   // before any upload / workspace
    int64_t chunk = B < 256 ? B : 256;
    // Windowed path (round 3; 6 .. 63 rows): the windowed factorisation with its per-window stores, then L applied window by window
    bool windowed = !ctx->opt.no_block && !ctx->opt.force_fallback && !ctx->opt.scan_config[0] && s.R <= 63 &&
                    pioran_block_fits(s.R, s.J);
    if (windowed) {
        rc = ensure_btab(ds, s);
        if (rc == PIORAN_ERR_UNSUPPORTED) windowed = false;
        else if (rc) return done(rc);
    }
    if (windowed) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            while (chunk > 1 && pioran_block_store_workspace_doubles(chunk, N, s.R, 3) * sizeof(double) > ws_allow(ctx, free_b) + ctx->bwork.cap) chunk /= 2;
        rc = ensure(ctx, ctx->bwork, pioran_block_store_workspace_doubles(chunk, N, s.R, 3) * sizeof(double));
        if (!rc) rc = ensure(ctx, ctx->bscratch, (size_t)chunk * (size_t)N * sizeof(double));   // xi
        if (!rc) rc = ensure(ctx, ctx->bst, chunk * sizeof(int32_t));
        if (rc == PIORAN_ERR_ALLOC) { windowed = false; chunk = B < 256 ? B : 256; }
        else if (rc) return done(rc);
    }
    const size_t cn = (size_t)chunk * (size_t)N * sizeof(double);
    if ((rc = ensure(ctx, ctx->bY, cn))) return done(rc);     // noise
    if ((rc = ensure(ctx, ctx->bS2, cn))) return done(rc);    // realisations
    if ((rc = ensure(ctx, ctx->bout, chunk * sizeof(double)))) return done(rc);
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        const int64_t nb = B - b0 < chunk ? B - b0 : chunk;
        if ((rc = upload(ctx, ctx->bA, A + b0 * J, (size_t)nb * J * sizeof(double)))) return done(rc);
        if ((rc = upload(ctx, ctx->bB, Bc + b0 * J, (size_t)nb * J * sizeof(double)))) return done(rc);
        if ((rc = upload(ctx, ctx->bY, q + b0 * N, (size_t)nb * N * sizeof(double)))) return done(rc);
        ScanParams p{};
        p.N = ds->N; p.J = s.J; p.R = s.R; p.B = nb;
        p.standard_rows = s.row_layout; p.n_complex = s.n_complex;
        p.rec_stride = 3 * (int64_t)(s.R + 2) + 2;
        p.tab = s.tab; p.rowmap = s.rowmap; p.t = ds->t; p.y = ds->y; p.s2 = ds->s2;
        p.A = (const double*)ctx->bA.p; p.Bc = (const double*)ctx->bB.p; p.C = s.dc; p.D = s.dd;
        p.out = (double*)ctx->bout.p;
        p.noise = (const double*)ctx->bY.p; p.ysim = (double*)ctx->bS2.p;
        p.opt = &ctx->opt;
        if (windowed) {
            p.gw = (double*)ctx->bwork.p;
            p.status = (int32_t*)ctx->bst.p;
            g_last_kernel = "block (windowed simulation)";
            rc = pioran_launch_block_sim(p, s.btab, (double*)ctx->bscratch.p, ctx->stream);
        } else {
            g_last_kernel = "wide (step-by-step simulation)";
            rc = pioran_launch_scan_wide_sim(p, ctx->stream);
        }
        if (rc) { ctx->last_err = "simulation launch failed"; return done(rc); }
        if (hipMemcpyAsync(y_out + b0 * N, ctx->bS2.p, (size_t)nb * N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            ctx_sync(ctx) != PIORAN_OK) {
            ctx->last_err = "simulation copy-back failed";
            return done(PIORAN_ERR_HIP);
        }
    }
    return done(PIORAN_OK);
}

// (c, d) per draw in every term, several draws: per-draw windowed tables, all draws of a chunk in one launch of every kernel
// (PIORAN_ERR_UNSUPPORTED: the shape does not fit the windowed kernel — the caller goes draw by draw)
static int simulate_perdraw_windowed(pioran_ctx* ctx, int64_t N, int64_t B, int64_t J, const double* A, const double* Bc, const double* C,
                                     const double* Dd, const double* t, const double* sigma2, const double* q, double* y_out)
{
    if (ctx->opt.no_block || ctx->opt.force_fallback || ctx->opt.scan_config[0] || !pioran_block_fits((int32_t)(2 * J), (int32_t)J))
        return PIORAN_ERR_UNSUPPORTED;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    pioran_ds* ds = nullptr;
    std::vector<double> zeros((size_t)N, 0.0);
    int rc = pioran_dataset_create(ctx, N, t, zeros.data(), sigma2, &ds);
    if (rc) return rc;
    auto done = [&](int code) { pioran_dataset_destroy(ds); return code; };
    PrepState& s = ds->host;
    if ((rc = prepare_state(ds, s, J, C, Dd, nullptr))) return done(rc);    // row map with both rows of every term
    const int64_t bt = (int64_t)pioran_block_table_doubles(N, s.R, s.J);
    int64_t chunk = B < 256 ? B : 256;
    {
        size_t free_b = 0, total_b = 0;
        auto need = [&](int64_t nb) { return ((size_t)nb * (size_t)bt + pioran_block_store_workspace_doubles(nb, N, s.R, 3) + 3 * (size_t)nb * (size_t)N) * sizeof(double); };
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            while (chunk > 1 && need(chunk) > ws_allow(ctx, free_b) + ctx->bwork.cap + ctx->bscratch.cap + ctx->bq.cap) chunk /= 2;
    }
    for (;;) {
        rc = ensure(ctx, ctx->bwork, pioran_block_store_workspace_doubles(chunk, N, s.R, 3) * sizeof(double));
        if (!rc) rc = ensure(ctx, ctx->bscratch, (size_t)chunk * (size_t)bt * sizeof(double));
        if (!rc) rc = ensure(ctx, ctx->bq, (size_t)chunk * (size_t)N * sizeof(double));     // xi
        if (rc != PIORAN_ERR_ALLOC || chunk == 1) break;
        chunk /= 2;
    }
    if (rc) return done(rc);
    const size_t cn = (size_t)chunk * (size_t)N * sizeof(double);
    if ((rc = ensure(ctx, ctx->bY, cn))) return done(rc);     // noise
    if ((rc = ensure(ctx, ctx->bS2, cn))) return done(rc);    // realisations
    if ((rc = ensure(ctx, ctx->bout, chunk * sizeof(double)))) return done(rc);
    if ((rc = ensure(ctx, ctx->bst, chunk * sizeof(int32_t)))) return done(rc);
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        const int64_t nb = B - b0 < chunk ? B - b0 : chunk;
        const size_t nbj = (size_t)nb * J * sizeof(double);
        if ((rc = upload(ctx, ctx->bA, A + b0 * J, nbj))) return done(rc);
        if ((rc = upload(ctx, ctx->bB, Bc + b0 * J, nbj))) return done(rc);
        if ((rc = upload(ctx, ctx->bC, C + b0 * J, nbj))) return done(rc);
        if ((rc = upload(ctx, ctx->bD, Dd + b0 * J, nbj))) return done(rc);
        if ((rc = upload(ctx, ctx->bY, q + b0 * N, (size_t)nb * N * sizeof(double)))) return done(rc);
        double* btab = (double*)ctx->bscratch.p;
        if ((rc = pioran_launch_block_table_batch(N, s.R, s.J, nb, s.rowmap, ds->t, (const double*)ctx->bC.p, (const double*)ctx->bD.p, ds->y, ds->s2,
                                                  btab, bt, ctx->stream))) return done(rc);
        ScanParams p{};
        p.opt = &ctx->opt;
        p.N = N; p.J = s.J; p.R = s.R; p.B = nb; p.standard_rows = 1;
        p.rec_stride = 3 * (int64_t)(s.R + 2) + 2;
        p.tab_draw_stride = bt;
        p.rowmap = s.rowmap; p.t = ds->t; p.y = ds->y; p.s2 = ds->s2;
        p.A = (const double*)ctx->bA.p; p.Bc = (const double*)ctx->bB.p; p.C = (const double*)ctx->bC.p; p.D = (const double*)ctx->bD.p;
        p.out = (double*)ctx->bout.p; p.status = (int32_t*)ctx->bst.p;
        p.noise = (const double*)ctx->bY.p; p.ysim = (double*)ctx->bS2.p;
        p.gw = (double*)ctx->bwork.p;
        g_last_kernel = "block (windowed simulation, per-draw tables)";
        rc = pioran_launch_block_sim(p, btab, (double*)ctx->bq.p, ctx->stream);
        if (rc) { ctx->last_err = "windowed simulation launch failed"; return done(rc); }
        if (hipMemcpyAsync(y_out + b0 * N, ctx->bS2.p, (size_t)nb * N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            ctx_sync(ctx) != PIORAN_OK) {
            ctx->last_err = "simulation copy-back failed";
            return done(PIORAN_ERR_HIP);
        }
    }
    return done(PIORAN_OK);
}

int pioran_celerite_simulate(pioran_ctx* ctx, int64_t N, int64_t B, int64_t J, const double* A, const double* Bc,
                             const double* C, const double* Dd, int cd_shared, const double* t, const double* sigma2,
                             const double* q, double* y_out)
{
    if (!ctx || N < 1 || B < 1 || J < 1 || !A || !Bc || !C || !Dd || !t || !sigma2 || !q || !y_out) return PIORAN_ERR_ARG;
    if (cd_shared || B == 1) return simulate_shared(ctx, N, B, J, A, Bc, C, Dd, t, sigma2, q, y_out);
    {
        const int rc = simulate_perdraw_windowed(ctx, N, B, J, A, Bc, C, Dd, t, sigma2, q, y_out);
        if (rc != PIORAN_ERR_UNSUPPORTED) return rc;
    }
    for (int64_t b = 0; b < B; ++b) {
        const int rc = simulate_shared(ctx, N, 1, J, A + b * J, Bc + b * J, C + b * J, Dd + b * J, t, sigma2, q + b * N, y_out + b * N);
        if (rc) return rc;
    }
    return PIORAN_OK;
}

const char* pioran_celerite_config_name(int64_t R)
{
    if (R < 0) return g_last_kernel;                 // kernel family of the calling thread's last launch: block, block+pd, wide, scan, fallback
    if (R <= 0) return pioran_scan_config_name(0);   // what the calling thread's last throughput-layout launch ran on
    if (R > pioran_wide_supported_rows()) return "fallback";
    if (R > pioran_scan_supported_rows()) return "wide";   // 80 .. 143 rows: the lean latency kernel (80 with a shared table and a large batch: the scan)
    return pioran_scan_config_name((int)R);
}

// ---- in-process farm ----------------------------------------------------------------------------
struct pioran_farm {
    std::vector<pioran_ctx*> ctx;
    std::vector<pioran_ds*> ds;
};

int pioran_farm_create(int ngpu, const int* devices, int64_t N, const double* t, const double* y, const double* sigma2,
                       pioran_farm** out)
{
    if (!out || ngpu < 1 || ngpu > 64 || !devices || N < 1 || !t || !y || !sigma2) return PIORAN_ERR_ARG;
    *out = nullptr;
    pioran_farm* f = new (std::nothrow) pioran_farm;
    if (!f) return PIORAN_ERR_ALLOC;
    int rc = PIORAN_OK;
    for (int g = 0; g < ngpu && rc == PIORAN_OK; ++g) {
        pioran_ctx* c = nullptr;
        pioran_ds* d = nullptr;
        rc = pioran_ctx_create(devices[g], &c);
        if (rc == PIORAN_OK) {
            f->ctx.push_back(c);
            rc = pioran_dataset_create(c, N, t, y, sigma2, &d);
            if (rc == PIORAN_OK) f->ds.push_back(d);
        }
    }
    if (rc != PIORAN_OK) { pioran_farm_destroy(f); return rc; }
    *out = f;
    return PIORAN_OK;
}

int pioran_farm_destroy(pioran_farm* f)
{
    if (!f) return PIORAN_ERR_ARG;
    for (auto* d : f->ds) pioran_dataset_destroy(d);
    for (auto* c : f->ctx) pioran_ctx_destroy(c);
    delete f;
    return PIORAN_OK;
}

int pioran_farm_size(const pioran_farm* f) { return f ? (int)f->ds.size() : PIORAN_ERR_ARG; }

static int farm_batch_impl(pioran_farm* f, int64_t B, int64_t J, const double* A, const double* Bc, const double* C,
                           const double* Dd, int cd_shared, const double* mu, const double* nu, const double* shift,
                           const double* Y, const double* S2, double* out, int32_t* status)
{
    if (!f || f->ds.empty() || B < 1 || J < 1 || !A || !Bc || !C || !Dd || !out) return PIORAN_ERR_ARG;
    if ((Y == nullptr) != (S2 == nullptr) || (Y && shift)) return PIORAN_ERR_ARG;
    const int64_t N = f->ds[0]->N;
    const int64_t G = (int64_t)f->ds.size();
    const int64_t base = B / G, extra = B % G;
    std::vector<int> rcs(G, PIORAN_OK);
    std::vector<std::thread> th;
    for (int64_t g = 0; g < G; ++g) {
        const int64_t lo = g * base + (g < extra ? g : extra), nb = base + (g < extra ? 1 : 0);
        if (nb == 0) continue;
        th.emplace_back([=, &rcs]() {
            const double* Cg = cd_shared ? C : C + lo * J;
            const double* Dg = cd_shared ? Dd : Dd + lo * J;
            const double* mug = mu ? mu + lo : nullptr;
            const double* nug = nu ? nu + lo : nullptr;
            int32_t* stg = status ? status + lo : nullptr;
            rcs[g] = shift ? pioran_celerite_logl_batch_shift(f->ds[g], nb, J, A + lo * J, Bc + lo * J, Cg, Dg, cd_shared, mug,
                                                              nug, shift + lo, out + lo, stg)
                           : pioran_celerite_logl_batch(f->ds[g], nb, J, A + lo * J, Bc + lo * J, Cg, Dg, cd_shared, mug, nug,
                                                        Y ? Y + lo * N : nullptr, S2 ? S2 + lo * N : nullptr, out + lo, stg);
        });
    }
    for (auto& t_ : th) t_.join();
    for (int rc : rcs)
        if (rc != PIORAN_OK) return rc;
    return PIORAN_OK;
}

int pioran_farm_logl_batch(pioran_farm* f, int64_t B, int64_t J, const double* A, const double* Bc, const double* C,
                           const double* Dd, int cd_shared, const double* mu, const double* nu, const double* shift,
                           double* out, int32_t* status)
{
    return farm_batch_impl(f, B, J, A, Bc, C, Dd, cd_shared, mu, nu, shift, nullptr, nullptr, out, status);
}

int pioran_farm_logl_batch_series(pioran_farm* f, int64_t B, int64_t J, const double* A, const double* Bc, const double* C,
                                  const double* Dd, int cd_shared, const double* mu, const double* nu, const double* Y,
                                  const double* S2, double* out, int32_t* status)
{
    if (!Y || !S2) return PIORAN_ERR_ARG;
    return farm_batch_impl(f, B, J, A, Bc, C, Dd, cd_shared, mu, nu, nullptr, Y, S2, out, status);
}

// ---- dense solver -------------------------------------------------------------------------------
// ascending time stamps enable the factorised covariance build (dense.hip); anything else takes the direct one
static int is_sorted(const double* t, int64_t N)
{
    for (int64_t i = 1; i < N; ++i)
        if (!(t[i] >= t[i - 1])) return 0;
    return 1;
}

static int dense_stage(pioran_ctx* ctx, int64_t N, int64_t J, const double* a, const double* b, const double* c,
                       const double* d, const double* t, const double* y, const double* sigma2, double** dv)
{
    // one staging buffer: a b c d (J each) | t y s2 (N each) | out (1) | info
    const size_t nd = 4 * (size_t)J + 3 * (size_t)N + 2;
    int rc = ensure(ctx, ctx->bwork, nd * sizeof(double));
    if (rc) return rc;
    double* base = (double*)ctx->bwork.p;
    const double* src[7] = {a, b, c, d, t, y, sigma2};
    size_t off = 0;
    for (int i = 0; i < 7; ++i) {
        const size_t n = i < 4 ? (size_t)J : (size_t)N;
        dv[i] = base + off;
        if (src[i])
            HIPCHK(ctx, hipMemcpyAsync(dv[i], src[i], n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        else
            HIPCHK(ctx, hipMemsetAsync(dv[i], 0, n * sizeof(double), ctx->stream));
        off += n;
    }
    dv[7] = base + off;      // out
    dv[8] = base + off + 1;  // info (int32 in the first 4 bytes)
    int64_t Mp, ld;
    pioran_dense_dims(N, &Mp, &ld);
    return ensure(ctx, ctx->bK, ((size_t)Mp * (size_t)ld + PIORAN_DENSE_WS) * sizeof(double));
}

static int dense_nll_impl(pioran_ctx* ctx, int64_t N, int64_t J, const double* a, const double* b, const double* c,
                          const double* d, const double* t, const double* y, const double* sigma2, double* out,
                          int32_t* info, float* phase_ms)
{
    if (!ctx || N < 1 || J < 1 || !a || !b || !c || !d || !t || !y || !sigma2 || !out) return PIORAN_ERR_ARG;
    if (N > 46000) return PIORAN_ERR_UNSUPPORTED;  // slab would exceed ~17 GB
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PendingGuard pending_guard(ctx);
    double* dv[9];
    int rc = dense_stage(ctx, N, J, a, b, c, d, t, y, sigma2, dv);
    if (rc) return rc;
    if (phase_ms) HIPCHK(ctx, hipEventRecord(ctx->ev[12], ctx->stream));
    rc = pioran_dense_nll_device(N, (int32_t)J, dv[0], dv[1], dv[2], dv[3], dv[4], dv[5], dv[6], (double*)ctx->bK.p,
                                 phase_ms ? &ctx->ev[13] : nullptr, dv[7], (int32_t*)dv[8], is_sorted(t, N), ctx->stream, 0.0, 1.0, &ctx->opt.dense);
    if (rc) { ctx->last_err = "dense kernel launch failed"; return rc; }
    int32_t hinfo = 0;
    if ((rc = download(ctx, out, dv[7], sizeof(double)))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(&hinfo, dv[8], sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    SYNC(ctx);
    if (info) *info = hinfo;
    if (phase_ms)
        for (int i = 0; i < 3; ++i) HIPCHK(ctx, hipEventElapsedTime(&phase_ms[i], ctx->ev[12 + i], ctx->ev[13 + i]));
    return PIORAN_OK;
}

int pioran_dense_nll(pioran_ctx* ctx, int64_t N, int64_t J, const double* a, const double* b, const double* c,
                     const double* d, const double* t, const double* y, const double* sigma2, double* out,
                     int32_t* info)
{
    return dense_nll_impl(ctx, N, J, a, b, c, d, t, y, sigma2, out, info, nullptr);
}

int pioran_dense_nll_batch(pioran_ctx* ctx, int64_t N, int64_t J, int64_t B, const double* A, const double* Bc, const double* C,
                           const double* Dd, int cd_shared, const double* t, const double* y, const double* sigma2,
                           const double* mu, const double* nu, double* out, int32_t* info)
{
    if (!ctx || N < 1 || J < 1 || B < 1 || !A || !Bc || !C || !Dd || !t || !y || !sigma2 || !out) return PIORAN_ERR_ARG;
    if (N > 46000) return PIORAN_ERR_UNSUPPORTED;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PendingGuard pending_guard(ctx);
    int rc;
    int64_t Mp, ld;
    pioran_dense_dims(N, &Mp, &ld);
    const size_t slab = (size_t)Mp * (size_t)ld + PIORAN_DENSE_WS;
    // Several factorisations per launch (round 3, late): a single N = 4096 factorisation is a chain of 64 latency-bound steps that
    // leaves most of the chip idle; with gridDim.z = nb matrices every kernel of the chain is launched once per BATCH (16 concurrent
    // streams of single-matrix launches gave 0.96 ms per factorisation, tools/sweep_dense_streams.py).  As many slabs as fit in a
    // third of the free memory, at most 32 (option dense_streams: fewer).
    const int64_t max_batch = ctx->opt.dense_streams > 0 && ctx->opt.dense_streams <= 64 ? ctx->opt.dense_streams : 32;
    int64_t ns = B < max_batch ? B : max_batch;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            while (ns > 1 && (size_t)ns * slab * sizeof(double) > free_b / 3 + ctx->bK.cap) --ns;
    }
    while ((rc = ensure(ctx, ctx->bK, (size_t)ns * slab * sizeof(double))) == PIORAN_ERR_ALLOC && ns > 1) --ns;
    if (rc) return rc;
    // staging: A Bc [B][J] | C D ([J] or [B][J]) | t y s2 [N] | mu nu [B] | out [B] | info [B] (int32)
    const size_t ncd = cd_shared ? (size_t)J : (size_t)B * J;
    const size_t nd = 2 * (size_t)B * J + 2 * ncd + 3 * (size_t)N + 4 * (size_t)B;
    if ((rc = ensure(ctx, ctx->bwork, nd * sizeof(double)))) return rc;
    double* dA = (double*)ctx->bwork.p; double* dB = dA + (size_t)B * J; double* dC = dB + (size_t)B * J; double* dD = dC + ncd;
    double* dt = dD + ncd; double* dy = dt + N; double* ds2 = dy + N; double* dmu = ds2 + N; double* dnu = dmu + B;
    double* dout = dnu + B; int32_t* dinfo = (int32_t*)(dout + B);
    HIPCHK(ctx, hipMemcpyAsync(dA, A, (size_t)B * J * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dB, Bc, (size_t)B * J * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dC, C, ncd * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dD, Dd, ncd * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dt, t, (size_t)N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dy, y, (size_t)N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ds2, sigma2, (size_t)N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (mu) HIPCHK(ctx, hipMemcpyAsync(dmu, mu, (size_t)B * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (nu) HIPCHK(ctx, hipMemcpyAsync(dnu, nu, (size_t)B * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    const int sorted = is_sorted(t, N);
    for (int64_t b0 = 0; b0 < B; b0 += ns) {
        const int64_t nb = B - b0 < ns ? B - b0 : ns;
        rc = pioran_dense_nll_device_batch(nb, N, (int32_t)J, dA + b0 * J, dB + b0 * J, cd_shared ? dC : dC + b0 * J, cd_shared ? dD : dD + b0 * J,
                                           cd_shared ? 0 : J, dt, dy, ds2, (double*)ctx->bK.p, (int64_t)slab, mu ? dmu + b0 : nullptr,
                                           nu ? dnu + b0 : nullptr, dout + b0, dinfo + b0, sorted, ctx->stream, &ctx->opt.dense);
        if (rc) { ctx->last_err = "dense kernel launch failed"; return rc; }
    }
    if ((rc = download(ctx, out, dout, (size_t)B * sizeof(double)))) return rc;
    if (info) if ((rc = download(ctx, info, dinfo, (size_t)B * sizeof(int32_t)))) return rc;
    SYNC(ctx);
    return PIORAN_OK;
}

int pioran_dense_nll_timed(pioran_ctx* ctx, int64_t N, int64_t J, const double* a, const double* b, const double* c,
                           const double* d, const double* t, const double* y, const double* sigma2, double* out,
                           int32_t* info, float* phase_ms)
{
    if (!phase_ms) return PIORAN_ERR_ARG;
    return dense_nll_impl(ctx, N, J, a, b, c, d, t, y, sigma2, out, info, phase_ms);
}

// predict_direct / predict_cov (src/direct_solver.jl:28-119): y == nullptr: covariance only; cov_out == nullptr: mean only
static int dense_predict_impl(pioran_ctx* ctx, int64_t N, int64_t J, const double* a, const double* b, const double* c,
                              const double* d, const double* t, const double* y, const double* sigma2, int64_t M,
                              const double* tau, double* mean_out, double* cov_out, int32_t* info)
{
    if (!ctx || N < 1 || J < 1 || M < 1 || !a || !b || !c || !d || !t || !sigma2 || !tau) return PIORAN_ERR_ARG;
    if ((y == nullptr) != (mean_out == nullptr) || (!mean_out && !cov_out)) return PIORAN_ERR_ARG;
    const int64_t Mp = (N + 63) / 64 * 64, Mq = (M + 63) / 64 * 64, Mtot = Mp + Mq;
    if (Mtot > 46000) return PIORAN_ERR_UNSUPPORTED;  // slab would exceed ~17 GB
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PendingGuard pending_guard(ctx);
    // [t | NaN padding | tau | NaN padding]: a NaN time is an identity row/column of the augmented matrix
    std::vector<double> te((size_t)Mtot, std::nan("")), s2e((size_t)Mtot, 0.0), ye((size_t)Mtot, 0.0);
    std::memcpy(te.data(), t, (size_t)N * sizeof(double));
    std::memcpy(s2e.data(), sigma2, (size_t)N * sizeof(double));
    std::memcpy(te.data() + Mp, tau, (size_t)M * sizeof(double));
    if (y) std::memcpy(ye.data(), y, (size_t)N * sizeof(double));
    double* dv[9];
    int rc = dense_stage(ctx, Mtot, J, a, b, c, d, te.data(), ye.data(), s2e.data(), dv);
    if (rc) return rc;
    // dense_stage copies asynchronously from the vectors above: they must outlive the copies
    SYNC(ctx);
    if ((rc = ensure(ctx, ctx->bout, (size_t)M * sizeof(double)))) return rc;
    rc = pioran_dense_predict_cov_device(N, M, (int32_t)J, dv[0], dv[1], dv[2], dv[3], dv[4], dv[6], (double*)ctx->bK.p,
                                         (int32_t*)dv[8], y ? dv[5] : nullptr, y ? (double*)ctx->bout.p : nullptr, ctx->stream);
    if (rc) { ctx->last_err = "dense prediction launch failed"; return rc; }
    const int64_t ld = Mtot + 64;
    int32_t hinfo = 0;
    if (cov_out)
        HIPCHK(ctx, hipMemcpy2DAsync(cov_out, (size_t)M * sizeof(double), (const double*)ctx->bK.p + Mp + Mp * ld,
                                     (size_t)ld * sizeof(double), (size_t)M * sizeof(double), (size_t)M, hipMemcpyDeviceToHost,
                                     ctx->stream));
    if (mean_out) if ((rc = download(ctx, mean_out, ctx->bout.p, (size_t)M * sizeof(double)))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(&hinfo, dv[8], sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    SYNC(ctx);
    if (cov_out) {
        // the slab holds the lower triangle (column k, rows i >= k): mirror it
        for (int64_t k = 0; k < M; ++k)
            for (int64_t i = k + 1; i < M; ++i) cov_out[i * M + k] = cov_out[k * M + i];
        if (hinfo != 0)
            for (int64_t e = 0; e < M * M; ++e) cov_out[e] = std::nan("");
    }
    if (info) *info = hinfo;
    return PIORAN_OK;
}

int pioran_dense_predict_cov(pioran_ctx* ctx, int64_t N, int64_t J, const double* a, const double* b, const double* c,
                             const double* d, const double* t, const double* sigma2, int64_t M, const double* tau,
                             double* cov_out, int32_t* info)
{
    if (!cov_out) return PIORAN_ERR_ARG;
    return dense_predict_impl(ctx, N, J, a, b, c, d, t, nullptr, sigma2, M, tau, nullptr, cov_out, info);
}

int pioran_dense_predict(pioran_ctx* ctx, int64_t N, int64_t J, const double* a, const double* b, const double* c,
                         const double* d, const double* t, const double* y, const double* sigma2, int64_t M,
                         const double* tau, double* mean_out, double* cov_out, int32_t* info)
{
    if (!y || !mean_out) return PIORAN_ERR_ARG;
    return dense_predict_impl(ctx, N, J, a, b, c, d, t, y, sigma2, M, tau, mean_out, cov_out, info);
}

int pioran_dense_covariance(pioran_ctx* ctx, int64_t N, int64_t J, const double* a, const double* b, const double* c,
                            const double* d, const double* t, const double* sigma2, double* K_out)
{
    if (!ctx || N < 1 || J < 1 || !a || !b || !c || !d || !t || !K_out) return PIORAN_ERR_ARG;
    if (N > 46000) return PIORAN_ERR_UNSUPPORTED;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    double* dv[9];
    int rc = dense_stage(ctx, N, J, a, b, c, d, t, nullptr, sigma2, dv);
    if (rc) return rc;
    rc = pioran_dense_build_device(N, (int32_t)J, dv[0], dv[1], dv[2], dv[3], dv[4], dv[5], dv[6], (double*)ctx->bK.p,
                                   is_sorted(t, N), ctx->stream);
    if (rc) { ctx->last_err = "dense build launch failed"; return rc; }
    int64_t Mp, ld;
    pioran_dense_dims(N, &Mp, &ld);
    HIPCHK(ctx, hipMemcpy2DAsync(K_out, (size_t)N * sizeof(double), ctx->bK.p, (size_t)ld * sizeof(double),
                                 (size_t)N * sizeof(double), (size_t)N, hipMemcpyDeviceToHost, ctx->stream));
    SYNC(ctx);
    // the slab holds the lower triangle (column-major): mirror it, K is symmetric (src/direct_solver.jl:9-14 fills both)
    for (int64_t k = 0; k < N; ++k)
        for (int64_t i = k + 1; i < N; ++i) K_out[k + i * N] = K_out[i + k * N];
    return PIORAN_OK;
}


}  // extern "C"
