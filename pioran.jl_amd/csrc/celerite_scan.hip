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
//   * the column loop broadcasts w_k, u_k, phi_k of the row-owning lane to the 16 lanes of the DPP
//     row with v_mov_b64_dpp row_newbcast (no LDS, no readlane), so q = S u needs no cross-lane
//     reduction inside a DPP row; only u'q and u'f are 16-lane DPP butterflies;
//   * with CBR > 1 every DPP row r holds the rows rotated by NSRC*r lanes, so the SAME instruction
//     stream (broadcast source lane N, register slot m) walks a DIFFERENT column block in each DPP
//     row; the partial q of the CBR column blocks are summed with ds_bpermute.
// FP64 VALU bound (no MFMA: the update is rank-1 per draw, nothing is shared across draws).
#include "common.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>

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

// lane N of each 16-lane DPP row -> all lanes of that row (one v_mov_b64_dpp)
template <int N>
__device__ __forceinline__ double row_bcast(double x)
{
    // `old` undefined: every lane is written (row_newbcast has no invalid lanes), so no init mov
    return __builtin_amdgcn_mov_dpp(x, 0x150 + N, 0xf, 0xf, true);
}

template <int CTRL>
__device__ __forceinline__ double dpp_perm(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// sum over the 16 lanes of a DPP row; every lane gets the bit-identical total
__device__ __forceinline__ double row16_sum(double x)
{
    x += dpp_perm<0xB1>(x);   // quad_perm [1,0,3,2]
    x += dpp_perm<0x4E>(x);   // quad_perm [2,3,0,1]
    x += dpp_perm<0x141>(x);  // row_half_mirror
    x += dpp_perm<0x140>(x);  // row_mirror
    return x;
}

template <int CBR>
__device__ __forceinline__ double group_sum(double x)
{
    x = row16_sum(x);
    if constexpr (CBR >= 2) x += __shfl_xor(x, 16);
    if constexpr (CBR >= 4) x += __shfl_xor(x, 32);
    return x;
}

template <int RPL, int CBR, int NSRC, bool SHARED_TAB>
__global__ void __launch_bounds__(256) celerite_scan_kernel(const ScanParams p)
{
    static_assert(NSRC * CBR <= 16, "source lanes must fit a DPP row");
    constexpr int G = 16 * CBR;          // lanes per draw
    constexpr int EPW = 64 / G;          // draws per wavefront
    constexpr int NC = NSRC * RPL;       // columns held per lane
    constexpr int YLAM = NSRC * CBR - 1; // the y row is the LAST row slot: logical lane YLAM, slot RPL-1
    constexpr int YS = RPL - 1;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int e = lane / G;
    const int r = (lane % G) >> 4;           // DPP row inside the draw = column block
    const int l = lane & 15;
    const int lam = (l + NSRC * r) & 15;     // logical lane: which rows this lane owns
    const bool contributes = l < NSRC;       // each row is counted once in u'q
    const bool isy = lam == YLAM;            // this lane's slot YS is the y row
    const int64_t b_raw = ((int64_t)blockIdx.x * 4 + wave) * EPW + e;
    const bool active = b_raw < p.B;
    const int64_t b = active ? b_raw : p.B - 1;

    const int J = p.J, Jp = J + 2, R = p.R;  // table columns J (inert pad) and J+1 (y row)
    const int64_t N = p.N;

    int term[RPL];
    bool ksin[RPL];
    double al[RPL], be[RPL];
    [[maybe_unused]] double cc[RPL], dd[RPL];
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        const int j = lam * RPL + i;
        if (j < R) {
            const int rm = p.rowmap[j];
            term[i] = rm & 0x3fffffff;
            ksin[i] = (rm >> 30) & 1;
            const double a = p.A[b * J + term[i]], bb = p.Bc[b * J + term[i]];
            // u = a co + b si (cos row) | a si - b co (sin row)      celerite_solver.jl:59-60
            al[i] = ksin[i] ? -bb : a;
            be[i] = ksin[i] ? a : bb;
            if constexpr (!SHARED_TAB) {
                cc[i] = p.C[b * J + term[i]];
                dd[i] = p.D[b * J + term[i]];
            }
        } else {
            // inert padding row (u = 0, v = 1, phi = 0), or the y row (u = 0, v = y_n - mu, phi = 1)
            term[i] = (isy && i == YS) ? J + 1 : J;
            ksin[i] = false;
            al[i] = 0.0;
            be[i] = 0.0;
            if constexpr (!SHARED_TAB) { cc[i] = 0.0; dd[i] = 0.0; }
        }
    }
    double suma = 0.0;  // :21
    for (int j = 0; j < J; ++j) suma += p.A[b * J + j];
    const double mu = p.mu ? p.mu[b] : 0.0;
    const double nu = p.nu ? p.nu[b] : 1.0;
    const bool has_nu = p.nu != nullptr;
    const double* yv = p.Y ? p.Y + b * N : p.y;
    const double* sv = p.S2 ? p.S2 + b * N : p.s2;

    int p1 = 0, p2 = 0;  // lanes holding the same rows in the other column blocks
    if constexpr (CBR >= 2) p1 = e * G + (r ^ 1) * 16 + ((lam - NSRC * (r ^ 1)) & 15);
    if constexpr (CBR >= 4) p2 = e * G + (r ^ 2) * 16 + ((lam - NSRC * (r ^ 2)) & 15);

    double co[RPL], si[RPL], ph[RPL];
    auto load_step = [&](int64_t n, double (&co_)[RPL], double (&si_)[RPL], double (&ph_)[RPL], double& y_, double& s2_) {
        if constexpr (SHARED_TAB) {
            const double* rec = p.tab + n * 3 * Jp;
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                co_[i] = rec[term[i]];
                si_[i] = rec[Jp + term[i]];
                ph_[i] = rec[2 * Jp + term[i]];
            }
        } else {
            const double tn = p.t[n];
            const double dt = n > 0 ? tn - p.t[n - 1] : 0.0;
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                if (term[i] < J) {
                    double s_, c_;
                    sincos(dd[i] * tn, &s_, &c_);  // :52-53
                    co_[i] = c_;
                    si_[i] = s_;
                    ph_[i] = exp(-cc[i] * dt);     // :54
                } else {
                    co_[i] = term[i] == J ? 1.0 : 0.0;
                    si_[i] = 0.0;
                    ph_[i] = term[i] == J ? 0.0 : 1.0;
                }
            }
        }
        y_ = yv[n];
        s2_ = sv[n];
    };

    double yn, s2n;
    load_step(0, co, si, ph, yn, s2n);

    // ---- first row, :27-42 and :126-128 ----
    double S[RPL][NC];
#pragma unroll
    for (int i = 0; i < RPL; ++i)
#pragma unroll
        for (int c = 0; c < NC; ++c) S[i][c] = 0.0;
    double w[RPL], v[RPL];
    double Dn = suma + (has_nu ? nu * s2n : s2n);
    double rD = 1.0 / Dn;
#pragma unroll
    for (int i = 0; i < RPL; ++i) v[i] = ksin[i] ? si[i] : co[i];
    if (isy) v[YS] = yn - mu;                // z_1 = y_1      :128
#pragma unroll
    for (int i = 0; i < RPL; ++i) w[i] = v[i] * rD;
    double Pm = Dn;      // running product of |D| (sign of D_1 kept: log of a negative D_1 is NaN, :126)
    int Pe = 0;
    {
        int ex;
        Pm = frexp(Pm, &ex);
        Pe += ex;
    }
    double quad = v[YS] * v[YS] * rD;        // meaningful in the y-row lanes only
    bool nonpd = !(Dn > 0.0);

    double co2[RPL], si2[RPL], ph2[RPL], yn2, s2n2;
    if (N > 1) load_step(1, co2, si2, ph2, yn2, s2n2);

    for (int64_t n = 1; n < N; ++n) {
#pragma unroll
        for (int i = 0; i < RPL; ++i) { co[i] = co2[i]; si[i] = si2[i]; ph[i] = ph2[i]; }
        yn = yn2;
        s2n = s2n2;
        // prefetch the next step's table row; independent of the recurrence
        load_step(n + 1 < N ? n + 1 : n, co2, si2, ph2, yn2, s2n2);

        double u[RPL], g[RPL], qt[RPL];
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            u[i] = al[i] * co[i] + be[i] * si[i];
            g[i] = Dn * w[i];                       // dn = D[n-1] * V[j,n-1]   :73
            qt[i] = 0.0;
            v[i] = ksin[i] ? si[i] : co[i];
        }
        if (isy) v[YS] = yn - mu;

        // ---- S update + q = S u over this DPP row's column block ----
        static_for<0, NSRC>([&](auto Nc) {
            constexpr int NN = decltype(Nc)::value;
            static_for<0, RPL>([&](auto Mc) {
                constexpr int MM = decltype(Mc)::value;
                constexpr int c = NN * RPL + MM;
                const double wk = row_bcast<NN>(w[MM]);
                const double uk = row_bcast<NN>(u[MM]);
                const double pk = row_bcast<NN>(ph[MM]);
#pragma unroll
                for (int i = 0; i < RPL; ++i) {
                    const double m = fma(g[i], wk, S[i][c]);   // S + dn * V[k,n-1]          :78
                    const double sn = (ph[i] * pk) * m;        // phi_j phi_k ( ... )        :78,85
                    S[i][c] = sn;
                    qt[i] = fma(sn, uk, qt[i]);                // (S u)_j                    :80-82,86-89
                }
            });
        });
        if constexpr (CBR >= 2) {
#pragma unroll
            for (int i = 0; i < RPL; ++i) qt[i] += __shfl(qt[i], p1);
        }
        if constexpr (CBR >= 4) {
#pragma unroll
            for (int i = 0; i < RPL; ++i) qt[i] += __shfl(qt[i], p2);
        }
        double sp = 0.0;
#pragma unroll
        for (int i = 0; i < RPL; ++i) sp += u[i] * qt[i];        // u'Su                       :83,88
        if (!contributes) sp = 0.0;
        const double s = group_sum<CBR>(sp);

        Dn = suma + (has_nu ? nu * s2n : s2n) - s;               // :92
        rD = 1.0 / Dn;
        const double z = v[YS] - qt[YS];                         // y row: z_n = y_n - u'f      :141
#pragma unroll
        for (int i = 0; i < RPL; ++i) w[i] = (v[i] - qt[i]) * rD;                        // :89,96
        nonpd |= !(Dn > 0.0);
        Pm *= fabs(Dn);                                          // log(abs(D[n]))  :140
        int ex;
        Pm = frexp(Pm, &ex);
        Pe += ex;
        quad = fma(z * z, rD, quad);                             // z_n^2 / D_n  (== y'K^-1 y, :333)
    }

    if (active && isy && r == 0) {
        const double logdet = log(Pm) + (double)Pe * 0.6931471805599453094;
        const double res = -0.5 * logdet - 0.5 * (double)N * 1.8378770664093454836 - 0.5 * quad;
        p.out[b] = res;
        if (p.status) p.status[b] = !isfinite(res) ? 2 : (nonpd ? 1 : 0);
    }
}

using LaunchFn = void (*)(const ScanParams&, dim3, hipStream_t);

template <int RPL, int CBR, int NSRC>
void launch_cfg(const ScanParams& p, dim3 grid, hipStream_t st)
{
    if (p.tab)
        hipLaunchKernelGGL((celerite_scan_kernel<RPL, CBR, NSRC, true>), grid, dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((celerite_scan_kernel<RPL, CBR, NSRC, false>), grid, dim3(256), 0, st, p);
}

struct ScanConfig {
    const char* name;
    int rpl, cbr, nsrc;
    LaunchFn fn;
    int capacity() const { return rpl * cbr * nsrc - 1; }  // one row slot carries y
};

#define CFG(RPL, CBR, NSRC) {"rpl" #RPL "_cbr" #CBR "_nsrc" #NSRC, RPL, CBR, NSRC, &launch_cfg<RPL, CBR, NSRC>}
// preference order: first entry whose capacity >= R wins (unless PIORAN_SCAN_CONFIG names another)
const ScanConfig kConfigs[] = {
    CFG(1, 1, 5),  CFG(1, 1, 9),  CFG(1, 1, 13), CFG(1, 1, 16),           // R <= 15
    CFG(2, 1, 9),  CFG(2, 1, 11), CFG(2, 1, 13), CFG(2, 1, 15), CFG(2, 1, 16),  // R <= 31
    CFG(3, 2, 6),  CFG(3, 2, 7),  CFG(3, 2, 8),                           // R <= 47
    CFG(4, 4, 4),                                                         // R <= 63
    CFG(5, 4, 4),                                                         // R <= 79
    // alternatives kept for tuning runs (selected by name)
    CFG(3, 1, 14), CFG(3, 4, 4), CFG(2, 2, 8),
};
#undef CFG
constexpr int kNumPreferred = 14;

const ScanConfig* pick_config(int R)
{
    if (const char* env = std::getenv("PIORAN_SCAN_CONFIG")) {
        for (const auto& c : kConfigs)
            if (!std::strcmp(env, c.name) && c.capacity() >= R) return &c;
    }
    for (int i = 0; i < kNumPreferred; ++i)
        if (kConfigs[i].capacity() >= R) return &kConfigs[i];
    return nullptr;
}

}  // namespace

int pioran_scan_supported_rows() { return 79; }

const char* pioran_scan_config_name(int R)
{
    const ScanConfig* c = pick_config(R);
    return c ? c->name : "fallback";
}

int pioran_launch_scan(const ScanParams& p, hipStream_t stream)
{
    const ScanConfig* c = pick_config(p.R);
    if (!c) return PIORAN_ERR_UNSUPPORTED;
    const int epw = 64 / (16 * c->cbr);
    const int64_t per_block = 4 * epw;
    const int64_t blocks = (p.B + per_block - 1) / per_block;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return PIORAN_ERR_ARG;
    c->fn(p, dim3((unsigned)blocks), stream);
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}
