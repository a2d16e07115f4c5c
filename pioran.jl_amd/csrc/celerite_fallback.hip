// Any-rank fallback of the celerite scan (R = 2J rows beyond what the register-resident kernel
// holds, e.g. the reference benchmark's j = 64 terms, benchmark/benchmarks.jl:17).
// One wavefront per draw; the R x R state S lives in an HBM scratch slab (k-major so that the 64
// lanes, which own rows j = lane, lane+64, ..., touch consecutive addresses); the per-row vectors
// live in LDS.  Same recurrence and the same operation order as celerite_scan.hip; HBM-bound
// (16 R^2 bytes per step and draw).  Correctness path, not the fast path.
#include "common.h"

namespace {
constexpr int RMAX = 512;

__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) x += __shfl_xor(x, m);
    return x;
}

__global__ void __launch_bounds__(64) celerite_fallback_kernel(const ScanParams p, int64_t b0)
{
    __shared__ double w_s[RMAX], f_s[RMAX], u_s[RMAX], ph_s[RMAX], v_s[RMAX], g_s[RMAX];
    __shared__ double al_s[RMAX], be_s[RMAX], cc_s[RMAX], dd_s[RMAX];
    __shared__ int term_s[RMAX];
    const int lane = threadIdx.x;
    const int64_t b = b0 + blockIdx.x;
    const int J = p.J, R = p.R, Rp = R + 2;
    const int64_t RS = p.rec_stride;  // shared table record stride, see table.hip
    const int64_t N = p.N;
    double* S = p.scratch + (int64_t)blockIdx.x * ((int64_t)R * R);

    for (int j = lane; j < R; j += 64) {
        const int rm = p.rowmap[j];
        const int tj = rm & 0xfffff;
        const bool ks = (rm >> 30) & 1;
        const double a = p.A[b * J + tj], bb = p.Bc[b * J + tj];
        term_s[j] = rm;
        al_s[j] = a;              // u = a v + (+-b) x  with (v, x) = (cos, sin) | (sin, cos)
        be_s[j] = ks ? -bb : bb;
        if (!p.tab) { cc_s[j] = p.C[b * J + tj]; dd_s[j] = p.D[b * J + tj]; }
        f_s[j] = 0.0;
        for (int k = 0; k < R; ++k) S[(int64_t)k * R + j] = 0.0;
    }
    double suma = 0.0;
    for (int j = 0; j < J; ++j) suma += p.A[b * J + j];
    const double mu = p.mu ? p.mu[b] : 0.0;
    const double nu = p.nu ? p.nu[b] : 1.0;
    const double* yv = p.Y ? p.Y + b * N : p.y;
    const double* sv = p.S2 ? p.S2 + b * N : p.s2;

    // (v, x, phi) of row j at step n: v is the row's own trig value, x the other one
    auto trig = [&](int j, int64_t n, double& v, double& x, double& ph) {
        if (p.tab) {
            const double* rec = p.tab + n * RS;
            v = rec[j]; x = rec[Rp + j]; ph = rec[2 * Rp + j];
        } else {
            const double tn = p.t[n];
            double si, co;
            sincos(dd_s[j] * tn, &si, &co);
            const bool ks = (term_s[j] >> 30) & 1;
            v = ks ? si : co;
            x = ks ? co : si;
            ph = n > 0 ? exp(-cc_s[j] * (tn - p.t[n - 1])) : 0.0;
        }
    };

    double Dn = suma + (p.nu ? nu * sv[0] : sv[0]);
    double rD = 1.0 / Dn;
    for (int j = lane; j < R; j += 64) {
        double v, x, ph;
        trig(j, 0, v, x, ph);
        w_s[j] = v * rD;
    }
    double z = yv[0] - mu;
    int Pe = 0;
    double Pm = frexp(Dn, &Pe);
    double quad = z * z * rD;
    bool nonpd = !(Dn > 0.0);
    __syncthreads();

    for (int64_t n = 1; n < N; ++n) {
        double zzp = 0.0;
        for (int j = lane; j < R; j += 64) {
            double v, x, ph;
            trig(j, n, v, x, ph);
            const double u = al_s[j] * v + be_s[j] * x;
            u_s[j] = u;
            v_s[j] = v;
            ph_s[j] = ph;
            g_s[j] = Dn * w_s[j];
            const double f = (f_s[j] + w_s[j] * z) * ph;
            f_s[j] = f;
            zzp += u * f;
        }
        __syncthreads();
        double sp = 0.0;
        double qloc[RMAX / 64];
        int jj = 0;
        for (int j = lane; j < R; j += 64, ++jj) {
            const double gj = g_s[j], pj = ph_s[j];
            double q = 0.0;
            for (int k = 0; k < R; ++k) {
                const double m = fma(gj, w_s[k], S[(int64_t)k * R + j]);
                const double sn = (pj * ph_s[k]) * m;
                S[(int64_t)k * R + j] = sn;
                q = fma(sn, u_s[k], q);
            }
            qloc[jj] = q;
            sp += u_s[j] * q;
        }
        const double s = wave_sum(sp);
        const double zz = wave_sum(zzp);
        __syncthreads();  // every lane has finished reading w_s (previous W) before it is replaced
        Dn = suma + (p.nu ? nu * sv[n] : sv[n]) - s;
        rD = 1.0 / Dn;
        jj = 0;
        for (int j = lane; j < R; j += 64, ++jj) w_s[j] = (v_s[j] - qloc[jj]) * rD;
        z = (yv[n] - mu) - zz;
        nonpd |= !(Dn > 0.0);
        Pm *= fabs(Dn);
        int ex;
        Pm = frexp(Pm, &ex);
        Pe += ex;
        quad = fma(z * z, rD, quad);
        __syncthreads();
    }
    if (lane == 0) {
        const double logdet = log(Pm) + (double)Pe * 0.6931471805599453094;
        const double res = -0.5 * logdet - 0.5 * (double)N * 1.8378770664093454836 - 0.5 * quad;
        p.out[b] = res;
        if (p.status) p.status[b] = !isfinite(res) ? 2 : (nonpd ? 1 : 0);
    }
}
}  // namespace

size_t pioran_fallback_scratch_doubles(int R) { return (size_t)R * (size_t)R; }

// p.scratch must hold `chunk` slabs of R*R doubles; the batch is walked in chunks.
int pioran_launch_scan_fallback(const ScanParams& p, hipStream_t stream)
{
    if (p.R > RMAX) return PIORAN_ERR_UNSUPPORTED;
    const int64_t chunk = 1024;
    for (int64_t b0 = 0; b0 < p.B; b0 += chunk) {
        const int64_t nb = p.B - b0 < chunk ? p.B - b0 : chunk;
        hipLaunchKernelGGL(celerite_fallback_kernel, dim3((unsigned)nb), dim3(64), 0, stream, p, b0);
        if (hipGetLastError() != hipSuccess) return PIORAN_ERR_HIP;
    }
    return PIORAN_OK;
}
