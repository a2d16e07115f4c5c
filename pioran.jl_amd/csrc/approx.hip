// approx on the device: theta -> (a_j, b_j) of the SumOfCelerite kernel, fused in front of the scan.
//
// Restates, per draw, src/psd.jl:214-289 (approx) for a continuum-only PSD model:
//   spectral grid f_j = f0 (fM/f0)^(j/(J-1)),  f0 = f_min/S_low, fM = f_max*S_high          :217-218,77-79
//   p_j = P(f_j) / P(f_0)                                   get_normalised_psd             :52-56
//   amplitudes = B \ p,  B_jk = 1 / (1 + (f_j/f_k)^{4|6})  build_approx / psd_decomp      :73-112
//   integ = analytic integral over [f_min, f_max] (integral_sho / integral_drwcelerite) or
//           the variance form                              get_norm_psd                   :301-324,375-395
//   SHO:          a = b = A_j f_j pi / sqrt2                                               :249-252
//   DRWCelerite:  a = A_j f_j pi / 3, b = sqrt3 a  ++  (a, 0)                             :264-275
// PSD features (QPO, src/psd.jl:15-27, 228-241, 254-261): n_qpo Lorentzians (S0, f0, Q) per draw become one celerite term each,
//   a = S0 w0 Q / 4, b = a / Delta, c = w0 / (2 Q), d = c Delta (Delta = sqrt(4 Q^2 - 1), w0 = 2 pi f0), a and b divided by the
// continuum's P(f_0) like the amplitudes, their integral (integrate_psd_feature :356-358) added to the normalisation when
// is_integrated_power, and appended as (2a, 2b, c, d) after the continuum terms; their (c, d) differ per draw.
// c_j, d_j of the continuum depend only on the grid and stay on the host side of the ABI (pioran_dataset_prepare).
// PSD models are Tonari.jl's closed forms (test/test_psd.jl:3-13).  The J x J matrix B is LU-factorised once
// on the host (partial pivoting, like Julia's `\`); each draw is one thread doing 2 J^2 flops of triangular
// solves in its own output row.  Cost ~ J (2 pow + log + atan2) + 2 J^2 flop per draw: microseconds per batch.
#include "common.h"

#include <cmath>
#include <vector>

namespace {

__device__ __forceinline__ double psd_model(int model, const double* th, double f)
{
    // 0: SingleBendingPowerLaw(alpha1, f1, alpha2); 1: DoubleBendingPowerLaw(alpha1, f1, alpha2, f2, alpha3)
    const double x = f / th[1];
    double v = pow(x, -th[0]) / (1.0 + pow(x, th[2] - th[0]));
    if (model == 1) v /= 1.0 + pow(f / th[3], th[4] - th[2]);
    return v;
}

__global__ void __launch_bounds__(64) approx_kernel(int64_t B, int model, int P, int J, int basis, int integrated,
                                                    double f_min, double f_max, const double* __restrict__ sp,
                                                    const double* __restrict__ LU, const int32_t* __restrict__ piv,
                                                    const double* __restrict__ theta, const double* __restrict__ norm,
                                                    int n_qpo, const double* __restrict__ qpo, double* __restrict__ A,
                                                    double* __restrict__ Bc, double* __restrict__ Cq, double* __restrict__ Dq)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int Jc = basis == 0 ? J : 2 * J;       // celerite terms of the continuum
    const int Jt = Jc + n_qpo;                   // + one per feature
    double* x = A + b * Jt;                      // work in place in the output row (first J entries)
    const double* th = theta + b * P;
    const double p0 = psd_model(model, th, sp[0]);
    for (int j = 0; j < J; ++j) x[j] = psd_model(model, th, sp[j]) / p0;
    // row interchanges, then L y = P p (unit lower), U x = y
    for (int j = 0; j < J; ++j) {
        const int pj = piv[j];
        if (pj != j) { const double tmp = x[j]; x[j] = x[pj]; x[pj] = tmp; }
    }
    for (int i = 1; i < J; ++i) {
        double acc = x[i];
        for (int k = 0; k < i; ++k) acc = fma(-LU[i * J + k], x[k], acc);
        x[i] = acc;
    }
    for (int i = J - 1; i >= 0; --i) {
        double acc = x[i];
        for (int k = i + 1; k < J; ++k) acc = fma(-LU[i * J + k], x[k], acc);
        x[i] = acc / LU[i * J + i];
    }
    // normalisation (src/psd.jl:375-395)
    double integ;
    if (integrated) {
        double hi = 0.0, lo = 0.0;
        if (basis == 0) {            // integral_sho :301-305
            const double s2 = 1.4142135623730951;
            for (int j = 0; j < J; ++j) {
                const double c = sp[j], nrm = c * x[j] / (4.0 * s2);
                const double ph = (f_max * f_max + s2 * c * f_max + c * c) / (f_max * f_max - s2 * c * f_max + c * c);
                const double pl = (f_min * f_min + s2 * c * f_min + c * c) / (f_min * f_min - s2 * c * f_min + c * c);
                hi += nrm * (log(ph) + 2.0 * atan2(c * s2 * f_max, c * c - f_max * f_max));
                lo += nrm * (log(pl) + 2.0 * atan2(c * s2 * f_min, c * c - f_min * f_min));
            }
        } else {                     // integral_drwcelerite :318-324
            const double s3 = 1.7320508075688772;
            for (int j = 0; j < J; ++j) {
                const double c = sp[j], nrm = x[j] * c / 3.0;
                const double ph = (f_max * f_max + s3 * c * f_max + c * c) / (f_max * f_max - s3 * c * f_max + c * c);
                const double pl = (f_min * f_min + s3 * c * f_min + c * c) / (f_min * f_min - s3 * c * f_min + c * c);
                hi += nrm * (atan(f_max / c) + 0.5 * atan2(f_max * f_max - c * c, c * f_max) + s3 / 4.0 * log(ph));
                lo += nrm * (atan(f_min / c) + 0.5 * atan2(f_min * f_min - c * c, c * f_min) + s3 / 4.0 * log(pl));
            }
        }
        integ = hi - lo;
        // features: integral of the celerite power spectrum between f_min and f_max   (src/psd.jl:330-334, 356-358)
        for (int q = 0; q < n_qpo; ++q) {
            const double S0 = qpo[(b * n_qpo + q) * 3], f0q = qpo[(b * n_qpo + q) * 3 + 1], Q = qpo[(b * n_qpo + q) * 3 + 2];
            const double De = sqrt(4.0 * Q * Q - 1.0), w0 = 2.0 * M_PI * f0q;
            const double fa = S0 * w0 * Q / 4.0 / p0, fb = fa / De, fc = w0 / Q / 2.0, fd = fc * De;
            auto icel = [&](double xx) {
                const double num = fc * fc + (fd + 2.0 * M_PI * xx) * (fd + 2.0 * M_PI * xx);
                const double den = fc * fc + (fd - 2.0 * M_PI * xx) * (fd - 2.0 * M_PI * xx);
                return (2.0 * fa * (atan2(fc, fd - 2.0 * M_PI * xx) - atan2(fc, fd + 2.0 * M_PI * xx)) + fb * log(num / den)) / (2.0 * M_PI);
            };
            integ += icel(f_max) - icel(f_min);
        }
    } else {
        double acc = 0.0;
        for (int j = 0; j < J; ++j) acc += x[j] * sp[j];
        integ = basis == 0 ? acc * M_PI / 1.4142135623730951 : acc * 2.0 * M_PI / 3.0;
    }
    const double scale = norm[b] / integ;
    double* bb = Bc + b * Jt;
    for (int q = 0; q < n_qpo; ++q) {            // (2a, 2b, c, d) of the feature terms   (:254-261, 276-283)
        const double S0 = qpo[(b * n_qpo + q) * 3], f0q = qpo[(b * n_qpo + q) * 3 + 1], Q = qpo[(b * n_qpo + q) * 3 + 2];
        const double De = sqrt(4.0 * Q * Q - 1.0), w0 = 2.0 * M_PI * f0q;
        const double fa = S0 * w0 * Q / 4.0 / p0 * scale, fc = w0 / Q / 2.0;
        x[Jc + q] = 2.0 * fa;
        bb[Jc + q] = 2.0 * (fa / De);
        Cq[b * Jt + Jc + q] = fc;
        Dq[b * Jt + Jc + q] = fc * De;
    }
    if (basis == 0) {
        for (int j = 0; j < J; ++j) {
            const double a = x[j] * scale * sp[j] * M_PI / 1.4142135623730951;
            x[j] = a;
            bb[j] = a;
        }
    } else {
        for (int j = 0; j < J; ++j) {
            const double a = x[j] * scale * sp[j] * M_PI / 3.0;
            x[j] = a;
            x[J + j] = a;
            bb[j] = 1.7320508075688772 * a;
            bb[J + j] = 0.0;
        }
    }
}

}  // namespace

// Host side: the spectral grid, its LU factorisation and the (c, d) that go with it.
int pioran_approx_setup_host(int64_t J, int basis, double f_min, double f_max, double S_low, double S_high,
                             std::vector<double>& sp, std::vector<double>& LU, std::vector<int32_t>& piv,
                             std::vector<double>& c, std::vector<double>& d, std::vector<int32_t>& real_term)
{
    if (J < 2 || J > 256 || !(f_min > 0.0) || !(f_max > f_min)) return PIORAN_ERR_ARG;
    const double f0 = f_min / S_low, fM = f_max * S_high;
    sp.resize(J);
    for (int64_t j = 0; j < J; ++j) sp[j] = f0 * std::pow(fM / f0, (double)j / (double)(J - 1));   // :77-79
    const double pw = basis == 0 ? 4.0 : 6.0;
    LU.assign(J * J, 0.0);
    for (int64_t j = 0; j < J; ++j)
        for (int64_t k = 0; k < J; ++k) LU[j * J + k] = 1.0 / (1.0 + std::pow(sp[j] / sp[k], pw));   // :82-96
    piv.resize(J);
    for (int64_t k = 0; k < J; ++k) {   // LU with partial pivoting (dgetrf, as Julia's `\` on a square matrix)
        int64_t p = k;
        double best = std::fabs(LU[k * J + k]);
        for (int64_t i = k + 1; i < J; ++i)
            if (std::fabs(LU[i * J + k]) > best) { best = std::fabs(LU[i * J + k]); p = i; }
        if (best == 0.0) return PIORAN_ERR_ARG;
        piv[k] = (int32_t)p;
        if (p != k)
            for (int64_t q = 0; q < J; ++q) std::swap(LU[k * J + q], LU[p * J + q]);
        for (int64_t i = k + 1; i < J; ++i) {
            const double l = LU[i * J + k] / LU[k * J + k];
            LU[i * J + k] = l;
            for (int64_t q = k + 1; q < J; ++q) LU[i * J + q] -= l * LU[k * J + q];
        }
    }
    if (basis == 0) {       // :250
        c.resize(J); d.resize(J); real_term.assign(J, 0);
        for (int64_t j = 0; j < J; ++j) c[j] = d[j] = std::sqrt(2.0) * M_PI * sp[j];
    } else {                // :266-273
        c.resize(2 * J); d.resize(2 * J); real_term.assign(2 * J, 0);
        for (int64_t j = 0; j < J; ++j) {
            c[j] = M_PI * sp[j];
            d[j] = std::sqrt(3.0) * c[j];
            c[J + j] = 2.0 * c[j];
            d[J + j] = 0.0;
            real_term[J + j] = 1;
        }
    }
    return PIORAN_OK;
}

int pioran_launch_approx(int64_t B, int model, int P, int J, int basis, int integrated, double f_min, double f_max,
                         const double* sp, const double* LU, const int32_t* piv, const double* theta, const double* norm,
                         int n_qpo, const double* qpo, double* A, double* Bc, double* Cq, double* Dq, hipStream_t stream)
{
    const int64_t blocks = (B + 63) / 64;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return PIORAN_ERR_ARG;
    if (n_qpo > 0 && (!qpo || !Cq || !Dq)) return PIORAN_ERR_ARG;
    hipLaunchKernelGGL(approx_kernel, dim3((unsigned)blocks), dim3(64), 0, stream, B, model, P, J, basis, integrated, f_min,
                       f_max, sp, LU, piv, theta, norm, n_qpo, qpo, A, Bc, Cq, Dq);
    return hipGetLastError() == hipSuccess ? PIORAN_OK : PIORAN_ERR_HIP;
}
