// Dense solver path (log_likelihood_direct, src/direct_solver.jl:6-21) — placeholder until the
// covariance build + MFMA Cholesky land; the entry points exist so the ABI is complete and they
// fail loudly.
#include "../../include/pioran_hip.h"
#include "common.h"

extern "C" {
int pioran_dense_nll(pioran_ctx*, int64_t, int64_t, const double*, const double*, const double*, const double*,
                     const double*, const double*, const double*, double*, int32_t*)
{
    return PIORAN_ERR_UNSUPPORTED;
}
int pioran_dense_covariance(pioran_ctx*, int64_t, int64_t, const double*, const double*, const double*,
                            const double*, const double*, const double*, double*)
{
    return PIORAN_ERR_UNSUPPORTED;
}
}
