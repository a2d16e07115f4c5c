"""pioran.jl_amd — MI355X-native ScalableGP log-likelihood hot path (drop-in for that path of Pioran.jl).

Directory name has a dot, so import it through the repo-root shim: `import pioran_jl_amd as pj`.
"""
from . import _lib, farm  # noqa: F401
from .gp import (Context, CustomMean, Dataset, Farm, FiniteScalableGP, PosteriorGP, ScalableGP, cov, default_context,
                 log_likelihood, log_likelihood_direct, logl, logpdf, logpdf_batch, mean, posterior, predict, predict_cov, predict_direct,
                 rand, rand_posterior, simulate, std)
from .kernels import (CARMA, Celerite, Exp, ScaledKernel, SemiSeparable, SHO, SumOfCelerite, SumOfSemiSeparable,
                      SumOfTerms, celerite_coefs)
from .psd import (QPO, DoubleBendingPowerLaw, SingleBendingPowerLaw, approx, approx_batch, approx_batch_vjp, build_approx,
                  convert_feature, get_approx_coefficients, get_norm_psd, psd_decomp, separate_psd)

__all__ = [n for n in dir() if not n.startswith("_")]
