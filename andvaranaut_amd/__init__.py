"""andvaranaut_amd: MI355X-native GP log-marginal-likelihood backend behind andvaranaut's GPMCMC surface."""
from .backend import MiGP, pack_theta, parse_kernel  # noqa: F401
from .gpmcmc import GPMCMC, load_object, save_object  # noqa: F401
from .transform import (affine, arcsinh, boxcox, boxcoxf, kumaraswamy, logarithm, maxmin, meanstd, minshift,  # noqa: F401
                        normal, preserve_zero, sal, sinharcsinh, stddev, stdshift, uniform, wgp)
