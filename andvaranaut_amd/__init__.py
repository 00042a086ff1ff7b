"""andvaranaut_amd: MI355X-native GP log-marginal-likelihood backend behind andvaranaut's GPMCMC surface."""
from .backend import MiGP, pack_theta, parse_kernel  # noqa: F401
from .gpmcmc import GPMCMC  # noqa: F401
from .transform import affine, logarithm, maxmin, meanstd, normal, stddev, uniform  # noqa: F401
