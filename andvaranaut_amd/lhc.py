"""Latin-hypercube sampling of the input priors, the recipe of LHC.__latin_sample (lhc.py:40-47):
scipy.stats.qmc.LatinHypercube(d, optimization="random-cd") in the unit cube, then each prior's ppf.
The reference accepts ``seed`` and ignores it (lhc.py:40-43); here it is passed
to the sampler so that benchmarks are reproducible, and ``seed=None`` behaves like the reference."""
import numpy as np
from scipy.stats import qmc


def latin_sample(priors, nsamps, seed=None, optimization="random-cd"):
    sampler = qmc.LatinHypercube(d=len(priors), optimization=optimization, seed=seed)
    points = sampler.random(n=nsamps)
    xsamps = np.zeros_like(points)
    for j, prior in enumerate(priors):
        xsamps[:, j] = prior.ppf(points[:, j])
    return xsamps
