"""Assembly kernel time per covariance family and input dimension (device timers)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import MiGP
from oracle import gp_oracle as orc
N = 16384
for d in (2, 16, 64):
    X, y = orc.synth_problem(N, d, seed=0)
    for kern in ("RBF", "Matern52", "Matern32", "Exponential", "RatQuad", "RBF+Matern52"):
        nk = kern.count("+") + 1
        theta = orc.synth_theta(d, nkern=nk)
        gp = MiGP(X, y, kern, need_grad=False)
        gp.set_profiling(1)
        gp.lml(theta); gp.lml(theta)
        t = gp.timers()["assemble_ms"]
        gb = (4.0 * N * (N + 1) + 8.0 * N * d) / 1e9
        print(f"d={d:3d} {kern:14s} assemble {t:.3f} ms  {gb / t:.2f} TB/s (lower triangle)", flush=True)
        gp.close()
