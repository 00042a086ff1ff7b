"""A factorisation and a few sweeps of M prediction points at one size, for rocprofv3 --kernel-trace --stats (K8 kernels)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import MiGP
from bench import synth_problem
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
d = int(sys.argv[2]) if len(sys.argv) > 2 else 16
M = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
X, y = synth_problem(N, d, seed=0)
theta = np.concatenate([np.exp(np.linspace(np.log(0.4), np.log(1.5), d)), [1.7], [1.0], [1e-4, 1e-6]])
gp = MiGP(X, y, "Matern52")
Xn = np.random.default_rng(0).uniform(0, 1, (M, d))
out = {}
for via in (False, True):
    gp.predict(theta, Xn, via_inverse=via)
    t0 = time.perf_counter()
    for _ in range(3):
        mu, var = gp.predict(theta, Xn, via_inverse=via)
    out["via_U" if via else "triangular_solve"] = (time.perf_counter() - t0) / 3 * 1e3
print(N, d, M, {k: round(v, 2) for k, v in out.items()}, "ms per sweep;", float(mu[0]), float(var[0]))
gp.close()
