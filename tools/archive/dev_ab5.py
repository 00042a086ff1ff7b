"""A/B: bulk trailing updates at one workgroup per CU once the trailing matrix is <= thr tile columns (option 8)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from andvaranaut_amd import MiGP
from oracle import gp_oracle as orc
for N in (16384, 8192, 4096):
    d = 16
    X, y = orc.synth_problem(N, d, seed=0)
    theta = orc.synth_theta(d)
    gp = MiGP(X, y, "Matern52", need_grad=False)
    ref = gp.lml(theta)
    res = {}
    for rnd in range(3):
        for thr in (0, 32, 48, 64, 80, 96, 128):
            gp.set_option(8, thr)
            v = gp.lml(theta)
            assert abs(v - ref) < 1e-9 * abs(ref), (v, ref)
            t0 = time.perf_counter()
            for _ in range(4): gp.lml(theta)
            res.setdefault(thr, []).append((time.perf_counter() - t0) / 4 * 1e3)
    print(N, "plain launches:", {k: round(float(np.min(v)), 2) for k, v in res.items()}, flush=True)
    gp.close()
