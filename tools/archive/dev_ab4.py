"""A/B: 64x64-tile routing threshold (option 7)."""
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
    gp.set_option(7, 0)
    ref = gp.lml(theta)
    res = {}
    for rnd in range(3):
        for thr in (0, 128, 256, 384, 512, 768, 1024, 1536):
            gp.set_option(7, thr)
            v = gp.lml(theta)
            assert abs(v - ref) < 1e-9 * abs(ref), (v, ref)
            t0 = time.perf_counter()
            for _ in range(3): gp.lml(theta)
            res.setdefault(thr, []).append((time.perf_counter() - t0) / 3 * 1e3)
    print(N, {thr: round(float(np.median(v)), 2) for thr, v in res.items()})
    gp.close()
