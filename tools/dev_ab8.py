"""A/B: CU count of the persistent look-ahead bulk kernel early (option 9) / late (option 10) with the switch at
option 11 remaining tile columns.  Plain launches."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import MiGP
from oracle import gp_oracle as orc
for N in (16384, 8192):
    d = 8
    X, y = orc.synth_problem(N, d, seed=0)
    theta = orc.synth_theta(d)
    gp = MiGP(X, y, "RBF", need_grad=False)
    gp.set_option(3, 0)
    ref = gp.lml(theta)
    res = {}
    for rnd in range(2):
        for early in (224, 232):
            for late, thr in ((224, 0), (192, 64), (192, 80), (160, 48), (160, 64), (128, 48), (128, 32), (192, 96), (160, 80)):
                gp.set_option(9, early); gp.set_option(10, late); gp.set_option(11, thr)
                v = gp.lml(theta); v = gp.lml(theta)
                assert abs(v - ref) <= 1e-10 * abs(ref), (v, ref)
                t0 = time.perf_counter()
                for _ in range(4): gp.lml(theta)
                res.setdefault((early, late, thr), []).append((time.perf_counter() - t0) / 4 * 1e3)
    print(N, {k: round(min(v), 2) for k, v in res.items()}, flush=True)
    gp.close()
