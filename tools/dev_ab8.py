"""A/B: fixed super-panel width x bulk kernel choice (option 2, option 9), plain launches."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import MiGP
from oracle import gp_oracle as orc
for N in (8192, 16384):
    d = 8
    X, y = orc.synth_problem(N, d, seed=0)
    theta = orc.synth_theta(d)
    gp = MiGP(X, y, "RBF", need_grad=False)
    gp.set_option(3, 0)
    res = {}
    for rnd in range(2):
        for W in (0, 2, 4, 8, 16):
            for wide in (0, 1):
                gp.set_option(2, W); gp.set_option(9, wide)
                gp.lml(theta)
                t0 = time.perf_counter()
                for _ in range(4): gp.lml(theta)
                res.setdefault((W, wide), []).append((time.perf_counter() - t0) / 4 * 1e3)
    print(N, {k: round(min(v), 2) for k, v in res.items()}, flush=True)
    gp.close()
