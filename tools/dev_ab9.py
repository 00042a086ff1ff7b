"""A/B: tail bulk kernel = variant B one-per-CU (option 8) vs variant C (options 10 = 1, 11 = thr).  Plain launches."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import MiGP
from oracle import gp_oracle as orc
for N in (16384, 8192, 4096):
    d = 8
    X, y = orc.synth_problem(N, d, seed=0)
    theta = orc.synth_theta(d)
    gp = MiGP(X, y, "RBF", need_grad=False)
    gp.set_option(3, 0)
    res = {}
    for rnd in range(3):
        for cfg in ((64, 0, 0), (0, 0, 0), (0, 1, 48), (0, 1, 64), (0, 1, 96), (0, 1, 128), (0, 1, 1 << 20)):
            gp.set_option(8, cfg[0]); gp.set_option(10, cfg[1]); gp.set_option(11, cfg[2])
            gp.lml(theta)
            t0 = time.perf_counter()
            for _ in range(4): gp.lml(theta)
            res.setdefault(cfg, []).append((time.perf_counter() - t0) / 4 * 1e3)
    print(N, {k: round(min(v), 3) for k, v in res.items()}, flush=True)
    gp.close()
