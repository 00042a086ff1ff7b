"""A/B of the tail panel width (W = 2 below thr2 tile columns), plain launches for every arm."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import MiGP
from oracle import gp_oracle as orc
for N in (4096, 8192, 16384):
    d = 8
    X, y = orc.synth_problem(N, d, seed=0)
    theta = orc.synth_theta(d)
    gp = MiGP(X, y, "RBF", need_grad=False)
    ref = gp.lml(theta)
    configs = [(72, 0), (72, 16), (72, 32), (72, 64), (72, 72), (40, 0), (40, 40)]
    res = {c: [] for c in configs}
    for rnd in range(3):
        for c in configs:
            gp.set_option(5, c[0]); gp.set_option(6, c[1])
            v = gp.lml(theta)
            assert abs(v - ref) < 1e-9 * abs(ref), (v, ref)
            t0 = time.perf_counter()
            for _ in range(4): gp.lml(theta)
            res[c].append((time.perf_counter() - t0) / 4 * 1e3)
    print(N, {c: round(min(v), 3) for c, v in res.items()}, flush=True)
    gp.close()
