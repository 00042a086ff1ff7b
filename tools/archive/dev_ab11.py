"""A/B: W = 16 super-panels while more than thr0 tile columns remain (option 4).  Plain launches."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import MiGP
from oracle import gp_oracle as orc
N, d = 16384, 8
X, y = orc.synth_problem(N, d, seed=0)
theta = orc.synth_theta(d)
gp = MiGP(X, y, "RBF", need_grad=False)
ref = gp.lml(theta)
res = {}
for rnd in range(3):
    for thr0 in (1 << 20, 120, 112, 96, 80, 64):
        gp.set_option(4, thr0)
        v = gp.lml(theta)
        assert abs(v - ref) <= 1e-10 * abs(ref)
        t0 = time.perf_counter()
        for _ in range(4): gp.lml(theta)
        res.setdefault(thr0, []).append((time.perf_counter() - t0) / 4 * 1e3)
print(N, {k: round(min(v), 3) for k, v in res.items()}, flush=True)
