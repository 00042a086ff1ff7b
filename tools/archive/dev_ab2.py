"""Interleaved A/B of super-panel width schedules."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from andvaranaut_amd import MiGP
from oracle import gp_oracle as orc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
d = 16
X, y = orc.synth_problem(N, d, seed=0)
theta = orc.synth_theta(d)
gp = MiGP(X, y, "Matern52", need_grad=False)
ref = gp.lml(theta)
configs = [(1 << 20, 72, 0), (1 << 20, 72, 16), (1 << 20, 72, 32), (1 << 20, 96, 0), (1 << 20, 48, 0), (1 << 20, 56, 24), (1 << 20, 40, 0), (1 << 20, 64, 0)]
res = {c: [] for c in configs}
for rnd in range(3):
    for c in configs:
        for i in range(3): gp.set_option(4 + i, c[i])
        v = gp.lml(theta)
        assert abs(v - ref) < 1e-9 * abs(ref), (v, ref)
        t0 = time.perf_counter()
        for _ in range(3): gp.lml(theta)
        res[c].append((time.perf_counter() - t0) / 3 * 1e3)
for c in configs:
    print(f"thresholds {c}: median {np.median(res[c]):.2f} ms  min {min(res[c]):.2f}")
