"""Interleaved A/B of Cholesky options in one process (same box, same clocks)."""
import sys, time, itertools
import numpy as np
sys.path.insert(0, ".")
from andvaranaut_amd import MiGP
from oracle import gp_oracle as orc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
d = 16
X, y = orc.synth_problem(N, d, seed=0)
theta = orc.synth_theta(d)
gp = MiGP(X, y, "Matern52", need_grad=False)
gp.lml(theta)
configs = [(la, var, w, g) for la in (1,) for var in (0, 1) for w in (4, 8) for g in (0, 1)]
res = {c: [] for c in configs}
for rnd in range(3):
    for c in configs:
        gp.set_option(0, c[0]); gp.set_option(1, c[1]); gp.set_option(2, c[2]); gp.set_option(3, c[3])
        gp.lml(theta)
        t0 = time.perf_counter()
        for _ in range(3): gp.lml(theta)
        res[c].append((time.perf_counter() - t0) / 3 * 1e3)
for c in configs:
    print(f"lookahead={c[0]} variant={'AB'[c[1]]} W={c[2]} graph={c[3]}: median {np.median(res[c]):.2f} ms  min {min(res[c]):.2f}")
