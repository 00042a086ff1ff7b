"""A/B: CU-masked bulk stream (option 8 = reserved CUs) x exclusive-CU leaf (option 9) x hipGraph (option 3)."""
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
    gp.set_option(3, 0)
    ref = gp.lml(theta)
    res = {}
    for rnd in range(2):
        for graph in (0, 1):
            for rsv, excl in ((0, 0), (8, 0), (8, 1), (16, 1), (32, 1), (64, 1)):
                gp.set_option(3, graph); gp.set_option(8, rsv); gp.set_option(9, excl)
                v = gp.lml(theta); v = gp.lml(theta)
                assert abs(v - ref) < 1e-9 * abs(ref), (v, ref)
                t0 = time.perf_counter()
                for _ in range(4): gp.lml(theta)
                res.setdefault((graph, rsv, excl), []).append((time.perf_counter() - t0) / 4 * 1e3)
    print(N, {k: round(float(np.min(v)), 2) for k, v in res.items()}, flush=True)
    gp.close()
