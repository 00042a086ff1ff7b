"""A/B: persistent look-ahead bulk kernel on n CUs with half-CU LDS (option 9 = n | 0x1000) + exclusive-CU leaf (option 12)."""
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
    ref = gp.lml(theta)
    res = {}
    for rnd in range(2):
        for graph in (0, 1):
            for wide, excl in ((0, 0), (0x1000 | 248, 1), (0x1000 | 240, 1), (0x1000 | 248, 0), (0x1000 | 224, 1), (248, 1), (1, 1), (0, 1)):
                gp.set_option(3, graph); gp.set_option(9, wide); gp.set_option(10, wide); gp.set_option(12, excl)
                v = gp.lml(theta); v = gp.lml(theta)
                assert abs(v - ref) <= 1e-10 * abs(ref), (v, ref)
                t0 = time.perf_counter()
                for _ in range(4): gp.lml(theta)
                res.setdefault((graph, hex(wide), excl), []).append((time.perf_counter() - t0) / 4 * 1e3)
    print(N, {k: round(min(v), 2) for k, v in res.items()}, flush=True)
    gp.close()
