"""A/B: look-ahead bulk updates on the 8-wave one-workgroup-per-CU GEMM (option 9) with graph on/off."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import MiGP
from oracle import gp_oracle as orc
for N in (1000, 4096, 8192, 16384):
    d = 8
    X, y = orc.synth_problem(N, d, seed=0)
    theta = orc.synth_theta(d)
    gp = MiGP(X, y, "RBF", need_grad=True)
    res = {}
    vals = {}
    for rnd in range(3):
        for graph in (0, 1):
            for split in (0, 1):
                gp.set_option(3, graph); gp.set_option(9, split)
                v = gp.lml(theta); v = gp.lml(theta)
                vals[(graph, split)] = v
                t0 = time.perf_counter()
                for _ in range(5): gp.lml(theta)
                res.setdefault((graph, split), []).append((time.perf_counter() - t0) / 5 * 1e3)
    g0 = gp.lml_grad(theta)
    print(N, {k: round(min(v), 3) for k, v in res.items()}, "max rel diff:", max(abs(v - vals[(0, 0)]) / abs(vals[(0, 0)]) for v in vals.values()), flush=True)
    gp.close()
