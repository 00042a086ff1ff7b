"""Latency of small problems (the reference's everyday sizes): LML and LML+gradient per evaluation."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import MiGP
from oracle import gp_oracle as orc
for N in (100, 128, 256, 512, 1024, 2048):
    d = 4
    X, y = orc.synth_problem(N, d, seed=0)
    theta = orc.synth_theta(d)
    gp = MiGP(X, y, "RBF")
    for _ in range(3): gp.lml(theta); gp.lml_grad(theta)
    t0 = time.perf_counter()
    for _ in range(50): gp.lml(theta)
    t1 = time.perf_counter()
    for _ in range(50): gp.lml_grad(theta)
    t2 = time.perf_counter()
    print(f"N={N:5d}: lml {(t1-t0)/50*1e6:8.1f} us   lml+grad {(t2-t1)/50*1e6:8.1f} us", flush=True)
    gp.close()
