"""Sweep timing: blocked triangular solve vs one GEMM with U = L^-T (mi_gp_predict vs mi_gp_predict_u)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import MiGP
from oracle import gp_oracle as orc
for N, d in ((4000, 6), (16384, 16)):
    X, y = orc.synth_problem(N, d, seed=0)
    theta = orc.synth_theta(d)
    gp = MiGP(X, y, "Matern52")
    for M in (96, 1000, 10000):
        Xn = np.random.default_rng(0).uniform(0, 1, (M, d))
        gp.predict(theta, Xn, via_inverse=False); gp.predict(theta, Xn, via_inverse=True)
        t0 = time.perf_counter(); [gp.predict(theta, Xn, via_inverse=False) for _ in range(3)]; a = (time.perf_counter() - t0) / 3
        t0 = time.perf_counter(); [gp.predict(theta, Xn, via_inverse=True) for _ in range(3)]; b = (time.perf_counter() - t0) / 3
        print(f"N={N} M={M}: triangular solve {a*1e3:.2f} ms | via U {b*1e3:.2f} ms", flush=True)
    gp.factor(theta)
    t0 = time.perf_counter(); gp.predict(theta, Xn[:128], via_inverse=True); c = time.perf_counter() - t0
    print(f"N={N}: first sweep after a new factorisation (forms U) {c*1e3:.2f} ms", flush=True)
    gp.close()
