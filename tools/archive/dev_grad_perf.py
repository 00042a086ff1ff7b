import sys, time
import numpy as np
sys.path.insert(0, ".")
from andvaranaut_amd import MiGP
from oracle import gp_oracle as orc
for (N, d, kern) in [(16384, 16, "Matern52"), (8192, 8, "RBF"), (4096, 8, "RBF")]:
    X, y = orc.synth_problem(N, d, seed=0)
    theta = orc.synth_theta(d)
    gp = MiGP(X, y, kern)
    gp.lml_grad(theta); gp.lml(theta)
    gp.set_profiling(1)
    gp.lml_grad(theta); tm = gp.timers()
    gp.set_profiling(0)
    t0 = time.perf_counter()
    for _ in range(5): gp.lml_grad(theta)
    tg = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    for _ in range(5): gp.lml(theta)
    tl = (time.perf_counter() - t0) / 5
    print(f"N={N} d={d} {kern}: lml {tl*1e3:.2f} ms ({1/tl:.1f}/s)  lml+grad {tg*1e3:.2f} ms ({1/tg:.1f}/s)  phases: chol {tm['cholesky_ms']:.1f} trtri {tm['trtri_ms']:.1f} lauum {tm['lauum_ms']:.1f} contract {tm['contract_ms']:.1f}")
    gp.close()
