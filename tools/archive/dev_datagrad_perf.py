"""Cost of the data-side gradient entry points on top of an LML+gradient evaluation."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import MiGP
from oracle import gp_oracle as orc
for N, d in ((16384, 16), (8192, 8), (2048, 4)):
    X, y = orc.synth_problem(N, d, seed=0)
    theta = orc.synth_theta(d)
    gp = MiGP(X, y, "Matern52")
    for _ in range(2): gp.lml_grad_data(theta)
    t0 = time.perf_counter()
    for _ in range(3): gp.lml_grad(theta)
    t1 = time.perf_counter()
    for _ in range(3): gp.lml_grad_data(theta, want_x=False)
    t2 = time.perf_counter()
    for _ in range(3): gp.lml_grad_data(theta, want_x=True)
    t3 = time.perf_counter()
    xn = np.random.default_rng(0).uniform(0.1, 0.9, (1, d))
    gp.predict_grad(theta, xn)
    t4 = time.perf_counter()
    for _ in range(5): gp.predict_grad(theta, xn, refactor=False)
    t5 = time.perf_counter()
    print(f"N={N}: lml_grad {(t1-t0)/3*1e3:.2f} ms | +alpha {(t2-t1)/3*1e3:.2f} | +alpha+grad_x {(t3-t2)/3*1e3:.2f} | predict_grad(1 pt, resident factor) {(t5-t4)/5*1e3:.2f} ms", flush=True)
    gp.close()
