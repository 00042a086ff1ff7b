import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import MiGP
from oracle import gp_oracle as orc
def t(fn, reps=20):
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e3
for N in (1024, 2048, 512):
    X, y = orc.synth_problem(N, 4, seed=0)
    theta = orc.synth_theta(4)
    gp = MiGP(X, y, "RBF")
    for _ in range(3): gp.lml(theta)
    a = t(lambda: gp.lml(theta))
    for _ in range(3): gp.lml_grad(theta)
    b = t(lambda: gp.lml_grad(theta))
    c = t(lambda: gp.lml(theta))
    gp.set_option(3, 0)
    d = t(lambda: gp.lml(theta)); e = t(lambda: gp.lml_grad(theta))
    gp.set_option(3, 1)
    f = t(lambda: gp.lml(theta))
    print(f"N={N}: lml graph {a:.3f} ms | grad graph {b:.3f} | lml graph again {c:.3f} | plain lml {d:.3f} grad {e:.3f} | lml graph after plain {f:.3f}", flush=True)
    gp.close()
