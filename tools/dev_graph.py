import sys
import numpy as np
sys.path.insert(0, ".")
from andvaranaut_amd import MiGP
from oracle import gp_oracle as orc
for N in (1, 300):
    d = 1
    X, y = orc.synth_problem(max(N,3), d, seed=N); X, y = X[:N], y[:N]
    theta = orc.synth_theta(d, gv=1e-3)
    gp = MiGP(X, y, "RBF", need_grad=False)
    print("N", N, "lml x3:", gp.lml(theta), gp.lml(theta), gp.lml(theta), "ref", orc.lml(X, y, ["RBF"], [], theta))
    print("  factor x3:", gp.factor(theta), gp.factor(theta), gp.factor(theta))
    Xn = np.random.default_rng(0).random((5, d))
    print("  predict:", gp.predict(theta, Xn)[0][:2], "then factor:", gp.factor(theta), gp.factor(theta))
    print("  lml again:", gp.lml(theta), gp.lml(theta))
    gp.close()
