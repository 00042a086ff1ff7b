"""Input dimensions beyond 128: LML, gradient and conditional against the oracle; the data-side gradients refuse loudly."""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
from andvaranaut_amd import MiGP
from oracle import gp_oracle as orc
for d in (129, 200, 513):
    for kernel in ("RBF", "Matern52+RBF"):
        kerns = kernel.split("+"); ops = ["+"] * (len(kerns) - 1)
        X, y = orc.synth_problem(300, d, seed=d)
        theta = orc.synth_theta(d, nkern=len(kerns))
        theta[: len(kerns) * d] *= 6.0  # keep K away from the identity in high dimension
        gp = MiGP(X, y, kernel)
        v, g = gp.lml_grad(theta)
        rv, rg = orc.lml_grad(X, y, kerns, ops, theta)
        sc = np.maximum(np.abs(rg), 1e-3 * np.abs(rg).max())
        Xn = np.random.default_rng(0).random((7, d))
        mu, var = gp.predict(theta, Xn)
        rmu, rvar = orc.predict(X, y, Xn, kerns, ops, theta)
        msg = ""
        try:
            gp.lml_grad_data(theta)
        except RuntimeError as e:
            msg = str(e)[:60]
        try:
            gp.predict_grad(theta, Xn[:2])
        except RuntimeError as e:
            msg += " | " + str(e)[:60]
        print(d, kernel, abs(v - rv) / abs(rv), np.max(np.abs(g - rg) / sc), np.abs(mu - rmu).max(), np.abs(var - rvar).max(), msg)
        gp.close()
