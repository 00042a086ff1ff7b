import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from oracle import gp_oracle as orc
from andvaranaut_amd import MiGP
import test_gpu_random_sweep as t
for seed in [int(a) for a in sys.argv[1:]] or [17]:
    rng = np.random.default_rng(1000 + seed)
    N, d, kerns, ops, M = t._random_case(rng)
    kernel = kerns[0] + "".join(o + k for o, k in zip(ops, kerns[1:]))
    X, y = orc.synth_problem(max(N, 3), d, seed=seed); X, y = X[:N], y[:N]
    theta = orc.synth_theta(d, nkern=len(kerns), gv=10.0 ** rng.uniform(-5, -2))
    theta[: len(kerns) * d] *= rng.uniform(0.7, 1.6, len(kerns) * d)
    gp = MiGP(X, y, kernel)
    v2, g, gy, gx = gp.lml_grad_data(theta)
    _, rgy, rgx = orc.lml_grad_data(X, y, kerns, ops, theta)
    scale = np.maximum(np.abs(rgx), 1e-3 * np.abs(rgx).max())
    err = np.abs(gx - rgx) / scale
    i = np.unravel_index(np.argmax(err), err.shape)
    print(seed, kernel, N, d, "max rel err", err.max(), "at", i, "gx", gx[i], "ref", rgx[i], "max|ref|", np.abs(rgx).max())
    print("  errs > 1e-6:", int((err > 1e-6).sum()), "of", err.size, " median", np.median(err))
    # nearest-neighbour distance of the worst point
    dd = np.sqrt(((X - X[i[0]]) ** 2).sum(1)); dd[i[0]] = 9
    print("  nearest neighbour of point", i[0], "at distance", dd.min())
    gp.close()
