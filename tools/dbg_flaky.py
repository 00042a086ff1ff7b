import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from andvaranaut_amd import _lib
if len(sys.argv) > 1 and sys.argv[1].endswith(".so"):
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from oracle import gp_oracle as orc
from andvaranaut_amd import MiGP
import test_gpu_random_sweep as t
seed = 17
rng = np.random.default_rng(1000 + seed)
N, d, kerns, ops, M = t._random_case(rng)
kernel = kerns[0] + "".join(o + k for o, k in zip(ops, kerns[1:]))
X, y = orc.synth_problem(max(N, 3), d, seed=seed); X, y = X[:N], y[:N]
theta = orc.synth_theta(d, nkern=len(kerns), gv=10.0 ** rng.uniform(-5, -2))
theta[: len(kerns) * d] *= rng.uniform(0.7, 1.6, len(kerns) * d)
_, rg = orc.lml_grad(X, y, kerns, ops, theta)
_, rgy, rgx = orc.lml_grad_data(X, y, kerns, ops, theta)
scale = np.maximum(np.abs(rgx), 1e-3 * np.abs(rgx).max())
for rep in range(int(os.environ.get("REPS", "12"))):
    gp = MiGP(X, y, kernel)
    out = []
    for inner in range(3):
        v2, g, gy, gx = gp.lml_grad_data(theta)
        out.append((float(np.max(np.abs(gx - rgx) / scale)), float(np.max(np.abs(g - rg)) / np.abs(rg).max()), float(np.max(np.abs(gy - rgy)) / np.abs(rgy).max())))
    print(rep, " ".join(f"gx {a:.2e} g {b:.2e} gy {c:.2e} |" for a, b, c in out), flush=True)
    gp.close()
