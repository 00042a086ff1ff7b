"""Stress: create / use / destroy handles of several sizes in one process (graph capture on every new handle)."""
import faulthandler, os, sys, time
faulthandler.enable()
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import _lib
if os.environ.get("MIGP_LIB"):
    _lib.LIB_PATH = os.environ["MIGP_LIB"]
from andvaranaut_amd import MiGP
from bench import synth_problem, theta_sequence
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
data = {N: synth_problem(N, 8, seed=0) for N in (1024, 2048, 4096, 8192, 16384)}
for it in range(rounds):
    for N in (2048, 4096, 8192, 16384, 1024):
        X, y = data[N]
        gp = MiGP(X, y, "RBF")
        th = theta_sequence(8, 6, seed=it)
        vals = []
        for i in range(3):
            vals.append(gp.lml(th[i]))
            vals.append(gp.lml_grad(th[i])[0])
        assert all(np.isfinite(v) for v in vals)
        tm = gp.timers()
        print(it, N, f"{vals[0]:.6f}", flush=True)
        gp.close()
print("stress ok")
