"""Phase times of one LML + gradient evaluation (HIP events on the handle's stream): python tools/dev_grad_phases.py N d"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import MiGP
from bench import synth_problem, theta_sequence
N, d = int(sys.argv[1]), int(sys.argv[2])
X, y = synth_problem(N, d, seed=0)
gp = MiGP(X, y, "RBF", need_grad=True)
th = theta_sequence(d, 4, seed=0)
gp.lml_grad(th[0])
gp.set_profiling(1)
for i in range(3):
    gp.lml_grad(th[i])
    tm = gp.timers()
print(N, {k: round(v, 3) for k, v in tm.items() if k.endswith("_ms")})
f = N ** 3 / 3.0
print(f"  U = L^-T: {f / (tm['trtri_ms'] * 1e-3) * 1e-12:.1f} TFLOP/s,  K^-1 = U U^T: {f / (tm['lauum_ms'] * 1e-3) * 1e-12:.1f} TFLOP/s")
