"""Time one library build: python tools/dev_ab_lib.py LIB N d kernel.  Run alternately (A B A B) from a
shell loop to compare two builds of libmi_gp.so on the same box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import _lib
if sys.argv[1] != "default":
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from andvaranaut_amd import MiGP
from bench import synth_problem, theta_sequence
N, d, kern = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
X, y = synth_problem(N, d, seed=0)
gp = MiGP(X, y, kern, need_grad=False)
th = theta_sequence(d, 8, seed=0)
ts = []
for rnd in range(6):
    gp.lml(th[0])
    t0 = time.perf_counter()
    for i in range(10):
        gp.lml(th[i % 8])
    ts.append((time.perf_counter() - t0) / 10 * 1e3)
print(f"{os.path.basename(sys.argv[1]):>20s} N={N} median {np.median(ts):.3f} ms min {min(ts):.3f}", flush=True)
