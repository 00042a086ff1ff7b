"""Run a few LML (and LML+gradient) evaluations at one size, for rocprofv3 --kernel-trace --stats."""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import MiGP
from oracle import gp_oracle as orc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
d = int(sys.argv[2]) if len(sys.argv) > 2 else 8
what = sys.argv[3] if len(sys.argv) > 3 else "lml"
X, y = orc.synth_problem(N, d, seed=0)
theta = orc.synth_theta(d)
gp = MiGP(X, y, "RBF", need_grad=(what != "lml"))
for kv in filter(None, os.environ.get("MIGP_OPTS", "").split(",")):  # e.g. MIGP_OPTS=0=2,16=0
    gp.set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
import time
for i in range(8):
    t0 = time.perf_counter()
    v = gp.lml(theta) if what == "lml" else gp.lml_grad(theta)[0]
    dt = time.perf_counter() - t0
print(N, what, v, f"{dt*1e3:.2f} ms")
gp.close()
