"""A few batched evaluations for rocprofv3 --kernel-trace --stats: python tools/trace_batch.py N d K [grad]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import MiGP
from bench import synth_problem, theta_sequence
N, d, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
grad = len(sys.argv) > 4
X, y = synth_problem(N, d, seed=0)
gp = MiGP(X, y, "RBF", need_grad=grad)
th = np.array(theta_sequence(d, 32, seed=0))
for kv in filter(None, os.environ.get("MIGP_OPTS", "").split(",")):
    gp.set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
for i in range(4):
    v = gp.lml_grad_batch(th[:K])[0] if grad else gp.lml_batch(th[:K])
print(N, K, v[:2])
