"""Host-side cost per posterior evaluation at small N: device call alone vs the whole logp_dlogp (priors, transforms, ctypes)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import MiGP
from andvaranaut_amd.priors import HyperModel
from bench import synth_problem
N, d = int(sys.argv[1]), int(sys.argv[2])
X, y = synth_problem(N, d, seed=0)
gp = MiGP(X, y, "RBF", need_grad=True)
model = HyperModel(d, ["RBF"], noise=True)
q = model.initial_point()
f = lambda q_: model.logp_dlogp(q_, gp.lml_grad)
f(q)
reps = 2000
t0 = time.perf_counter()
for _ in range(reps): f(q)
t1 = time.perf_counter()
th = np.concatenate([np.full(d, 0.7), [1.7], [1.0], [1e-4], [1e-6]])
gp.lml_grad(th)
t2 = time.perf_counter()
for _ in range(reps): gp.lml_grad(th)
t3 = time.perf_counter()
for _ in range(reps): gp.lml(th)
t4 = time.perf_counter()
print(f"N={N} d={d}: logp_dlogp {1e6*(t1-t0)/reps:.1f} us, MiGP.lml_grad {1e6*(t3-t2)/reps:.1f} us, MiGP.lml {1e6*(t4-t3)/reps:.1f} us")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(500): f(q)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
