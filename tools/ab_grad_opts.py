"""A/B of option sets on LML + gradient evaluations in one process (interleaved), with bit comparison of the gradients.
    python tools/ab_grad_opts.py N d kernel "30=0" "30=16" ..."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import MiGP
from bench import synth_problem, theta_sequence
N, d, kern = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
sets = sys.argv[4:]
X, y = synth_problem(N, d, seed=0)
gp = MiGP(X, y, kern)
th = theta_sequence(d, 8, seed=0)
res, grads = {s: [] for s in sets}, {}
reps = 6 if N <= 8192 else 3
for rnd in range(4):
    for s in sets:
        for kv in s.split(","):
            gp.set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
        grads[s] = gp.lml_grad(th[0])
        t0 = time.perf_counter()
        for i in range(reps):
            gp.lml_grad(th[i % 8])
        res[s].append((time.perf_counter() - t0) / reps * 1e3)
for s in sets:
    same = grads[s][0] == grads[sets[0]][0] and np.array_equal(grads[s][1], grads[sets[0]][1])
    print(f"N={N} {kern} lml+grad [{s:>12s}] median {np.median(res[s]):8.3f} ms  min {min(res[s]):8.3f}  bits equal to [{sets[0]}]: {same}", flush=True)
