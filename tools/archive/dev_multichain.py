"""Aggregate LML+gradient evaluations/s with k independent handles (chains) driven by k host threads on ONE GPU."""
import sys, os, time, threading
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import MiGP
from oracle import gp_oracle as orc
for N, d in ((2048, 4), (4096, 8), (8192, 8), (16384, 16)):
    X, y = orc.synth_problem(N, d, seed=0)
    theta = orc.synth_theta(d)
    for k in (1, 2, 3, 4):
        if N == 16384 and k > 2: continue
        gps = [MiGP(X, y, "RBF") for _ in range(k)]
        for g in gps:
            for _ in range(2): g.lml_grad(theta)
        reps = 20 if N <= 4096 else 8
        def work(g):
            for _ in range(reps): g.lml_grad(theta)
        ths = [threading.Thread(target=work, args=(g,)) for g in gps]
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        dt = time.perf_counter() - t0
        print(f"N={N} chains={k}: {k*reps/dt:8.1f} LML+grad evals/s aggregate ({dt/reps*1e3:.2f} ms per round)", flush=True)
        for g in gps: g.close()
