"""Aggregate LML(+gradient) evaluations/s of T handles driven by T host threads on ONE GPU: python tools/dev_concurrent.py N d [grad]"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import MiGP
from bench import synth_problem, theta_sequence
N, d = int(sys.argv[1]), int(sys.argv[2])
grad = len(sys.argv) > 3
th = theta_sequence(d, 8, seed=0)
reps = 20 if N <= 8192 else 8
for T in [int(t) for t in os.environ.get("HANDLES", "1,2,3,4").split(",")]:
    gps = []
    for t in range(T):
        X, y = synth_problem(N, d, seed=t)
        gps.append(MiGP(X, y, "RBF", need_grad=grad))
        for kv in filter(None, os.environ.get("MIGP_OPTS", "").split(",")):  # e.g. MIGP_OPTS=0=0: single-stream handles
            gps[-1].set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
    def work(gp):
        f = (lambda t_: gp.lml_grad(t_)[0]) if grad else gp.lml
        for i in range(reps):
            f(th[i % 8])
    for gp in gps:
        (gp.lml_grad(th[0]) if grad else gp.lml(th[0]))
    ths = [threading.Thread(target=work, args=(gp,)) for gp in gps]
    t0 = time.perf_counter()
    for t_ in ths: t_.start()
    for t_ in ths: t_.join()
    dt = time.perf_counter() - t0
    print(f"N={N} {'lml+grad' if grad else 'lml'} handles={T}: {T * reps / dt:8.1f} evals/s aggregate, {dt / reps * 1e3:7.3f} ms per round", flush=True)
    for gp in gps: gp.close()
