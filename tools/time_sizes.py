"""Per-evaluation latency of the LML and the LML + gradient over the problem sizes of the BASELINE configs."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import MiGP
from bench import synth_problem, theta_sequence

cases = [(128, 2, "RBF"), (1024, 4, "RBF"), (2048, 8, "RBF"), (4096, 8, "RBF"), (8192, 8, "RBF"), (16384, 16, "Matern52")]
if len(sys.argv) > 1:
    cases = [c for c in cases if str(c[0]) in sys.argv[1].split(",")]
out = {}
for N, d, kern in cases:
    X, y = synth_problem(N, d, seed=0)
    gp = MiGP(X, y, kern)
    for kv in os.environ.get("MIGP_OPTS", "").split(","):  # e.g. MIGP_OPTS=13=0,7=2048
        if kv:
            gp.set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
    th = theta_sequence(d, 24, seed=0)
    for i in range(4):
        gp.lml(th[i]); gp.lml_grad(th[i])
    reps = 20 if N <= 8192 else 10
    t0 = time.perf_counter()
    for i in range(reps):
        gp.lml(th[4 + i])
    t1 = time.perf_counter()
    for i in range(reps):
        gp.lml_grad(th[4 + i])
    t2 = time.perf_counter()
    tm = gp.timers()
    out[N] = {"d": d, "kernel": kern, "lml_ms": (t1 - t0) / reps * 1e3, "lml_grad_ms": (t2 - t1) / reps * 1e3,
              "chol_tflops": N ** 3 / 3 / ((t1 - t0) / reps) * 1e-12}
    print(N, json.dumps(out[N]), flush=True)
    gp.close()
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "time_sizes.json"), "w"), indent=1)
