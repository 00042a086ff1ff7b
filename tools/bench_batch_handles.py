"""H handles (one host thread each) x K problems per batched call on ONE GPU: aggregate LML evaluations/s.
python tools/bench_batch_handles.py N d kernel [--grad] H1xK1 H2xK2 ..."""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import MiGP
from bench import synth_problem, theta_sequence
grad = "--grad" in sys.argv
args = [a for a in sys.argv[1:] if a != "--grad"]
N, d, kern = int(args[0]), int(args[1]), args[2]
X, y = synth_problem(N, d, seed=0)
th = np.array(theta_sequence(d, 64, seed=0))
for spec in args[3:]:
    H, K = (int(v) for v in spec.split("x"))
    gps = [MiGP(X, y, kern, need_grad=grad) for _ in range(H)]
    f = (lambda g, T: g.lml_grad_batch(T)[0]) if grad else (lambda g, T: g.lml_batch(T))
    for g in gps:
        f(g, th[:K])
    reps = max(3, int(600 / max(N / 1024, 1) ** 2 / K))
    def run(g):
        for i in range(reps):
            f(g, th[(i * K) % 40:(i * K) % 40 + K])
    ts = [threading.Thread(target=run, args=(g,)) for g in gps]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    print(f"N={N} {'LML+grad' if grad else 'LML'} {H} handle(s) x K={K}: {H * K * reps / dt:9.1f} evals/s", flush=True)
    for g in gps:
        g.close()
