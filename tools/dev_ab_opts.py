"""A/B of option sets in ONE process, interleaved: python tools/dev_ab_opts.py N d kernel "8=64" "8=200" ...
Each set: rounds x reps LML evaluations (and LML + gradient with --grad), medians reported."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import MiGP
from bench import synth_problem, theta_sequence
args = [a for a in sys.argv[1:] if a != "--grad" and not a.startswith("--batch")]
grad = "--grad" in sys.argv
batch = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("--batch=")), 0)  # --batch=K: the batched entry points, ms per call
N, d, kern = int(args[0]), int(args[1]), args[2]
sets = args[3:]
X, y = synth_problem(N, d, seed=0)
gp = MiGP(X, y, kern, need_grad=grad)
th = theta_sequence(d, max(8, 8 * batch), seed=0)
DEFAULTS = {0: 1, 2: 0, 4: 1 << 20, 5: 0, 6: 0, 7: 1024, 8: 1 << 20, 9: 1, 14: 8, 16: 1, 18: 1536, 19: 1024, 20: 72, 21: 8, 26: 2, 30: 16, 31: 48, 32: 2048, 35: 32, 37: 24, 38: 8, 45: 1, 46: 31, 47: 2000}
res = {s: [] for s in sets}
vals = {}
enq = {}
reps = 10 if N <= 8192 else 5
f = (lambda t: gp.lml_grad(t)[0]) if grad else gp.lml
if batch:
    tb = [np.array(th[i * batch:(i + 1) * batch]) for i in range(8)]
    th = tb
    f = (lambda T: gp.lml_grad_batch(T)[0][0]) if grad else (lambda T: gp.lml_batch(T)[0])
for rnd in range(4):
    for s in sets:
        for k, v in DEFAULTS.items():
            gp.set_option(k, v)
        for kv in s.split(","):
            if kv and kv != "default":
                gp.set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
        vals[s] = f(th[0])
        t0 = time.perf_counter()
        for i in range(reps):
            f(th[i % 8])
        res[s].append((time.perf_counter() - t0) / reps * 1e3)
        enq[s] = gp.timers().get("enqueue_ms", 0.0)
for s in sets:
    v = sorted(res[s])
    print(f"N={N} {kern} {'lml+grad' if grad else 'lml'}{' K=%d' % batch if batch else ''} [{s:>14s}] median {np.median(v):8.3f} ms  min {v[0]:8.3f}  max {v[-1]:8.3f}  enqueue {enq[s]:6.3f} ms  value {vals[s]!r}", flush=True)
