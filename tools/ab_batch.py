"""A/B of whole library BUILDS on the BATCHED entry points, alternating subprocesses (see tools/ab_lib.py):
    python tools/ab_batch.py "4096 8 RBF 8" "4096 8 RBF 16 grad" -- libA.so libB.so        (spec: N d kernel K [grad])"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
WORKER = r'''
import os, sys, time
import numpy as np
lib = os.path.abspath(sys.argv[2])
own = os.path.dirname(os.path.dirname(lib))
sys.path.insert(0, own if os.path.basename(os.path.dirname(lib)) == "andvaranaut_amd" and os.path.exists(os.path.join(own, "bench.py")) else sys.argv[1])
import andvaranaut_amd._lib as L
L.LIB_PATH = lib
from andvaranaut_amd import MiGP
from bench import synth_problem, theta_sequence
N, d, kern, K = int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], int(sys.argv[6])
grad = len(sys.argv) > 7 and sys.argv[7] == "grad"
X, y = synth_problem(N, d, seed=0)
gp = MiGP(X, y, kern, need_grad=grad)
th = np.array(theta_sequence(d, 64, seed=0))
f = (lambda T: gp.lml_grad_batch(T)[0]) if grad else gp.lml_batch
for i in range(2):
    v = f(th[:K])
res = []
reps = max(2, int(40 / K / max(N / 4096, 1) ** 3))
for rnd in range(5):
    t0 = time.perf_counter()
    for i in range(reps):
        v = f(th[(i * K) % 48:(i * K) % 48 + K])
    res.append(K * reps / (time.perf_counter() - t0))
print(np.median(res), float(v[0]))
'''


def main():
    args = sys.argv[1:]
    specs, libs = args[: args.index("--")], args[args.index("--") + 1:]
    for spec in specs:
        res = {l: [] for l in libs}
        vals = {}
        for rnd in range(3):
            for l in libs:
                out = subprocess.run([sys.executable, "-c", WORKER, ROOT, l] + spec.split(), capture_output=True, text=True)
                if out.returncode != 0:
                    print(out.stderr[-500:])
                    continue
                t, v = out.stdout.strip().splitlines()[-1].split()
                res[l].append(float(t))
                vals[l] = v
        for l in libs:
            print(f"{spec:>24s}  {os.path.basename(l):24s} median {np.median(res[l]):9.1f} evals/s   runs {['%.1f' % x for x in res[l]]}  value {vals.get(l)}", flush=True)


if __name__ == "__main__":
    main()
