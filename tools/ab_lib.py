"""A/B of whole library BUILDS on one box: alternating subprocesses, one per (library, round).
    python tools/ab_lib.py "4096 8 RBF" "8192 8 RBF" -- tools/ab/r05/andvaranaut_amd/libmi_gp.so andvaranaut_amd/libmi_gp.so
(a library given as <root>/andvaranaut_amd/libmi_gp.so with a bench.py beside the package runs through THAT root's Python package)
Each subprocess loads the given libmi_gp.so (andvaranaut_amd._lib.LIB_PATH), evaluates the LML a few times and prints the
median; three rounds per library, interleaved, so that box-to-box and drift effects cancel."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
WORKER = r'''
import os, sys, time
import numpy as np
lib = os.path.abspath(sys.argv[2])
own = os.path.dirname(os.path.dirname(lib))  # <root>/andvaranaut_amd/libmi_gp.so: that root's Python package (an older C-ABI)
sys.path.insert(0, own if os.path.basename(os.path.dirname(lib)) == "andvaranaut_amd" and os.path.exists(os.path.join(own, "bench.py")) else sys.argv[1])
import andvaranaut_amd._lib as L
L.LIB_PATH = lib
from andvaranaut_amd import MiGP
from bench import synth_problem, theta_sequence
N, d, kern = int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
grad = len(sys.argv) > 6 and sys.argv[6] == "grad"   # spec "N d kernel grad": LML + gradient evaluations
X, y = synth_problem(N, d, seed=0)
gp = MiGP(X, y, kern, need_grad=grad)
th = theta_sequence(d, 8, seed=0)
if grad:
    _lml = gp.lml
    gp.lml = lambda t: gp.lml_grad(t)[0]
for i in range(3):
    gp.lml(th[i])
res = []
reps = (10 if N <= 8192 else 5) if not grad else (6 if N <= 8192 else 3)
for rnd in range(5):
    t0 = time.perf_counter()
    for i in range(reps):
        v = gp.lml(th[i % 8])
    res.append((time.perf_counter() - t0) / reps * 1e3)
print(np.median(res), v)
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
            print(f"{spec:>22s}  {os.path.basename(l):28s} median {np.median(res[l]):8.3f} ms   runs {['%.3f' % x for x in res[l]]}  value {vals.get(l)}", flush=True)


if __name__ == "__main__":
    main()
