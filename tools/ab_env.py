"""A/B of environment settings on one box: alternating subprocesses, one per (setting, round).
    python tools/ab_env.py "4096 8 RBF" "8192 8 RBF" -- "" "MIGP_CU_RESERVE=1" "MIGP_OPTS=32=0"
MIGP_OPTS=id=value,id=value is applied with mi_gp_set_option; any other variable goes to the worker's environment."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
WORKER = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
from andvaranaut_amd import MiGP
from bench import synth_problem, theta_sequence
N, d, kern = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
grad = len(sys.argv) > 5 and sys.argv[5] == "grad"
X, y = synth_problem(N, d, seed=0)
gp = MiGP(X, y, kern, need_grad=grad)
for kv in filter(None, os.environ.get("MIGP_OPTS", "").split(",")):
    gp.set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
th = theta_sequence(d, 8, seed=0)
f = (lambda t: gp.lml_grad(t)[0]) if grad else gp.lml
for i in range(3):
    f(th[i])
res = []
reps = 10 if N <= 8192 else 5
for rnd in range(5):
    t0 = time.perf_counter()
    for i in range(reps):
        v = f(th[i % 8])
    res.append((time.perf_counter() - t0) / reps * 1e3)
print(np.median(res), v)
'''


def main():
    args = sys.argv[1:]
    specs, envs = args[: args.index("--")], args[args.index("--") + 1:]
    for spec in specs:
        res = {e: [] for e in envs}
        vals = {}
        for rnd in range(3):
            for e in envs:
                env = dict(os.environ)
                for kv in filter(None, e.split(";")):
                    env[kv.split("=", 1)[0]] = kv.split("=", 1)[1]
                out = subprocess.run([sys.executable, "-c", WORKER, ROOT] + spec.split(), capture_output=True, text=True, env=env)
                if out.returncode != 0:
                    print(out.stderr[-500:])
                    continue
                t, v = out.stdout.strip().splitlines()[-1].split()
                res[e].append(float(t))
                vals[e] = v
        for e in envs:
            print(f"{spec:>22s}  [{e:32s}] median {np.median(res[e]):8.3f} ms   runs {['%.3f' % x for x in res[e]]}  value {vals.get(e)}", flush=True)


if __name__ == "__main__":
    main()
