"""Aggregate evaluations/s of the batched entry points on ONE GPU: python tools/bench_batch.py [N d kernel] [--grad] [--ks 1,2,4,8]
Compares with the one-at-a-time entry point (K = 1 row) at the same thetas."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import MiGP
from bench import synth_problem, theta_sequence
args = [a for a in sys.argv[1:] if not a.startswith("--")]
grad = "--grad" in sys.argv
ks = [int(x) for x in sys.argv[sys.argv.index("--ks") + 1].split(",")] if "--ks" in sys.argv else [1, 2, 4, 8, 16]
if "--ks" in sys.argv:
    args = [a for a in args if a != sys.argv[sys.argv.index("--ks") + 1]]
N, d, kern = (int(args[0]), int(args[1]), args[2]) if len(args) >= 3 else (4096, 8, "RBF")
X, y = synth_problem(N, d, seed=0)
gp = MiGP(X, y, kern, need_grad=grad)
for kv in filter(None, os.environ.get("MIGP_OPTS", "").split(",")):  # e.g. MIGP_OPTS=7=512
    gp.set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
th = np.array(theta_sequence(d, 64, seed=0))
single = (lambda t: gp.lml_grad(t)[0]) if grad else gp.lml
single(th[0])
reps = max(3, int(2000 / max(N / 1024, 1) ** 2 / 4))
t0 = time.perf_counter()
for i in range(reps):
    single(th[i % 64])
t1 = (time.perf_counter() - t0) / reps
out = {"N": N, "d": d, "kernel": kern, "what": "lml+grad" if grad else "lml", "single_ms": t1 * 1e3, "single_evals_per_s": 1 / t1, "batch": {}}
print(f"N={N} {kern} {'LML+grad' if grad else 'LML'}: one at a time {t1 * 1e3:.3f} ms = {1 / t1:.1f} evals/s", flush=True)
for k in ks:
    f = (lambda T: gp.lml_grad_batch(T)[0]) if grad else gp.lml_batch
    try:
        f(th[:k])
    except Exception as e:  # noqa: BLE001 - out of memory at large N * K
        print(f"  K={k}: {e}")
        break
    r = max(2, reps // k)
    t0 = time.perf_counter()
    for i in range(r):
        f(th[(i * k) % 48:(i * k) % 48 + k])
    tb = (time.perf_counter() - t0) / r
    out["batch"][k] = {"ms_per_batch": tb * 1e3, "evals_per_s": k / tb}
    print(f"  K={k:3d}: {tb * 1e3:8.3f} ms per batch = {k / tb:8.1f} evals/s  ({k / tb * t1:.2f}x)", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/r04_batch_{N}_{'grad' if grad else 'lml'}.json", "w"), indent=1)
