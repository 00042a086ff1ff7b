"""Single-rank timing of the sharded driver's LML and LML+gradient next to the single-GPU path."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, "/root/repo")
from andvaranaut_amd import MiGP  # noqa: E402
from andvaranaut_amd.distributed import DistGP  # noqa: E402
from bench import synth_problem, theta_sequence  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
d = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
X, y = synth_problem(N, d, seed=0)
th = theta_sequence(d, reps + 1, seed=0)
gp = DistGP(X, y, "Matern52")
for name, fn in (("lml", gp.lml), ("lml_grad", gp.lml_grad)):
    fn(th[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        r = fn(th[1 + i])
    torch.cuda.synchronize()
    print(f"DistGP world=1 N={N} {name}: {(time.perf_counter() - t0) / reps * 1e3:.1f} ms", flush=True)
# phases of the gradient
val = gp.lml(th[0], 0, _keep=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
gp._u_owned()
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"  U rows (trsm): {(t1 - t0) * 1e3:.1f} ms  ({N**3 / 3 / (t1 - t0) * 1e-12:.1f} TFLOP/s)")
v2, g2 = gp.lml_grad(th[reps])
del gp
torch.cuda.empty_cache()
one = MiGP(X, y, "Matern52")
one.lml_grad(th[0])
t0 = time.perf_counter()
for i in range(reps):
    v, g = one.lml_grad(th[1 + i])
print(f"MiGP lml_grad: {(time.perf_counter() - t0) / reps * 1e3:.1f} ms")
print("max rel grad diff", np.max(np.abs(g - g2) / np.maximum(np.abs(g), 1e-3 * np.abs(g).max())))
