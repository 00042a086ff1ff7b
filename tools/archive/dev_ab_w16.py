"""A/B: 16-tile super-panels (k = 2048 bulk updates) while more than `thr` tile columns remain (mi_gp_set_option 4)."""
import sys
import time

sys.path.insert(0, "/root/repo")
from andvaranaut_amd import MiGP  # noqa: E402
from bench import synth_problem, theta_sequence  # noqa: E402

N, d = 16384, 16
X, y = synth_problem(N, d, seed=0)
th = theta_sequence(d, 14, seed=0)
gp = MiGP(X, y, "Matern52", need_grad=False)
for thr in (1 << 20, 112, 104, 96, 88, 1 << 20):
    gp.set_option(4, thr)
    for i in range(3):
        gp.lml(th[i])
    t0 = time.perf_counter()
    vals = [gp.lml(th[3 + i]) for i in range(10)]
    print(f"W=16 above {thr}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms  lml[0]={vals[0]:.10e}", flush=True)
