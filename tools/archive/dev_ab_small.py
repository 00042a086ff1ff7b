"""A/B: launches with fewer 128x128 tiles than `thr` run on the 64x64-tile kernel (mi_gp_set_option 7)."""
import sys
import time

sys.path.insert(0, "/root/repo")
from andvaranaut_amd import MiGP  # noqa: E402
from bench import synth_problem, theta_sequence  # noqa: E402

for N, d, kern in ((8192, 8, "RBF"), (16384, 16, "Matern52")):
    X, y = synth_problem(N, d, seed=0)
    th = theta_sequence(d, 14, seed=0)
    gp = MiGP(X, y, kern, need_grad=False)
    for thr in (1024, 768, 512, 384, 1536, 1024):
        gp.set_option(7, thr)
        for i in range(3):
            gp.lml(th[i])
        t0 = time.perf_counter()
        for i in range(20):
            gp.lml(th[3 + i % 10])
        print(f"N={N} small-tile threshold {thr}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", flush=True)
    gp.set_option(7, 1024)
    gp.close()
