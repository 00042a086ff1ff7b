"""A/B: leaf + strip in one launch  for the last `fuse` tile columns (mi_gp_set_option 13; 0 = never) against separate launches."""
import sys
import time

sys.path.insert(0, "/root/repo")
from andvaranaut_amd import MiGP  # noqa: E402
from bench import synth_problem, theta_sequence  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [128, 1024, 4096, 8192, 16384]
for N in sizes:
    d, kern = (16, "Matern52") if N >= 16384 else (8, "RBF")
    X, y = synth_problem(N, d, seed=0)
    th = theta_sequence(d, 14, seed=0)
    gp = MiGP(X, y, kern, need_grad=False)
    ref = None
    for fuse in (0, 64, 0, 64):
        gp.set_option(13, fuse)
        for i in range(3):
            gp.lml(th[i])
        t0 = time.perf_counter()
        vals = [gp.lml(th[3 + i % 10]) for i in range(20)]
        dt = (time.perf_counter() - t0) / 20
        vals = vals[:10]
        ref = ref or vals
        print(f"N={N} fuse={fuse}: {dt * 1e3:.3f} ms  identical={vals == ref}  lml[0]={vals[0]:.12e}", flush=True)
    gp.close()
