"""A/B: band-column-major tile order of the bulk trapezoid launches (mi_gp_set_option 14 = band height in tile rows, 0 = row-major)."""
import sys
import time

sys.path.insert(0, "/root/repo")
from andvaranaut_amd import MiGP  # noqa: E402
from bench import synth_problem, theta_sequence  # noqa: E402

N, d = 16384, 16
X, y = synth_problem(N, d, seed=0)
th = theta_sequence(d, 14, seed=0)
gp = MiGP(X, y, "Matern52", need_grad=False)
ref = None
for band in (0, 8, 4, 16, 2, 0, 8):
    gp.set_option(14, band)
    for i in range(3):
        gp.lml(th[i])
    t0 = time.perf_counter()
    vals = [gp.lml(th[3 + i % 10]) for i in range(20)][:10]
    dt = (time.perf_counter() - t0) / 20
    ref = ref or vals
    gp.set_option(0, 0); gp.set_profiling(2)
    gp.lml(th[3]); tm = gp.timers()
    gp.set_profiling(0); gp.set_option(0, 1)
    print(f"band={band}: {dt * 1e3:.3f} ms  identical={vals == ref}  B-kernel alone {tm['gemm_b_flops'] / tm['gemm_b_ms'] * 1e-9:.1f} TFLOP/s", flush=True)
