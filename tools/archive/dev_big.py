import sys, time
import numpy as np
sys.path.insert(0, ".")
from andvaranaut_amd import MiGP
from bench import synth_problem
N, d = int(sys.argv[1]), 32
X, y = synth_problem(N, d, seed=0)
ls = np.exp(np.linspace(np.log(0.8), np.log(3.0), d))
theta = np.concatenate([ls, [1.7], [1.0], [1e-4, 1e-6]])
gp = MiGP(X, y, "RBF", need_grad=False)
t0 = time.perf_counter(); v = gp.lml(theta); t1 = time.perf_counter()
print(f"N={N} first eval {t1-t0:.2f} s lml={v:.10e} info={gp.info}")
t0 = time.perf_counter(); v2 = gp.lml(theta); t1 = time.perf_counter()
print(f"N={N} second eval {t1-t0:.3f} s lml={v2:.10e}  chol TFLOP/s (whole eval) {N**3/3/(t1-t0)*1e-12:.1f}")
