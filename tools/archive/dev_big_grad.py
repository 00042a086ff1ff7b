"""LML + gradient at BASELINE config 4's shape on ONE GPU (K, U = L^-T and K^-1: 3 x 34 GB of the 288 GB HBM)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import MiGP
from bench import synth_problem
N, d = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 32
X, y = synth_problem(N, d, seed=0)
ls = np.exp(np.linspace(np.log(0.8), np.log(3.0), d))
theta = np.concatenate([ls, [1.7], [1.0], [1e-4, 1e-6]])
gp = MiGP(X, y, "RBF", need_grad=True)
t0 = time.perf_counter(); v, g = gp.lml_grad(theta); t1 = time.perf_counter()
print(f"N={N} first lml_grad {t1-t0:.2f} s lml={v:.10e} info={gp.info} |g|={np.linalg.norm(g):.6e}", flush=True)
t0 = time.perf_counter(); v2, g2 = gp.lml_grad(theta); t1 = time.perf_counter()
print(f"N={N} second lml_grad {t1-t0:.3f} s  ({N**3/(t1-t0)*1e-12:.1f} TFLOP/s over N^3 flops)  same={v2 == v and np.array_equal(g, g2)}")
# directional finite difference of the LML along the gradient (two more factorisations)
h = 1e-4
u = g / np.linalg.norm(g)
th_p, th_m = theta.copy(), theta.copy()
th_p[:-1] += h * u[:-1] * theta[:-1]; th_m[:-1] -= h * u[:-1] * theta[:-1]
fd = (gp.lml(th_p) - gp.lml(th_m)) / (2 * h)
an = float(np.dot(g[:-1], u[:-1] * theta[:-1]))
print(f"directional derivative: analytic {an:.8e}  finite difference {fd:.8e}  rel diff {abs(an-fd)/abs(an):.2e}")
