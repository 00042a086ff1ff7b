"""Time one BO iteration's parts at a non-toy size (batched predict on 10 000 LHS points, refinement on
the differentiable predictive, refit)."""
import sys, os, time
import numpy as np
import scipy.stats as st
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import GPMCMC, MiGP, uniform
if len(sys.argv) > 2 and sys.argv[2] == "device":  # A/B: small predict calls through device buffers (round 5's route)
    MiGP.PINNED_IO_MAX_POINTS = 0
d, n0 = 6, int(sys.argv[1]) if len(sys.argv) > 1 else 4000
priors = [st.uniform(loc=0, scale=1) for _ in range(d)]
fun = lambda x: np.array([np.sum((x - 0.3) ** 2) + 0.3 * np.sin(6 * np.sum(x))])
g = GPMCMC(kernel="Matern52", noise=True, xconrevs=[uniform(p) for p in priors], yconrevs=[None], nx=d, ny=1, priors=priors,
           target=fun, verbose=False)
rng = np.random.default_rng(0)
x = rng.uniform(0, 1, (n0, d)); y = np.array([fun(r) for r in x])
g.set_data(x, y)
t0 = time.perf_counter(); g.fit(method="map"); t_fit = time.perf_counter() - t0
xs = rng.uniform(0, 1, (10000, d))
g.yopt = np.min(g.y)
t0 = time.perf_counter(); e = g.predict(xs, EI=True, EIopt="min"); t_sweep = time.perf_counter() - t0
pot = g._bo_potential("EI", "min", True, 1e-6)
x0 = xs[np.argmax(e[:, 0])]
pot(x0)
t0 = time.perf_counter()
for _ in range(20): pot(x0)
t_pot = (time.perf_counter() - t0) / 20
np.random.seed(0)
t0 = time.perf_counter(); xo, yo = g.BO(opt_type="min", max_iter=2, predict_samps=10000); t_bo = time.perf_counter() - t0
print(f"N={n0} d={d}: MAP fit {t_fit:.2f} s | EI sweep over 10000 points {t_sweep*1e3:.1f} ms | one refinement step (value+gradient) {t_pot*1e3:.2f} ms | "
      f"2 BO iterations (propose, refine, evaluate, refit) {t_bo:.2f} s; best {float(yo):.4f} (start {float(np.min(y)):.4f})")
