import sys, os, time, cProfile, pstats
import numpy as np
import scipy.stats as st
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import GPMCMC, uniform
d, n0 = 6, 4000
priors = [st.uniform(loc=0, scale=1) for _ in range(d)]
fun = lambda x: np.array([np.sum((x - 0.3) ** 2) + 0.3 * np.sin(6 * np.sum(x))])
g = GPMCMC(kernel="Matern52", noise=True, xconrevs=[uniform(p) for p in priors], yconrevs=[None], nx=d, ny=1, priors=priors, target=fun, verbose=False)
rng = np.random.default_rng(0)
x = rng.uniform(0, 1, (n0, d)); y = np.array([fun(r) for r in x])
g.set_data(x, y); g.fit(method="map"); g.yopt = np.min(g.y)
xs = rng.uniform(0, 1, (10000, d))
for _ in range(3): g.predict(xs, EI=True, EIopt="min")
t0 = time.perf_counter()
for _ in range(5): g.predict(xs, EI=True, EIopt="min")
print("sweep", (time.perf_counter() - t0) / 5 * 1e3, "ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(5): g.predict(xs, EI=True, EIopt="min")
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
