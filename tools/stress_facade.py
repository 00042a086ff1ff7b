"""Exercise the GPMCMC surface in many configurations end to end (catch exceptions, check sanity)."""
import sys, os, time, traceback
import numpy as np
import scipy.stats as st
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import GPMCMC, normal, uniform, wgp, logarithm

priors = [st.uniform(loc=0, scale=2), st.uniform(loc=1, scale=0.5)]
fun = lambda x: np.array([np.exp(0.5 * (x[0] ** 2 - x[0] - x[1] ** 2 * x[0] + x[1]))])
xt = np.random.default_rng(5).uniform([0, 1], [2, 1.5], (40, 2))
yt = np.array([fun(r) for r in xt])
def rel(g): return float(np.sqrt(np.mean(((g.predict(xt) - yt) / yt) ** 2)))
def mk(kernel, noise, ycon=None, xcon=None, n=60):
    g = GPMCMC(kernel=kernel, noise=noise, xconrevs=xcon or [uniform(priors[0]), uniform(priors[1])], yconrevs=[ycon], nx=2, ny=1,
               priors=priors, target=fun, verbose=False)
    g.sample(n, seed=1)
    return g
ok = True
def case(name, fn):
    global ok
    t0 = time.perf_counter()
    try:
        out = fn()
        print(f"[ok]   {name}: {out}  ({time.perf_counter()-t0:.1f}s)", flush=True)
    except Exception:
        ok = False
        print(f"[FAIL] {name}\n{traceback.format_exc()}", flush=True)

def c1():
    g = mk("RatQuad", True); g.fit(method="map", truncate=True); return rel(g), sorted(g.hypers)
def c2():
    g = mk("RBF*Matern32+Exponential", False); g.fit(method="map"); return rel(g)
def c3():
    g = mk("Matern52", True, ycon=logarithm()); g.fit(method="mcmc_map", draws=40, tune=40, chains=2, random_seed=0); return rel(g)
def c4():
    g = mk("RBF", True); g.fit(method="mcmc_mean", draws=40, tune=40, chains=2, random_seed=0, truncate=True)
    y, yv = g.predict(xt, return_var=True, normvar=True); y2 = g.predict(xt, revert=False); g.yopt = np.min(g.y); e = g.predict(xt, EI=True, EIopt="min")
    return rel(g), float(yv.mean()), y2.shape, float(e.max())
def c5():
    yw = wgp(["logarithm", "meanstd", "sinharcsinh"], np.array([0.0, 1.0]), y=np.ones(3) + np.arange(3))
    g = mk("Matern52", True, ycon=yw); g.fit(method="map", cwgp=True)
    np.random.seed(0); x, y = g.BO(opt_type="min", max_iter=3, predict_samps=2000, cwgp=True); return rel(g), float(y), g.nsamp
def c6():
    g = mk("RBF", True, n=50); g.fit()
    np.random.seed(1); data, xo = g.inverse_opt(np.array([fun(np.array([1.1, 1.3]))]), method="mcmc_mean", draws=60, tune=60, chains=2, random_seed=2)
    return xo.tolist(), float(g.predict(np.array([xo]))[0, 0]), float(fun(np.array([1.1, 1.3]))[0])
def c7():
    g = mk("RBF", True, n=50); g.fit(); g.train_test(0.8); g.change_model(kernel="Matern32", noise=False); g.fit(restarts=2); g.del_samples(5); g.fit(method="none"); return rel(g), g.nsamp
def c8():
    xw = [wgp(["uniform", "kumaraswamy"], np.array([1.0, 1.0]), y=np.linspace(0, 2, 5), xdist=priors[0]), uniform(priors[1])]
    g = mk("RBF", True, xcon=xw); g.fit(method="map", iwgp=True, truncate=True); np.random.seed(3); g.BO(opt_type="max", opt_method="map", method="eps-RS", max_iter=2, iwgp=True); return rel(g), g.nsamp
for nm, f in (("ratquad truncate map", c1), ("composite no-noise map", c2), ("log-warp mcmc_map", c3), ("mcmc_mean truncate + predict modes", c4),
              ("cwgp fit + BO(cwgp)", c5), ("inverse_opt mcmc", c6), ("model changes / restarts / none", c7), ("iwgp truncate + BO map eps-RS", c8)):
    case(nm, f)
print("ALL OK" if ok else "FAILURES")
