"""BASELINE configs 3 and 5 end to end through the host drivers (numbers quoted in DESIGN.md)."""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
from andvaranaut_amd import MiGP
from andvaranaut_amd.priors import HyperModel
from andvaranaut_amd.optimize import find_MAP
from andvaranaut_amd.nuts import sample_chain
from bench import synth_problem

out = {}
# config 3: Matern-5/2, N=16384, d=16, MAP hyper-parameter loop (LML + gradient per step)
N, d = 16384, 16
X, y = synth_problem(N, d, seed=0)
gp = MiGP(X, y, "Matern52")
model = HyperModel(d, ["Matern52"], noise=True, jitter=1e-6)
f = lambda q: model.logp_dlogp(q, gp.lml_grad, jacobian=False)  # what pm.find_MAP optimises
f(model.initial_point())
t0 = time.perf_counter()
q, info = find_MAP(f, model.initial_point(), maxeval=60)
dt = time.perf_counter() - t0
pt = model.point_dict(q)
out["config3_map"] = {"N": N, "d": d, "evals": info["nfev"], "seconds": dt, "evals_per_s": info["nfev"] / dt,
                      "logp": info["logp"], "kv": float(pt["kv"]), "gv": float(pt["gv"]), "l_minmax": [float(pt["l"].min()), float(pt["l"].max())]}
print(json.dumps(out["config3_map"]), flush=True)
gp.close()
# config 5: RBF, N=8192, d=8, one NUTS chain per GPU (short run)
N, d = 8192, 8
X, y = synth_problem(N, d, seed=1)
gp = MiGP(X, y, "RBF")
model = HyperModel(d, ["RBF"], noise=True, jitter=1e-6)
f = lambda q: model.logp_dlogp(q, gp.lml_grad)
qmap, _ = find_MAP(lambda q: model.logp_dlogp(q, gp.lml_grad, jacobian=False), model.initial_point(), maxeval=40)
t0 = time.perf_counter()
r = sample_chain(f, qmap, draws=30, tune=30, seed=0)
dt = time.perf_counter() - t0
out["config5_nuts"] = {"N": N, "d": d, "draws": 30, "tune": 30, "leapfrogs": r["n_leapfrog"], "seconds": dt,
                       "grad_evals_per_s": r["n_leapfrog"] / dt, "mean_tree_depth": r["mean_tree_depth"], "diverging": r["diverging"],
                       "lp_mean": float(np.mean(r["lp"]))}
print(json.dumps(out["config5_nuts"]), flush=True)
if out["config5_nuts"]["diverging"]:
    print(f"WARNING: {out['config5_nuts']['diverging']} divergent transitions: this run is too short for the step size to adapt "
          "(tools/run_config5_nuts.py is the reference-length run)", flush=True)
# the same with three chains side by side on this GPU (what GPMCMC.fit(method='mcmc_*') does with more chains than GPUs)
import threading
others = [MiGP(X, y, "RBF") for _ in range(2)]
handles = [gp] + others
res = [None] * 3
def chain(i):
    h = handles[i]
    res[i] = sample_chain(lambda q: model.logp_dlogp(q, h.lml_grad), qmap, draws=30, tune=30, seed=i)
ths = [threading.Thread(target=chain, args=(i,)) for i in range(3)]
t0 = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
dt3 = time.perf_counter() - t0
nl = sum(r_["n_leapfrog"] for r_ in res)
out["config5_nuts_three_chains_one_gpu"] = {"chains": 3, "leapfrogs": nl, "seconds": dt3, "grad_evals_per_s": nl / dt3,
                                             "diverging": [r_["diverging"] for r_ in res]}
print(json.dumps(out["config5_nuts_three_chains_one_gpu"]), flush=True)
if any(out["config5_nuts_three_chains_one_gpu"]["diverging"]):
    print(f"WARNING: divergent transitions {out['config5_nuts_three_chains_one_gpu']['diverging']} in the three-chain run", flush=True)
json.dump(out, open("gpurun_out/configs_3_5.json", "w"), indent=1)
