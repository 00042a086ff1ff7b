"""BASELINE config 5 at length: one NUTS chain (RBF, N=8192, d=8) with PyMC's default adaptation schedule scaled down."""
import sys, time, json, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import MiGP
from andvaranaut_amd.priors import HyperModel
from andvaranaut_amd.nuts import sample_chain
from bench import synth_problem
N, d = 8192, 8
X, y = synth_problem(N, d, seed=1)
gp = MiGP(X, y, "RBF")
model = HyperModel(d, ["RBF"], noise=True, jitter=1e-6)
f = lambda q: model.logp_dlogp(q, gp.lml_grad)
t0 = time.perf_counter()
r = sample_chain(f, model.initial_point(), draws=150, tune=150, seed=0)
dt = time.perf_counter() - t0
pts = [model.point_dict(q) for q in r["q"]]
out = {"N": N, "d": d, "draws": 150, "tune": 150, "leapfrogs": int(r["n_leapfrog"]), "seconds": dt,
       "grad_evals_per_s": r["n_leapfrog"] / dt, "mean_tree_depth": float(r["mean_tree_depth"]), "diverging": int(np.sum(r["diverging"])),
       "step_size": float(r["step_size"]), "lp_mean": float(np.mean(r["lp"])),
       "posterior_mean": {"kv": float(np.mean([p["kv"] for p in pts])), "gv": float(np.mean([p["gv"] for p in pts])),
                          "l": np.mean([p["l"] for p in pts], axis=0).tolist()}}
print(json.dumps(out))
json.dump(out, open("gpurun_out/config5_long.json", "w"), indent=1)
