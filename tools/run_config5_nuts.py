"""BASELINE config 5 at the reference's run length: pm.sample's defaults are 1000 tuning + 1000 kept draws and >= 2
chains (gpmcmc.py:351; [3P] pymc.sample).  Three NUTS chains (RBF, N=8192, d=8) share ONE GPU on three handles driven
by three host threads (how GPMCMC.fit(method='mcmc_*') schedules more chains than GPUs); every leapfrog step is one
device LML + gradient evaluation.  Records divergences, step sizes, split R-hat and a bulk effective sample size per
hyper-parameter, and the distance between the posterior mean and the MAP point -> profiles/r03_config5_nuts.json.

    python tools/run_config5_nuts.py [--draws 1000 --tune 1000 --chains 3 --out gpurun_out/config5_nuts.json]"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def split_rhat(x):
    """Gelman et al. (BDA3) split R-hat of x[chain, draw]."""
    c, n = x.shape
    h = n // 2
    s = np.concatenate([x[:, :h], x[:, h:2 * h]], axis=0)
    m, n2 = s.shape
    means, var = s.mean(1), s.var(1, ddof=1)
    W = var.mean()
    B = n2 * means.var(ddof=1)
    return float(np.sqrt(((n2 - 1) / n2 * W + B / n2) / W))


def ess_bulk(x):
    """Effective sample size from the chains' autocorrelation (Geyer's initial positive sequence on the pooled estimate)."""
    c, n = x.shape
    xc = x - x.mean(1, keepdims=True)
    acov = np.zeros(n)
    for k in range(c):
        f = np.fft.rfft(xc[k], 2 * n)
        acov += np.fft.irfft(f * np.conj(f))[:n] / n
    acov /= c
    W = acov[0] * n / (n - 1)
    B_over_n = x.mean(1).var(ddof=1) if c > 1 else 0.0
    var_plus = acov[0] + B_over_n
    rho = 1.0 - (W - acov) / var_plus
    tau, t = -1.0, 0
    while t + 1 < n:
        pair = rho[t] + rho[t + 1]
        if pair < 0:
            break
        tau += 2.0 * pair
        t += 2
    return float(c * n / max(tau, 1.0 / np.log10(c * n)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--draws", type=int, default=1000)
    ap.add_argument("--tune", type=int, default=1000)
    ap.add_argument("--chains", type=int, default=3)
    ap.add_argument("--out", default="gpurun_out/config5_nuts.json")
    ap.add_argument("--batched", action="store_true", help="ONE handle; the chains meet in one mi_gp_lml_grad_batch call per leapfrog step")
    args = ap.parse_args()

    from andvaranaut_amd import MiGP
    from andvaranaut_amd.nuts import sample_chain
    from andvaranaut_amd.optimize import find_MAP
    from andvaranaut_amd.priors import HyperModel
    from bench import synth_problem

    N, d = args.n, args.d
    X, y = synth_problem(N, d, seed=1)
    model = HyperModel(d, ["RBF"], noise=True, jitter=1e-6)
    handles = [MiGP(X, y, "RBF") for _ in range(1 if args.batched else args.chains)]
    ev = None
    if args.batched:
        from andvaranaut_amd.batching import BatchedEvaluator
        ev = BatchedEvaluator(handles[0].lml_grad_batch, args.chains)
    t0 = time.perf_counter()
    qmap, info = find_MAP(lambda q: model.logp_dlogp(q, handles[0].lml_grad, jacobian=False), model.initial_point())
    t_map = time.perf_counter() - t0
    print(f"MAP: logp {info['logp']:.4f} after {info.get('nfev', '?')} evaluations, {t_map:.1f} s", flush=True)

    seeds = np.random.SeedSequence(2026).spawn(args.chains)
    res, errs = [None] * args.chains, []

    def lane(c):
        try:
            fn = ev.evaluate if ev is not None else handles[c].lml_grad
            f = lambda q: model.logp_dlogp(q, fn)  # noqa: E731  (pm.sample: density WITH the Jacobian)
            res[c] = sample_chain(f, model.initial_point(), draws=args.draws, tune=args.tune, seed=seeds[c],
                                  progressbar=(c == 0))
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))
        finally:
            if ev is not None:
                ev.leave()

    t0 = time.perf_counter()
    ths = [threading.Thread(target=lane, args=(c,)) for c in range(args.chains)]
    for t in ths:
        t.start()
    while any(t.is_alive() for t in ths):  # a line a minute: early tuning iterations can take minutes per hundred
        for t in ths:
            t.join(timeout=60.0 / len(ths))
        if ev is not None and any(t.is_alive() for t in ths):
            print(f"  t={time.perf_counter() - t0:7.0f} s: {ev.rounds} batched rounds, {ev.evaluations} evaluations", flush=True)
    dt = time.perf_counter() - t0
    if errs:
        raise SystemExit("a chain failed: " + "; ".join(errs))

    q = np.stack([r["q"] for r in res])  # [chain, draw, nq] unconstrained (log) space
    names = ["log gv"] + [f"log l[{m}]" for m in range(d)] + ["log kv"]  # PyMC creation order (gpmcmc.py:198,207,208)
    assert q.shape[2] == len(names), (q.shape, names)
    rhat = {n_: split_rhat(q[:, :, i]) for i, n_ in enumerate(names)}
    ess = {n_: ess_bulk(q[:, :, i]) for i, n_ in enumerate(names)}
    pooled = q.reshape(-1, q.shape[2])
    sd = pooled.std(0, ddof=1)
    nleap = int(sum(r["n_leapfrog"] for r in res))
    pts = [model.point_dict(v) for v in pooled[:: max(1, len(pooled) // 500)]]
    out = {
        "workload": f"RBF GP hyper-parameter posterior, N={N} d={d}, {args.chains} NUTS chains sharing one MI355X "
                    + ("(ONE handle, one batched evaluation per leapfrog round: blockIdx.z = chain)" if args.batched else "(one handle + host thread each)"),
        "batched_rounds": (ev.rounds if ev is not None else None), "batched_mean_k": (ev.evaluations / max(ev.rounds, 1) if ev is not None else None),
        "reference": "pm.sample defaults (gpmcmc.py:351): 1000 tune + 1000 draws, target_accept 0.8, max_treedepth 10",
        "draws": args.draws, "tune": args.tune, "chains": args.chains,
        "seconds": dt, "leapfrog_steps": nleap, "grad_evals_per_s_all_chains": nleap / dt,
        "diverging_after_tuning": [int(r["diverging"]) for r in res],
        "step_size": [float(r["step_size"]) for r in res],
        "mean_tree_depth": [float(r["mean_tree_depth"]) for r in res],
        "split_rhat": rhat, "rhat_max": max(rhat.values()),
        "ess_bulk": ess, "ess_min": min(ess.values()),
        "posterior_mean_q": dict(zip(names, pooled.mean(0).tolist())),
        "posterior_sd_q": dict(zip(names, sd.tolist())),
        "map_q": dict(zip(names, np.asarray(qmap).tolist())),
        "map_minus_mean_in_posterior_sd": dict(zip(names, ((np.asarray(qmap) - pooled.mean(0)) / sd).tolist())),
        "lp_mean_per_chain": [float(np.mean(r["lp"])) for r in res],
        "posterior_mean_natural": {"kv": float(np.mean([p["kv"] for p in pts])), "gv": float(np.mean([p["gv"] for p in pts])),
                                   "l": np.mean([p["l"] for p in pts], axis=0).tolist()},
        "map_seconds": t_map,
    }
    if sum(out["diverging_after_tuning"]) > 0:
        print(f"WARNING: {sum(out['diverging_after_tuning'])} divergent transitions after tuning", flush=True)
    if out["rhat_max"] >= 1.01:
        print(f"WARNING: split R-hat {out['rhat_max']:.4f} >= 1.01", flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("seconds", "leapfrog_steps", "grad_evals_per_s_all_chains", "diverging_after_tuning",
                                           "step_size", "rhat_max", "ess_min")}), flush=True)
    for h in handles:
        h.close()


if __name__ == "__main__":
    main()
