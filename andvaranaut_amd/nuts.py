"""No-U-Turn sampler: the host half of pm.sample as GPMCMC.__fit uses it (gpmcmc.py:350-361).
[3P] PyMC's default for continuous models is NUTS with ``init="jitter+adapt_diag"``: multinomial
trajectory sampling with the generalised U-turn criterion, dual-averaging step-size adaptation to
``target_accept`` (0.8), a diagonal mass matrix re-estimated from the tuning draws, ``max_treedepth``
10, 1000 tuning + 1000 kept draws.  This is an independent implementation of that published
algorithm (Hoffman & Gelman 2014; Betancourt 2017); it cannot reproduce PyMC's random stream, only
its posterior statistics.  Every leapfrog step costs one device evaluation of (logp, dlogp)."""
import numpy as np


class Trace:
    """Minimal stand-in for the arviz.InferenceData fields the reference reads
    (gpmcmc.py:404-430): ``posterior[name]`` and ``sample_stats['lp']`` as [chain, draw, ...]."""

    def __init__(self, posterior, sample_stats):
        self.posterior = posterior
        self.sample_stats = sample_stats


class _DualAveraging:
    def __init__(self, eps0, target, gamma=0.05, t0=10.0, kappa=0.75):
        self.mu = np.log(10.0 * eps0)
        self.target, self.gamma, self.t0, self.kappa = target, gamma, t0, kappa
        self.hbar, self.log_eps_bar, self.t = 0.0, 0.0, 0

    def update(self, accept):
        self.t += 1
        w = 1.0 / (self.t + self.t0)
        self.hbar = (1 - w) * self.hbar + w * (self.target - accept)
        log_eps = self.mu - np.sqrt(self.t) / self.gamma * self.hbar
        eta = self.t ** (-self.kappa)
        self.log_eps_bar = eta * log_eps + (1 - eta) * self.log_eps_bar
        return np.exp(log_eps)

    def final(self):
        return np.exp(self.log_eps_bar)


class _Welford:
    def __init__(self, n):
        self.k, self.mean, self.m2 = 0, np.zeros(n), np.zeros(n)

    def add(self, x):
        self.k += 1
        d = x - self.mean
        self.mean += d / self.k
        self.m2 += d * (x - self.mean)

    def variance(self):
        # regularised like Stan / PyMC: shrink towards 1e-3 for few samples
        var = self.m2 / max(self.k - 1, 1)
        return (self.k / (self.k + 5.0)) * var + 1e-3 * (5.0 / (self.k + 5.0))


def _leapfrog(f, q, p, g, eps, minv):
    p = p + 0.5 * eps * g
    q = q + eps * minv * p
    v, g = f(q)
    p = p + 0.5 * eps * g
    return q, p, v, g


def _find_reasonable_eps(f, q, v, g, minv, rng):
    eps = 0.1
    p = rng.standard_normal(q.size) / np.sqrt(minv)
    h0 = v - 0.5 * np.sum(minv * p * p)
    _, p1, v1, _ = _leapfrog(f, q, p, g, eps, minv)
    h1 = v1 - 0.5 * np.sum(minv * p1 * p1) if np.isfinite(v1) else -np.inf
    a = 1.0 if (h1 - h0) > np.log(0.5) else -1.0
    for _ in range(50):
        if not a * (h1 - h0) > -a * np.log(2.0):
            break
        eps *= 2.0 ** a
        _, p1, v1, _ = _leapfrog(f, q, p, g, eps, minv)
        h1 = v1 - 0.5 * np.sum(minv * p1 * p1) if np.isfinite(v1) else -np.inf
    return eps


def _nuts_step(f, q, v, g, eps, minv, rng, max_treedepth):
    """One NUTS transition (multinomial sampling, generalised U-turn)."""
    p0 = rng.standard_normal(q.size) / np.sqrt(minv)
    h0 = v - 0.5 * np.sum(minv * p0 * p0)
    # tree state: leftmost / rightmost (q, p, g), proposal (q, v, g), log-sum of weights, rho
    ql, pl, gl = q, p0, g
    qr, pr, gr = q, p0, g
    qprop, vprop, gprop = q, v, g
    logw = 0.0
    rho = p0.copy()
    acc_sum, n_acc, diverged, depth = 0.0, 0, False, 0

    def build(qe, pe, ge, direction, depth):
        nonlocal acc_sum, n_acc, diverged
        if depth == 0:
            q1, p1, v1, g1 = _leapfrog(f, qe, pe, ge, direction * eps, minv)
            h1 = v1 - 0.5 * np.sum(minv * p1 * p1) if np.isfinite(v1) else -np.inf
            dh = h1 - h0
            if not np.isfinite(dh):
                dh = -np.inf
            acc_sum += np.exp(min(dh, 0.0)) if dh > -np.inf else 0.0
            n_acc += 1
            if dh < -1000.0:
                diverged = True
            return q1, p1, g1, q1, p1, g1, q1, v1, g1, dh, p1.copy(), not diverged
        a = build(qe, pe, ge, direction, depth - 1)
        (ql_, pl_, gl_, qr_, pr_, gr_, qp, vp, gp, lw, rh, ok) = a
        if not ok:
            return a
        if direction == 1:
            b = build(qr_, pr_, gr_, direction, depth - 1)
            qr_, pr_, gr_ = b[3], b[4], b[5]
        else:
            b = build(ql_, pl_, gl_, direction, depth - 1)
            ql_, pl_, gl_ = b[0], b[1], b[2]
        lw2, rh2, ok2 = b[9], b[10], b[11]
        lw_tot = np.logaddexp(lw, lw2)
        if ok2 and np.log(rng.random()) < lw2 - lw_tot:
            qp, vp, gp = b[6], b[7], b[8]
        rh_tot = rh + rh2
        ok = ok2 and (np.dot(rh_tot, minv * pl_) > 0) and (np.dot(rh_tot, minv * pr_) > 0)
        return ql_, pl_, gl_, qr_, pr_, gr_, qp, vp, gp, lw_tot, rh_tot, ok

    while depth < max_treedepth:
        direction = 1 if rng.random() < 0.5 else -1
        if direction == 1:
            t = build(qr, pr, gr, direction, depth)
            qr, pr, gr = t[3], t[4], t[5]
        else:
            t = build(ql, pl, gl, direction, depth)
            ql, pl, gl = t[0], t[1], t[2]
        lw2, rh2, ok2 = t[9], t[10], t[11]
        if not ok2:
            break
        if np.log(rng.random()) < lw2 - logw:
            qprop, vprop, gprop = t[6], t[7], t[8]
        logw = np.logaddexp(logw, lw2)
        rho = rho + rh2
        depth += 1
        if not ((np.dot(rho, minv * pl) > 0) and (np.dot(rho, minv * pr) > 0)):
            break
    return qprop, vprop, gprop, acc_sum / max(n_acc, 1), depth, diverged, n_acc


def sample_chain(logp_dlogp, q0, draws=1000, tune=1000, target_accept=0.8, max_treedepth=10, seed=None,
                 progressbar=False):
    """One NUTS chain.  Returns dict(q=[draws,nq], lp=[draws], stats...)."""
    rng = np.random.default_rng(seed)
    nq = len(q0)
    q = np.array(q0, dtype=np.float64) + rng.uniform(-1.0, 1.0, nq)  # "jitter" of jitter+adapt_diag
    v, g = logp_dlogp(q)
    tries = 0
    while not np.isfinite(v) and tries < 20:
        q = np.array(q0, dtype=np.float64) + rng.uniform(-1.0, 1.0, nq) * 0.5 ** tries
        v, g = logp_dlogp(q)
        tries += 1
    if not np.isfinite(v):
        raise FloatingPointError("could not find a finite starting point for NUTS")
    minv = np.ones(nq)  # inverse mass = estimated posterior variance
    eps = _find_reasonable_eps(logp_dlogp, q, v, g, minv, rng)
    da = _DualAveraging(eps, target_accept)
    wf = _Welford(nq)
    # adaptation windows (Stan-style doubling, which is what adapt_diag converges to)
    start_buf, end_buf, win = int(0.15 * tune), int(0.1 * tune), max(int(0.05 * tune), 10)
    next_win = start_buf + win
    qs, lps, depths, nleap, div = np.empty((draws, nq)), np.empty(draws), [], 0, 0
    for it in range(tune + draws):
        q, v, g, acc, depth, diverged, nl = _nuts_step(logp_dlogp, q, v, g, eps, minv, rng, max_treedepth)
        nleap += nl
        if it < tune:
            eps = da.update(acc)
            if start_buf <= it < tune - end_buf:
                wf.add(q)
                if it + 1 == next_win and wf.k > 5:
                    minv = wf.variance()
                    wf = _Welford(nq)
                    win *= 2
                    next_win = it + 1 + win
                    if next_win + win > tune - end_buf:
                        next_win = tune - end_buf
                    eps = _find_reasonable_eps(logp_dlogp, q, v, g, minv, rng)
                    da = _DualAveraging(eps, target_accept)
            if it == tune - 1:
                eps = da.final()
        else:
            k = it - tune
            qs[k], lps[k] = q, v
            depths.append(depth)
            div += int(diverged)
        if progressbar and (it + 1) % 100 == 0:
            print(f"  NUTS {it + 1}/{tune + draws} eps={eps:.3g} depth={depth} lp={v:.4f}", flush=True)
    return {"q": qs, "lp": lps, "step_size": eps, "n_leapfrog": nleap, "diverging": div,
            "mean_tree_depth": float(np.mean(depths)) if depths else 0.0}
