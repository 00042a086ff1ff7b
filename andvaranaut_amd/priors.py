"""Hyper-parameter priors, their unconstraining transforms and the joint log-posterior that the
reference builds with PyMC inside GPMCMC.__fit (gpmcmc.py:193-208):

    noise:     gv ~ HalfNormal(sigma=1e-3)                     (log transform)   [:198]
               gv ~ Truncated(Normal(0, 1e-3), 1e-15, 1.0)     (interval)        [:195-196]  truncate=True
    no noise:  gv = 0                                                             [:200]
    l  ~ LogNormal(0, 1)            shape nx*nkern             (log)             [:207]
    l  ~ TruncatedNormal(0.5, 0.15, 1e-3, 100)                 (interval)        [:202-203]  truncate=True
    kv ~ LogNormal(0.56, 0.75)      shape nkern                (log)             [:208]
    kv ~ TruncatedNormal(1.0, 0.15, 1e-1, 100)                 (interval)        [:204-205]  truncate=True
    alpha ~ LogNormal(0.56, 0.75)   RatQuad only               (log)             [:288]
    iwgp     ~ LogNormal(0, 0.25) | TruncatedNormal(1, 1, 1e-3, 5)    input-warp parameters   [:217-221]
    cwgp_pos ~ LogNormal(0, 0.25) | TruncatedNormal(1, 1, 1e-3, 5)    positive output-warp parameters [:251-256]
    cwgp     ~ Normal(0, 1)       | TruncatedNormal(0, 1, -10, 10)    the other output-warp parameters [:257-262]

The logp / moment / transform formulas restate PyMC 5.9.2 (pymc/distributions/continuous.py,
pymc/logprob/transforms.py), which is pinned only as ``pymc <= 5.9.2`` by the reference and is not
installed here.  Variables are ordered as PyMC creates them in __fit: gv, l, kv, (iwgp), (cwgp_pos), (cwgp), (alpha).

[3P] pm.find_MAP compiles the model's logp and dlogp with ``jacobian=False`` (pymc/tuning/starting.py): the
optimiser moves the unconstrained vector but maximises the density of the constrained variables, without the
log-determinant of the transform; pm.sample / NUTS uses the full transformed density.  ``logp_dlogp`` takes
that switch as ``jacobian``."""
import numpy as np
from scipy.special import erf, expit

LOG_SQRT_2PI = 0.5 * np.log(2.0 * np.pi)


def _log_sigmoid(x):
    return -np.logaddexp(0.0, -x)


class LogNormal:
    transform = "log"

    def __init__(self, mu, sigma):
        self.mu, self.sigma = float(mu), float(sigma)

    def logp(self, x):
        z = (np.log(x) - self.mu) / self.sigma
        return -0.5 * z * z - LOG_SQRT_2PI - np.log(self.sigma) - np.log(x)

    def dlogp(self, x):
        return -((np.log(x) - self.mu) / self.sigma ** 2 + 1.0) / x

    def moment(self):
        return np.exp(self.mu + 0.5 * self.sigma ** 2)


class Normal:
    transform = None  # unconstrained: no transformed twin in the point dictionary

    def __init__(self, mu, sigma):
        self.mu, self.sigma = float(mu), float(sigma)

    def logp(self, x):
        z = (x - self.mu) / self.sigma
        return -0.5 * z * z - LOG_SQRT_2PI - np.log(self.sigma)

    def dlogp(self, x):
        return -(x - self.mu) / self.sigma ** 2

    def moment(self):
        return self.mu


class HalfNormal:
    transform = "log"

    def __init__(self, sigma):
        self.sigma = float(sigma)

    def logp(self, x):
        return -0.5 * (x / self.sigma) ** 2 + 0.5 * np.log(2.0 / np.pi) - np.log(self.sigma)

    def dlogp(self, x):
        return -x / self.sigma ** 2

    def moment(self):
        return self.sigma


class TruncatedNormal:
    transform = "interval"

    def __init__(self, mu, sigma, lower, upper):
        self.mu, self.sigma, self.lower, self.upper = float(mu), float(sigma), float(lower), float(upper)
        a = (self.lower - self.mu) / self.sigma
        b = (self.upper - self.mu) / self.sigma
        self.lognorm = np.log(0.5 * (erf(b / np.sqrt(2.0)) - erf(a / np.sqrt(2.0))))

    def logp(self, x):
        z = (x - self.mu) / self.sigma
        return -0.5 * z * z - LOG_SQRT_2PI - np.log(self.sigma) - self.lognorm

    def dlogp(self, x):
        return -(x - self.mu) / self.sigma ** 2

    def moment(self):
        return 0.5 * (self.lower + self.upper)


def forward(dist, x):
    """Constrained -> unconstrained value."""
    if dist.transform is None:
        return np.asarray(x, dtype=np.float64)
    if dist.transform == "log":
        return np.log(x)
    return np.log(x - dist.lower) - np.log(dist.upper - x)


def backward(dist, q):
    """Unconstrained -> constrained value, d x / d q, log |dx/dq| and d log|dx/dq| / dq."""
    if dist.transform is None:
        return q, np.ones_like(q), np.zeros_like(q), np.zeros_like(q)
    if dist.transform == "log":
        x = np.exp(q)
        return x, x, q, np.ones_like(q)
    w = dist.upper - dist.lower
    s = expit(q)
    x = dist.lower + w * s
    return x, w * s * (1.0 - s), np.log(w) + _log_sigmoid(q) + _log_sigmoid(-q), 1.0 - 2.0 * s


class HyperModel:
    """Free hyper-parameters of one GP fit, in PyMC creation order, with the map between the flat
    unconstrained vector the optimiser / sampler moves and the C-ABI theta vector."""

    def __init__(self, nx, kerns, noise=True, truncate=False, jitter=1e-6, n_iwgp=0, n_cwgp_pos=0, n_cwgp=0):
        self.nx, self.kerns, self.nkern = int(nx), list(kerns), len(kerns)
        self.noise, self.truncate, self.jitter = bool(noise), bool(truncate), float(jitter)
        self.vars = []  # (name, dist, size)
        if noise:
            gv = TruncatedNormal(0.0, 1e-3, 1e-15, 1.0) if truncate else HalfNormal(1e-3)
            self.vars.append(("gv", gv, 1, True))
        if truncate:
            self.vars.append(("l", TruncatedNormal(0.5, 0.15, 1e-3, 100.0), nx * self.nkern, False))
            self.vars.append(("kv", TruncatedNormal(1.0, 0.15, 1e-1, 100.0), self.nkern, False))
        else:
            self.vars.append(("l", LogNormal(0.0, 1.0), nx * self.nkern, False))
            self.vars.append(("kv", LogNormal(0.56, 0.75), self.nkern, False))
        warp_pos = (lambda: TruncatedNormal(1.0, 1.0, 1e-3, 5.0)) if truncate else (lambda: LogNormal(0.0, 0.25))
        if n_iwgp > 0:
            self.vars.append(("iwgp", warp_pos(), int(n_iwgp), False))
        if n_cwgp_pos > 0:
            self.vars.append(("cwgp_pos", warp_pos(), int(n_cwgp_pos), False))
        if n_cwgp > 0:
            self.vars.append(("cwgp", TruncatedNormal(0.0, 1.0, -10.0, 10.0) if truncate else Normal(0.0, 1.0),
                              int(n_cwgp), False))
        if "RatQuad" in self.kerns:
            self.vars.append(("alpha", LogNormal(0.56, 0.75), 1, True))
        self.nq = sum(v[2] for v in self.vars)
        self.ntheta = self.nkern * self.nx + 2 * self.nkern + 2

    def transformed_name(self, name, dist):
        return name if dist.transform is None else f"{name}_{dist.transform}__"

    def initial_point(self):
        """[3P] model.initial_point(): transformed moments of the priors (find_MAP's default start)."""
        return np.concatenate([np.full(size, forward(dist, dist.moment())) for _, dist, size, _ in self.vars])

    def split(self, q):
        out, o = {}, 0
        for name, dist, size, _ in self.vars:
            out[name] = np.asarray(q[o : o + size], dtype=np.float64)
            o += size
        return out

    def constrain(self, q):
        """dict of natural-scale values from the unconstrained vector."""
        parts = self.split(q)
        return {name: backward(dist, parts[name])[0] for name, dist, _, _ in self.vars}

    def theta(self, values):
        """C-ABI theta [l, kv, alpha, gv, jitter] from natural-scale values."""
        nk, nx = self.nkern, self.nx
        th = np.empty(self.ntheta)
        th[: nk * nx] = values["l"]
        th[nk * nx : nk * nx + nk] = values["kv"]
        th[nk * nx + nk : nk * nx + 2 * nk] = values["alpha"][0] if "alpha" in values else 1.0
        th[nk * nx + 2 * nk] = values["gv"][0] if "gv" in values else 0.0
        th[nk * nx + 2 * nk + 1] = self.jitter
        return th

    def _theta_grad_slices(self, gtheta):
        nk, nx = self.nkern, self.nx
        g = {"l": gtheta[: nk * nx], "kv": gtheta[nk * nx : nk * nx + nk]}
        ga = gtheta[nk * nx + nk : nk * nx + 2 * nk]
        g["alpha"] = np.array([sum(ga[i] for i, k in enumerate(self.kerns) if k == "RatQuad")])
        g["gv"] = np.array([gtheta[nk * nx + 2 * nk]])
        return g

    def logp_dlogp(self, q, lml_grad, jacobian=True, likelihood=None):
        """Joint log-posterior in the unconstrained space and its gradient.
        ``lml_grad(theta) -> (lml, dlml/dtheta)`` is the device callable (MiGP.lml_grad).  With warp
        variables in the model, ``likelihood(values, theta) -> (loglik, dloglik/dtheta, {name: dloglik/dvalue})``
        replaces it: it warps the data with the current parameters, evaluates the LML (+ warp Jacobian) on the
        device and returns the extra gradients.  ``jacobian=False`` is what pm.find_MAP optimises."""
        parts = self.split(q)
        values, dxdq, prior, dprior = {}, {}, 0.0, {}
        with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
            for name, dist, _, _ in self.vars:
                x, dx, lj, dlj = backward(dist, parts[name])
                values[name], dxdq[name] = x, dx
                prior += np.sum(dist.logp(x))
                dprior[name] = dist.dlogp(x) * dx
                if jacobian:
                    prior += np.sum(lj)
                    dprior[name] = dprior[name] + dlj
        theta = self.theta(values)
        if not (np.all(np.isfinite(theta)) and np.isfinite(prior)):
            return -np.inf, np.zeros(self.nq)  # overflowed transform far out in the tails: PyMC's logp is -inf there
        if likelihood is None:
            lml, gtheta = lml_grad(theta)
            gextra = {}
        else:
            lml, gtheta, gextra = likelihood(values, theta)
        if not np.isfinite(lml):
            return -np.inf, np.zeros(self.nq)
        gl = self._theta_grad_slices(gtheta)
        gl.update(gextra)
        grad = np.concatenate([dprior[name] + gl[name] * dxdq[name] for name, _, _, _ in self.vars])
        return prior + lml, grad

    def point_dict(self, q):
        """PyMC-style point: transformed and natural values, e.g. {'l_log__':..., 'l':..., 'kv':...}
        (the keys recorded at tutorial/tutorial.ipynb:529)."""
        parts = self.split(q)
        out = {}
        for name, dist, size, scalar in self.vars:
            x = backward(dist, parts[name])[0]
            tq = parts[name]
            if dist.transform is not None:
                out[self.transformed_name(name, dist)] = np.array(tq[0]) if scalar else tq.copy()
            out[name] = np.array(x[0]) if scalar else x.copy()
        return out

    def q_from_point(self, point):
        """Inverse of point_dict from natural values (used by fit(method='none') and MAP polish)."""
        qs = []
        for name, dist, size, _ in self.vars:
            x = np.atleast_1d(np.asarray(point[name], dtype=np.float64))
            qs.append(forward(dist, np.broadcast_to(x, (size,))))
        return np.concatenate(qs)
