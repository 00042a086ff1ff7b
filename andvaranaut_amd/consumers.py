"""Consumers of the fitted GP that the reference keeps on GPMCMC: Bayesian optimisation (``BO``,
gpmcmc.py:601-906) and the Bayesian inverse solver (``inverse_opt``, gpmcmc.py:1040-1217).  Both build a
second PyMC model whose free variables are the INPUTS x and differentiate the GP through them; here the
device supplies those derivatives (mi_gp_predict_grad for the single-point predictive of BO's refinement,
mi_gp_grad_x for the joint (N + nobs) likelihood of inverse_opt) and the same L-BFGS-B / NUTS drivers as
``fit`` move x in PyMC's transformed space."""
import numpy as np
from scipy.optimize import Bounds, differential_evolution

from .backend import MiGP
from .lhc import latin_sample
from .nuts import Trace, sample_chain
from .optimize import find_MAP
from .priors import Normal, TruncatedNormal, backward, forward


class Uniform:
    """pm.Uniform(lower, upper): constant density, interval transform."""

    transform = "interval"

    def __init__(self, lower, upper):
        self.lower, self.upper = float(lower), float(upper)

    def logp(self, x):
        return -np.log(self.upper - self.lower) * np.ones_like(x)

    def dlogp(self, x):
        return np.zeros_like(x)

    def moment(self):
        return 0.5 * (self.lower + self.upper)


def _loc_scale(frozen):
    _, loc, scale = frozen.dist._parse_args(*frozen.args, **frozen.kwds)
    return float(loc), float(scale)


def pymc_prior(frozen, allow_truncnorm=False):
    """scipy frozen distribution -> the PyMC prior the reference creates for an input variable
    (gpmcmc.py:702-728 for BO, :1053-1094 for inverse_opt)."""
    name = frozen.dist.name
    if name == "uniform":
        loc, scale = _loc_scale(frozen)
        return Uniform(loc, loc + scale)
    if name == "norm":
        loc, scale = _loc_scale(frozen)
        return Normal(loc, scale)
    if name == "truncnorm" and allow_truncnorm:
        loc, scale = _loc_scale(frozen)
        lo, hi = frozen.support()
        return TruncatedNormal(loc, scale, lo, hi)
    raise Exception("Prior distribution conversion from scipy to pymc not implemented")


class InputModel:
    """Free variables x0..x{nx-1} with their priors; ``potential(x) -> (value, d value / d x)`` is added to the
    prior log-density (pm.Potential).  Same conventions as HyperModel: L-BFGS-B / NUTS move the transformed
    vector, find_MAP drops the transform Jacobian."""

    def __init__(self, dists):
        self.dists = list(dists)
        self.nq = len(self.dists)

    def initial_point(self):
        return np.array([float(forward(d, np.array(d.moment()))) for d in self.dists])

    def q_from_x(self, x):
        return np.array([float(forward(d, np.array(float(xi)))) for d, xi in zip(self.dists, x)])

    def x_from_q(self, q):
        return np.array([float(backward(d, np.array(float(qi)))[0]) for d, qi in zip(self.dists, q)])

    def logp_dlogp(self, q, potential, jacobian=True):
        x, dxdq, val, grad = np.empty(self.nq), np.empty(self.nq), 0.0, np.zeros(self.nq)
        for j, d in enumerate(self.dists):
            xj, dx, lj, dlj = backward(d, np.array(float(q[j])))
            x[j], dxdq[j] = float(xj), float(dx)
            val += float(d.logp(xj))
            grad[j] = float(d.dlogp(xj)) * dxdq[j]
            if jacobian:
                val += float(lj)
                grad[j] += float(dlj)
        pv, pg = potential(x)
        if not np.isfinite(pv):
            return -np.inf, np.zeros(self.nq)
        return val + pv, grad + np.asarray(pg) * dxdq

    def point_dict(self, q):
        out = {}
        x = self.x_from_q(q)
        for j, d in enumerate(self.dists):
            if d.transform is not None:
                out[f"x{j}_{d.transform}__"] = np.array(q[j])
            out[f"x{j}"] = np.array(x[j])
        return out


def _con_and_der(conrev, v):
    """con(v) and d con / d v for one input conversion (analytic ``der`` when the class has it)."""
    c = float(conrev.con(np.array([v]))[0])
    if hasattr(conrev, "der"):
        return c, float(np.asarray(conrev.der(np.array([v])))[0])
    h = 1e-6 * max(1.0, abs(v))
    return c, float((conrev.con(np.array([v + h]))[0] - conrev.con(np.array([v - h]))[0]) / (2 * h))


class ConsumersMixin:
    # ------------------------------------------------------------------ shared drivers
    def _drive_input_model(self, imodel, potential, method, start_x=None, random_start=False, **kwargs):
        """find_MAP or NUTS over the inputs; returns (mp dict with natural 'x{j}', data)."""
        f_map = lambda q: imodel.logp_dlogp(q, potential, jacobian=False)  # noqa: E731
        if method == "map":
            if start_x is not None:
                q0 = imodel.q_from_x(start_x)
            elif random_start:  # start = {ky: np.random.normal()} in the transformed space (gpmcmc.py:827,1168)
                q0 = np.random.normal(size=imodel.nq)
            else:
                q0 = imodel.initial_point()
            q, info = find_MAP(f_map, q0, maxeval=kwargs.get("maxeval", 5000), progressbar=kwargs.get("progressbar", False))
            return imodel.point_dict(q), info
        if method not in ("mcmc_mean", "mcmc_map"):
            raise Exception("method must be one of map, mcmc_map, or mcmc_mean")
        f = lambda q: imodel.logp_dlogp(q, potential, jacobian=True)  # noqa: E731
        chains = int(kwargs.get("chains", 2))
        seeds = np.random.SeedSequence(kwargs.get("random_seed")).spawn(chains)
        res = [sample_chain(f, imodel.initial_point(), draws=kwargs.get("draws", 1000), tune=kwargs.get("tune", 1000),
                            target_accept=kwargs.get("target_accept", 0.8), seed=seeds[c]) for c in range(chains)]
        qs = np.stack([r["q"] for r in res])
        posterior = {}
        for c in range(chains):
            for k in range(qs.shape[1]):
                for name, val in imodel.point_dict(qs[c, k]).items():
                    posterior.setdefault(name, np.empty((chains, qs.shape[1])))[c, k] = val
        data = Trace(posterior, {"lp": np.stack([r["lp"] for r in res])})
        if method == "mcmc_mean":
            return self.mean_extract(data), data
        mp = self.map_extract(data)
        try:
            q, _ = find_MAP(f_map, imodel.q_from_x([mp[f"x{j}"] for j in range(imodel.nq)]))
            mp = imodel.point_dict(q)
        except Exception:
            pass
        return mp, data

    def _append_evaluated(self, xsamp):
        """Evaluate the target (and the mean function) at new points and grow every data array
        (gpmcmc.py:878-888, 1196-1204)."""
        xs, ys = self._GPMCMC__evaluate(xsamp, self.target)
        if len(xs) == 0:
            raise RuntimeError("target evaluation failed at the proposed point")
        xm, ym = self._GPMCMC__evaluate(xs, self.mean)
        self.x = np.r_[self.x, xs]
        self.y = np.r_[self.y, ys]
        self.xc = np.r_[self.xc, self.__xconrev__(xs)]
        self.yc = np.r_[self.yc, self.__yconrev__(ys)]
        self.ym = np.r_[self.ym, ym.reshape(len(xs), self.ny)]
        self.nsamp = len(self.x)
        return xs, ys, ym

    def __yconrev__(self, yin, mode="con"):
        yout = np.zeros_like(yin)
        if mode == "con":
            yout[:, 0] = self.yconrevs[0].con(yin[:, 0])
        elif mode == "rev":
            yout[:, 0] = self.yconrevs[0].rev(yin[:, 0])
        else:
            raise Exception("Error: Mode must be one of con or rev")
        return yout

    def __xconrev__(self, xin, mode="con"):
        xout = np.zeros_like(xin)
        for i in range(self.nx):
            if mode == "con":
                xout[:, i] = self.xconrevs[i].con(xin[:, i])
            elif mode == "rev":
                xout[:, i] = self.xconrevs[i].rev(xin[:, i])
            else:
                raise Exception("Error: Mode must be one of con or rev")
        return xout

    # ------------------------------------------------------------------ BO
    def _bo_potential(self, method, opt_type, normvar, jitter):
        """pm.Potential of BO's single-point model (gpmcmc.py:766-815) as x -> (value, gradient): the
        Gauss-Hermite reverted mean (+-), the (normalised) variance, or the expected improvement."""
        import torch

        theta = self._theta_from_hypers(self.hypers, jitter)
        xi_np, wi_np = np.polynomial.hermite.hermgauss(8)
        xi, wi = torch.from_numpy(xi_np), torch.from_numpy(wi_np)
        state = {"fresh": True}
        ycon = self.yconrevs[0]

        def potential(x):
            cd = [_con_and_der(self.xconrevs[j], float(x[j])) for j in range(self.nx)]
            xin = np.array([[c for c, _ in cd]])
            dxin = np.array([dv for _, dv in cd])
            # the reference's single-point variance is kstarstar - v^T v, without the noise term (gpmcmc.py:784-785)
            mu, var, dmu, dvar = self._ensure_gp().predict_grad(theta, xin, pred_noise=False, refactor=state["fresh"])
            state["fresh"] = False
            m_t = torch.tensor(float(mu[0]), dtype=torch.float64, requires_grad=True)
            v_t = torch.tensor(max(float(var[0]), 1e-300), dtype=torch.float64, requires_grad=True)
            yi = torch.sqrt(2 * v_t) * xi + m_t * torch.ones_like(xi)
            yir = ycon.revmc(yi) if hasattr(ycon, "revmc") else ycon.rev(yi)
            if self.mean != self.zero_mean:  # evaluated at the converted point, as written at gpmcmc.py:790
                yir = yir + float(np.atleast_1d(self.mean(xin[0]))[0])
            ypmean = (wi * yir).sum() / np.sqrt(np.pi)
            if method in ("eps-RS", "exploit"):
                pot = ypmean if opt_type == "max" else -ypmean
            elif method == "explore":
                ym2 = (wi * yir ** 2).sum() / np.sqrt(np.pi)
                pot = ym2 - ypmean ** 2
                if normvar:
                    pot = pot / ypmean ** 2
            elif method == "EI":
                diff = (yir - self.yopt) if opt_type == "max" else (self.yopt - yir)
                pot = (wi * torch.clamp(diff, min=0.0)).sum() / np.sqrt(np.pi)
            else:
                raise Exception("method must be one of eps-RS ,EI, exploit, or explore")
            pot.backward()
            gm = 0.0 if m_t.grad is None else float(m_t.grad)
            gv = 0.0 if v_t.grad is None else float(v_t.grad)
            return float(pot.detach()), (gm * dmu[0] + gv * dvar[0]) * dxin

        return potential

    def BO(self, opt_type="min", opt_method="predict", fit_method="map", max_iter=16, method="EI", eps=0.1, iwgp=False,
           cwgp=False, jitter=1e-6, conv=0.01, predict_samps=10000, normvar=True, refine=True, **kwargs):
        """Bayesian optimisation loop of gpmcmc.py:601-906: propose (batched prediction over an LHC sample or
        differential evolution, optionally refined by a MAP on the differentiable single-point predictive, or a
        MAP / MCMC on it alone), evaluate the target, refit, until the proposal stops moving."""
        if self.ny > 1:
            raise Exception("Bayesian minimisation only implemented for single output")
        if opt_type == "max":
            xoptf, yoptf = np.argmax, np.max
        elif opt_type == "min":
            xoptf, yoptf = np.argmin, np.min
        else:
            raise Exception("Error: opt_type argument must be one of max or min")
        self.xopt = self.x[xoptf(self.y[:, 0]), :]
        self.yopt = yoptf(self.y)
        if self.verbose:
            print("Running Bayesian minimisation...")
            print(f"Current optima is {self.yopt} at x point {self.xopt}")
        if self.m is None:
            raise Exception("Model must be fitted before running Bayesian optimisation")
        if method == "exploit":
            eps = 0.0
        if method not in ("eps-RS", "EI", "exploit", "explore"):
            raise Exception("method must be one of eps-RS ,EI, exploit, or explore")
        lbs = np.array([p.ppf(1e-8) for p in self.priors])
        ubs = np.array([p.isf(1e-8) for p in self.priors])
        bnds = Bounds(lbs, ubs)
        xsampold = np.array([[1e300 for _ in range(self.nx)]])
        for it in range(max_iter):
            if self.verbose:
                print(f"Iteration {it + 1}")
            xsamp = None
            if opt_method in ("DE", "predict"):
                verb, self.verbose = self.verbose, False
                try:
                    def optf(x):
                        x = np.atleast_2d(np.asarray(x, dtype=np.float64))
                        if method in ("eps-RS", "exploit"):
                            ym = self.predict(x)
                            return ym[:, 0] if opt_type == "min" else -ym[:, 0]
                        if method == "explore":
                            _, yv = self.predict(x, return_var=True, normvar=normvar)
                            return -yv[:, 0]
                        return -self.predict(x, EI=True, EIopt=opt_type)[:, 0]

                    roll = np.random.rand()
                    if method != "eps-RS" or roll > eps:
                        if opt_method == "DE":
                            # population evaluated in one batched prediction per generation
                            res = differential_evolution(lambda xs: optf(xs.T), bnds, vectorized=True,
                                                         updating="deferred")
                            xsamp, fopt = np.array([res.x]), res.fun
                        else:
                            # the reference draws these with scipy's "random-cd" discrepancy optimisation
                            # (lhc.py:42), 14 s of host time for 10 000 points; a plain Latin hypercube serves an
                            # arg-min sweep as well and keeps the iteration on the device's time scale
                            xsamps = latin_sample(self.priors, predict_samps, seed=int(np.random.randint(2 ** 31 - 1)),
                                                  optimization="random-cd" if predict_samps <= 2000 else None)
                            ysamps = optf(xsamps)
                            xsamp, fopt = np.array([xsamps[np.argmin(ysamps), :]]), np.min(ysamps)
                        if verb:
                            print(f"Function opt is {float(fopt):0.3f}")
                    else:
                        xsamp = np.array([[p.rvs() for p in self.priors]])
                finally:
                    self.verbose = verb
            if opt_method not in ("DE", "predict") or (opt_method == "predict" and refine):
                imodel = InputModel([pymc_prior(p) for p in self.priors])
                potential = self._bo_potential(method, opt_type, normvar, jitter)
                roll = np.random.rand()
                if method != "eps-RS" or roll > eps:
                    if opt_method == "map" or (opt_method == "predict" and refine):
                        if opt_method == "map":
                            mp, _ = self._drive_input_model(imodel, potential, "map", random_start=True, **kwargs)
                        else:
                            if self.verbose:
                                print(f"Refining {xsamp[0, :]}")
                            mp, _ = self._drive_input_model(imodel, potential, "map", start_x=xsamp[0], **kwargs)
                    else:
                        mp, _ = self._drive_input_model(imodel, potential, opt_method, **kwargs)
                    xsamp = np.array([[float(mp[f"x{j}"]) for j in range(self.nx)]])
                else:
                    xsamp = np.array([[p.rvs() for p in self.priors]])
            xdiff = np.sum(np.abs(xsamp - xsampold) / np.abs(xsampold)) / self.nx
            if xdiff < conv:
                if self.verbose:
                    print(f"Convergence at relative tolerance {xdiff} achieved with point {xsamp}")
                break
            if self.verbose and it > 0:
                print(f"Relative convergence in sample: {xdiff}")
            xsampold = xsamp
            ypred = self.predict(xsamp)
            if self.verbose:
                print(f"Predicted {ypred} at x point {xsamp}")
            xs, ys, ym = self._append_evaluated(xsamp)
            if self.verbose:
                print(f"New sample is {ys + ym} at x point {xs}")
            self.xopt = self.x[xoptf(self.y[:, 0]), :]
            self.yopt = yoptf(self.y)
            if fit_method == "map":
                try:
                    self.fit(method=fit_method, iwgp=iwgp, cwgp=cwgp, start=self.hypers)
                except Exception:
                    self.fit(method=fit_method, iwgp=iwgp, cwgp=cwgp)
            else:
                self.fit(method=fit_method, iwgp=iwgp, cwgp=cwgp)
        return self.xopt, self.yopt

    # ------------------------------------------------------------------ inverse_opt
    def _gh_stats_inv(self, y, yv, deg=8):
        """Variance of the CONVERTED observation by Gauss-Hermite quadrature (gpmcmc.py:573-585).  As written
        there the loop overwrites its result, so the value of the last observation is returned for all."""
        xi, wi = np.polynomial.hermite.hermgauss(deg)
        yvcon = 0.0
        for i in range(len(y)):
            yi = np.sqrt(2 * yv[i, 0]) * xi + y[i, 0]
            yir = self.yconrevs[0].con(yi)
            ym = np.sum(wi * yir) / np.sqrt(np.pi)
            yvcon = np.sum(wi * yir ** 2) / np.sqrt(np.pi) - ym ** 2
        return yvcon

    def inverse_opt(self, yobs, yvarobs=None, method="map", evaluate_opt=False, jitter=1e-6, **kwargs):
        """Bayesian inverse solve of gpmcmc.py:1040-1217: posterior over the input x that produced the
        observation(s) ``yobs`` -- the GP likelihood of the training set extended by ``nobs`` rows that all sit
        at the unknown x, times the input priors.  The (N + nobs) Cholesky, its LML and dLML/dX run on the device."""
        if self.m is None:
            raise Exception("Model must be fitted before running Bayesian optimisation")
        if self.verbose:
            print("Running Bayesian inverse solver...")
        imodel = InputModel([pymc_prior(p, allow_truncnorm=True) for p in self.priors])
        yobs = np.asarray(yobs, dtype=np.float64).reshape(-1, 1)
        nobs, n = len(yobs), self.nsamp
        yin = np.zeros(n + nobs)
        yin[:-nobs] = self.yc[:, 0]
        yin[-nobs:] = self.yconrevs[0].con(yobs[:, 0])
        # diagonal added to K: standard deviations, as written at gpmcmc.py:1134-1146,1158
        ynoise = np.zeros(n + nobs)
        gv = float(np.atleast_1d(self.hypers["gv"])[0]) if self.noise else 0.0
        ynoise[:-nobs] = np.sqrt(gv + jitter)
        if yvarobs is not None:
            ynoise[-nobs:] = np.sqrt(self._gh_stats_inv(yobs, np.asarray(yvarobs, dtype=np.float64).reshape(-1, 1)))
        yfull = np.r_[self.y[:, 0], yobs[:, 0]]
        yder = np.asarray(self.yconrevs[0].der(yfull)) if hasattr(self.yconrevs[0], "der") else np.ones_like(yfull)
        logjac = float(np.sum(np.log(yder)))
        theta = self._theta_from_hypers(self.hypers, 0.0)
        nk = self.nkern
        theta[nk * self.nx + 2 * nk] = 0.0  # the noise enters through ynoise only
        xaug = np.zeros((n + nobs, self.nx))
        xaug[:-nobs] = self.xc
        gpi = MiGP(xaug, yin, self.kernel, device=self.device)
        try:
            gpi.set_diag(ynoise)

            def potential(x):
                cd = [_con_and_der(self.xconrevs[j], float(x[j])) for j in range(self.nx)]
                xaug[-nobs:, :] = np.array([c for c, _ in cd])
                gpi.update_data(X=xaug)
                val, _, _, gx = gpi.lml_grad_data(theta, want_x=True)
                if not np.isfinite(val):
                    return -np.inf, np.zeros(self.nx)
                return val + logjac, gx[-nobs:].sum(axis=0) * np.array([dv for _, dv in cd])

            mp, data = self._drive_input_model(imodel, potential, method, random_start=True, **kwargs)
        finally:
            gpi.close()
        xopt = np.array([[float(mp[f"x{j}"]) for j in range(self.nx)]])
        ypred = self.predict(xopt)
        if self.verbose:
            print(f"Predicted {ypred} at x point {xopt}")
        if evaluate_opt:
            xs, ys, ym = self._append_evaluated(xopt)
            if self.verbose:
                print(f"Actual evaluation is {ys + ym} at x point {xs}")
            return data, xopt[0, :], ys[0]
        return data, xopt[0, :]
