"""GPMCMC: the reference's GP-surrogate facade (andvaranaut/gpmcmc.py:30) with its hot path -- the
GP log marginal likelihood, its hyper-parameter gradient and the posterior conditional -- running on
an MI355X through libmi_gp.so instead of PyMC / PyTensor / SciPy-LAPACK.

Same constructor, ``set_data`` / ``sample`` / ``fit`` / ``predict`` / ``change_model`` signatures and the
same ``hypers`` dictionary keys as the reference (gpmcmc.py:31-32,122,158,175-177,472,522-523;
keys recorded at tutorial/tutorial.ipynb:529), including the warped fits ``fit(iwgp=True)`` /
``fit(cwgp=True)`` whose warp parameters are optimised / sampled with the hyper-parameters
(gpmcmc.py:211-279,319): the device supplies dLML/dX and dLML/dy, torch.autograd carries them through the
warps.  ``BO`` and ``inverse_opt`` live in consumers.py.  Out of scope here:
dask-parallel target execution, plots."""
import copy
import os
import re
import threading
from time import time as stopwatch

import numpy as np

from .backend import MiGP, parse_kernel
from .consumers import ConsumersMixin
from .lhc import latin_sample
from .nuts import Trace, sample_chain
from .optimize import find_MAP
from .priors import HyperModel
from .transform import _none_conrev, wgp

# handles that may evaluate at once per device with in-kernel polls on their cross-stream edges (include/mi_gp.h, option 26)
MAX_POLLING_HANDLES = 6


def save_object(obj, fname):
    """core.py:21-23: whole-object checkpoint with cloudpickle (a GPMCMC drops its device handle, see __getstate__)."""
    import cloudpickle

    with open(fname, "wb") as f:
        cloudpickle.dump(obj, f)


def load_object(fname):
    """core.py:24-27."""
    import cloudpickle

    with open(fname, "rb") as f:
        return cloudpickle.load(f)


class GPMCMC(ConsumersMixin):
    def __init__(self, xconrevs=None, yconrevs=None, kernel="RBF", noise=True, mean=0, nx=None, ny=None,
                 priors=None, target=None, parallel=False, nproc=1, constraints=None, rundir=None, verbose=True,
                 pulse=1, device=0):
        # argument checks of _core.__init__ (core.py:57-86)
        if (not isinstance(nx, int)) or (nx < 1):
            raise Exception("Error: must specify an integer number of input dimensions > 0")
        if (not isinstance(ny, int)) or (ny < 1):
            raise Exception("Error: must specify an integer number of output dimensions > 0")
        if (not isinstance(priors, list)) or (len(priors) != nx):
            raise Exception("Error: must provide list of scipy.stats univariate priors of length nx")
        if any(getattr(p, "__module__", None) != "scipy.stats._distn_infrastructure" for p in priors):
            raise Exception("Error: must provide list of scipy.stats univariate priors of length nx")
        if not callable(target):
            raise Exception("Error: must provide target function which produces output from specified inputs")
        if not isinstance(parallel, bool):
            raise Exception("Error: parallel must be type bool.")
        if parallel:
            raise NotImplementedError("dask-parallel target evaluation (core.py:105-134) is outside this backend")
        keys = ["constraints", "lower_bounds", "upper_bounds"]
        if (constraints is not None) and ((not isinstance(constraints, dict)) or not all(k in constraints for k in keys)):
            raise Exception(f"Error: provided constraints must be a dictionary with keys {keys} and list items.")
        self.nx, self.ny, self.priors, self.target = nx, ny, priors, target
        self.parallel, self.nproc, self.pulse = parallel, nproc, pulse
        self.constraints, self.verbose = constraints, verbose
        self.rundir = "runs" if rundir is None else rundir
        self.device = int(device)
        self.nsamp = 0
        self.x = np.empty((0, nx))
        self.y = np.empty((0, ny))
        self.xc = copy.deepcopy(self.x)
        self.yc = copy.deepcopy(self.y)
        self.__conrev_check(xconrevs, yconrevs)
        self.ym = copy.deepcopy(self.y)
        self.change_model(kernel, noise, mean)
        self.train = None
        self.test = None

    # ------------------------------------------------------------------ data plumbing
    def zero_mean(self, x):
        return np.zeros(self.ny)

    def __conrev_check(self, xconrevs, yconrevs):
        """gpmcmc.py:98-119."""
        xconrevs = [None] * self.nx if xconrevs is None else xconrevs
        yconrevs = [None] * self.ny if yconrevs is None else yconrevs
        if not isinstance(xconrevs, list) or len(xconrevs) != self.nx:
            raise Exception("Error: xconrevs must be None or list of conversion/reversion classes of size nx")
        if not isinstance(yconrevs, list) or len(yconrevs) != self.ny:
            raise Exception("Error: yconrevs must be None or list of conversion/reversion classes of size ny")
        for lst in (xconrevs, yconrevs):
            for j, c in enumerate(lst):
                if c is None:
                    lst[j] = _none_conrev()
                elif (not callable(c.con)) or (not callable(c.rev)):
                    raise Exception("Error: Provided data conversion/reversion function not callable.")
        self.xconrevs, self.yconrevs = xconrevs, yconrevs

    def __evaluate(self, xsamps, fun):
        """Serial branch of _core.__vector_solver (core.py:137-215): failed / non-finite rows are dropped."""
        keep_x, ys = [], []
        for i in range(len(xsamps)):
            try:
                yout = np.atleast_1d(np.asarray(fun(xsamps[i, :]), dtype=np.float64))
            except Exception as e:
                print(f"Warning: Target function evaluation failed at sample {i} with x values: {xsamps[i, :]}; "
                      f"error message: {e}")
                continue
            if yout.shape != (self.ny,):
                raise Exception("Error: number of target function outputs is not equal to ny")
            if np.any(np.isnan(yout)) or np.any(np.abs(yout) == np.inf):
                print(f"Warning: Target function evaluation returned inf/nan at sample with x values: {xsamps[i, :]}")
                continue
            keep_x.append(xsamps[i, :])
            ys.append(yout)
        if not ys:
            return np.empty((0, self.nx)), np.empty((0, self.ny))
        return np.array(keep_x), np.array(ys)

    def __check_constraints(self, xsamps):
        """core.py:218-246, applied to proposed samples before the target runs (lhc.py:30-31).  As in the reference the
        mask entry of a sample is overwritten by every constraint in turn, so the LAST constraint decides (quirk kept)."""
        n0 = len(xsamps)
        mask = np.ones(n0, dtype=bool)
        for i, xj in enumerate(xsamps):
            for e, f in enumerate(self.constraints["constraints"]):
                flag = True
                res = f(xj)
                lo, hi = self.constraints["lower_bounds"][e], self.constraints["upper_bounds"][e]
                if isinstance(lo, list):
                    for k, b in enumerate(lo):
                        if res[k] < b:
                            flag = False
                    for k, b in enumerate(hi):
                        if res[k] > b:
                            flag = False
                elif res < lo or res > hi:
                    flag = False
                mask[i] = flag
                if not flag:
                    print(f"Sample {i + 1} with x values {xj} removed due to invalidaing constraint {e + 1}.")
        xsamps = xsamps[mask]
        if len(xsamps) < n0:
            print(f"{n0 - len(xsamps)} samples removed due to violating constraints.")
        return xsamps

    def __reconvert(self):
        self.xc = np.empty_like(self.x)
        self.yc = np.empty_like(self.y)
        for i in range(self.nx):
            self.xc[:, i] = self.xconrevs[i].con(self.x[:, i])
        for i in range(self.ny):
            self.yc[:, i] = self.yconrevs[i].con(self.y[:, i] - self.ym[:, i])

    def __mean_eval(self):
        xm, ym = self.__evaluate(self.x, self.mean)
        if len(xm) != len(self.x):
            raise Exception("Mean function not valid at every x point in dataset")
        self.ym = ym.reshape(len(self.x), self.ny)

    def sample(self, nsamps, seed=None):
        """gpmcmc.py:158-172 + lhc.py:24-37."""
        if not isinstance(nsamps, int) or (nsamps < 1):
            raise Exception("Error: nsamps argument must be an integer > 0")
        if self.verbose:
            print(f"Evaluating {nsamps} latin hypercube samples...")
        xs = latin_sample(self.priors, nsamps, seed)
        if self.constraints is not None:
            xs = self.__check_constraints(xs)
        xs, ys = self.__evaluate(xs, self.target)
        self.x = np.r_[self.x, xs]
        self.y = np.r_[self.y, ys]
        self.nsamp = len(self.x)
        self.__mean_eval()
        self.__reconvert()
        self.train = self.test = None

    def set_data(self, x, y):
        """gpmcmc.py:122-137 + lhc.py:113-131."""
        if not isinstance(x, np.ndarray) or len(x.shape) != 2 or x.dtype != "float64" or x.shape[1] != self.nx:
            raise Exception("Error: Setting data requires a 2d numpy array of float64 inputs")
        if not isinstance(y, np.ndarray) or len(y.shape) != 2 or y.dtype != "float64" or y.shape[1] != self.ny:
            raise Exception("Error: Setting data requires a 2d numpy array of float64 outputs")
        for i in range(self.nx):
            intv = self.priors[i].interval(1.0)
            if not all(x[:, i] >= intv[0]) or not all(x[:, i] <= intv[1]):
                raise Exception("Error: provided x data must fit within provided input distribution ranges.")
        self.x, self.y, self.nsamp = x, y, len(x)
        self.__mean_eval()
        self.__reconvert()
        self.train = self.test = None

    def del_samples(self, ndels=None, method="coarse_lhc", idx=None):
        """Drop samples (gpmcmc.py:57-72 + lhc.py:50-97): the ndels points nearest to a fresh Latin-hypercube
        sample ('coarse_lhc'), ndels at random, or the given indexes ('specific'); converted copies follow."""
        if method == "coarse_lhc":
            if not isinstance(ndels, int) or ndels < 1:
                raise Exception("Error: must specify positive int for ndels")
            xsamps = latin_sample(self.priors, ndels)
            for i in range(ndels):
                k = int(np.argmin(np.linalg.norm(self.x - xsamps[i], axis=1)))
                self.x, self.y = np.delete(self.x, k, axis=0), np.delete(self.y, k, axis=0)
                self.xc, self.yc = np.delete(self.xc, k, axis=0), np.delete(self.yc, k, axis=0)
                self.ym = np.delete(self.ym, k, axis=0)
        elif method == "random":
            if not isinstance(ndels, int) or ndels < 1:
                raise Exception("Error: must specify positive int for ndels")
            keep = np.random.choice(np.arange(len(self.x)), size=len(self.x) - ndels, replace=False)
            self.x, self.y, self.xc, self.yc, self.ym = (a[keep] for a in (self.x, self.y, self.xc, self.yc, self.ym))
        elif method == "specific":
            if not isinstance(idx, (int, list)):
                raise Exception("Error: must specify int or list of ints for idx")
            mask = np.ones(len(self.x), dtype=bool)
            mask[idx] = False
            self.x, self.y, self.xc, self.yc, self.ym = (a[mask] for a in (self.x, self.y, self.xc, self.yc, self.ym))
        else:
            raise Exception("Error: method must be one of 'coarse_lhc','random','specific'")
        self.nsamp = len(self.x)
        self.train = self.test = None

    def change_conrevs(self, xconrevs=None, yconrevs=None):
        self.__conrev_check(xconrevs, yconrevs)
        self.__reconvert()

    def change_xconrevs(self, xconrevs=None):
        self.__conrev_check(xconrevs, self.yconrevs)
        self.__reconvert()

    def change_yconrevs(self, yconrevs=None):
        self.__conrev_check(self.xconrevs, yconrevs)
        self.__reconvert()

    def train_test(self, training_frac=0.9):
        from sklearn.model_selection import train_test_split

        self.nsamp = len(self.x)
        self.train, self.test = train_test_split(np.arange(self.nsamp), train_size=training_frac)

    def test_stats(self, revert=True, iwgp=False, cwgp=False, method="none", jitter=1e-6):
        """The numeric half of ``test_plots`` (gpmcmc.py:933-976; the plots are out of scope): condition the model on the
        training split (``train_test``; refitted there with ``method``, 'none' = the stored hypers), predict the held-out
        points and return RMSE, mean absolute / percentage error and R^2 with the arrays ``returndat`` hands back."""
        if getattr(self, "train", None) is None:
            self.train_test()
        xtrain, xtest = self.x[self.train, :], self.x[self.test, :]
        ytrain, ytest = self.y[self.train, :], self.y[self.test, :]
        ymtrain, ymtest = self.ym[self.train, :], self.ym[self.test, :]
        m, gp, hypers, _ = self.__fit(xtrain, ytrain - ymtrain, method, iwgp, cwgp, jitter)
        try:
            xctest = np.column_stack([self.xconrevs[i].con(xtest[:, i]) for i in range(self.nx)])
            saved = self.m, self.hypers
            self.m, self.hypers = m, hypers
            try:
                mu, var = gp.predict(self._theta_from_hypers(hypers, jitter), xctest, pred_noise=True)
            finally:
                self.m, self.hypers = saved
        finally:
            gp.close()  # (self.gp was released by __fit: the next predict rebuilds it on the full data)
        ypred, yvars = mu.reshape((-1, 1)), var.reshape((-1, 1))
        if revert:
            yt = ytest[:, 0]
            ypred, yvars = self.__gh_stats(xtest, ypred, yvars, normvar=False)
            meany = np.mean(self.y)
        else:
            yt = self.yconrevs[0].con(ytest[:, 0] - ymtest[:, 0])
            meany = np.mean(self.yconrevs[0].con((self.y - self.ym)[:, 0]))
        ypred, yvars = ypred[:, 0], yvars[:, 0]
        out = {"rmse": float(np.sqrt(np.mean((ypred - yt) ** 2))), "mea": float(np.mean(np.abs(ypred - yt))),
               "mpe": float(np.mean(np.abs(ypred - yt) / np.abs(yt))),
               "r2": float(1 - np.sum((ypred - yt) ** 2) / np.sum((yt - meany) ** 2)),
               "xtest": xtest if revert else xctest, "ytest": yt, "ypred": ypred, "yvars": yvars}
        if self.verbose:
            print(f"RMSE for y is: {out['rmse']:0.5e}")
            print(f"Mean absoulte error for y is: {out['mea']:0.5e}")
            print(f"Mean percentage error for y is: {out['mpe']:0.5%}")
            print(f"R^2 for y is: {out['r2']:0.5f}")
        return out

    def y_dist(self, mode="hist_kde", nsamps=None, return_data=False, surrogate=True, seed=None):
        """Uncertainty propagation through the surrogate (gpmcmc.py:140-151; tutorial.ipynb cell 34): a Latin-hypercube sample of the
        input priors is pushed through ``predict`` (one device sweep via U = L^-T for large ``nsamps``).  The density plots of
        LHC.__y_dist (lhc.py:100-110) are outside this backend's scope: ``mode`` is validated as there and otherwise ignored.
        ``surrogate=False``: the underlying data set.  ``return_data=True`` returns (x, y) as the reference does."""
        modes = ["hist", "kde", "ecdf", "hist_kde"]
        if mode not in modes:
            raise Exception(f"Error: selected mode must be one of {modes}")
        if not isinstance(surrogate, bool):
            raise Exception("Error: surrogate argument must be of type bool")
        if not surrogate:
            return (self.x, self.y) if return_data else None
        xsamps = latin_sample(self.priors, nsamps, seed)
        ypreds = self.predict(xsamps)
        if return_data:
            return xsamps, ypreds

    def relative_importances(self, logscale=False):
        """Inverse length scales of the fitted model (gpmcmc.py:1030-1037 draws them as a bar chart; here the values are returned)."""
        if self.hypers is None:
            raise Exception("Error: fit the GP before asking for relative importances")
        ri = 1.0 / np.asarray(self.hypers["l"], dtype=np.float64)
        return np.log(ri) if logscale else ri

    def test_plots(self, revert=True, yplots=True, xplots=True, logscale=False, iwgp=False, cwgp=False, method="none",
                   errorbars=True, saveyfig=None, xlab=None, ylab=None, returndat=False):
        """Signature of the reference's ``test_plots`` (gpmcmc.py:933-1026) for drop-in callers: the train / test fit and the
        four printed figures are ``test_stats``; the matplotlib figures themselves are outside this backend's scope, so the
        plotting arguments are accepted and ignored.  ``returndat=True`` returns (xtest, ytest, ypred, yvars) as there."""
        st_ = self.test_stats(revert=revert, iwgp=iwgp, cwgp=cwgp, method=method)
        if returndat:
            return st_["xtest"], st_["ytest"], st_["ypred"], st_["yvars"]

    def change_model(self, kernel=None, noise=None, mean=None):
        """gpmcmc.py:472-519: kernel string grammar, noise flag, mean function; scrubs the fitted model."""
        kernel = self.kernel if kernel is None else kernel
        noise = self.noise if noise is None else noise
        if mean is not None:
            self.mean = self.zero_mean if (not callable(mean) and mean == 0) else mean
            if len(self.x) > 0:
                self.__mean_eval()
                self.__reconvert()
        kerns, ops = parse_kernel(kernel)
        if not isinstance(noise, bool):
            raise Exception("Error: noise must be of type bool")
        self.kernel, self.kerns, self.ops, self.nkern, self.noise = kernel, kerns, ops, len(kerns), noise
        self.__release()
        self.m = None
        self.hypers = None

    def __release(self):
        gp = getattr(self, "gp", None)
        if gp is not None:
            gp.close()
        self.gp = None

    # -- checkpoint / resume: the reference pickles the whole object (core.py:21-27 save_object / load_object).  Here
    # x, y, the conrevs, `m` (the HyperModel) and `hypers` are the state; the device handle is dropped on pickling
    # and rebuilt from them on the next predict / BO / inverse_opt.
    def __getstate__(self):
        state = dict(self.__dict__)
        state["gp"] = None
        return state

    def _ensure_gp(self):
        if self.gp is None and self.hypers is not None and self.m is not None:
            xin, yin = self._converted(self.x, self.y - self.ym)
            self.gp = MiGP(xin, yin, self.kernel, device=self.device)
        return self.gp

    # ------------------------------------------------------------------ fit
    def fit(self, method="map", return_data=False, iwgp=False, cwgp=False, jitter=1e-6, truncate=False,
            restarts=1, **kwargs):
        """gpmcmc.py:175-182."""
        self.m, self.gp, self.hypers, data = self.__fit(self.x, self.y - self.ym, method, iwgp, cwgp, jitter,
                                                        truncate, restarts, **kwargs)
        if return_data:
            return data

    def _converted(self, x, y):
        xin = np.zeros_like(x)
        for i in range(self.nx):
            xin[:, i] = self.xconrevs[i].con(x[:, i])
        yin = self.yconrevs[0].con(y[:, 0])  # only y[:,0] is modelled (gpmcmc.py:279)
        return np.ascontiguousarray(xin), np.ascontiguousarray(yin)

    # -- warps whose parameters are model variables (gpmcmc.py:211-279, 433-462)
    def cwgp_set(self, params, mode="numpy", y=None):
        """Set (mode='numpy') or build (otherwise) the output warp with new parameters (gpmcmc.py:433-441)."""
        if y is None:
            y = self.y - self.ym
        warper = wgp(self.yconrevs[0].warping_names, params, y[:, 0], mode=mode)
        if mode == "numpy":
            self.change_yconrevs([warper])
        else:
            return warper

    def iwgp_set(self, params, mode="numpy", x=None):
        """Set / build the input warps with new parameters (gpmcmc.py:443-462)."""
        if x is None:
            x = self.x
        xconrevs, rc = [], 0
        for i in range(self.nx):
            if isinstance(self.xconrevs[i], wgp):
                ran = len(self.xconrevs[i].params)
                xconrevs.append(wgp(self.xconrevs[i].warping_names, params[rc : rc + ran], y=x[:, i],
                                    xdist=self.priors[i], mode=mode))
                rc += ran
            else:
                xconrevs.append(self.xconrevs[i])
        if mode == "numpy":
            self.change_xconrevs(xconrevs=xconrevs)
        else:
            return xconrevs

    def _warp_sizes(self, iwgp, cwgp):
        n_i = n_pos = n_free = 0
        if iwgp:
            n_i = sum(c.np for c in self.xconrevs if isinstance(c, wgp))
            if n_i == 0:
                raise Exception("Error: iwgp set to true but none of xconrevs are wgp classes")
        if cwgp:
            if not isinstance(self.yconrevs[0], wgp):
                raise Exception("Error: cwgp set to true but yconrevs class is not wgp")
            if self.yconrevs[0].np == 0:
                raise Exception("Error: cwgp set to true but wgp class has no tuneable parameters")
            n_pos = int(np.sum(self.yconrevs[0].pos[: self.yconrevs[0].np]))
            n_free = self.yconrevs[0].np - n_pos
        return n_i, n_pos, n_free

    def _cwgp_params(self, pos_vals, free_vals):
        """Interleave the positive and unconstrained output-warp parameters in warp order (gpmcmc.py:263-275)."""
        out, rc, rcpos = [], 0, 0
        for i in range(self.yconrevs[0].np):
            if self.yconrevs[0].pos[i]:
                out.append(pos_vals[rcpos])
                rcpos += 1
            else:
                out.append(free_vals[rc])
                rc += 1
        return out

    def _warp_likelihood(self, gp, x, y, xin0, iwgp, cwgp):
        """likelihood(values, theta) for HyperModel.logp_dlogp with warp variables: warp the data with the
        current parameters (torch, host), evaluate LML + its theta / X / y gradients on the device, add the
        warp Jacobian sum(log y') (gpmcmc.py:319) and pull the data gradients back onto the warp parameters."""
        import torch

        xt = torch.from_numpy(np.ascontiguousarray(x))
        yt = torch.from_numpy(np.ascontiguousarray(y))

        def likelihood(values, theta):
            leaves = {}
            xin_t = yin_t = yder_t = None
            if iwgp:
                leaves["iwgp"] = torch.tensor(values["iwgp"], dtype=torch.float64, requires_grad=True)
                warpers = self.iwgp_set(leaves["iwgp"], mode="torch", x=xt)
                cols = [warpers[i].conmc(xt[:, i]) if isinstance(warpers[i], wgp) else torch.from_numpy(xin0[:, i])
                        for i in range(self.nx)]
                xin_t = torch.stack(cols, dim=1)
            if cwgp:
                for nm in ("cwgp_pos", "cwgp"):
                    if nm in values:
                        leaves[nm] = torch.tensor(values[nm], dtype=torch.float64, requires_grad=True)
                rvs = self._cwgp_params(leaves.get("cwgp_pos"), leaves.get("cwgp"))
                warper = self.cwgp_set(torch.stack(list(rvs)), mode="torch", y=yt)
                yin_t = warper.conmc(yt[:, 0])
                yder_t = warper.dermc(yt[:, 0])
                if not bool(torch.all(torch.isfinite(yin_t))) or not bool(torch.all(yder_t > 0)):
                    return -np.inf, None, {}
            gp.update_data(X=None if xin_t is None else xin_t.detach().numpy(),
                           y=None if yin_t is None else yin_t.detach().numpy())
            val, gth, gy, gx = gp.lml_grad_data(theta, want_x=iwgp)
            if not np.isfinite(val):
                return -np.inf, None, {}
            s = torch.zeros((), dtype=torch.float64)
            if cwgp:
                s = s + (torch.from_numpy(gy) * yin_t).sum() + torch.log(yder_t).sum()
                val = val + float(torch.log(yder_t).sum().detach())
            if iwgp:
                s = s + (torch.from_numpy(gx) * xin_t).sum()
            s.backward()
            return val, gth, {k: v.grad.numpy().copy() for k, v in leaves.items()}

        return likelihood

    def __fit(self, x, y, method, iwgp, cwgp, jitter=1e-6, truncate=False, restarts=1, **kwargs):
        n_i, n_pos, n_free = self._warp_sizes(iwgp, cwgp)
        model = HyperModel(self.nx, self.kerns, noise=self.noise, truncate=truncate, jitter=jitter, n_iwgp=n_i,
                           n_cwgp_pos=n_pos, n_cwgp=n_free)
        xin, yin = self._converted(x, y)
        self.__release()
        gp = MiGP(xin, yin, self.kernel, device=self.device)
        lik = self._warp_likelihood(gp, x, y, xin, iwgp, cwgp) if (iwgp or cwgp) else None
        # pm.find_MAP maximises without the transform Jacobian, pm.sample with it (priors.py header)
        fun_map = lambda q: model.logp_dlogp(q, gp.lml_grad, jacobian=False, likelihood=lik)  # noqa: E731
        data = None
        if method == "map":
            best, mp = -np.inf, None
            nrest = max(int(restarts), 1)
            last_error = None
            for _ in range(nrest):
                # the reference builds a random start and never passes it (gpmcmc.py:330-332): every
                # restart begins at the model's initial point, so they coincide; kept as is
                try:
                    q0 = model.initial_point()
                    if kwargs.get("start") is not None:  # pm.find_MAP(start=...), e.g. BO's warm refits (gpmcmc.py:899)
                        q0 = model.q_from_point(kwargs["start"])
                    q, info = find_MAP(fun_map, q0, progressbar=kwargs.get("progressbar", False),
                                       maxeval=kwargs.get("maxeval", 5000))
                except RuntimeError:
                    raise  # device / library errors (MiGP._check) are never a "failed restart"
                except Exception as e:
                    if nrest == 1:
                        raise  # single fit: the exception propagates as in gpmcmc.py:343-345
                    print("Restart failed")
                    last_error = e
                    continue
                if info["logp"] > best:
                    best, mp, data = info["logp"], model.point_dict(q), info
            if mp is None:
                raise RuntimeError("find_MAP failed in every restart") from last_error
            if self.verbose:
                print(f"MAP: {data['nfev']} evaluations, logp = {data['logp']:,.5g}")
        elif method == "none":
            mp = self.hypers
        elif method in ("mcmc_mean", "mcmc_map"):
            data = self.__sample(model, gp, x, y, xin, yin, iwgp, cwgp, **kwargs)
            if method == "mcmc_mean":
                mp = self.mean_extract(data)
            else:
                mp = self.map_extract(data)
                try:
                    q, _ = find_MAP(fun_map, model.q_from_point(mp))
                    mp = model.point_dict(q)
                except Exception:
                    pass
        else:
            raise Exception("method must be one of map, mcmc_map, or mcmc_mean")
        # freeze the warps at the fitted parameters and leave the converted data on the device (gpmcmc.py:362-399)
        if iwgp or cwgp:
            if method != "none":
                if iwgp:
                    self.iwgp_set(np.atleast_1d(mp["iwgp"]))
                if cwgp:
                    self.cwgp_set(np.array(self._cwgp_params(np.atleast_1d(mp.get("cwgp_pos", [])),
                                                              np.atleast_1d(mp.get("cwgp", [])))))
            xin, yin = self._converted(x, y)
            gp.update_data(X=xin, y=yin)
        return model, gp, mp, data

    def __sample(self, model, gp, x, y, xin, yin, iwgp=False, cwgp=False, draws=1000, tune=1000, chains=None,
                 cores=None, target_accept=0.8, random_seed=None, max_treedepth=10, progressbar=False, devices=None,
                 chains_per_device=None, batched=None, **_):
        """pm.sample(**kwargs) of gpmcmc.py:351: independent NUTS chains, one device handle per concurrent chain
        (one chain per GPU when several are visible: SURVEY.md section 8e).  Chains that share a GPU run on up to
        ``chains_per_device`` handles at once (default: up to 3 while their buffers fit): below N ~ 10^4 one evaluation
        is bound by the serial panel chain and leaves most of the chip idle -- three concurrent handles on one MI355X
        deliver 2.7x the evaluations/s at N=1024, 2.2x at N=4096, 1.25x at N=8192 (tools/archive/dev_concurrent.py), and every
        chain's draws are the same as when it runs alone (evaluations are deterministic per handle).
        Round 4: without warp parameters (and unless ``chains_per_device`` / ``batched=False`` ask for lanes) the chains of a
        device share ONE handle and meet in one batched evaluation per leapfrog step (blockIdx.z = chain): N=4096 LML
        1440 evaluations/s with eight chains against 1000 with three handles, N=2048 LML + gradient 2800 against 1860."""
        import torch

        chains = max(2, min(4, os.cpu_count() or 2)) if chains is None else int(chains)
        ndev = torch.cuda.device_count()
        devices = list(devices) if devices is not None else [(self.device + i) % max(ndev, 1) for i in range(chains)]
        seeds = np.random.SeedSequence(random_seed).spawn(chains)
        results = [None] * chains
        errors = []

        by_dev = {}
        for c in range(chains):
            by_dev.setdefault(devices[c % len(devices)], []).append(c)

        def lanes_for(dev, nchain):
            """How many handles work side by side on this device."""
            if chains_per_device is not None:
                return max(1, min(int(chains_per_device), nchain))
            npad = (len(yin) + 127) // 128 * 128
            # one gradient-capable handle: K, U = L^-T and K^-1; the leaf inverses (128 x 128 per tile column); X, the
            # data-side gradient partials (up to 8 column splits of n x d) and the small vectors
            need = (3 * (npad + 128) * (npad + 16) + (npad // 128) * 128 * 128 + 10 * npad * xin.shape[1] + 8 * npad) * 8
            free, _total = torch.cuda.mem_get_info(dev)
            base = 1 if dev == self.device else 0  # the first lane's handle exists already -- on self.device only
            return max(1, min(3, nchain, base + int(0.8 * free // need)))

        # a lane = one host thread + one handle; the chains of a device are dealt round-robin to its lanes
        def run_lane(dev, cs, h_existing, shared, nlanes=1):
            try:
                h = h_existing or MiGP(xin, yin, self.kernel, device=dev)
                # Lanes that SHARE a GPU give up the look-ahead stream up to 64 tile columns (two streams start at 4 since
                # round 6): the other lanes fill the idle CUs anyway and a hand-off between two streams costs ~5-10 us
                # (three handles, LML + gradient: N = 6144 176 -> 202 evaluations/s, N = 8192 94 -> 97; from 72 tile columns on
                # there is nothing in it).  The super-panel width is pinned to the one the two-stream driver would pick, so
                # the arithmetic -- and every draw -- is bit-identical to the default schedule.
                ntc = (len(yin) + 127) // 128
                pinned = shared and 4 <= ntc <= 64
                before26 = None
                if shared and not pinned and nlanes > MAX_POLLING_HANDLES:
                    # Two-stream lanes enqueue in-kernel polls ahead of the writes they wait for (include/mi_gp.h, option 26):
                    # tested with six handles evaluating at once per device.  Beyond that the lanes use event edges from the
                    # start (same bits) instead of finding out through a poll limit.
                    before26 = h.get_option(26, 2)
                    h.set_option(26, 0)
                if pinned:
                    # the caller's own handle gets its previous settings back afterwards (library defaults: look-ahead by
                    # size = 1, super-panel width by size = 0)
                    before = (h.get_option(2, 0), h.get_option(0, 1))
                    if 20 <= ntc <= 60:  # (api_gp.hip NARROW_PANELS_MAX_TILES: above it both schedules use 8-tile super-panels;
                        h.set_option(2, 4)  # up to 31 tile columns the whole problem runs in column mode: no panels at all)
                    h.set_option(0, 0)
                try:
                    lik = self._warp_likelihood(h, x, y, xin, iwgp, cwgp) if (iwgp or cwgp) else None
                    f = lambda q: model.logp_dlogp(q, h.lml_grad, likelihood=lik)  # noqa: E731
                    for c in cs:
                        results[c] = sample_chain(f, model.initial_point(), draws=draws, tune=tune,
                                                  target_accept=target_accept, max_treedepth=max_treedepth, seed=seeds[c],
                                                  progressbar=progressbar and c == 0)
                finally:
                    if h_existing is None:
                        h.close()
                    else:
                        if pinned:
                            h.set_option(2, before[0])
                            h.set_option(0, before[1])
                        if before26 is not None:
                            h.set_option(26, before26)
            except Exception as e:  # noqa: BLE001 - reported by the caller's thread
                errors.append(e)

        # Batched mode (round 4): the chains of a device share ONE handle and meet once per leapfrog step in ONE batched
        # device call (batching.BatchedEvaluator -> mi_gp_lml_grad_batch, blockIdx.z = chain) -- without warp parameters
        # only: warped chains evaluate different DATA, not just different theta.  Same draws as the unbatched schedule.
        def run_batched(dev, cs, h_existing):
            try:
                from .batching import BatchedEvaluator

                h = h_existing or MiGP(xin, yin, self.kernel, device=dev)
                ev = BatchedEvaluator(h.lml_grad_batch, len(cs))

                def one(c):
                    try:
                        f = lambda q: model.logp_dlogp(q, ev.evaluate)  # noqa: E731
                        results[c] = sample_chain(f, model.initial_point(), draws=draws, tune=tune, target_accept=target_accept,
                                                  max_treedepth=max_treedepth, seed=seeds[c], progressbar=progressbar and c == 0)
                    except Exception as e:  # noqa: BLE001
                        errors.append(e)
                    finally:
                        ev.leave()

                ts = [threading.Thread(target=one, args=(c,)) for c in cs]
                for t in ts:
                    t.start()
                for t in ts:
                    t.join()
                if h_existing is None:
                    h.close()
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        def batch_fits(dev, nchain):
            """One handle's batch buffers: nchain copies of K, U and K^-1 (the lanes path sizes itself the same way)."""
            npad = (len(yin) + 127) // 128 * 128
            need = nchain * (3 * (npad + 128) * (npad + 16) + (npad // 128) * 128 * 128) * 8
            free, _total = torch.cuda.mem_get_info(dev)
            return need <= 0.8 * free

        threads = []
        use_batch = (batched if batched is not None else True) and not (iwgp or cwgp)
        for dev, cs in by_dev.items():
            if use_batch and len(cs) > 1 and chains_per_device is None and batch_fits(dev, len(cs)):
                threads.append(threading.Thread(target=run_batched, args=(dev, cs, gp if dev == self.device else None)))
                continue
            k = lanes_for(dev, len(cs))
            for lane in range(k):
                mine = cs[lane::k]
                if mine:
                    threads.append(threading.Thread(target=run_lane, args=(dev, mine, gp if (dev == self.device and lane == 0) else None, k > 1, k)))
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise RuntimeError(f"a NUTS chain failed: {errors[0]}") from errors[0]
        if any(r is None for r in results):
            raise RuntimeError("a NUTS chain failed")
        posterior = {}
        qs = np.stack([r["q"] for r in results])  # [chain, draw, nq]
        for c in range(chains):
            for k in range(qs.shape[1]):
                pt = model.point_dict(qs[c, k])
                for name, val in pt.items():
                    if name not in posterior:
                        posterior[name] = np.empty((chains, qs.shape[1]) + np.shape(val))
                    posterior[name][c, k] = val
        stats = {"lp": np.stack([r["lp"] for r in results]),
                 "step_size": np.array([r["step_size"] for r in results]),
                 "n_leapfrog": np.array([r["n_leapfrog"] for r in results]),
                 "diverging": np.array([r["diverging"] for r in results])}
        return Trace(posterior, stats)

    def mean_extract(self, data):
        """Posterior means over chains and draws (gpmcmc.py:404-412)."""
        return {k: np.array(v.mean(axis=(0, 1))) for k, v in data.posterior.items()}

    def map_extract(self, data):
        """Draw with the largest log-posterior (gpmcmc.py:415-430)."""
        lp = data.sample_stats["lp"].reshape(-1)
        k = int(np.argmax(lp))
        if self.verbose:
            print(f"Max log posterior: {lp[k]}")
        mp = {}
        for name, v in data.posterior.items():
            flat = v.reshape((-1,) + v.shape[2:])
            mp[name] = np.array(flat[k])
        return mp

    # ------------------------------------------------------------------ predict
    def _theta_from_hypers(self, hyps, jitter):
        model = self.m
        vals = {"l": np.atleast_1d(hyps["l"]), "kv": np.atleast_1d(hyps["kv"])}
        if self.noise:
            vals["gv"] = np.atleast_1d(hyps["gv"])
        if "alpha" in hyps:
            vals["alpha"] = np.atleast_1d(hyps["alpha"])
        th = model.theta(vals)
        th[-1] = jitter
        return th

    def predict(self, x, return_var=False, convert=True, revert=True, normvar=False, jitter=1e-6, EI=False,
                EIopt=None, deg=8):
        """gpmcmc.py:522-542."""
        if self._ensure_gp() is None or self.hypers is None:
            raise Exception("Error: fit the GP before predicting")
        if convert:
            xarg = np.zeros_like(x)
            for i in range(self.nx):
                xarg[:, i] = self.xconrevs[i].con(x[:, i])
        else:
            xarg = copy.deepcopy(x)
            x = copy.deepcopy(x)
            for i in range(self.nx):
                x[:, i] = self.xconrevs[i].rev(x[:, i])
        if self.verbose:
            print("Predicting...")
        t0 = stopwatch()
        mu, var = self.gp.predict(self._theta_from_hypers(self.hypers, jitter), xarg, pred_noise=True)
        if self.verbose:
            print(f"Time taken: {stopwatch() - t0:0.2f} s")
        y, yv = mu.reshape((-1, 1)), var.reshape((-1, 1))
        if revert:
            y, yv = self.__gh_stats(x, y, yv, normvar, deg, EI=EI, EIopt=EIopt)
        return (y, yv) if return_var else y

    def __gh_stats(self, x, y, yv, normvar=True, deg=8, EI=False, EIopt=None):
        """Gauss-Hermite mean / variance (or EI) of the reverted variable (gpmcmc.py:545-569),
        vectorised over the prediction points instead of the reference's per-point Python loop."""
        xi, wi = np.polynomial.hermite.hermgauss(deg)
        yi = np.sqrt(2.0 * yv) * xi[None, :] + y  # [M, deg]
        means = np.array([self.mean(x[i, :])[0] for i in range(len(x))]) if self.mean != self.zero_mean else 0.0
        yir = self.yconrevs[0].rev(yi) + np.reshape(means, (-1, 1) if np.ndim(means) else ())
        if EI:
            ydiff = (yir - self.yopt) if EIopt == "max" else (self.yopt - yir)
            first = np.where(ydiff > 0.0, ydiff, 0.0)
        else:
            first = yir
        ymean = (first @ wi) / np.sqrt(np.pi)
        ym2 = ((yir ** 2) @ wi) / np.sqrt(np.pi)
        yout = ymean.reshape((-1, 1))
        yvout = (ym2 - ymean ** 2).reshape((-1, 1))
        if normvar:
            yvout = yvout / np.power(yout, 2)
        return yout, yvout

    def __del__(self):
        try:
            self.__release()
        except Exception:
            pass
