"""Input / output conversion classes and the composite warp ``wgp``, with the
argument meaning of the reference classes of the same name (transform.py:139-143 ``normal``, :193-207
``logarithm``, :208-229 ``affine``, :230-264 ``meanstd`` / ``minshift`` / ``stddev`` / ``stdshift``,
:265-288 ``maxmin`` / ``uniform``, :289-318 ``arcsinh``, :319-345 ``boxcox`` / ``boxcoxf``, :346-366
``sinharcsinh``, :367-392 ``sal``, :394-417 ``kumaraswamy``, :419-428 ``preserve_zero``, :430-574 ``wgp``).

Each class exposes ``con`` (to the GP's working space), ``rev`` (back) and ``der`` (d con / d y).  The
reference keeps a second, PyTensor copy of every formula (``conmc`` / ``revmc`` / ``dermc``) so that PyMC can
differentiate through warps whose parameters are sampled (iwgp / cwgp).  Here one set of formulas serves
both: the same methods accept NumPy arrays or torch tensors (float64, host), and the ``*mc`` names are
aliases; the posterior code differentiates the torch form with ``torch.autograd`` and hands the device the
warped data (andvaranaut_amd/gpmcmc.py)."""
import copy

import numpy as np
import scipy.stats as st

try:  # torch is the autodiff engine for sampled warp parameters; the NumPy forms work without it
    import torch
except Exception:  # pragma: no cover
    torch = None


def _is_t(*xs):
    return torch is not None and any(isinstance(x, torch.Tensor) for x in xs)


def _f(name_np, name_t=None):
    name_t = name_t or name_np

    def fn(x, *a):
        if _is_t(x, *a):
            x = x if isinstance(x, torch.Tensor) else torch.as_tensor(x, dtype=torch.float64)
            return getattr(torch, name_t)(x, *a)
        return getattr(np, name_np)(x, *a)

    return fn


_log, _exp, _sqrt, _abs, _sign = _f("log"), _f("exp"), _f("sqrt"), _f("abs"), _f("sign")
_sinh, _cosh, _arcsinh = _f("sinh"), _f("cosh"), _f("arcsinh", "asinh")
_mean, _min, _max = _f("mean"), _f("min"), _f("max")


def _std(x):
    return torch.std(x, unbiased=False) if _is_t(x) else np.std(x)  # population form, as np.std / pt.std


def _pow(x, p):
    if _is_t(x, p):
        x = x if isinstance(x, torch.Tensor) else torch.as_tensor(x, dtype=torch.float64)
        return torch.pow(x, p)
    return np.power(x, p)


def _ones_like(x):
    return torch.ones_like(x) if _is_t(x) else np.ones_like(x)


def _positive(v, name):
    """The reference raises inside a bare try/except that swallows it again (transform.py:214-218,295-302):
    positivity is a convention enforced by the priors of GPMCMC.__fit, not by these classes."""
    return None


class _mc_aliases:
    """conmc / revmc / dermc: the reference's PyTensor twins; here the same code paths (torch-capable)."""

    def conmc(self, y):
        return self.con(y)

    def revmc(self, y):
        return self.rev(y)

    def dermc(self, y):
        return self.der(y)


class _none_conrev(_mc_aliases):
    """Identity conversion (gpmcmc.py:23-27)."""

    def con(self, x):
        return x

    def rev(self, x):
        return x

    def der(self, x):
        return _ones_like(x)


class affine(_mc_aliases):
    """con(y) = a + b*y  (transform.py:208-229)."""

    def __init__(self, a, b):
        self.a = a
        self.b = b
        _positive(b, "b")
        self.default_priors = [st.norm(), st.norm()]

    def con(self, y):
        return self.a + self.b * y

    def rev(self, y):
        return (y - self.a) / self.b

    def der(self, y):
        return self.b * _ones_like(y)


class meanstd(affine):
    """Standardise by the sample mean and (population) standard deviation (transform.py:230-239)."""

    def __init__(self, y, mode="numpy"):
        mean, std = _mean(y), _std(y)
        self.a = -mean / std
        self.b = 1 / std


class minshift(affine):
    """Shift by ``safety`` times the sample minimum (transform.py:240-247)."""

    def __init__(self, y, mode="numpy", safety=1000):
        self.a = -_min(y) * safety
        self.b = 1.0


class stddev(affine):
    def __init__(self, y, mode="numpy"):
        self.a = 0
        self.b = 1 / _std(y)


class stdshift(affine):
    """con(y) = a + y / std(y) with a tunable shift (transform.py:256-264)."""

    def __init__(self, a, y, mode="numpy"):
        self.a = a
        self.b = 1 / _std(y)
        self.default_priors = [st.norm()]


class maxmin(affine):
    """Map the sample range to [safety, 1-safety] (or [-1+.., 1-..] if centred) (transform.py:265-280)."""

    def __init__(self, x, centred=False, safety=0.01, mode="numpy"):
        xmin, xmax = _min(x), _max(x)
        xminus = (xmax - xmin) / (1 - 2 * safety)
        xplus = xmax + xmin
        if centred:
            self.a = -xplus / xminus
            self.b = 2 / xminus
        else:
            self.a = -xmin / xminus + safety
            self.b = 1 / xminus


class uniform(affine):
    """Map a scipy uniform prior's support to [safety, 1-safety] (transform.py:281-288)."""

    def __init__(self, dist, safety=1e-10):
        intv = dist.interval(1.0)
        xminus = (intv[1] - intv[0]) / (1 - 2 * safety)
        self.a = -intv[0] / xminus + safety
        self.b = 1 / xminus


class preserve_zero(affine):
    """Scale by the standard deviation such that ``yzero`` maps to zero (transform.py:419-428)."""

    def __init__(self, y, yzero, mode="numpy"):
        ystd = _std(y)
        self.a = -yzero / ystd
        self.b = 1 / ystd


class normal(_mc_aliases):
    """Standardise by a scipy prior's mean and std (transform.py:36-38, 139-143)."""

    def __init__(self, dist):
        self.mean = dist.mean()
        self.std = dist.std()

    def con(self, x):
        return (x - self.mean) / self.std

    def rev(self, x):
        return x * self.std + self.mean

    def der(self, x):
        return _ones_like(x) / self.std


class logarithm(_mc_aliases):
    """Log output warp (transform.py:193-207)."""

    def con(self, y):
        return _log(y)

    def rev(self, y):
        return _exp(y)

    def der(self, y):
        return 1 / y


class arcsinh(_mc_aliases):
    """con(y) = a + b asinh((y - c) / d)  (transform.py:289-318)."""

    def __init__(self, a, b, c, d):
        self.a, self.b, self.c, self.d = a, b, c, d
        self.default_priors = [st.norm(), st.norm(), st.norm(), st.norm()]
        _positive(b, "b")
        _positive(d, "d")

    def con(self, y):
        return self.a + self.b * _arcsinh((y - self.c) / self.d)

    def rev(self, y):
        return self.c + self.d * _sinh((y - self.a) / self.b)

    def der(self, y):
        return self.b / _sqrt(_pow(self.d, 2) + _pow(y - self.c, 2))


class boxcox(_mc_aliases):
    """Sign-preserving Box-Cox with lambda shifted so that 0 is (almost) the identity (transform.py:319-339)."""

    def __init__(self, lamb):
        self.lamb = lamb
        self.default_priors = [st.norm(loc=0)]

    def con(self, y):
        lambp = self.lamb + 1
        return (_sign(y) * _pow(_abs(y), lambp) - 1) / lambp

    def rev(self, y):
        lambp = self.lamb + 1
        term = y * lambp + 1
        return _sign(term) * _pow(_abs(term), 1 / lambp)

    def der(self, y):
        return _pow(_abs(y), self.lamb)


class boxcoxf(boxcox):
    """Box-Cox fitted by scikit-learn (transform.py:340-345); NumPy data only."""

    def __init__(self, y):
        from sklearn.preprocessing import PowerTransformer

        powt = PowerTransformer(method="box-cox", standardize=False)
        powt.fit(np.asarray(y).reshape(-1, 1))
        self.lamb = powt.lambdas_[0]


class sinharcsinh(_mc_aliases):
    """con(y) = sinh(b asinh(y) - a)  (transform.py:346-366)."""

    def __init__(self, a, b):
        self.a, self.b = a, b
        _positive(b, "b")
        self.default_priors = [st.norm(), st.norm()]

    def con(self, y):
        return _sinh(self.b * _arcsinh(y) - self.a)

    def rev(self, y):
        return _sinh((_arcsinh(y) + self.a) / self.b)

    def der(self, y):
        return self.b * _cosh(self.b * _arcsinh(y) - self.a) / _sqrt(1 + _pow(y, 2))


class sal(_mc_aliases):
    """Sinh-arcsinh followed by an affine map: con(y) = c + d sinh(b asinh(y) - a)  (transform.py:367-392)."""

    def __init__(self, a, b, c, d):
        self.a, self.b, self.c, self.d = a, b, c, d
        _positive(b, "b")
        _positive(d, "d")
        self.default_priors = [st.norm(), st.norm(), st.norm(), st.norm()]

    def con(self, y):
        return self.c + self.d * _sinh(self.b * _arcsinh(y) - self.a)

    def rev(self, y):
        return _sinh((_arcsinh((y - self.c) / self.d) + self.a) / self.b)

    def der(self, y):
        return self.b * self.d * _cosh(self.b * _arcsinh(y) - self.a) / _sqrt(1 + _pow(y, 2))


class kumaraswamy(_mc_aliases):
    """Input warp with the Kumaraswamy CDF on [0, 1]  (transform.py:394-417)."""

    def __init__(self, a, b):
        self.a, self.b = a, b
        _positive(a, "a")
        _positive(b, "b")
        self.default_priors = [st.norm(), st.norm()]

    def con(self, x):
        return 1 - _pow(1 - _pow(x, self.a), self.b)

    def rev(self, x):
        return _pow(1 - _pow(1 - x, 1 / self.b), 1 / self.a)

    def der(self, x):
        return self.a * self.b * _pow(x, self.a - 1) * _pow(1 - _pow(x, self.a), self.b - 1)


class wgp(_mc_aliases):
    """Composite warp: the listed warps applied left to right, their tunable parameters taken in order from
    ``params`` (transform.py:430-574).  ``pos[i]`` marks parameters that must be positive (they get the
    log-normal prior in GPMCMC.__fit, gpmcmc.py:251-272), ``np`` is the number of tunable parameters and
    ``pid[k]`` the parameter count consumed up to and including warp k.  Data-dependent members (meanstd,
    maxmin, stddev, stdshift, minshift, pzero, boxcoxf) are fitted to ``y`` as warped by the members before
    them, so they move with the tunable parameters exactly as in the reference's ``mode='pytensor'`` path:
    pass torch tensors for ``params`` / ``y`` to differentiate through the whole chain."""

    allowed = ["affine", "logarithm", "arcsinh", "boxcox", "sinharcsinh", "sal", "meanstd", "boxcoxf", "uniform",
               "maxmin", "kumaraswamy", "pzero", "stddev", "stdshift", "minshift"]

    def __init__(self, warpings, params, y=None, xdist=None, mode="numpy"):
        self.warping_names = warpings
        self.warpings = []
        self.params = params
        self.pid = np.zeros(len(warpings), dtype=np.int32)
        self.pos = np.zeros(len(params), dtype=np.bool_)
        self.default_priors = []
        pc = 0
        yzero = 0.0
        yc = None
        if y is not None:
            yc = y if _is_t(y) else copy.deepcopy(y)

        def need_y(name):
            if y is None:
                raise Exception(f"Must supply y array to use {name}")

        for k, name in enumerate(warpings):
            if name not in self.allowed:
                raise Exception(f"Only {self.allowed} classes allowed")
            if name == "affine":
                w = affine(params[pc], params[pc + 1])
                self.pos[pc : pc + 2] = [False, True]
                self.default_priors.extend(w.default_priors)
                pc += 2
            elif name == "logarithm":
                w = logarithm()
            elif name == "arcsinh":
                w = arcsinh(params[pc], params[pc + 1], params[pc + 2], params[pc + 3])
                self.pos[pc : pc + 4] = [False, True, False, True]
                self.default_priors.extend(w.default_priors)
                pc += 4
            elif name == "boxcox":
                w = boxcox(lamb=params[pc])
                self.pos[pc : pc + 1] = [False]
                self.default_priors.extend(w.default_priors)
                pc += 1
            elif name == "sinharcsinh":
                w = sinharcsinh(params[pc], params[pc + 1])
                self.pos[pc : pc + 2] = [False, True]
                self.default_priors.extend(w.default_priors)
                pc += 2
            elif name == "sal":
                w = sal(params[pc], params[pc + 1], params[pc + 2], params[pc + 3])
                self.pos[pc : pc + 4] = [False, True, False, True]
                self.default_priors.extend(w.default_priors)
                pc += 4
            elif name == "kumaraswamy":
                w = kumaraswamy(params[pc], params[pc + 1])
                self.pos[pc : pc + 2] = [True, True]
                self.default_priors.extend(w.default_priors)
                pc += 2
            elif name == "stdshift":
                need_y("stddev")
                w = stdshift(params[pc], yc, mode=mode)
                self.pos[pc] = False
                self.default_priors.extend(w.default_priors)
                pc += 1
            elif name == "meanstd":
                need_y("meanstd")
                w = meanstd(yc, mode=mode)
            elif name == "minshift":
                need_y("minshift")
                w = minshift(yc, mode=mode)
            elif name == "stddev":
                need_y("stddev")
                w = stddev(yc, mode=mode)
            elif name == "boxcoxf":
                need_y("fitted box cox")
                w = boxcoxf(y=yc)
            elif name == "uniform":
                if xdist is None:
                    raise Exception("Must supply x distribution to use uniform")
                w = uniform(xdist)
            elif name == "maxmin":
                need_y("maxmin")
                w = maxmin(yc, mode=mode)
            else:  # pzero
                need_y("pzero")
                w = preserve_zero(yc, yzero, mode=mode)
            self.warpings.append(w)
            self.pid[k] = pc
            if y is not None:
                yc = w.con(yc)
                with np.errstate(all="ignore"):  # log(0) etc.: only 'pzero' ever reads it
                    yzero = w.con(yzero)
        self.np = pc

    def con(self, y):
        res = y
        for w in self.warpings:
            res = w.con(res)
        return res

    def rev(self, y):
        res = y
        for w in reversed(self.warpings):
            res = w.rev(res)
        return res

    def der(self, y):
        res = _ones_like(y)
        x = y if _is_t(y) else copy.deepcopy(y)
        for w in self.warpings:
            res = res * w.der(x)
            x = w.con(x)
        return res
