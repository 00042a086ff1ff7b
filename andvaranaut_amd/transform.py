"""Input / output conversion classes (the subset of the reference's transform.py that the GP hot
path consumes: SURVEY.md section 8f-3).  Each class exposes ``con`` (to the GP's working space),
``rev`` (back) and, for output warps, ``der`` (d con / d y), with the same argument meaning as the
reference classes of the same name (transform.py:139-143 ``normal``, :193-207 ``logarithm``,
:208-229 ``affine``, :230-239 ``meanstd``, :265-280 ``maxmin``, :281-288 ``uniform``).
Only NumPy forms are provided: the PyTensor twins (conmc/revmc/dermc) exist in the reference so
that PyMC can differentiate through warps whose parameters are sampled (iwgp / cwgp), which this
backend does not implement yet."""
import numpy as np


class _none_conrev:
    """Identity conversion (gpmcmc.py:23-27)."""

    def con(self, x):
        return x

    def rev(self, x):
        return x

    def der(self, x):
        return np.ones_like(x)


class affine:
    """con(y) = a + b*y  (transform.py:208-229)."""

    def __init__(self, a, b):
        if not b > 0.0:
            raise Exception("Parameter b must be positive")
        self.a = a
        self.b = b

    def con(self, y):
        return self.a + self.b * y

    def rev(self, y):
        return (y - self.a) / self.b

    def der(self, y):
        return self.b * np.ones_like(y)


class meanstd(affine):
    """Standardise by the sample mean and (population) standard deviation (transform.py:230-239)."""

    def __init__(self, y):
        mean, std = np.mean(y), np.std(y)
        self.a = -mean / std
        self.b = 1 / std


class stddev(affine):
    def __init__(self, y):
        self.a = 0
        self.b = 1 / np.std(y)


class maxmin(affine):
    """Map the sample range to [safety, 1-safety] (or [-1+.., 1-..] if centred) (transform.py:265-280)."""

    def __init__(self, x, centred=False, safety=0.01):
        xmin, xmax = np.min(x), np.max(x)
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


class normal:
    """Standardise by a scipy prior's mean and std (transform.py:36-38, 139-143)."""

    def __init__(self, dist):
        self.mean = dist.mean()
        self.std = dist.std()

    def con(self, x):
        return (x - self.mean) / self.std

    def rev(self, x):
        return x * self.std + self.mean

    def der(self, x):
        return np.ones_like(x) / self.std


class logarithm:
    """Log output warp (transform.py:193-207)."""

    def con(self, y):
        return np.log(y)

    def rev(self, y):
        return np.exp(y)

    def der(self, y):
        return 1 / y
