"""CPU oracle for the GP log-marginal-likelihood hot path  --  TEST INFRASTRUCTURE ONLY.

This module is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The shipped package (``andvaranaut_amd``) must never import anything from ``oracle/``.

It is a NumPy/SciPy restatement of the arithmetic the reference reaches through
PyMC/PyTensor/SciPy (none of which are vendored in /root/reference or installed
here).  Each function cites the reference call site it follows; the third-party
algorithm (pymc<=5.9.2, pytensor 2.17.x; pinned only as ``pymc <= 5.9.2`` at
/root/reference/pyproject.toml:17) is restated from its published sources and is
marked [3P].

PARITY PINNING: the reference ships no tests, golden vectors or fixtures for this
path, and it cannot be imported here (missing pymc, pytensor,
arviz, dask, netCDF4, seaborn: an ordinary ModuleNotFoundError, not a denial).  The
oracle is therefore pinned by (tests/test_oracle_golden.py):
  * closed-form N=1 / N=2 values,
  * 50-digit mpmath evaluations that share no code with this file
    (oracle/gen_golden.py -> tests/golden/mpmath_*.json),
  * scikit-learn's GaussianProcessRegressor as an independent implementation,
  * complex-step / finite-difference gradient checks,
  * the two numeric pins the tutorial notebook records (tutorial.ipynb:366).
That is "parity unpinned" in the strict sense of the task contract (no reference-run
outputs exist to compare with); DESIGN.md says so too.
"""
import numpy as np
import scipy.linalg as sla

KERNEL_IDS = {"RBF": 0, "Matern52": 1, "Matern32": 2, "Exponential": 3, "RatQuad": 4}
SQRT5 = np.sqrt(5.0)
SQRT3 = np.sqrt(3.0)


# --------------------------------------------------------------------------- K1
def square_dist(X, Xs, ls):
    """[3P] pymc/gp/cov.py Stationary.square_dist (expansion form + clip), reached
    from gpmcmc.py:284-299.  ``X * (1/ls)``, not ``X / ls``, exactly as PyMC does."""
    Xa = X * (1.0 / ls)
    X2 = np.sum(np.square(Xa), 1)
    if Xs is None:
        sqd = -2.0 * np.dot(Xa, Xa.T) + (X2.reshape(-1, 1) + X2.reshape(1, -1))
    else:
        Xb = Xs * (1.0 / ls)
        Xs2 = np.sum(np.square(Xb), 1)
        sqd = -2.0 * np.dot(Xa, Xb.T) + (X2.reshape(-1, 1) + Xs2.reshape(1, -1))
    return np.clip(sqd, 0.0, np.inf)


def euclidean_dist(r2):
    """[3P] Stationary.euclidean_dist: sqrt(r2 + 1e-12)."""
    return np.sqrt(r2 + 1e-12)


def base_kernel(name, r2, alpha=None):
    """[3P] ExpQuad/Matern52/Matern32/Exponential/RatQuad .full (gpmcmc.py:283-299)."""
    if name == "RBF":
        return np.exp(-0.5 * r2)
    if name == "RatQuad":
        return np.power(1.0 + 0.5 * r2 * (1.0 / alpha), -1.0 * alpha)
    r = euclidean_dist(r2)
    if name == "Matern52":
        return (1.0 + SQRT5 * r + 5.0 / 3.0 * np.square(r)) * np.exp(-1.0 * SQRT5 * r)
    if name == "Matern32":
        return (1.0 + SQRT3 * r) * np.exp(-SQRT3 * r)
    if name == "Exponential":
        return np.exp(-0.5 * r)
    raise ValueError(name)


def base_kernel_dr2(name, r2, alpha=None):
    """d k / d r2 of the base kernels (analytic; used by the K7 gradient restatement)."""
    if name == "RBF":
        return -0.5 * np.exp(-0.5 * r2)
    if name == "RatQuad":
        return -0.5 * np.power(1.0 + 0.5 * r2 / alpha, -alpha - 1.0)
    r = euclidean_dist(r2)
    if name == "Matern52":
        return -(5.0 / 6.0) * (1.0 + SQRT5 * r) * np.exp(-SQRT5 * r)
    if name == "Matern32":
        return -1.5 * np.exp(-SQRT3 * r)
    if name == "Exponential":
        return -0.25 * np.exp(-0.5 * r) / r
    raise ValueError(name)


def split_theta(theta, d, nkern):
    """theta layout of the C-ABI (include/mi_gp.h): [l(nkern*d), kv(nkern), alpha(nkern), gv, jitter]."""
    theta = np.asarray(theta, dtype=np.float64)
    ls = theta[: nkern * d].reshape(nkern, d)
    kv = theta[nkern * d : nkern * d + nkern]
    alpha = theta[nkern * d + nkern : nkern * d + 2 * nkern]
    gv = theta[nkern * d + 2 * nkern]
    jitter = theta[nkern * d + 2 * nkern + 1]
    return ls, kv, alpha, gv, jitter


def pack_theta(ls, kv, gv, jitter, alpha=None):
    ls = np.atleast_2d(np.asarray(ls, dtype=np.float64))
    kv = np.atleast_1d(np.asarray(kv, dtype=np.float64))
    if alpha is None:
        alpha = np.ones_like(kv)
    alpha = np.atleast_1d(np.asarray(alpha, dtype=np.float64))
    return np.concatenate([ls.ravel(), kv, alpha, [gv, jitter]])


def component_matrices(X, Xs, kerns, ls, kv, alpha):
    """kv[i] * Cov_i(ls_i) for each component (gpmcmc.py:282-299)."""
    comps, r2s = [], []
    for i, name in enumerate(kerns):
        r2 = square_dist(X, Xs, ls[i])
        r2s.append(r2)
        comps.append(kv[i] * base_kernel(name, r2, alpha[i]))
    return comps, r2s


def combine(comps, ops):
    """Left-to-right '+' / '*' on full matrices (gpmcmc.py:302-307)."""
    K = comps[0]
    for i in range(1, len(comps)):
        if ops[i - 1] == "+":
            K = K + comps[i]
        elif ops[i - 1] == "*":
            K = K * comps[i]
        else:
            raise ValueError(ops[i - 1])
    return K


def kernel_matrix(X, Xs, kerns, ops, theta):
    d = X.shape[1]
    ls, kv, alpha, _, _ = split_theta(theta, d, len(kerns))
    comps, _ = component_matrices(X, Xs, kerns, ls, kv, alpha)
    return combine(comps, ops)


def kernel_diag(kerns, ops, theta, d):
    """[3P] Stationary.diag == 1, so the composite diag is the +/* fold of kv."""
    _, kv, _, _, _ = split_theta(theta, d, len(kerns))
    v = kv[0]
    for i in range(1, len(kerns)):
        v = v + kv[i] if ops[i - 1] == "+" else v * kv[i]
    return v


# ----------------------------------------------------------------------- K2..K6
def noisy_cov(X, kerns, ops, theta, form="marginal", extra_diag=None):
    """K2.  form='marginal': [3P] Marginal._build_marginal_likelihood:
    (Kxx + WhiteNoise(sigma)) + jitter*I with sigma=sqrt(gv) (gpmcmc.py:321-323), so the
    diagonal gets sqrt(gv)**2.  form='explicit': K + I*(jitter+gv) (gpmcmc.py:312).
    form='conditional': (Kxx + jitter*I) + Knx ([3P] Marginal._build_conditional)."""
    d = X.shape[1]
    _, _, _, gv, jitter = split_theta(theta, d, len(kerns))
    K = kernel_matrix(X, None, kerns, ops, theta)
    n = K.shape[0]
    idx = np.arange(n)
    if form == "marginal":
        s = np.sqrt(gv)
        K[idx, idx] += s * s
        K[idx, idx] += jitter
    elif form == "explicit":
        K[idx, idx] += jitter + gv
    elif form == "conditional":
        s = np.sqrt(gv)
        K[idx, idx] += jitter
        K[idx, idx] += s * s
    else:
        raise ValueError(form)
    if extra_diag is not None:  # inverse_opt's K += diag(ynoise) (gpmcmc.py:1158)
        K[idx, idx] += np.asarray(extra_diag, dtype=np.float64)
    return K


def lml(X, y, kerns, ops, theta, form="marginal", return_parts=False, extra_diag=None):
    """K3,K4,K6.  [3P] MvNormal.logp via quaddist_chol (scipy.linalg.cholesky lower +
    solve_triangular), == gpmcmc.py:313-318 without the warp Jacobian.
    Non-PD -> -inf ([3P] check_parameters 'posdef')."""
    K = noisy_cov(X, kerns, ops, theta, form, extra_diag)
    n = len(y)
    try:
        L = sla.cholesky(K, lower=True)
    except sla.LinAlgError:
        return (-np.inf, None, None) if return_parts else -np.inf
    beta = sla.solve_triangular(L, y, lower=True)
    quad = np.sum(beta ** 2)
    logdet = np.sum(np.log(np.diag(L)))
    val = -0.5 * n * np.log(2.0 * np.pi) - 0.5 * quad - logdet
    if return_parts:
        return val, L, beta
    return val


def lml_grad(X, y, kerns, ops, theta, form="marginal"):
    """K7.  Analytic  dLML/dtheta_k = 1/2 tr((alpha alpha^T - K^-1) dK/dtheta_k)  for the
    natural parameters in the C-ABI layout (d/d jitter is returned too; it equals d/d gv).
    The reference gets the same quantity by reverse-mode autodiff inside pm.find_MAP /
    pm.sample (gpmcmc.py:345,351)."""
    d = X.shape[1]
    nk = len(kerns)
    ls, kv, alpha, gv, jitter = split_theta(theta, d, nk)
    val, L, beta = lml(X, y, kerns, ops, theta, form, return_parts=True)
    g = np.zeros(nk * d + 2 * nk + 2)
    if L is None:
        return val, g
    n = len(y)
    a = sla.solve_triangular(L, beta, lower=True, trans="T")
    Kinv = sla.cho_solve((L, True), np.eye(n))
    W = np.outer(a, a) - Kinv
    comps, r2s = component_matrices(X, None, kerns, ls, kv, alpha)
    # coefficient dK/dK_c of the left-to-right fold
    pref = [None] * nk
    T = comps[0]
    pref[0] = np.ones_like(T)
    for i in range(1, nk):
        pref[i] = np.ones_like(T) if ops[i - 1] == "+" else T.copy()
        T = T + comps[i] if ops[i - 1] == "+" else T * comps[i]
    for c in range(nk):
        coef = pref[c]
        for i in range(c + 1, nk):
            if ops[i - 1] == "*":
                coef = coef * comps[i]
        WC = W * coef
        dk = kv[c] * base_kernel_dr2(kerns[c], r2s[c], alpha[c])
        dk = np.where(r2s[c] > 0.0, dk, 0.0)  # derivative of clip(.,0,inf)
        G = WC * dk
        for m in range(d):
            diff = (X[:, m : m + 1] - X[:, m : m + 1].T) * (1.0 / ls[c, m])
            g[c * d + m] = 0.5 * np.sum(G * (-2.0 * diff ** 2 / ls[c, m]))
        g[nk * d + c] = 0.5 * np.sum(WC * comps[c]) / kv[c]
        if kerns[c] == "RatQuad":
            u = 0.5 * r2s[c] / alpha[c]
            dka = comps[c] * (-np.log1p(u) + u / (1.0 + u))
            g[nk * d + nk + c] = 0.5 * np.sum(WC * dka)
    g[nk * d + 2 * nk] = 0.5 * np.trace(W)
    g[nk * d + 2 * nk + 1] = 0.5 * np.trace(W)
    return val, g


def lml_grad_data(X, y, kerns, ops, theta, form="marginal", extra_diag=None):
    """Data-side gradients of the LML: (LML, dLML/dy, dLML/dX).
        dLML/dy = -alpha,   dLML/dx_im = sum_j (alpha_i alpha_j - Kinv_ij) dK_ij/dx_im
    (row i and column i of 1/2 tr(W dK) contribute equally).  The reference gets them by autodiff when
    warp parameters (cwgp / iwgp, gpmcmc.py:211-279) or observation inputs (inverse_opt,
    gpmcmc.py:1096-1101) are random variables of the PyMC model."""
    d = X.shape[1]
    nk = len(kerns)
    ls, kv, alpha, gv, jitter = split_theta(theta, d, nk)
    val, L, beta = lml(X, y, kerns, ops, theta, form, return_parts=True, extra_diag=extra_diag)
    n = len(y)
    if L is None:
        return val, np.zeros(n), np.zeros((n, d))
    a = sla.solve_triangular(L, beta, lower=True, trans="T")
    Kinv = sla.cho_solve((L, True), np.eye(n))
    W = np.outer(a, a) - Kinv
    comps, r2s = component_matrices(X, None, kerns, ls, kv, alpha)
    pref = [None] * nk
    T = comps[0]
    pref[0] = np.ones_like(T)
    for i in range(1, nk):
        pref[i] = np.ones_like(T) if ops[i - 1] == "+" else T.copy()
        T = T + comps[i] if ops[i - 1] == "+" else T * comps[i]
    gX = np.zeros((n, d))
    for c in range(nk):
        coef = pref[c]
        for i in range(c + 1, nk):
            if ops[i - 1] == "*":
                coef = coef * comps[i]
        dk = kv[c] * base_kernel_dr2(kerns[c], r2s[c], alpha[c])
        dk = np.where(r2s[c] > 0.0, dk, 0.0)
        G = W * coef * dk
        for m in range(d):
            diff = X[:, m : m + 1] - X[:, m : m + 1].T
            gX[:, m] += np.sum(G * diff, axis=1) * (2.0 / ls[c, m] ** 2)
    return val, -a, gX


# --------------------------------------------------------------------------- K8
def predict(X, y, Xnew, kerns, ops, theta, pred_noise=True):
    """K8.  [3P] Marginal._build_conditional(diag=True) as called at gpmcmc.py:593-594;
    the same algebra is written out in-tree at gpmcmc.py:766-778."""
    d = X.shape[1]
    _, _, _, gv, _ = split_theta(theta, d, len(kerns))
    K = noisy_cov(X, kerns, ops, theta, form="conditional")
    L = sla.cholesky(K, lower=True)
    Kxs = kernel_matrix(X, Xnew, kerns, ops, theta)
    A = sla.solve_triangular(L, Kxs, lower=True)
    v = sla.solve_triangular(L, y, lower=True)
    mu = A.T @ v
    var = kernel_diag(kerns, ops, theta, d) - np.sum(np.square(A), 0)
    if pred_noise:
        s = np.sqrt(gv)
        var = var + s * s
    return mu, var


# ---------------------------------------------------------- synthetic workloads
def predict_grad(X, y, Xnew, kerns, ops, theta):
    """Gradients of the conditional mean / variance w.r.t. each prediction point (converted inputs):
    d mu/dx* = sum_i alpha_i dk(x_i,x*)/dx*,  d var/dx* = -2 sum_i w_i dk(x_i,x*)/dx*,  w = K^-1 k(X,x*).
    The reference differentiates the same single-point predictive by PyTensor autodiff (gpmcmc.py:766-801)."""
    d = X.shape[1]
    nk = len(kerns)
    ls, kv, alpha, gv, jitter = split_theta(theta, d, nk)
    K = noisy_cov(X, kerns, ops, theta, form="conditional")
    L = sla.cholesky(K, lower=True)
    a = sla.cho_solve((L, True), y)
    M = Xnew.shape[0]
    dmu, dvar = np.zeros((M, d)), np.zeros((M, d))
    for p in range(M):
        xs = Xnew[p : p + 1]
        comps, r2s = component_matrices(X, xs, kerns, ls, kv, alpha)  # n x 1 each
        kstar = combine(comps, ops)[:, 0]
        w = sla.cho_solve((L, True), kstar)
        pref = [None] * nk
        T = comps[0]
        pref[0] = np.ones_like(T)
        for i in range(1, nk):
            pref[i] = np.ones_like(T) if ops[i - 1] == "+" else T.copy()
            T = T + comps[i] if ops[i - 1] == "+" else T * comps[i]
        for c in range(nk):
            coef = pref[c]
            for i in range(c + 1, nk):
                if ops[i - 1] == "*":
                    coef = coef * comps[i]
            dk = kv[c] * base_kernel_dr2(kerns[c], r2s[c], alpha[c])
            dk = np.where(r2s[c] > 0.0, dk, 0.0)
            g = (coef * dk)[:, 0]
            for m in range(d):
                dkx = g * 2.0 * (xs[0, m] - X[:, m]) / ls[c, m] ** 2
                dmu[p, m] += a @ dkx
                dvar[p, m] += -2.0 * (w @ dkx)
    return dmu, dvar


def synth_problem(N, d, seed=0):
    """SURVEY.md section 8d inputs: LHS in [0,1]^d (mirrors lhc.py:42-43), standardised
    y = sin(3 sum x) + sum x^2/d + N(0,1e-4)."""
    from scipy.stats import qmc

    X = qmc.LatinHypercube(d, seed=seed).random(N)
    rng = np.random.default_rng(seed)
    f = np.sin(3.0 * X.sum(1)) + (X ** 2).sum(1) / d
    yv = f + rng.normal(0.0, 1e-2, N)
    yv = (yv - yv.mean()) / yv.std()
    return np.ascontiguousarray(X), np.ascontiguousarray(yv)


def synth_theta(d, nkern=1, kv=1.7, gv=1e-4, jitter=1e-6):
    ls = np.tile(np.exp(np.linspace(np.log(0.4), np.log(1.5), d)), (nkern, 1))
    return pack_theta(ls, np.full(nkern, kv), gv, jitter)
