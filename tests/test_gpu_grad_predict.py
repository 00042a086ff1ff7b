"""Parity of the HIP gradient (K7) and conditional (K8) paths with the oracle and the mpmath goldens."""
import numpy as np
import pytest

from conftest import case_theta

pytestmark = pytest.mark.gpu


def _mods():
    import torch

    assert torch.cuda.is_available()
    from andvaranaut_amd import MiGP
    from oracle import gp_oracle as orc

    return MiGP, orc


def _split(kernel):
    return kernel.replace("*", "+").split("+"), [c for c in kernel if c in "+*"]


GRAD_CASES = [(1, 1, "RBF"), (2, 2, "Matern52"), (100, 2, "RBF"), (129, 3, "Matern32"), (300, 3, "Matern52"),
              (1000, 4, "Matern32+RBF"), (640, 5, "RBF*Matern52"), (700, 3, "RatQuad"), (515, 33, "RBF"),
              (900, 2, "RBF+Matern52*Exponential"), (2500, 6, "Matern52")]


@pytest.mark.parametrize("N,d,kernel", GRAD_CASES)
def test_lml_grad_matches_oracle(N, d, kernel):
    MiGP, orc = _mods()
    X, y = orc.synth_problem(max(N, 3), d, seed=N + d)
    X, y = X[:N], y[:N]
    kerns, ops = _split(kernel)
    theta = orc.synth_theta(d, nkern=len(kerns), gv=1e-3)
    gp = MiGP(X, y, kernel)
    val, g = gp.lml_grad(theta)
    ref, gref = orc.lml_grad(X, y, kerns, ops, theta)
    tol = 1e-8 if "Exponential" in kernel else 1e-10
    assert abs(val - ref) <= tol * abs(ref)
    # fp64 tolerance on the gradient: 1e-8 of its largest component (conditioning ~1e4 here)
    gtol = 1e-6 if "Exponential" in kernel else 1e-8
    assert np.abs(g - gref).max() <= gtol * max(np.abs(gref).max(), 1e-300), (g, gref)
    gp.close()


def test_grad_matches_mpmath_golden(mp_cases):
    MiGP, _ = _mods()
    for c in mp_cases:
        if "grad" not in c:
            continue
        X, y = np.array(c["X"]), np.array(c["y"])
        kernel = c["kerns"][0]
        for o, k in zip(c["ops"], c["kerns"][1:]):
            kernel += o + k
        gp = MiGP(X, y, kernel)
        _, g = gp.lml_grad(case_theta(c))
        ref = np.array([float(v) for v in c["grad"]])
        nk, d = len(c["kerns"]), c["d"]
        for i, kern in enumerate(c["kerns"]):
            if kern != "RatQuad":
                ref[nk * d + nk + i] = 0.0
        got = g[: nk * d + 2 * nk + 1]
        scale = np.maximum(np.abs(ref), 1e-3 * np.abs(ref).max())
        tol = 1e-6 if "Exponential" in c["kerns"] else 1e-8
        assert np.all(np.abs(got - ref) / scale < tol), (c["name"], got, ref)
        assert g[-1] == g[-2]  # d/d jitter == d/d gv
        gp.close()


def test_grad_against_finite_differences_of_the_gpu_lml():
    MiGP, orc = _mods()
    N, d = 400, 4
    X, y = orc.synth_problem(N, d, seed=3)
    theta = orc.synth_theta(d, gv=1e-2)
    gp = MiGP(X, y, "Matern52")
    _, g = gp.lml_grad(theta)
    for i in [0, d - 1, d, d + 2]:
        h = 1e-6 * max(1.0, abs(theta[i]))
        tp, tm = theta.copy(), theta.copy()
        tp[i] += h
        tm[i] -= h
        fd = (gp.lml(tp) - gp.lml(tm)) / (2 * h)
        assert abs(fd - g[i]) <= 2e-5 * max(1.0, abs(fd)), (i, fd, g[i])
    gp.close()


PRED_CASES = [(1, 1, "RBF", 5), (100, 2, "RBF", 10), (300, 3, "Matern52", 129), (1000, 4, "Matern32+RBF", 1000),
              (640, 5, "RBF*Matern52", 77), (2048, 8, "RBF", 300)]


@pytest.mark.parametrize("N,d,kernel,M", PRED_CASES)
def test_predict_matches_oracle(N, d, kernel, M):
    MiGP, orc = _mods()
    X, y = orc.synth_problem(max(N, 3), d, seed=N)
    X, y = X[:N], y[:N]
    kerns, ops = _split(kernel)
    theta = orc.synth_theta(d, nkern=len(kerns), gv=1e-3)
    Xn = np.random.default_rng(M).random((M, d))
    gp = MiGP(X, y, kernel, need_grad=False)
    mu, var = gp.predict(theta, Xn)
    rmu, rvar = orc.predict(X, y, Xn, kerns, ops, theta)
    assert np.allclose(mu, rmu, rtol=1e-9, atol=1e-9)
    assert np.allclose(var, rvar, rtol=1e-8, atol=1e-11)
    mu2, var2 = gp.predict(theta, Xn, pred_noise=False, chunk=128)  # chunked path, no noise
    assert np.allclose(mu2, rmu, rtol=1e-9, atol=1e-9)
    assert np.allclose(var2, rvar - np.sqrt(1e-3) ** 2, rtol=1e-8, atol=1e-11)
    gp.close()


def test_predict_matches_mpmath_golden(mp_cases):
    MiGP, _ = _mods()
    for c in mp_cases:
        X, y, Xn = np.array(c["X"]), np.array(c["y"]), np.array(c["Xnew"])
        kernel = c["kerns"][0]
        for o, k in zip(c["ops"], c["kerns"][1:]):
            kernel += o + k
        gp = MiGP(X, y, kernel, need_grad=False)
        mu, var = gp.predict(case_theta(c), Xn)
        tol = 1e-7 if "Exponential" in c["kerns"] else 1e-9
        assert np.allclose(mu, [float(v) for v in c["mu"]], rtol=tol, atol=tol), c["name"]
        assert np.allclose(var, [float(v) for v in c["var"]], rtol=tol, atol=tol), c["name"]
        gp.close()


def test_predict_requires_positive_definite():
    MiGP, orc = _mods()
    X = np.zeros((10, 1))
    gp = MiGP(X, np.ones(10), "RBF", need_grad=False)
    with pytest.raises(FloatingPointError):
        gp.predict(orc.pack_theta([[1.0]], [1.0], 0.0, -1e-3), np.zeros((2, 1)))
    gp.close()


def test_interpolation_property_at_training_points():
    """With tiny noise the conditional mean at the training inputs reproduces y and the variance ~ gv."""
    MiGP, orc = _mods()
    N, d = 500, 2
    X, y = orc.synth_problem(N, d, seed=9)
    theta = orc.pack_theta(np.full((1, d), 0.3), [1.5], 1e-8, 1e-8)
    gp = MiGP(X, y, "Matern52", need_grad=False)
    mu, var = gp.predict(theta, X)
    assert np.abs(mu - y).max() < 2e-2  # residual = gv * alpha, small but not zero (y carries noise)
    assert np.all(var > 0) and var.max() < 1e-6
    gp.close()


@pytest.mark.parametrize("N,d,kernel,M", [(1, 1, "RBF", 3), (300, 3, "Matern52", 700), (1000, 4, "Matern32+RBF", 129), (2500, 6, "RBF", 5000)])
def test_predict_through_the_inverse_factor_matches_oracle(N, d, kernel, M):
    """mi_gp_predict_u (A = K* U as one GEMM) against the oracle's conditional and against the triangular-solve path."""
    MiGP, orc = _mods()
    X, y = orc.synth_problem(max(N, 3), d, seed=N + d)
    X, y = X[:N], y[:N]
    kerns, ops = _split(kernel)
    theta = orc.synth_theta(d, nkern=len(kerns), gv=1e-3)
    Xn = np.random.default_rng(M).uniform(0.0, 1.0, (M, d))
    gp = MiGP(X, y, kernel)
    mu_u, var_u = gp.predict(theta, Xn, via_inverse=True)
    mu_t, var_t = gp.predict(theta, Xn, via_inverse=False)
    mu_o, var_o = orc.predict(X, y, Xn, kerns, ops, theta)
    assert np.allclose(mu_u, mu_o, rtol=1e-8, atol=1e-9) and np.allclose(var_u, var_o, rtol=1e-7, atol=1e-10)
    assert np.allclose(mu_u, mu_t, rtol=1e-9, atol=1e-10) and np.allclose(var_u, var_t, rtol=1e-8, atol=1e-11)
    # default routing: large sweeps go through U, and an LML evaluation in between invalidates it
    gp.lml(theta)
    mu_d, _ = gp.predict(theta, Xn)
    assert np.allclose(mu_d, mu_o, rtol=1e-8, atol=1e-9)
    gp.close()


def test_pinned_host_route_of_small_predict_calls_returns_the_device_routes_bits(monkeypatch):
    """Round 6: up to MiGP.PINNED_IO_MAX_POINTS points travel through pinned host memory the kernels address directly (no
    torch copies or synchronisations around the call).  Same kernels on the same values: the blocked solve, the route through
    U = L^-T and predict_grad return the device-buffer route's bits -- 1 point, the limit, one more than the limit."""
    MiGP, orc = _mods()
    N, d = 1500, 5
    X, y = orc.synth_problem(N, d, seed=3)
    theta = orc.synth_theta(d)
    gp = MiGP(X, y, "Matern52")
    lim = MiGP.PINNED_IO_MAX_POINTS
    assert lim >= 16
    Xs = np.random.default_rng(0).random((lim + 1, d))
    got = {}
    for route in ("pinned", "device"):
        if route == "device":
            monkeypatch.setattr(MiGP, "PINNED_IO_MAX_POINTS", 0)
        for m in (1, 7, lim, lim + 1):
            got[route, m, "solve"] = gp.predict(theta, Xs[:m], via_inverse=False)
            got[route, m, "u"] = gp.predict(theta, Xs[:m], via_inverse=True)
        for m in (1, 7):
            got[route, m, "grad"] = gp.predict_grad(theta, Xs[:m], refactor=False)
    for (route, m, what), v in got.items():
        if route == "pinned":
            w = got["device", m, what]
            assert all(np.array_equal(a, b) for a, b in zip(v, w)), (m, what)
    rmu, rvar = orc.predict(X, y, Xs[:7], ["Matern52"], [], theta)
    assert np.allclose(got["pinned", 7, "solve"][0], rmu, rtol=1e-9, atol=1e-9)
    assert np.allclose(got["pinned", 7, "solve"][1], rvar, rtol=1e-8, atol=1e-11)
    gp.close()
