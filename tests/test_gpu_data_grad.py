"""Parity of the data-side gradients (dLML/dy = -alpha, dLML/dX) and of the per-point noise diagonal with
the oracle: the quantities the reference's autodiff supplies when warp parameters (cwgp / iwgp,
gpmcmc.py:211-279) or observation inputs (inverse_opt, gpmcmc.py:1096-1165) are model variables."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mods():
    import torch

    assert torch.cuda.is_available()
    from andvaranaut_amd import MiGP
    from oracle import gp_oracle as orc

    return MiGP, orc


def _split(kernel):
    return kernel.replace("*", "+").split("+"), [c for c in kernel if c in "+*"]


CASES = [(1, 1, "RBF"), (3, 2, "Matern52"), (64, 2, "RBF"), (65, 3, "Matern32"), (200, 16, "Matern52"),
         (513, 17, "RBF"), (700, 33, "Matern52"), (640, 5, "RBF*Matern52"), (900, 2, "RBF+Matern52*Exponential"),
         (300, 3, "RatQuad"), (1500, 8, "Matern32+RBF")]


@pytest.mark.parametrize("N,d,kernel", CASES)
def test_data_gradients_match_oracle(N, d, kernel):
    MiGP, orc = _mods()
    X, y = orc.synth_problem(max(N, 3), d, seed=3 * N + d)
    X, y = X[:N], y[:N]
    kerns, ops = _split(kernel)
    theta = orc.synth_theta(d, nkern=len(kerns), gv=1e-3)
    gp = MiGP(X, y, kernel)
    val, g, gy, gX = gp.lml_grad_data(theta)
    ref, gy_ref, gX_ref = orc.lml_grad_data(X, y, kerns, ops, theta)
    assert abs(val - ref) <= 1e-9 * abs(ref)
    tol = 1e-6 if "Exponential" in kernel else 1e-8
    assert np.abs(gy - gy_ref).max() <= tol * max(np.abs(gy_ref).max(), 1e-300)
    assert gX.shape == (N, d)
    assert np.abs(gX - gX_ref).max() <= tol * max(np.abs(gX_ref).max(), 1e-300), np.abs(gX - gX_ref).max()
    gp.close()


def test_data_gradients_finite_differences_through_the_device():
    """Independent of the oracle: central differences of the device LML in X and y."""
    MiGP, orc = _mods()
    X, y = orc.synth_problem(150, 3, seed=11)
    theta = orc.synth_theta(3, gv=1e-3)
    gp = MiGP(X, y, "Matern52")
    _, _, gy, gX = gp.lml_grad_data(theta)
    h = 1e-6
    for i, m in ((0, 0), (17, 2), (149, 1)):
        Xp, Xm = X.copy(), X.copy()
        Xp[i, m] += h
        Xm[i, m] -= h
        gp.update_data(X=Xp)
        fp = gp.lml(theta)
        gp.update_data(X=Xm)
        fm = gp.lml(theta)
        assert abs((fp - fm) / (2 * h) - gX[i, m]) <= 2e-5 * max(1.0, abs(gX[i, m]))
    gp.update_data(X=X)
    for i in (3, 99):
        yp, ym = y.copy(), y.copy()
        yp[i] += h
        ym[i] -= h
        gp.update_data(y=yp)
        fp = gp.lml(theta)
        gp.update_data(y=ym)
        fm = gp.lml(theta)
        assert abs((fp - fm) / (2 * h) - gy[i]) <= 2e-5 * max(1.0, abs(gy[i]))
    gp.close()


def test_update_data_matches_fresh_handle():
    MiGP, orc = _mods()
    X, y = orc.synth_problem(400, 4, seed=2)
    X2, y2 = orc.synth_problem(400, 4, seed=9)
    theta = orc.synth_theta(4)
    gp = MiGP(X, y, "RBF")
    a = gp.lml(theta)
    gp.update_data(X=X2, y=y2)
    b = gp.lml(theta)
    gp2 = MiGP(X2, y2, "RBF")
    assert b == gp2.lml(theta) and a != b
    gp.close()
    gp2.close()


@pytest.mark.parametrize("N,kernel", [(5, "RBF"), (300, "Matern52"), (1000, "RBF+Matern32")])
def test_extra_diagonal(N, kernel):
    """K + diag(v): LML, gradient and data gradients with a per-point noise vector (gpmcmc.py:1134-1158)."""
    MiGP, orc = _mods()
    d = 3
    X, y = orc.synth_problem(N, d, seed=N)
    kerns, ops = _split(kernel)
    theta = orc.synth_theta(d, nkern=len(kerns), gv=0.0, jitter=0.0)
    rng = np.random.default_rng(0)
    v = 10.0 ** rng.uniform(-5, -2, N)
    gp = MiGP(X, y, kernel)
    gp.set_diag(v)
    val, g, gy, gX = gp.lml_grad_data(theta)
    ref, gy_ref, gX_ref = orc.lml_grad_data(X, y, kerns, ops, theta, extra_diag=v)
    assert abs(val - ref) <= 1e-9 * abs(ref)
    assert np.abs(gy - gy_ref).max() <= 1e-8 * np.abs(gy_ref).max()
    assert np.abs(gX - gX_ref).max() <= 1e-8 * np.abs(gX_ref).max()
    assert abs(gp.lml(theta) - ref) <= 1e-9 * abs(ref)
    gp.set_diag(None)
    theta2 = orc.synth_theta(d, nkern=len(kerns))
    assert abs(gp.lml(theta2) - orc.lml(X, y, kerns, ops, theta2)) <= 1e-9 * abs(ref)
    gp.close()


def test_data_gradients_need_a_gradient_evaluation_first():
    MiGP, orc = _mods()
    X, y = orc.synth_problem(50, 2, seed=0)
    gp = MiGP(X, y, "RBF")
    gp.lml(orc.synth_theta(2))
    import ctypes

    out = np.empty(50)
    r = gp.lib.mi_gp_alpha(gp.h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    assert r == -1 and b"mi_gp_lml_grad" in gp.lib.mi_gp_last_error(gp.h)
    gp.close()


PG_CASES = [(1, 1, "RBF", 1), (50, 2, "Matern52", 3), (129, 3, "Matern32", 1), (700, 16, "Matern52", 5),
            (640, 5, "RBF*Matern52", 2), (900, 2, "RBF+Matern52*Exponential", 4), (300, 3, "RatQuad", 130),
            (2000, 33, "RBF", 2)]


@pytest.mark.parametrize("N,d,kernel,M", PG_CASES)
def test_predictive_gradients_match_oracle(N, d, kernel, M):
    """mi_gp_predict_grad: the differentiable single-point predictive of BO's refinement (gpmcmc.py:766-801)."""
    MiGP, orc = _mods()
    X, y = orc.synth_problem(max(N, 3), d, seed=N + 7 * d)
    X, y = X[:N], y[:N]
    kerns, ops = _split(kernel)
    theta = orc.synth_theta(d, nkern=len(kerns), gv=1e-3)
    Xn = np.random.default_rng(M).uniform(0.05, 0.95, (M, d))
    gp = MiGP(X, y, kernel)
    mu, var, dmu, dvar = gp.predict_grad(theta, Xn)
    mu_o, var_o = orc.predict(X, y, Xn, kerns, ops, theta)
    dmu_o, dvar_o = orc.predict_grad(X, y, Xn, kerns, ops, theta)
    assert np.allclose(mu, mu_o, rtol=1e-7, atol=1e-9) and np.allclose(var, var_o, rtol=1e-6, atol=1e-10)
    tol = 1e-5 if "Exponential" in kernel else 1e-7
    assert np.abs(dmu - dmu_o).max() <= tol * max(np.abs(dmu_o).max(), 1e-300)
    assert np.abs(dvar - dvar_o).max() <= tol * max(np.abs(dvar_o).max(), 1e-300)
    # a second call on the resident factorisation (other points) gives the same answers as a fresh one
    Xn2 = Xn[::-1].copy()
    _, _, dmu2, dvar2 = gp.predict_grad(theta, Xn2, refactor=False)
    assert np.allclose(dmu2, dmu[::-1], rtol=1e-12, atol=0) and np.allclose(dvar2, dvar[::-1], rtol=1e-12, atol=0)
    # and an LML evaluation in between does not leave a stale inverse behind
    gp.lml(theta)
    _, _, dmu3, _ = gp.predict_grad(theta, Xn)
    assert np.allclose(dmu3, dmu, rtol=1e-12, atol=0)
    gp.close()


def test_device_data_gradients_match_mpmath_golden():
    """The device entry points against the 50-digit mpmath vectors directly (tests/golden/mpmath_data_grad.json)."""
    import json
    import os

    MiGP, _ = _mods()
    with open(os.path.join(os.path.dirname(__file__), "golden", "mpmath_data_grad.json")) as f:
        cases = json.load(f)
    for c in cases:
        X, y, xs = np.array(c["X"]), np.array(c["y"]), np.array([c["xstar"]])
        theta = np.concatenate([np.array(c["ls"]).ravel(), c["kv"], c["alpha"], [c["gv"], c["jitter"]]])
        kernel = c["kerns"][0]
        for o, k in zip(c["ops"], c["kerns"][1:]):
            kernel += o + k
        gp = MiGP(X, y, kernel)
        _, _, gy, gX = gp.lml_grad_data(theta)
        _, _, dmu, dvar = gp.predict_grad(theta, xs, pred_noise=False)
        tol = 1e-6 if "Exponential" in c["kerns"] else 1e-8
        for got, key in ((gX, "gX"), (gy, "gy"), (dmu[0], "dmu"), (dvar[0], "dvar")):
            ref = np.array(c[key], dtype=object)
            ref = np.array([[float(v) for v in row] for row in ref]) if key == "gX" else np.array([float(v) for v in ref])
            assert np.abs(got - ref).max() <= tol * max(np.abs(ref).max(), 1e-300), (c["name"], key)
        gp.close()


@pytest.mark.parametrize("kernel,d", [("Exponential*Matern32+Matern32+RBF", 2), ("RBF+Matern52*Matern32+RBF", 3),
                                      ("Matern52+RBF*RBF", 17), ("RBF+Matern32", 40), ("Matern32*RBF+Exponential+Matern52", 5)])
def test_gradients_are_reproducible_for_every_component_count(kernel, d):
    """Round-3 regression: a build of grad_x_kernel<4, 1> (fully unrolled, 86 KB of code, 256 VGPRs) returned
    NONDETERMINISTIC garbage in dLML/dX -- entries of 1e13 that changed from call to call on identical inputs -- while LML,
    dLML/dtheta and dLML/dy of the same evaluation were right.  Fresh handles and repeated evaluations must return the
    same bits for every gradient entry point, and those must match the oracle."""
    MiGP, orc = _mods()
    N = 207
    X, y = orc.synth_problem(N, d, seed=17)
    kerns, ops = _split(kernel)
    rng = np.random.default_rng(d)
    for rescale in (1.0, np.sqrt(d / 2.0)):
        theta = orc.synth_theta(d, nkern=len(kerns), gv=1e-3)
        theta[: len(kerns) * d] *= rescale
        _, g_ref = orc.lml_grad(X, y, kerns, ops, theta)
        _, _, gX_ref = orc.lml_grad_data(X, y, kerns, ops, theta)
        Xn = rng.uniform(0.05, 0.95, (4, d))
        dmu_ref, dvar_ref = orc.predict_grad(X, y, Xn, kerns, ops, theta)
        first = None
        for _ in range(3):
            gp = MiGP(X, y, kernel)
            for _ in range(3):
                _, g, _, gX = gp.lml_grad_data(theta)
                _, _, dmu, dvar = gp.predict_grad(theta, Xn)
                if first is None:
                    first = (g.copy(), gX.copy(), dmu.copy(), dvar.copy())
                assert np.array_equal(g, first[0]) and np.array_equal(gX, first[1])
                assert np.array_equal(dmu, first[2]) and np.array_equal(dvar, first[3])
            gp.close()
        tol = 1e-5 if "Exponential" in kernel else 1e-7
        scale = np.maximum(np.abs(gX_ref), 1e-3 * np.abs(gX_ref).max())
        assert np.max(np.abs(first[1] - gX_ref) / scale) <= tol
        assert np.abs(first[0] - g_ref).max() <= tol * np.abs(g_ref).max()
        assert np.abs(first[2] - dmu_ref).max() <= tol * np.abs(dmu_ref).max()
        assert np.abs(first[3] - dvar_ref).max() <= tol * np.abs(dvar_ref).max()


@pytest.mark.parametrize("N,d,kernel", [(300, 129, "RBF"), (260, 200, "Matern52"), (200, 300, "RBF+Matern32")])
def test_data_and_predictive_gradients_beyond_128_input_dimensions(N, d, kernel):
    """The reference loops over any number of input dimensions (gpmcmc.py:235-237, 282-307); round 2 refused d > 128 in
    mi_gp_grad_x / mi_gp_predict_grad.  dLML/dX now runs one pass per window of 128 output dimensions, the predictive
    gradient sizes its LDS by d."""
    MiGP, orc = _mods()
    X, y = orc.synth_problem(N, d, seed=d)
    kerns, ops = _split(kernel)
    theta = orc.synth_theta(d, nkern=len(kerns), gv=1e-3)
    theta[: len(kerns) * d] *= np.sqrt(d / 2.0)  # length scales ~ sqrt(d): K is not numerically the identity
    gp = MiGP(X, y, kernel)
    val, g, gy, gX = gp.lml_grad_data(theta)
    ref, gy_ref, gX_ref = orc.lml_grad_data(X, y, kerns, ops, theta)
    assert abs(val - ref) <= 1e-9 * abs(ref)
    assert np.abs(gy - gy_ref).max() <= 1e-8 * np.abs(gy_ref).max()
    assert np.abs(gX - gX_ref).max() <= 1e-8 * np.abs(gX_ref).max()
    Xn = np.random.default_rng(d).uniform(0.05, 0.95, (3, d))
    mu, var, dmu, dvar = gp.predict_grad(theta, Xn)
    dmu_o, dvar_o = orc.predict_grad(X, y, Xn, kerns, ops, theta)
    assert np.abs(dmu - dmu_o).max() <= 1e-7 * np.abs(dmu_o).max()
    assert np.abs(dvar - dvar_o).max() <= 1e-7 * np.abs(dvar_o).max()
    gp.close()


def test_update_data_uploads_the_finite_member_of_a_pair_and_factor_refuses_bad_data():
    """update_data(X=bad, y=new) dropped y; a later update_data(X=good) then evaluated against the stale y."""
    import torch

    assert torch.cuda.is_available()
    from andvaranaut_amd import MiGP
    from oracle import gp_oracle as orc

    N, d = 300, 3
    X, y = orc.synth_problem(N, d, seed=2)
    theta = orc.synth_theta(d)
    gp = MiGP(X, y, "RBF")
    y2 = y[::-1].copy()
    Xbad = X.copy()
    Xbad[5, 1] = np.nan
    gp.update_data(X=Xbad, y=y2)
    assert gp.lml(theta) == -np.inf
    with pytest.raises(FloatingPointError):
        gp.factor(theta)
    gp.update_data(X=X)  # y2 must be resident by now
    ref = orc.lml(X, y2, ["RBF"], [], theta)
    val = gp.lml(theta)
    assert abs(val - ref) <= 1e-10 * abs(ref), (val, ref)
    # a bad y alone keeps the handle "bad" until a finite y arrives, whatever happens to X
    ybad = y.copy()
    ybad[0] = np.inf
    gp.update_data(y=ybad)
    gp.update_data(X=X)
    assert gp.lml(theta) == -np.inf
    gp.update_data(y=y)
    ref = orc.lml(X, y, ["RBF"], [], theta)
    assert abs(gp.lml(theta) - ref) <= 1e-10 * abs(ref)
    gp.close()
