"""Pin the CPU oracle (oracle/gp_oracle.py) against independent vectors: closed forms, 50-digit
mpmath, scikit-learn, finite differences and the tutorial notebook's recorded numbers."""
import numpy as np
import pytest

from conftest import case_theta, load_json
from oracle import gp_oracle as orc


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def test_closed_form_n1():
    cf = load_json("closed_form.json")["n1"]
    X = np.array([cf["x"]])
    y = np.array([cf["y"]])
    for kern in ("RBF", "Matern52"):
        theta = orc.pack_theta([cf["ls"]], [cf["kv"]], cf["gv"], cf["jitter"])
        val = orc.lml(X, y, [kern], [], theta, form="explicit")
        assert _rel(val, float(cf["lml_" + kern])) < 1e-13


def test_closed_form_n2():
    cf = load_json("closed_form.json")["n2"]
    X = np.array(cf["x"]).reshape(2, 1)
    y = np.array(cf["y"])
    theta = orc.pack_theta([[cf["l"]]], [cf["kv"]], cf["gv"], cf["jitter"])
    for form in ("explicit", "marginal"):
        assert _rel(orc.lml(X, y, ["RBF"], [], theta, form=form), float(cf["lml_RBF"])) < 1e-13


def test_lml_vs_mpmath(mp_cases):
    for c in mp_cases:
        X, y = np.array(c["X"]), np.array(c["y"])
        val = orc.lml(X, y, c["kerns"], c["ops"], case_theta(c))
        # tolerance scales with conditioning; these cases have gv=1e-3 so cond(K) <~ 1e5
        tol = 1e-9 if "Exponential" in c["kerns"] else 1e-11
        assert _rel(val, float(c["lml"])) < tol, c["name"]


def test_kernel_entries_vs_mpmath(mp_cases):
    for c in mp_cases:
        X = np.array(c["X"])
        K = orc.kernel_matrix(X, None, c["kerns"], c["ops"], case_theta(c))
        for i, j, v in c["K_samples"]:
            # the expansion-form distance loses ~1e-16 absolute in r2; Exponential/Matern turn that
            # into ~1e-10 near r = 0 through sqrt(r2 + 1e-12)
            tol = 2e-9 if any(k in ("Exponential", "Matern32", "Matern52") for k in c["kerns"]) else 1e-13
            assert abs(K[i, j] - float(v)) <= tol * max(1.0, abs(float(v))), (c["name"], i, j)


def test_grad_vs_mpmath(mp_cases):
    for c in mp_cases:
        if "grad" not in c:
            continue
        X, y = np.array(c["X"]), np.array(c["y"])
        _, g = orc.lml_grad(X, y, c["kerns"], c["ops"], case_theta(c))
        ref = np.array([float(v) for v in c["grad"]])
        nk, d = len(c["kerns"]), c["d"]
        got = g[: nk * d + 2 * nk + 1]
        for c_i, kern in enumerate(c["kerns"]):
            if kern != "RatQuad":  # alpha is inert for the other kernels
                ref[nk * d + nk + c_i] = 0.0
        scale = np.maximum(np.abs(ref), 1e-3 * np.abs(ref).max())
        tol = 1e-6 if "Exponential" in c["kerns"] else 1e-8
        assert np.all(np.abs(got - ref) / scale < tol), (c["name"], got, ref)


def test_predict_vs_mpmath(mp_cases):
    for c in mp_cases:
        X, y, Xn = np.array(c["X"]), np.array(c["y"]), np.array(c["Xnew"])
        mu, var = orc.predict(X, y, Xn, c["kerns"], c["ops"], case_theta(c))
        rmu = np.array([float(v) for v in c["mu"]])
        rvar = np.array([float(v) for v in c["var"]])
        tol = 1e-7 if "Exponential" in c["kerns"] else 1e-9
        assert np.allclose(mu, rmu, rtol=tol, atol=tol), c["name"]
        assert np.allclose(var, rvar, rtol=tol, atol=tol), c["name"]


@pytest.mark.parametrize("kern,nu", [("RBF", None), ("Matern52", 2.5), ("Matern32", 1.5)])
def test_lml_and_grad_vs_sklearn(kern, nu):
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel, Matern, WhiteKernel

    N, d = 128, 2
    X, y = orc.synth_problem(N, d, seed=3)
    theta = orc.synth_theta(d)
    ls, kv, _, gv, jitter = orc.split_theta(theta, d, 1)
    base = RBF(ls[0]) if nu is None else Matern(ls[0], nu=nu)
    k = ConstantKernel(kv[0]) * base + WhiteKernel(gv + jitter)
    gpr = GaussianProcessRegressor(kernel=k, alpha=0.0, optimizer=None).fit(X, y)
    ref, gref = gpr.log_marginal_likelihood(gpr.kernel_.theta, eval_gradient=True)
    val, g = orc.lml_grad(X, y, [kern], [], theta)
    assert _rel(val, ref) < 1e-10
    # sklearn differentiates w.r.t. log-parameters: [log kv, log ls..., log noise]
    g_log = np.concatenate([[g[d] * kv[0]], g[:d] * ls[0], [g[d + 2] * (gv + jitter)]])
    assert np.allclose(g_log, gref, rtol=1e-7, atol=1e-7)


def test_grad_finite_difference_composite():
    N, d = 60, 3
    X, y = orc.synth_problem(N, d, seed=5)
    kerns, ops = ["RBF", "Matern52", "RatQuad"], ["+", "*"]
    theta = orc.synth_theta(d, nkern=3, gv=1e-2)
    theta[3 * d + 3: 3 * d + 6] = [1.0, 1.0, 1.7]
    _, g = orc.lml_grad(X, y, kerns, ops, theta)
    for i in range(len(theta) - 1):
        h = 1e-6 * max(1.0, abs(theta[i]))
        tp, tm = theta.copy(), theta.copy()
        tp[i] += h
        tm[i] -= h
        fd = (orc.lml(X, y, kerns, ops, tp) - orc.lml(X, y, kerns, ops, tm)) / (2 * h)
        assert abs(fd - g[i]) <= 1e-5 * max(1.0, abs(fd)), (i, fd, g[i])


def test_non_pd_gives_minus_inf():
    X = np.zeros((4, 1))
    y = np.ones(4)
    theta = orc.pack_theta([[1.0]], [1.0], 0.0, -1e-3)
    assert orc.lml(X, y, ["RBF"], [], theta) == -np.inf


def _data_grad_cases():
    import json
    import os

    with open(os.path.join(os.path.dirname(__file__), "golden", "mpmath_data_grad.json")) as f:
        return json.load(f)


def _case_arrays(c):
    X, y, xs = np.array(c["X"]), np.array(c["y"]), np.array([c["xstar"]])
    nk, d = len(c["kerns"]), c["d"]
    theta = np.concatenate([np.array(c["ls"]).ravel(), c["kv"], c["alpha"], [c["gv"], c["jitter"]]])
    assert theta.shape == (nk * d + 2 * nk + 2,)
    return X, y, xs, theta


def test_data_gradients_match_mpmath_golden():
    """dLML/dX, dLML/dy and d mu/dx*, d var/dx* of the oracle against 50-digit numerical differentiation
    (oracle/gen_golden_data.py): pins the restatements that the device's data-gradient entry points are tested against."""
    for c in _data_grad_cases():
        X, y, xs, theta = _case_arrays(c)
        _, gy, gX = orc.lml_grad_data(X, y, c["kerns"], c["ops"], theta, form="explicit")
        gX_ref = np.array([[float(v) for v in row] for row in c["gX"]])
        gy_ref = np.array([float(v) for v in c["gy"]])
        tol = 1e-7 if "Exponential" in c["kerns"] else 1e-9
        assert np.abs(gX - gX_ref).max() <= tol * np.abs(gX_ref).max(), c["name"]
        assert np.abs(gy - gy_ref).max() <= tol * np.abs(gy_ref).max(), c["name"]
        dmu, dvar = orc.predict_grad(X, y, xs, c["kerns"], c["ops"], theta)
        dmu_ref = np.array([float(v) for v in c["dmu"]])
        dvar_ref = np.array([float(v) for v in c["dvar"]])
        assert np.abs(dmu[0] - dmu_ref).max() <= 10 * tol * max(np.abs(dmu_ref).max(), 1e-300), c["name"]
        assert np.abs(dvar[0] - dvar_ref).max() <= 10 * tol * max(np.abs(dvar_ref).max(), 1e-300), c["name"]


def _illcond_truth(c):
    """(theta, lml, gradient in C-ABI positions, their indices, cond) of an ill-conditioned golden case."""
    nk, d = len(c["kerns"]), c["d"]
    theta = case_theta(c)
    idx = list(range(nk * d)) + [nk * d + k for k in range(nk)] + [nk * d + 2 * nk]  # ls, kv, gv
    K = orc.noisy_cov(np.array(c["X"]), c["kerns"], c["ops"], theta)
    return theta, float(c["lml"]), np.array([float(v) for v in c["grad"]]), idx, np.linalg.cond(K)


def test_oracle_in_the_ill_conditioned_regime_vs_mpmath():
    """cond(K) 2e7 .. 8e8 (oracle/gen_golden_illcond.py): the NumPy oracle's forward error against the 50-digit truth stays
    within a few cond * eps -- the yardstick the GPU test measures the device against."""
    eps = np.finfo(float).eps
    for c in load_json("mpmath_illcond.json"):
        theta, lml, grad, idx, cond = _illcond_truth(c)
        assert 1e7 <= cond <= 1e9, (c["name"], cond)
        val, g = orc.lml_grad(np.array(c["X"]), np.array(c["y"]), c["kerns"], c["ops"], theta)
        assert abs(val - lml) <= 8 * cond * eps * max(abs(lml), 1.0), (c["name"], val, lml)
        err = np.abs(g[idx] - grad).max() / np.abs(grad).max()
        assert err <= 8 * cond * eps, (c["name"], err / (cond * eps))
