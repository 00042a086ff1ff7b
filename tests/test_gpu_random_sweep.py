"""Seeded random sweep over shapes and kernel structures: every device entry point of one handle (LML, gradient
w.r.t. theta / y / X, conditional, conditional gradient) against the oracle on the same inputs.  The fixed-case
tests pin the tile-boundary sizes; this one walks odd N, d and compositions nobody picked by hand."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NAMES = ["RBF", "Matern52", "Matern32", "Exponential"]


def _random_case(rng):
    nk = int(rng.integers(1, 5))
    if rng.random() < 0.15:
        kerns, ops = ["RatQuad"], []  # the reference only supports RatQuad on its own (gpmcmc.py:287)
    else:
        kerns = [NAMES[int(rng.integers(0, 3 if i else 4))] for i in range(nk)]
        ops = [("+", "*")[int(rng.integers(0, 2))] for _ in range(nk - 1)]
    N = int(rng.choice([int(rng.integers(1, 70)), int(rng.integers(70, 400)), int(rng.integers(400, 1300))]))
    d = int(rng.choice([1, 2, 3, int(rng.integers(4, 20)), int(rng.integers(20, 70))]))
    M = int(rng.integers(1, 300))
    return N, d, kerns, ops, M


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("SWEEP_SEEDS", "20"))))
def test_random_case_all_entry_points(seed):
    from andvaranaut_amd import MiGP
    from oracle import gp_oracle as orc

    rng = np.random.default_rng(1000 + seed)
    N, d, kerns, ops, M = _random_case(rng)
    kernel = kerns[0] + "".join(o + k for o, k in zip(ops, kerns[1:]))
    X, y = orc.synth_problem(max(N, 3), d, seed=seed)
    X, y = X[:N], y[:N]
    theta = orc.synth_theta(d, nkern=len(kerns), gv=10.0 ** rng.uniform(-5, -2))
    theta[: len(kerns) * d] *= rng.uniform(0.7, 1.6, len(kerns) * d)
    expo = "Exponential" in kerns
    # the quadratic form y^T K^-1 y carries a forward error of cond(K) * eps on BOTH sides (device and oracle); so do
    # alpha = K^-1 y and K^-1: all tolerances scale with the condition number of the noisy covariance (random
    # compositions in d = 1..3 with small noise reach 1e8 and beyond)
    cond = np.linalg.cond(orc.noisy_cov(X, kerns, ops, theta))
    tol = max(1e-8 if expo else 1e-10, 20.0 * cond * 2.2e-16)
    gp = MiGP(X, y, kernel)
    ref = orc.lml(X, y, kerns, ops, theta)
    val = gp.lml(theta)
    assert abs(val - ref) <= tol * max(abs(ref), 1.0), (kernel, N, d, val, ref)
    v2, g, gy, gx = gp.lml_grad_data(theta)
    _, rg = orc.lml_grad(X, y, kerns, ops, theta)
    _, rgy, rgx = orc.lml_grad_data(X, y, kerns, ops, theta)
    assert abs(v2 - ref) <= tol * max(abs(ref), 1.0)

    def close(a, b, rtol):
        scale = np.maximum(np.abs(b), 1e-3 * max(np.max(np.abs(b)), 1e-300))
        return np.max(np.abs(a - b) / scale) <= rtol

    # (per-component comparison with a floor of 1e-3 of the largest component: a forward error of c * cond * eps of the
    # LARGEST component shows up as up to 1000 c * cond * eps here.  Seed 9 (cond 6.4e7) measures 0.4-0.7 cond * eps of
    # the largest component = 180-370 cond * eps per component depending on the summation order inside the strip kernel:
    # the factor was 200 until round 4 changed that order.)
    gtol = max(1e-5 if expo else 1e-7, 500.0 * cond * 2.2e-16)
    assert close(g, rg, gtol), (kernel, N, d, g, rg)
    assert close(gy, rgy, gtol), (kernel, N, d)
    assert close(gx, rgx, 10 * gtol), (kernel, N, d)
    Xn = rng.random((M, d))
    mu, var = gp.predict(theta, Xn)
    rmu, rvar = orc.predict(X, y, Xn, kerns, ops, theta)
    ctol = max(1e-8, 200.0 * cond * 2.2e-16)
    assert np.allclose(mu, rmu, rtol=ctol, atol=ctol), (kernel, N, d)
    assert np.allclose(var, rvar, rtol=10 * ctol, atol=max(1e-10, ctol * 1e-2)), (kernel, N, d)
    m2 = min(M, 5)
    pm, pv, dm, dv = gp.predict_grad(theta, Xn[:m2])
    qdm, qdv = orc.predict_grad(X, y, Xn[:m2], kerns, ops, theta)
    assert np.allclose(pm, rmu[:m2], rtol=ctol, atol=ctol) and np.allclose(pv, rvar[:m2], rtol=10 * ctol, atol=max(1e-10, ctol * 1e-2))
    ptol = max(1e-4 if expo else 1e-6, 10 * gtol)
    assert np.allclose(dm, qdm, rtol=ptol, atol=ptol * max(np.abs(qdm).max(), 1e-12)), (kernel, N, d)
    assert np.allclose(dv, qdv, rtol=ptol, atol=ptol * max(np.abs(qdv).max(), 1e-12)), (kernel, N, d)
    gp.close()


def _random_case_wide(rng):
    """The lifted caps of round 3: up to eight components, input dimensions beyond 128."""
    nk = int(rng.integers(1, 9))
    kerns = [NAMES[int(rng.integers(0, 3))] for _ in range(nk)]
    ops = [("+", "*")[int(rng.integers(0, 2))] for _ in range(nk - 1)]
    N = int(rng.choice([int(rng.integers(2, 130)), int(rng.integers(130, 600))]))
    d = int(rng.choice([int(rng.integers(1, 40)), int(rng.integers(100, 160)), int(rng.integers(160, 320))]))
    return N, d, kerns, ops, int(rng.integers(1, 40))


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("SWEEP_SEEDS_WIDE", "12"))))
def test_random_case_many_components_and_dimensions(seed):
    from andvaranaut_amd import MiGP
    from oracle import gp_oracle as orc

    rng = np.random.default_rng(5000 + seed)
    N, d, kerns, ops, M = _random_case_wide(rng)
    kernel = kerns[0] + "".join(o + k for o, k in zip(ops, kerns[1:]))
    X, y = orc.synth_problem(max(N, 3), d, seed=seed)
    X, y = X[:N], y[:N]
    theta = orc.synth_theta(d, nkern=len(kerns), gv=10.0 ** rng.uniform(-4, -2))
    theta[: len(kerns) * d] *= rng.uniform(0.7, 1.6, len(kerns) * d) * np.sqrt(max(d, 2) / 2.0)
    theta[len(kerns) * d: len(kerns) * d + len(kerns)] = rng.uniform(0.5, 1.5, len(kerns))  # products of many kv stay O(1)
    cond = np.linalg.cond(orc.noisy_cov(X, kerns, ops, theta))
    tol = max(1e-10, 20.0 * cond * 2.2e-16)
    gtol = max(1e-7, 200.0 * cond * 2.2e-16)
    gp = MiGP(X, y, kernel)
    val, g, gy, gx = gp.lml_grad_data(theta)
    ref, rg = orc.lml_grad(X, y, kerns, ops, theta)
    _, rgy, rgx = orc.lml_grad_data(X, y, kerns, ops, theta)
    assert abs(val - ref) <= tol * max(abs(ref), 1.0), (kernel, N, d, val, ref)

    def close(a, b, rtol):
        scale = np.maximum(np.abs(b), 1e-3 * max(np.max(np.abs(b)), 1e-300))
        return np.max(np.abs(a - b) / scale) <= rtol

    assert close(g, rg, gtol) and close(gy, rgy, gtol) and close(gx, rgx, 10 * gtol), (kernel, N, d)
    Xn = rng.random((M, d))
    mu, var = gp.predict(theta, Xn)
    rmu, rvar = orc.predict(X, y, Xn, kerns, ops, theta)
    ctol = max(1e-8, 200.0 * cond * 2.2e-16)
    assert np.allclose(mu, rmu, rtol=ctol, atol=ctol) and np.allclose(var, rvar, rtol=10 * ctol, atol=max(1e-10, ctol * 1e-2))
    m2 = min(M, 4)
    _, _, dm, dv = gp.predict_grad(theta, Xn[:m2])
    qdm, qdv = orc.predict_grad(X, y, Xn[:m2], kerns, ops, theta)
    ptol = max(1e-6, 10 * gtol)
    assert np.allclose(dm, qdm, rtol=ptol, atol=ptol * max(np.abs(qdm).max(), 1e-12)), (kernel, N, d)
    assert np.allclose(dv, qdv, rtol=ptol, atol=ptol * max(np.abs(qdv).max(), 1e-12)), (kernel, N, d)
    v2, g2, _, gx2 = gp.lml_grad_data(theta)
    assert v2 == val and np.array_equal(g2, g) and np.array_equal(gx2, gx)
    gp.close()
