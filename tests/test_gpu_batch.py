"""Batched evaluation (mi_gp_set_batch / mi_gp_lml_batch / mi_gp_lml_grad_batch): K covariances of the same inputs, one
theta each, factorised in lockstep with blockIdx.z = problem -- what the reference's MAP restarts (gpmcmc.py:328-343) and
the chains of pm.sample (gpmcmc.py:351) evaluate side by side.  Parity with the oracle for K = 1, 3, 8, bit-equality with
the one-at-a-time entry points, a non-positive-definite member, sizes on both sides of the two-stream threshold."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _split(kernel):
    return kernel.replace("*", "+").split("+"), [c for c in kernel if c in "+*"]


def _thetas(orc, d, nk, k, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(k):
        th = orc.synth_theta(d, nkern=nk, gv=10.0 ** rng.uniform(-4.5, -2.5))
        th[: nk * d] *= rng.uniform(0.7, 1.5, nk * d)
        th[nk * d: nk * d + nk] *= rng.uniform(0.8, 1.3, nk)
        out.append(th)
    return np.array(out)


@pytest.mark.parametrize("N,d,kernel,K", [(300, 3, "Matern52", 1), (700, 5, "RBF", 3), (1500, 8, "RBF", 8),
                                          (1000, 2, "RBF+Matern32", 3), (2100, 4, "Matern52*RBF", 8),
                                          (5000, 6, "Matern52", 3)])
def test_batch_matches_oracle_and_the_single_entry_points(N, d, kernel, K):
    import torch

    assert torch.cuda.is_available()
    from andvaranaut_amd import MiGP
    from oracle import gp_oracle as orc

    X, y = orc.synth_problem(N, d, seed=N + K)
    kerns, ops = _split(kernel)
    th = _thetas(orc, d, len(kerns), K, seed=N)
    gp = MiGP(X, y, kernel)
    vals = gp.lml_batch(th)
    v2, grads = gp.lml_grad_batch(th)
    assert np.array_equal(vals, v2)
    for p in range(K):
        ref, gref = orc.lml_grad(X, y, kerns, ops, th[p])
        assert abs(vals[p] - ref) <= 1e-10 * abs(ref), (p, vals[p], ref)
        scale = np.maximum(np.abs(gref), 1e-3 * np.abs(gref).max())
        assert np.max(np.abs(grads[p] - gref) / scale) <= 1e-7, (p, grads[p], gref)
        one, gone = gp.lml_grad(th[p])
        assert one == vals[p] and np.array_equal(gone, grads[p]), (p, one, vals[p])  # same arithmetic per element
    assert np.array_equal(gp.lml_batch(th), vals)  # and again after the single evaluations
    gp.close()


def test_batch_with_a_non_positive_definite_member():
    from andvaranaut_amd import MiGP
    from oracle import gp_oracle as orc

    N, d = 900, 3
    X, y = orc.synth_problem(N, d, seed=5)
    th = _thetas(orc, d, 1, 4, seed=9)
    th[2, -1] = -10.0  # negative jitter: not positive definite
    gp = MiGP(X, y, "Matern52")
    vals, grads = gp.lml_grad_batch(th)
    assert vals[2] == -np.inf and np.all(grads[2] == 0.0) and gp.batch_info[2] > 0
    for p in (0, 1, 3):
        ref = orc.lml(X, y, ["Matern52"], [], th[p])
        assert abs(vals[p] - ref) <= 1e-10 * abs(ref)
        assert gp.batch_info[p] == 0
    gp.close()
