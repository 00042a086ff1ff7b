"""Parity of the HIP path (through the C-ABI) with the CPU oracle and the golden fixtures."""
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


# rtol on the LML itself: 1e-10 is the north-star tolerance; Exponential's sqrt(r2+1e-12) at r2 ~ 0
# amplifies 1e-16 rounding of the expansion-form distance, so it gets 1e-8 (CPU-vs-CPU differs too).
CASES = [
    (1, 1, "RBF", 1e-10), (2, 1, "RBF", 1e-10), (100, 2, "RBF", 1e-10), (127, 2, "Matern52", 1e-10),
    (128, 2, "RBF", 1e-10), (129, 3, "Matern32", 1e-10), (300, 3, "Matern52", 1e-10), (1024, 8, "RBF", 1e-10),
    (1000, 4, "Matern32+RBF", 1e-10), (640, 5, "RBF*Exponential", 1e-8), (2048, 4, "RatQuad", 1e-10),
    (515, 33, "RBF", 1e-10), (700, 2, "RBF+Matern52*Matern32", 1e-10), (4096, 8, "RBF", 1e-10),
]


@pytest.mark.parametrize("N,d,kernel,rtol", CASES)
def test_lml_matches_oracle(N, d, kernel, rtol):
    MiGP, orc = _mods()
    X, y = orc.synth_problem(max(N, 3), d, seed=N + d)
    X, y = X[:N], y[:N]
    kerns, ops = _split(kernel)
    theta = orc.synth_theta(d, nkern=len(kerns))
    gp = MiGP(X, y, kernel, need_grad=False)
    val = gp.lml(theta)
    ref = orc.lml(X, y, kerns, ops, theta)
    assert gp.info == 0
    assert abs(val - ref) <= rtol * abs(ref), (val, ref)
    logdet, quad = gp.lml_parts()
    _, L, beta = orc.lml(X, y, kerns, ops, theta, return_parts=True)
    assert abs(logdet - np.log(np.diag(L)).sum()) <= 1e-9 * max(1.0, abs(logdet))
    assert abs(quad - beta @ beta) <= 1e-8 * max(1.0, abs(quad))
    gp.close()


def test_lml_matches_mpmath_golden(mp_cases):
    MiGP, _ = _mods()
    for c in mp_cases:
        X, y = np.array(c["X"]), np.array(c["y"])
        kernel = c["kerns"][0]
        for o, k in zip(c["ops"], c["kerns"][1:]):
            kernel += o + k
        gp = MiGP(X, y, kernel, need_grad=False)
        val = gp.lml(case_theta(c))
        tol = 1e-9 if "Exponential" in c["kerns"] else 1e-10
        assert abs(val - float(c["lml"])) <= tol * abs(float(c["lml"])), c["name"]
        gp.close()


def test_theta_sweep_and_reuse_of_one_handle():
    """A MAP / MCMC loop re-evaluates one handle at many theta (gpmcmc.py:345,351)."""
    MiGP, orc = _mods()
    N, d = 777, 6
    X, y = orc.synth_problem(N, d, seed=11)
    gp = MiGP(X, y, "Matern52", need_grad=False)
    rng = np.random.default_rng(0)
    for _ in range(6):
        theta = orc.pack_theta(np.exp(rng.normal(0, 0.5, d)), [np.exp(rng.normal(0.56, 0.3))], 10 ** rng.uniform(-5, -2), 1e-6)
        val, ref = gp.lml(theta), orc.lml(X, y, ["Matern52"], [], theta)
        assert abs(val - ref) <= 1e-10 * abs(ref)
    gp.close()


def test_non_positive_definite_reports_info_and_minus_inf():
    MiGP, orc = _mods()
    X = np.zeros((200, 2))  # all points coincide: K = kv * ones, rank one
    y = np.ones(200)
    gp = MiGP(X, y, "RBF", need_grad=False)
    theta = orc.pack_theta([[1.0, 1.0]], [1.0], 0.0, -1e-3)
    assert gp.lml(theta) == -np.inf
    assert gp.info > 0
    # and the handle keeps working afterwards
    theta = orc.pack_theta([[1.0, 1.0]], [1.0], 1e-2, 1e-6)
    assert np.isfinite(gp.lml(theta)) and gp.info == 0
    gp.close()


def test_full_size_properties_n16384():
    """BASELINE config 3 shape (Matern-5/2, N=16384, d=16): size-independent checks.
    (a) determinism of two evaluations, (b) LML is invariant under a permutation of the data,
    (c) logdet/quad identities against a blockwise check: the first 2048 points evaluated alone must
    equal the oracle on that sub-problem (prefix of a Cholesky is the Cholesky of the prefix)."""
    MiGP, orc = _mods()
    N, d = 16384, 16
    X, y = orc.synth_problem(N, d, seed=0)
    theta = orc.synth_theta(d)
    gp = MiGP(X, y, "Matern52", need_grad=False)
    v1 = gp.lml(theta)
    v2 = gp.lml(theta)
    assert v1 == v2 and np.isfinite(v1)
    perm = np.random.default_rng(1).permutation(N)
    gp2 = MiGP(X[perm], y[perm], "Matern52", need_grad=False)
    v3 = gp2.lml(theta)
    assert abs(v3 - v1) <= 1e-9 * abs(v1), (v1, v3)
    gp2.close()
    # the factor's leading 2048 x 2048 block is the factor of the 2048-point problem
    Lgpu = gp.K_t[:2048, :2048].cpu().numpy()
    _, Lref, _ = orc.lml(X[:2048], y[:2048], ["Matern52"], [], theta, return_parts=True)
    assert np.allclose(np.tril(Lgpu), Lref, rtol=0, atol=1e-9)
    gp.close()


def test_config4_shape_fits_one_gpu():
    """BASELINE config 4's shape (RBF, N=65536, d=32): K is 34 GB and runs on one MI355X.
    Size-independent checks: finite, positive definite, bit-identical on re-evaluation, and the
    quadratic form / log-det split is consistent with the returned LML."""
    MiGP, _ = _mods()
    from bench import synth_problem

    N, d = 65536, 32
    X, y = synth_problem(N, d, seed=0)
    theta = np.concatenate([np.exp(np.linspace(np.log(0.8), np.log(3.0), d)), [1.7], [1.0], [1e-4, 1e-6]])
    gp = MiGP(X, y, "RBF", need_grad=True)  # K, U = L^-T and K^-1: 3 x 34 GB of the 288 GB
    v1 = gp.lml(theta)
    assert gp.info == 0 and np.isfinite(v1)
    logdet, quad = gp.lml_parts()
    assert abs(v1 - (-0.5 * N * np.log(2 * np.pi) - 0.5 * quad - logdet)) <= 1e-12 * abs(v1)
    # ORACLE-TIED at the full size: 256 random rows of L L^T, formed on the device in fp64 from the
    # factor the library left in K_t, against the same rows of the oracle's noisy covariance assembled on the host
    # (O(256 N d)); bound 64 N eps of the largest entry (observed: ~1e-13)
    import torch
    from oracle import gp_oracle as orc

    rows = np.sort(np.random.default_rng(7).choice(N, 256, replace=False))
    Kref = orc.kernel_matrix(X[rows], X, ["RBF"], [], theta)
    sg = np.sqrt(theta[-2])
    Kref[np.arange(256), rows] += sg * sg
    Kref[np.arange(256), rows] += theta[-1]
    with torch.cuda.device(gp.dev):
        rt = torch.from_numpy(rows).to(gp.dev)
        cols = torch.arange(N, device=gp.dev)
        Lr = gp.K_t[rt, :N] * (cols[None, :] <= rt[:, None])  # the rows' lower-triangular part (the strict upper triangle of K_t is never written)
        llt = torch.empty((256, N), dtype=torch.float64, device=gp.dev)
        for j0 in range(0, N, 8192):  # tril of 8192 rows of L at a time (4 GB), not of the whole 34 GB factor
            Lj = gp.K_t[j0:j0 + 8192, :N] * (cols[None, :] <= (j0 + torch.arange(8192, device=gp.dev))[:, None])
            llt[:, j0:j0 + 8192] = Lr @ Lj.T
            del Lj
        res = float((llt - torch.from_numpy(Kref).to(gp.dev)).abs().max())
        del llt, Lr
    assert res <= 64 * N * np.finfo(float).eps * np.abs(Kref).max(), res
    assert gp.lml(theta) == v1
    # SURVEY 8e "gradient at C4 scale": no distributed inverse is needed, the whole gradient path fits one GPU.
    # Size-independent check: the directional derivative along the gradient against a central difference of the LML.
    v2, g = gp.lml_grad(theta)
    assert abs(v2 - v1) <= 1e-12 * abs(v1) and np.all(np.isfinite(g))
    # ... and the oracle's rows times the device's alpha = K^-1 y reproduce y (residual relative to |K| |alpha| + |y|)
    _, _, neg_alpha, _ = gp.lml_grad_data(theta, want_x=False)
    alpha = -neg_alpha
    resid = np.abs(Kref @ alpha - y[rows])
    scale = np.abs(Kref) @ np.abs(alpha) + np.abs(y[rows])
    assert (resid / scale).max() <= 64 * N * np.finfo(float).eps, (resid / scale).max()
    u = g / np.linalg.norm(g)
    h = 1e-4
    tp, tm = theta.copy(), theta.copy()
    tp[:-1] += h * u[:-1] * theta[:-1]
    tm[:-1] -= h * u[:-1] * theta[:-1]
    fd = (gp.lml(tp) - gp.lml(tm)) / (2 * h)
    an = float(np.dot(g[:-1], u[:-1] * theta[:-1]))
    assert abs(fd - an) <= 1e-6 * abs(an), (fd, an)
    gp.close()


def test_four_component_kernel_and_many_dims():
    """Four components (the specialised gradient kernels' limit) and d > 64 (three LDS chunks of the input dimension)."""
    MiGP, orc = _mods()
    N, d = 400, 70
    X, y = orc.synth_problem(N, d, seed=5)
    kernel = "RBF+Matern52*Matern32+RatQuad"
    kerns, ops = ["RBF", "Matern52", "Matern32", "RatQuad"], ["+", "*", "+"]
    theta = orc.synth_theta(d, nkern=4, gv=1e-3)
    theta[:4 * d] *= 6.0  # length scales ~ sqrt(d) so the kernel is not numerically the identity
    theta[4 * d + 4: 4 * d + 8] = [1.0, 1.0, 1.0, 2.2]  # RatQuad alpha
    gp = MiGP(X, y, kernel)
    val, g = gp.lml_grad(theta)
    ref, gref = orc.lml_grad(X, y, kerns, ops, theta)
    assert abs(val - ref) <= 1e-10 * abs(ref), (val, ref)
    assert np.abs(g - gref).max() <= 1e-8 * np.abs(gref).max()
    Xn = np.random.default_rng(1).random((33, d))
    mu, var = gp.predict(theta, Xn)
    rmu, rvar = orc.predict(X, y, Xn, kerns, ops, theta)
    assert np.allclose(mu, rmu, rtol=1e-9, atol=1e-9) and np.allclose(var, rvar, rtol=1e-8, atol=1e-11)
    gp.close()


@pytest.mark.parametrize("kernel", ["RBF+Matern52*Matern32+RBF+Matern52", "Matern32*RBF+Matern52+RBF*Matern32+Matern52",
                                    "RBF+Matern52+Matern32+RBF*Matern52+Matern32+RBF+Matern52"])
def test_five_to_eight_component_kernels(kernel):
    """The reference folds any number of components (gpmcmc.py:282-307); MI_GP_MAX_KERN is 8 since round 3 (the gradient
    kernels take 5..8 components through one instantiation with a run-time count).  Every entry point against the oracle."""
    MiGP, orc = _mods()
    N, d = 333, 3
    X, y = orc.synth_problem(N, d, seed=9)
    kerns, ops = _split(kernel)
    nk = len(kerns)
    assert 5 <= nk <= 8
    theta = orc.synth_theta(d, nkern=nk, gv=1e-3)
    theta[: nk * d] *= np.random.default_rng(nk).uniform(0.8, 1.5, nk * d)
    theta[nk * d: nk * d + nk] = np.random.default_rng(nk + 1).uniform(0.6, 1.4, nk)
    gp = MiGP(X, y, kernel)
    val, g, gy, gX = gp.lml_grad_data(theta)
    ref, gref = orc.lml_grad(X, y, kerns, ops, theta)
    _, gy_ref, gX_ref = orc.lml_grad_data(X, y, kerns, ops, theta)
    assert abs(val - ref) <= 1e-10 * abs(ref), (val, ref)
    assert np.abs(g - gref).max() <= 1e-8 * np.abs(gref).max()
    assert np.abs(gy - gy_ref).max() <= 1e-8 * np.abs(gy_ref).max()
    assert np.abs(gX - gX_ref).max() <= 1e-8 * np.abs(gX_ref).max()
    Xn = np.random.default_rng(1).random((7, d))
    mu, var, dmu, dvar = gp.predict_grad(theta, Xn)
    rmu, rvar = orc.predict(X, y, Xn, kerns, ops, theta)
    dmu_o, dvar_o = orc.predict_grad(X, y, Xn, kerns, ops, theta)
    assert np.allclose(mu, rmu, rtol=1e-9, atol=1e-9) and np.allclose(var, rvar, rtol=1e-8, atol=1e-11)
    assert np.abs(dmu - dmu_o).max() <= 1e-7 * np.abs(dmu_o).max()
    assert np.abs(dvar - dvar_o).max() <= 1e-7 * np.abs(dvar_o).max()
    v2, g2, _, gX2 = gp.lml_grad_data(theta)
    assert v2 == val and np.array_equal(g2, g) and np.array_equal(gX2, gX)
    gp.close()
    with pytest.raises(Exception):
        MiGP(X, y, "+".join(["RBF"] * 9))


def test_handles_can_be_created_and_destroyed_repeatedly():
    MiGP, orc = _mods()
    import torch

    X, y = orc.synth_problem(600, 3, seed=1)
    theta = orc.synth_theta(3)
    ref = orc.lml(X, y, ["RBF"], [], theta)
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    for _ in range(20):
        gp = MiGP(X, y, "RBF")
        assert abs(gp.lml(theta) - ref) <= 1e-10 * abs(ref)
        gp.close()
        del gp
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() <= base + (1 << 20)


def test_tuning_options_do_not_change_results():
    MiGP, orc = _mods()
    X, y = orc.synth_problem(3000, 6, seed=2)
    theta = orc.synth_theta(6)
    gp = MiGP(X, y, "Matern52", need_grad=False)
    ref0 = gp.lml(theta)
    gp.set_option(2, 8)  # pin the super-panel width: by default it also depends on whether look-ahead is active (option 0)
    ref = gp.lml(theta)
    assert abs(ref - ref0) <= 1e-11 * abs(ref0)
    for what, value in [(0, 0), (0, 2), (8, 1 << 20), (8, 0), (14, 0), (14, 4), (16, 0), (16, 1), (7, 0), (7, 100000), (9, 0), (9, 1), (2, 2), (2, 4), (2, 8)]:
        gp.set_option(what, value)
        v = gp.lml(theta)
        if what in (2, 7, 9):  # the super-panel width regroups the k-sums of the updates, the tile size their MFMA order
            assert abs(v - ref) <= 1e-11 * abs(ref), (what, value, v, ref)
        else:               # pure scheduling knobs: same arithmetic in the same order per tile, bit-identical
            assert v == ref, (what, value, v, ref)
        assert gp.lml(theta) == v  # and again
    gp.set_option(2, 0)  # default width again: forced look-ahead narrows the super-panels of a problem this small
    gp.set_option(0, 2)
    assert abs(gp.lml(theta) - ref) <= 1e-11 * abs(ref)
    for gone in (1, 3, 10, 11, 12, 13, 15):  # round-1 experiments: GEMM variants, graph replay, persistent bulk, exclusive leaf, fused leaf + strip
        with pytest.raises(RuntimeError):
            gp.set_option(gone, 0)
    gp.close()


@pytest.mark.parametrize("N", [1000, 3000, 5200])
def test_rows_below_the_y_row_stay_zero(N):
    """K_dev's last tile row carries y^T in row Np and zeros below; every trapezoid update of the
    factorisation includes it.  The 64x64-tile kernel skips the lower, all-zero half of that tile row (round 6,
    GemmParams::dead_last_half): those rows must be exactly zero after an evaluation, the row of beta = L^-1 y finite, and
    the LML the oracle's (gpmcmc.py:313-318)."""
    import torch
    MiGP, orc = _mods()
    d = 5
    X, y = orc.synth_problem(N, d, seed=N)
    theta = orc.synth_theta(d)
    gp = MiGP(X, y, "RBF", need_grad=False)
    v = gp.lml(theta)
    ref = orc.lml(X, y, ["RBF"], [], theta)
    assert abs(v - ref) <= 1e-10 * abs(ref)
    npad = gp.np_
    torch.cuda.synchronize()
    below = gp.K_t[npad + 1:npad + 128, :npad]
    assert int(torch.count_nonzero(below).item()) == 0
    beta = gp.K_t[npad, :N]
    assert bool(torch.isfinite(beta).all().item()) and float(beta.abs().max().item()) > 0.0
    gp.close()


def test_shared_lane_schedule_is_bit_identical_to_the_default():
    """GPMCMC.fit runs chains that share a GPU on single-stream handles between 20 and 64 tile columns, with the
    super-panel width the two-stream default would pick pinned (options 2 = 4, 0 = 0): same arithmetic, same bits --
    every draw of a chain equals the default schedule's."""
    MiGP, orc = _mods()
    X, y = orc.synth_problem(5000, 5, seed=4)  # 40 tile columns
    theta = orc.synth_theta(5)
    gp = MiGP(X, y, "Matern52")
    v0, g0 = gp.lml_grad(theta)
    gp.set_option(2, 4)
    gp.set_option(0, 0)
    v1, g1 = gp.lml_grad(theta)
    assert v1 == v0 and np.array_equal(g1, g0)
    assert gp.lml(theta) == v0
    gp.set_option(2, 0)
    gp.set_option(0, 1)
    assert gp.lml(theta) == v0
    gp.close()


def test_options_are_per_handle():
    """Two handles with different launcher options in one process (fit(method='mcmc_*') drives one handle per GPU from
    one thread each): neither sees the other's knobs, both return the same bits."""
    MiGP, orc = _mods()
    X, y = orc.synth_problem(2500, 4, seed=9)
    theta = orc.synth_theta(4)
    a = MiGP(X, y, "RBF", need_grad=False)
    b = MiGP(X, y, "RBF", need_grad=False)
    b.set_option(7, 0)    # b: never the 64x64-tile kernel
    b.set_option(14, 0)   # b: row-major tile order
    va, vb = a.lml(theta), b.lml(theta)
    assert va == vb and a.lml(theta) == va and b.lml(theta) == vb
    a.close()
    b.close()


def test_handles_of_several_sizes_in_one_process():
    """Round-2 regression: with hipGraph replay hipGraphLaunch crashed intermittently (hip::Graph::UpdateStreams,
    profiles/r02_hipgraph_updatestreams_segv.txt) when handles of several sizes came and went in one process; replay is
    gone, the scenario stays (tools/stress_handles.py is the long form)."""
    MiGP, orc = _mods()
    data = {N: orc.synth_problem(N, 6, seed=N) for N in (1024, 2048, 4096)}
    keep = MiGP(*data[1024], "RBF")  # stays alive across the other handles' lifetimes
    theta = orc.synth_theta(6)
    ref = keep.lml(theta)
    for it in range(3):
        for N in (2048, 4096, 1024):
            gp = MiGP(*data[N], "RBF")
            v0 = gp.lml(theta)
            v1, g = gp.lml_grad(theta)
            assert v1 == v0 and np.isfinite(v0) and np.all(np.isfinite(g))
            assert gp.lml(theta) == v0
            gp.close()
            assert keep.lml(theta) == ref
    keep.close()
