"""The north star's literal parity sentence, tested directly: "LML matching CPU reference to rtol 1e-10 at N=16384"
(BASELINE.json; formula gpmcmc.py:311-318 == MvNormal.logp of Marginal.marginal_likelihood, gpmcmc.py:321-323), and the
gradient the MAP / NUTS drivers consume (gpmcmc.py:345,351) against the oracle's analytic gradient at config 5's size.
The oracle costs ~10 s (N=16384 LML) and ~20 s (N=8192 LML + gradient) on a one-GPU box's 16 host cores."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mods():
    import torch

    assert torch.cuda.is_available()
    from andvaranaut_amd import MiGP
    from oracle import gp_oracle as orc

    try:  # OpenBLAS with more threads than the cgroup quota allows is several times slower
        import threadpoolctl

        from bench import host_cores

        threadpoolctl.threadpool_limits(limits=min(host_cores(), 16))
    except Exception:
        pass
    return MiGP, orc


def test_headline_lml_n16384_matches_oracle_rtol_1e10():
    """BASELINE config 3 / the bench workload: Matern-5/2, N=16384, d=16, fp64, rtol 1e-10 on the LML itself; the
    log-determinant and the quadratic form are checked separately so that a cancellation between them cannot hide."""
    MiGP, orc = _mods()
    N, d = 16384, 16
    X, y = orc.synth_problem(N, d, seed=0)
    theta = orc.synth_theta(d)
    gp = MiGP(X, y, "Matern52", need_grad=False)
    val = gp.lml(theta)
    logdet, quad = gp.lml_parts()
    gp.close()
    ref, L, beta = orc.lml(X, y, ["Matern52"], [], theta, return_parts=True)
    rel = abs(val - ref) / abs(ref)
    print(f"N=16384 Matern52: device {val!r} oracle {ref!r} rel {rel:.3e}")
    assert rel <= 1e-10, (val, ref, rel)
    ref_logdet = float(np.log(np.diag(L)).sum())
    ref_quad = float(beta @ beta)
    assert abs(logdet - ref_logdet) <= 1e-10 * abs(ref_logdet), (logdet, ref_logdet)
    assert abs(quad - ref_quad) <= 1e-9 * abs(ref_quad), (quad, ref_quad)


def test_lml_grad_n8192_matches_oracle():
    """Config 5's per-chain evaluation (RBF, N=8192, d=8): every component of the analytic gradient within 1e-8 of the
    largest one, the LML within 1e-10."""
    MiGP, orc = _mods()
    N, d = 8192, 8
    X, y = orc.synth_problem(N, d, seed=1)
    theta = orc.synth_theta(d)
    gp = MiGP(X, y, "RBF")
    val, g = gp.lml_grad(theta)
    gp.close()
    ref, gref = orc.lml_grad(X, y, ["RBF"], [], theta)
    assert abs(val - ref) <= 1e-10 * abs(ref), (val, ref)
    err = np.abs(g - gref).max() / np.abs(gref).max()
    print(f"N=8192 RBF gradient: max component error {err:.3e} of the largest component")
    assert err <= 1e-8, (g, gref)


def test_predict_at_n8192_through_both_routes_matches_the_oracle():
    """K8 (gpmcmc.py:588-598, 766-778) above the sizes of tests/test_gpu_grad_predict.py: posterior mean and variance at
    N = 8192, M = 512 through the blocked triangular solve (mi_gp_predict) AND through U = L^-T with one triangular-k GEMM
    (mi_gp_predict_u, the route large BO sweeps take) against oracle.predict.  N = 16384, M = 1000:
    profiles/r04_fullsize_parity.json (tools/fullsize_parity.py predict)."""
    from andvaranaut_amd import MiGP
    from bench import reference_theta, synth_problem
    from oracle import gp_oracle as orc

    N, d, M = 8192, 8, 512
    X, y = synth_problem(N, d, seed=0)
    theta = reference_theta(d)
    Xn = np.random.default_rng(3).random((M, d))
    rmu, rvar = orc.predict(X, y, Xn, ["RBF"], [], theta)
    gp = MiGP(X, y, "RBF")
    for via in (False, True):
        mu, var = gp.predict(theta, Xn, via_inverse=via)
        assert np.max(np.abs(mu - rmu)) <= 1e-8 * np.abs(rmu).max(), (via, np.max(np.abs(mu - rmu)))
        assert np.max(np.abs(var - rvar) / rvar) <= 1e-6, (via, np.max(np.abs(var - rvar) / rvar))
    gp.close()


def test_ragged_large_size_through_the_split_and_merged_launches():
    """The round-4 driver paths that only large problems reach (api_gp.hip cholesky(): merged head from 72 trailing tile
    columns, split bulk update, single-stream tail) at a size that is NOT a multiple of anything: N = 10300 (81 tile columns,
    ragged last super-panel).  LML against the oracle at the north star's 1e-10, and the scheduling-only options
    (include/mi_gp.h: 18, 19, 21, 24, 26 bit-identical; 20 same k order per tile) must not change a bit.  gpmcmc.py:313."""
    from andvaranaut_amd import MiGP
    from bench import reference_theta, synth_problem
    from oracle import gp_oracle as orc

    N, d = 10300, 4
    X, y = synth_problem(N, d, seed=2)
    theta = reference_theta(d)
    ref = orc.lml(X, y, ["Matern52"], [], theta)
    gp = MiGP(X, y, "Matern52", need_grad=False)
    v0 = gp.lml(theta)
    assert abs(v0 - ref) <= 1e-10 * abs(ref), (v0, ref)
    for opts in ({18: 0}, {20: 0}, {21: 0}, {18: 0, 20: 0, 21: 0}, {18: 1024, 19: 256, 20: 40}, {26: 0}, {26: 1}, {26: 0, 21: 16}):
        for k, v in opts.items():
            gp.set_option(k, v)
        assert gp.lml(theta) == v0, opts
        for k, v in {18: 1536, 19: 1024, 20: 72, 21: 8, 26: 2}.items():
            gp.set_option(k, v)
    gp.close()
