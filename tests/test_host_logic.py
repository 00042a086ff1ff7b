"""Host-side logic (priors, transforms, MAP driver, NUTS, facade plumbing) on CPU.
The device callable is replaced by the oracle here -- tests are the one place allowed to do that."""
import numpy as np
import pytest
import scipy.stats as st

from conftest import ROOT, load_json
from oracle import gp_oracle as orc


def test_prior_logps_match_scipy():
    from andvaranaut_amd.priors import HalfNormal, LogNormal, TruncatedNormal

    x = np.array([0.03, 0.5, 1.7, 9.0])
    assert np.allclose(LogNormal(0.56, 0.75).logp(x), st.lognorm(s=0.75, scale=np.exp(0.56)).logpdf(x))
    assert np.allclose(HalfNormal(1e-3).logp(x * 1e-3), st.halfnorm(scale=1e-3).logpdf(x * 1e-3))
    a, b = (1e-3 - 0.5) / 0.15, (100.0 - 0.5) / 0.15
    assert np.allclose(TruncatedNormal(0.5, 0.15, 1e-3, 100.0).logp(x), st.truncnorm(a, b, loc=0.5, scale=0.15).logpdf(x))
    # [3P] PyMC moments used as find_MAP's start
    assert np.isclose(LogNormal(0.0, 1.0).moment(), np.exp(0.5))
    assert HalfNormal(1e-3).moment() == 1e-3
    assert TruncatedNormal(1.0, 0.15, 0.1, 100.0).moment() == 50.05


def test_transforms_roundtrip_and_jacobians():
    from andvaranaut_amd.priors import LogNormal, TruncatedNormal, backward, forward

    for dist in (LogNormal(0, 1), TruncatedNormal(0.5, 0.15, 1e-3, 100.0)):
        q = np.array([-2.0, -0.3, 0.0, 1.1, 3.0])
        x, dx, lj, dlj = backward(dist, q)
        assert np.allclose(forward(dist, x), q)
        h = 1e-6
        xp, _, ljp, _ = backward(dist, q + h)
        xm, _, ljm, _ = backward(dist, q - h)
        assert np.allclose((xp - xm) / (2 * h), dx, rtol=1e-6)
        assert np.allclose(lj, np.log(dx))
        assert np.allclose((ljp - ljm) / (2 * h), dlj, rtol=1e-5, atol=1e-8)


def test_tutorial_conversion_pins():
    """tutorial/tutorial.ipynb:366 records uniform / normal conversions of one LHS sample."""
    from andvaranaut_amd.transform import normal, uniform

    pins = load_json("tutorial_pins.json")
    xc0 = uniform(st.uniform(loc=0, scale=2)).con(np.array([pins["x"][0]]))[0]
    xc1 = normal(st.uniform(loc=1, scale=0.5)).con(np.array([pins["x"][1]]))[0]
    # the notebook prints 8 decimals of x; dividing by std = 0.144 amplifies that rounding
    assert abs(xc0 - pins["xc"][0]) < 6e-9 and abs(xc1 - pins["xc"][1]) < 5e-8
    u = uniform(st.uniform(loc=0, scale=2))
    assert np.allclose(u.rev(u.con(np.array([0.3, 1.9]))), [0.3, 1.9])


def _tutorial_target(x):
    """The tutorial's example target (tutorial.ipynb cell 5): 2 inputs, 1 output."""
    x1, x2 = x
    return np.array([x1 ** 2 - x1 - x2 ** 2 * x1 + x2])


def test_tutorial_pins_both_rows_through_the_facade():
    """Every number tutorial.ipynb cell 20 records: set_data on the facade converts x with the same uniform / normal
    conrevs (gpmcmc.py:235-237) and leaves y unconverted (yconrevs=[None], gpmcmc.py:279); the recorded y are the
    target's values at the recorded x (8 printed decimals -> 2e-8)."""
    from andvaranaut_amd import GPMCMC
    from andvaranaut_amd.transform import normal, uniform

    pins = load_json("tutorial_pins.json")
    c20 = pins["cell20"]
    space = [st.uniform(loc=0, scale=2), st.uniform(loc=1, scale=0.5)]
    g = GPMCMC(kernel="RBF", noise=False, xconrevs=[uniform(space[0]), normal(space[1])], yconrevs=[None], nx=2, ny=1,
               priors=space, target=_tutorial_target, parallel=False, nproc=1, verbose=False)
    x, y = np.array(c20["x"]), np.array(c20["y"])
    g.set_data(x, y)
    assert np.abs(g.xc - np.array(c20["xc"])).max() < 5e-8
    assert np.abs(g.yc - np.array(c20["yc"])).max() == 0.0
    ytar = np.array([_tutorial_target(r) for r in x])
    assert np.abs(ytar - y).max() < 2e-8


def test_tutorial_target_pairs():
    """tutorial.ipynb cells 8 / 10: eight (x, y) pairs the reference's LHC.sample recorded for its example target;
    GPMCMC.sample evaluates the target the same way (core.py:109-114 serial path)."""
    pins = load_json("tutorial_pins.json")["test_fun_pairs"]
    x, y = np.array(pins["x"]), np.array(pins["y"])
    got = np.array([_tutorial_target(r) for r in x])
    assert np.abs(got - y).max() < 2e-8
    from andvaranaut_amd import GPMCMC

    space = [st.uniform(loc=0, scale=2), st.uniform(loc=1, scale=0.5)]
    g = GPMCMC(kernel="RBF", noise=False, nx=2, ny=1, priors=space, target=_tutorial_target, parallel=False, nproc=1,
               verbose=False)
    g.sample(4, seed=0)
    assert g.x.shape == (4, 2) and np.abs(g.y - np.array([_tutorial_target(r) for r in g.x])).max() == 0.0
    assert np.all((g.x[:, 0] >= 0) & (g.x[:, 0] <= 2) & (g.x[:, 1] >= 1) & (g.x[:, 1] <= 1.5))


def _oracle_callable(X, y, kerns, ops):
    return lambda theta: orc.lml_grad(X, y, kerns, ops, theta)


@pytest.mark.parametrize("noise,truncate,kernel", [(True, False, "Matern52"), (False, False, "RBF"),
                                                   (True, True, "RBF+Matern32"), (True, False, "RatQuad")])
def test_joint_logp_gradient_by_finite_differences(noise, truncate, kernel):
    from andvaranaut_amd.backend import parse_kernel
    from andvaranaut_amd.priors import HyperModel

    N, d = 40, 2
    X, y = orc.synth_problem(N, d, seed=2)
    kerns, ops = parse_kernel(kernel)
    model = HyperModel(d, kerns, noise=noise, truncate=truncate, jitter=1e-6)
    f = lambda q: model.logp_dlogp(q, _oracle_callable(X, y, kerns, ops))  # noqa: E731
    q = model.initial_point() + 0.1 * np.random.default_rng(0).standard_normal(model.nq)
    v, g = f(q)
    assert np.isfinite(v)
    for i in range(model.nq):
        h = 1e-6
        qp, qm = q.copy(), q.copy()
        qp[i] += h
        qm[i] -= h
        fd = (f(qp)[0] - f(qm)[0]) / (2 * h)
        # without a noise term K + 1e-6 I has cond ~1e10 and the finite difference itself is only good to ~1e-4
        tol = 2e-5 if noise else 3e-4
        assert abs(fd - g[i]) <= tol * max(1.0, abs(fd)), (i, fd, g[i])
    pt = model.point_dict(q)
    assert "l" in pt and "kv" in pt and ("gv" in pt) == noise
    assert ("l_interval__" in pt) == truncate and ("l_log__" in pt) == (not truncate)
    assert np.allclose(model.q_from_point(pt), q)


def test_find_map_on_the_tutorial_problem():
    """BASELINE config 1 plumbing: tutorial function, N=100, d=2, RBF, noise=False, MAP
    (tutorial/tutorial.ipynb:61-68, 488, 566-569) with the oracle as the likelihood."""
    from andvaranaut_amd.lhc import latin_sample
    from andvaranaut_amd.optimize import find_MAP
    from andvaranaut_amd.priors import HyperModel
    from andvaranaut_amd.transform import normal, uniform

    priors = [st.uniform(loc=0, scale=2), st.uniform(loc=1, scale=0.5)]
    x = latin_sample(priors, 100, seed=1)
    y = x[:, 0] ** 2 - x[:, 0] - x[:, 1] ** 2 * x[:, 0] + x[:, 1]
    xin = np.column_stack([uniform(priors[0]).con(x[:, 0]), normal(priors[1]).con(x[:, 1])])
    model = HyperModel(2, ["RBF"], noise=False, jitter=1e-6)
    tr, te = np.arange(90), np.arange(90, 100)
    f = lambda q: model.logp_dlogp(q, _oracle_callable(xin[tr], y[tr], ["RBF"], []))  # noqa: E731
    q, info = find_MAP(f, model.initial_point())
    assert info["nfev"] < 200 and info["logp"] > f(model.initial_point())[0] + 100
    assert np.linalg.norm(f(q)[1]) < 1e-2 * max(1.0, abs(info["logp"]))
    pt = model.point_dict(q)
    assert set(pt) == {"l_log__", "l", "kv_log__", "kv"}  # keys recorded at tutorial.ipynb:529
    mu, _ = orc.predict(xin[tr], y[tr], xin[te], ["RBF"], [], model.theta(model.constrain(q)))
    rmse = np.sqrt(np.mean((mu - y[te]) ** 2))
    assert rmse < 2e-3, rmse  # the notebook records 1.4e-4 on its own unseeded sample


def test_nuts_recovers_a_correlated_gaussian():
    from andvaranaut_amd.nuts import sample_chain

    mu = np.array([1.0, -2.0, 0.5])
    A = np.array([[1.0, 0.6, 0.0], [0.6, 2.0, -0.4], [0.0, -0.4, 0.5]])
    P = np.linalg.inv(A)

    def f(q):
        d = q - mu
        return -0.5 * d @ P @ d, -P @ d

    r = sample_chain(f, np.zeros(3), draws=1500, tune=600, seed=3)
    qs = r["q"]
    assert r["diverging"] == 0
    assert np.allclose(qs.mean(0), mu, atol=0.15)
    assert np.allclose(np.cov(qs.T), A, atol=0.35)
    assert np.allclose(r["lp"], [f(q)[0] for q in qs])


def test_nuts_on_a_small_gp_posterior_and_extracts():
    """mcmc_mean / mcmc_map extraction semantics (gpmcmc.py:404-430) on oracle-backed chains."""
    from andvaranaut_amd.nuts import Trace, sample_chain
    from andvaranaut_amd.optimize import find_MAP
    from andvaranaut_amd.priors import HyperModel

    N, d = 30, 1
    X, y = orc.synth_problem(N, d, seed=4)
    model = HyperModel(d, ["RBF"], noise=True, jitter=1e-6)
    f = lambda q: model.logp_dlogp(q, _oracle_callable(X, y, ["RBF"], []))  # noqa: E731
    chains = [sample_chain(f, model.initial_point(), draws=150, tune=150, seed=s) for s in (1, 2)]
    qs = np.stack([c["q"] for c in chains])
    post = {}
    for c in range(2):
        for k in range(qs.shape[1]):
            for name, val in model.point_dict(qs[c, k]).items():
                post.setdefault(name, np.empty((2, qs.shape[1]) + np.shape(val)))[c, k] = val
    data = Trace(post, {"lp": np.stack([c["lp"] for c in chains])})
    qmap, info = find_MAP(f, model.initial_point())
    # the best draw is close to (and not above) the optimum, the posterior mean is in its basin
    assert data.sample_stats["lp"].max() <= info["logp"] + 1e-6
    assert data.sample_stats["lp"].max() > info["logp"] - 6.0
    assert abs(np.log(post["kv"]).mean() - qmap[model.nq - 1]) < 1.5


def test_facade_validates_arguments_and_fails_loudly_without_a_gpu():
    import torch

    from andvaranaut_amd import GPMCMC
    from andvaranaut_amd.transform import normal, uniform

    priors = [st.uniform(loc=0, scale=2), st.uniform(loc=1, scale=0.5)]
    fun = lambda x: np.array([x[0] ** 2 - x[0] - x[1] ** 2 * x[0] + x[1]])  # noqa: E731
    with pytest.raises(Exception):
        GPMCMC(nx=0, ny=1, priors=priors, target=fun)
    with pytest.raises(Exception):
        GPMCMC(nx=2, ny=1, priors=priors, target=fun, kernel="RBF+Foo")
    g = GPMCMC(kernel="Matern52+RBF", noise=False, xconrevs=[uniform(priors[0]), normal(priors[1])], yconrevs=[None],
               nx=2, ny=1, priors=priors, target=fun, verbose=False)
    assert g.kerns == ["Matern52", "RBF"] and g.ops == ["+"] and g.nkern == 2 and g.hypers is None
    g.sample(12, seed=0)
    assert g.x.shape == (12, 2) and g.y.shape == (12, 1) and g.xc.shape == (12, 2)
    assert np.allclose(g.xc[:, 0], uniform(priors[0]).con(g.x[:, 0])) and np.allclose(g.yc, g.y)
    x2 = np.random.default_rng(0).uniform([0, 1], [2, 1.5], (5, 2))
    g.set_data(x2, np.array([fun(r) for r in x2]))
    assert g.nsamp == 5
    with pytest.raises(Exception):
        g.set_data(x2 + 10.0, np.zeros((5, 1)))
    g.del_samples(idx=[0, 2], method="specific")
    assert g.nsamp == 3 and g.x.shape == g.xc.shape == (3, 2) and g.y.shape == g.yc.shape == g.ym.shape == (3, 1)
    assert np.allclose(g.x, x2[[1, 3, 4]])
    g.del_samples(1, method="random")
    g.del_samples(1)  # nearest to a fresh latin-hypercube point
    assert g.nsamp == 1
    with pytest.raises(Exception):
        g.del_samples(1, method="foo")
    g.set_data(x2, np.array([fun(r) for r in x2]))
    with pytest.raises(Exception, match="yconrevs class is not wgp"):  # gpmcmc.py:238-239
        g.fit(cwgp=True)
    with pytest.raises(Exception, match="none of xconrevs are wgp"):  # gpmcmc.py:232-233
        g.fit(iwgp=True)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):  # no CPU fallback for the hot path
            g.fit()


def test_constraints_filter_proposed_samples_like_the_reference(capsys):
    """core.py:218-246 through lhc.py:30-31: samples violating the constraint are dropped before the target runs;
    the reference's quirk that the last constraint's verdict overwrites earlier ones is kept."""
    from andvaranaut_amd import GPMCMC

    priors = [st.uniform(loc=0, scale=2), st.uniform(loc=1, scale=0.5)]
    calls = []

    def fun(x):
        calls.append(x.copy())
        return np.array([x[0] + x[1]])

    with pytest.raises(Exception, match="constraints must be a dictionary"):
        GPMCMC(nx=2, ny=1, priors=priors, target=fun, constraints=[lambda x: x[0]])
    with pytest.raises(Exception, match="constraints must be a dictionary"):
        GPMCMC(nx=2, ny=1, priors=priors, target=fun, constraints={"constraints": []})
    cons = {"constraints": [lambda x: x[0]], "lower_bounds": [0.0], "upper_bounds": [1.0]}
    g = GPMCMC(nx=2, ny=1, priors=priors, target=fun, constraints=cons, verbose=False)
    g.sample(40, seed=3)
    assert 0 < g.nsamp < 40 and np.all(g.x[:, 0] <= 1.0) and len(calls) == g.nsamp
    assert "samples removed due to violating constraints" in capsys.readouterr().out
    # list-valued bounds and two constraints: only the LAST constraint decides (reference quirk)
    cons2 = {"constraints": [lambda x: x[0], lambda x: [x[1], x[0] + x[1]]], "lower_bounds": [5.0, [1.0, 0.0]],
             "upper_bounds": [6.0, [1.25, 10.0]]}
    g2 = GPMCMC(nx=2, ny=1, priors=priors, target=fun, constraints=cons2, verbose=False)
    g2.sample(40, seed=4)
    assert 0 < g2.nsamp < 40 and np.all(g2.x[:, 1] <= 1.25)  # the first constraint (x0 in [5,6]) rejects everything, yet is overwritten


# ----------------------------------------------------------------------------------------------- warps
class _OracleGP:
    """Stands in for andvaranaut_amd.MiGP on the CPU: same methods, arithmetic by the oracle (tests only)."""

    def __init__(self, X, y, kerns, ops):
        self.X, self.y, self.kerns, self.ops = X.copy(), y.copy(), kerns, ops

    def update_data(self, X=None, y=None):
        if X is not None:
            self.X = np.array(X, dtype=np.float64)
        if y is not None:
            self.y = np.array(y, dtype=np.float64)

    def lml_grad(self, theta):
        return orc.lml_grad(self.X, self.y, self.kerns, self.ops, theta)

    def lml_grad_data(self, theta, want_x=True):
        val, g = self.lml_grad(theta)
        _, gy, gX = orc.lml_grad_data(self.X, self.y, self.kerns, self.ops, theta)
        return val, g, gy, gX


def test_wgp_round_trip_derivative_and_parameter_bookkeeping():
    from andvaranaut_amd.transform import wgp

    rng = np.random.default_rng(0)
    y = rng.lognormal(size=60) + 0.5
    names = ["logarithm", "meanstd", "sal", "boxcox", "arcsinh", "stddev"]
    params = np.array([0.1, 1.2, -0.2, 0.9, 0.05, 0.3, 1.1, -0.1, 0.8])
    w = wgp(names, params, y=y)
    assert w.np == 9 and list(w.pid) == [0, 0, 4, 5, 9, 9]
    assert list(w.pos) == [False, True, False, True, False, False, True, False, True]
    assert len(w.default_priors) == 9
    yc = w.con(y)
    assert np.allclose(w.rev(yc), y, rtol=1e-12)
    h = 1e-6
    assert np.allclose((w.con(y + h) - w.con(y - h)) / (2 * h), w.der(y), rtol=1e-7)
    assert np.allclose(w.conmc(y), yc) and np.allclose(w.dermc(y), w.der(y))
    # data-dependent members are fitted to the data as warped by the members before them (transform.py:543-548)
    assert abs(np.std(yc) - 1.0) < 1e-12
    with pytest.raises(Exception):
        wgp(["meanstd"], np.zeros(0))  # needs y
    with pytest.raises(Exception):
        wgp(["foo"], np.zeros(0), y=y)
    k = wgp(["uniform", "kumaraswamy"], np.array([1.3, 0.8]), y=rng.uniform(size=10), xdist=st.uniform(0, 1))
    x = rng.uniform(0.05, 0.95, 20)
    assert list(k.pos) == [True, True] and np.allclose(k.rev(k.con(x)), x, rtol=1e-12)


@pytest.mark.parametrize("iwgp,cwgp,truncate", [(False, True, False), (True, False, False), (True, True, False),
                                                (True, True, True)])
def test_warped_posterior_gradient_by_finite_differences(iwgp, cwgp, truncate):
    """logp of fit(iwgp=..., cwgp=...) (gpmcmc.py:211-279,311-319): gradient through the warps = device data
    gradients (here the oracle's) pulled back by torch.autograd, against central differences."""
    from andvaranaut_amd import GPMCMC
    from andvaranaut_amd.priors import HyperModel
    from andvaranaut_amd.transform import uniform, wgp

    rng = np.random.default_rng(1)
    priors = [st.uniform(loc=0, scale=2), st.uniform(loc=1, scale=0.5)]
    fun = lambda x: np.array([np.exp(np.sin(2 * x[0]) + x[1] ** 2)])  # noqa: E731
    x = np.column_stack([rng.uniform(0, 2, 30), rng.uniform(1, 1.5, 30)])
    y = np.array([fun(r) for r in x])
    xcon = [wgp(["uniform", "kumaraswamy"], np.array([1.0, 1.0]), y=x[:, 0], xdist=priors[0]), uniform(priors[1])]
    ycon = [wgp(["logarithm", "meanstd", "sal"], np.array([0.0, 1.0, 0.0, 1.0]), y=y[:, 0])]
    g = GPMCMC(kernel="Matern52", noise=True, xconrevs=xcon, yconrevs=ycon, nx=2, ny=1, priors=priors, target=fun,
               verbose=False)
    g.set_data(x, y)
    n_i, n_pos, n_free = g._warp_sizes(iwgp, cwgp)
    assert (n_i, n_pos, n_free) == (2 if iwgp else 0, 2 if cwgp else 0, 2 if cwgp else 0)
    model = HyperModel(2, ["Matern52"], noise=True, truncate=truncate, n_iwgp=n_i, n_cwgp_pos=n_pos, n_cwgp=n_free)
    xin, yin = g._converted(x, y)
    fake = _OracleGP(xin, yin, ["Matern52"], [])
    lik = g._warp_likelihood(fake, x, y, xin, iwgp, cwgp)
    for jac in (True, False):
        f = lambda q: model.logp_dlogp(q, None, jacobian=jac, likelihood=lik)  # noqa: E731
        q = model.initial_point() + 0.1 * rng.standard_normal(model.nq)
        v, grad = f(q)
        assert np.isfinite(v)
        for i in range(model.nq):
            h = 1e-6
            qp, qm = q.copy(), q.copy()
            qp[i] += h
            qm[i] -= h
            fd = (f(qp)[0] - f(qm)[0]) / (2 * h)
            assert abs(fd - grad[i]) <= 5e-5 * max(1.0, abs(fd)), (i, fd, grad[i])
    names = [v[0] for v in model.vars]
    assert names == ["gv", "l", "kv"] + (["iwgp"] if iwgp else []) + (["cwgp_pos", "cwgp"] if cwgp else [])
    pt = model.point_dict(model.initial_point())
    if cwgp:
        assert "cwgp" in pt and ("cwgp_interval__" in pt if truncate else "cwgp_log__" not in pt)


def test_find_map_objective_has_no_transform_jacobian():
    """[3P] pm.find_MAP compiles logp with jacobian=False; NUTS keeps it: they differ by sum(log|dx/dq|)."""
    from andvaranaut_amd.priors import HyperModel

    X, y = orc.synth_problem(20, 2, seed=0)
    model = HyperModel(2, ["RBF"], noise=True)
    q = model.initial_point() + 0.05
    cb = _oracle_callable(X, y, ["RBF"], [])
    a, ga = model.logp_dlogp(q, cb, jacobian=True)
    b, gb = model.logp_dlogp(q, cb, jacobian=False)
    assert abs((a - b) - np.sum(q)) < 1e-9  # all three blocks are log-transformed: log|dx/dq| = q
    assert np.allclose(ga - gb, 1.0)


def test_input_model_priors_and_gradient():
    """scipy -> PyMC prior conversion of BO / inverse_opt (gpmcmc.py:702-728,1053-1094) and the transformed
    log-density that find_MAP / NUTS see."""
    from andvaranaut_amd.consumers import InputModel, Uniform, pymc_prior
    from andvaranaut_amd.priors import Normal, TruncatedNormal

    u = pymc_prior(st.uniform(loc=1, scale=0.5))
    assert isinstance(u, Uniform) and (u.lower, u.upper) == (1.0, 1.5)
    u2 = pymc_prior(st.uniform(2.0, 3.0))
    assert (u2.lower, u2.upper) == (2.0, 5.0)
    nrm = pymc_prior(st.norm(loc=0.3, scale=2.0))
    assert isinstance(nrm, Normal) and (nrm.mu, nrm.sigma) == (0.3, 2.0)
    t = pymc_prior(st.truncnorm(-1.0, 2.0, loc=0.5, scale=0.2), allow_truncnorm=True)
    assert isinstance(t, TruncatedNormal) and np.allclose([t.lower, t.upper, t.mu, t.sigma], [0.3, 0.9, 0.5, 0.2])
    with pytest.raises(Exception):
        pymc_prior(st.truncnorm(-1.0, 2.0))  # BO converts uniform and normal priors only
    with pytest.raises(Exception):
        pymc_prior(st.beta(2, 3), allow_truncnorm=True)
    im = InputModel([u, nrm, t])
    pot = lambda x: (-np.sum((x - np.array([1.2, 0.1, 0.6])) ** 2), -2 * (x - np.array([1.2, 0.1, 0.6])))  # noqa: E731
    q = im.initial_point() + np.array([0.3, -0.2, 0.4])
    for jac in (True, False):
        v, g = im.logp_dlogp(q, pot, jacobian=jac)
        for i in range(3):
            h = 1e-6
            qp, qm = q.copy(), q.copy()
            qp[i] += h
            qm[i] -= h
            fd = (im.logp_dlogp(qp, pot, jacobian=jac)[0] - im.logp_dlogp(qm, pot, jacobian=jac)[0]) / (2 * h)
            assert abs(fd - g[i]) < 1e-6 * max(1.0, abs(fd))
    pt = im.point_dict(q)
    assert set(pt) == {"x0_interval__", "x0", "x1", "x2_interval__", "x2"}
    assert np.allclose(im.q_from_x(im.x_from_q(q)), q)
    # truncated-normal logp against scipy
    assert abs(float(t.logp(np.array(0.55))) - st.truncnorm(-1.0, 2.0, loc=0.5, scale=0.2).logpdf(0.55)) < 1e-12


def test_overflowed_transform_is_minus_infinity_not_an_error():
    """A leapfrog step far into the tails can overflow exp(q): the log-density there is -inf (rejected), the device
    entry point (which refuses non-finite theta) is not even called."""
    from andvaranaut_amd.priors import HyperModel

    model = HyperModel(2, ["RBF"], noise=True)
    q = model.initial_point()
    q[1] = 800.0  # exp overflows

    def boom(theta):
        raise AssertionError("device must not be called with a non-finite theta")

    with np.errstate(all="ignore"):
        v, g = model.logp_dlogp(q, boom)
    assert v == -np.inf and np.all(g == 0)


def test_gauss_hermite_reversion_matches_the_per_point_loop():
    """SURVEY 8a row A6 (gpmcmc.py:545-569): the vectorised __gh_stats against a literal restatement of the reference's
    per-point loop -- mean / variance, normvar on and off, EI towards max and min, a non-zero mean function and a
    non-trivial reversion (logarithmic warp)."""
    from andvaranaut_amd import GPMCMC
    from andvaranaut_amd.transform import logarithm

    priors = [st.uniform(loc=0, scale=2), st.uniform(loc=1, scale=0.5)]
    fun = lambda x: np.array([np.exp(x[0] - x[1])])  # noqa: E731
    rng = np.random.default_rng(3)
    M = 23
    x = rng.uniform([0, 1], [2, 1.5], (M, 2))
    mu = rng.normal(0.2, 0.8, (M, 1))
    var = rng.uniform(1e-4, 0.5, (M, 1))

    def loop(g, x, y, yv, normvar, deg, EI, EIopt):  # gpmcmc.py:545-569, line by line
        y, yv = y.copy(), yv.copy()
        xi, wi = np.polynomial.hermite.hermgauss(deg)
        for i in range(len(y)):
            yi = np.sqrt(2 * yv[i, 0]) * xi + y[i, 0]
            yir = g.yconrevs[0].rev(yi) + g.mean(x[i, :])
            if EI:
                ydiff = yir - g.yopt if EIopt == "max" else g.yopt - yir
                ydiff = np.where(ydiff > 0.0, ydiff, 0.0)
                y[i, 0] = 1 / np.sqrt(np.pi) * np.sum(wi * ydiff)
            else:
                y[i, 0] = 1 / np.sqrt(np.pi) * np.sum(wi * yir)
            ym2 = 1 / np.sqrt(np.pi) * np.sum(wi * np.power(yir, 2))
            yv[i, 0] = ym2 - y[i, 0] ** 2
        if normvar:
            yv /= np.power(y, 2)
        return y, yv

    for mean in (0, lambda xx: np.array([0.3 * xx[0] - 0.1])):
        g = GPMCMC(kernel="RBF", noise=True, yconrevs=[logarithm()], mean=mean, nx=2, ny=1, priors=priors, target=fun,
                   verbose=False)
        g.yopt = 0.9
        for normvar in (True, False):
            for deg in (8, 5):
                for EI, EIopt in ((False, None), (True, "max"), (True, "min")):
                    got = g._GPMCMC__gh_stats(x, mu.copy(), var.copy(), normvar, deg, EI=EI, EIopt=EIopt)
                    ref = loop(g, x, mu, var, normvar, deg, EI, EIopt)
                    assert np.allclose(got[0], ref[0], rtol=1e-13, atol=1e-15), (normvar, deg, EI, EIopt)
                    assert np.allclose(got[1], ref[1], rtol=1e-11, atol=1e-14), (normvar, deg, EI, EIopt)


def test_batched_evaluator_rendezvous_serves_every_client_its_own_row():
    """andvaranaut_amd/batching.py: K host threads (the NUTS chains of one GPU, gpmcmc.py:351) meet in one batched call per
    round; clients that finish early leave and the others go on in smaller batches; a failing batch raises in every client."""
    import threading

    from andvaranaut_amd.batching import BatchedEvaluator

    sizes = []

    def batch_fn(th):
        sizes.append(len(th))
        return th.sum(axis=1), 2.0 * th

    K = 5
    ev = BatchedEvaluator(batch_fn, K)
    got = {}

    def client(c):
        try:
            for step in range(3 + c):  # clients leave at different times
                th = np.array([c, step, 1.0])
                v, g = ev.evaluate(th)
                assert v == th.sum() and np.array_equal(g, 2.0 * th)
                got[(c, step)] = v
        finally:
            ev.leave()

    ts = [threading.Thread(target=client, args=(c,)) for c in range(K)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=30)
    assert not any(t.is_alive() for t in ts)
    assert len(got) == sum(3 + c for c in range(K)) and ev.evaluations == len(got)
    assert sizes[:3] == [5, 5, 5] and sizes[-1] == 1 and sorted(set(sizes)) == [1, 2, 3, 4, 5]

    def bad(th):
        raise ValueError("device lost")

    ev2 = BatchedEvaluator(bad, 2)
    errs = []

    def c2():
        try:
            ev2.evaluate(np.zeros(3))
        except RuntimeError as e:
            errs.append(e)
        finally:
            ev2.leave()

    ts = [threading.Thread(target=c2) for _ in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=30)
    assert len(errs) == 2


def test_sharded_timeline_model_orders_its_three_send_modes():
    """tools/emulate_rank.predict(): the timeline that turns per-step times of emulated ranks into a W-rank prediction
    (a model, not a measurement).  On synthetic step tables: (i) with a free link the prediction is the
    serial owner chain plus the bulk updates of the slowest rank; (ii) a piece-wise send is never later than the send
    behind the whole panel and never earlier than a free link; (iii) when the pieces are staged evenly and a piece's
    transfer is shorter than the gap to the next piece, only the LAST piece's transfer is exposed."""
    import importlib.util
    import os

    import numpy as np

    spec = importlib.util.spec_from_file_location("emulate_rank", os.path.join(ROOT, "tools", "emulate_rank.py"))
    em = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(em)
    world, pwt, npan, N = 4, 4, 16, 16 * 512
    recs = []
    for r in range(world):
        steps = np.zeros((npan + 1, 4))
        pieces = np.zeros((npan + 1, pwt))
        for j in range(npan):
            if (j + 1) % world == r and j + 1 < npan:
                steps[j, 0], steps[j, 1] = 0.2, 0.8            # update, factor (+ staging) of panel j + 1
                pieces[j] = 0.2 + 0.8 * np.arange(1, pwt + 1) / pwt   # staged evenly through the factorisation
            steps[j, 3] = 0.5                                  # this rank's bulk update of step j
        if r == 0:
            steps[npan, 1] = 0.8
            pieces[npan] = 0.8 * np.arange(1, pwt + 1) / pwt
        recs.append({"rank": r, "npan": npan, "steps": steps.tolist(), "pieces": pieces.tolist()})
    free = em.predict(world, recs, N, pwt, True, link_gbps=1e9, pipelined=True)["predicted_ms"]
    whole = em.predict(world, recs, N, pwt, True, link_gbps=100.0, pipelined=False)
    piece = em.predict(world, recs, N, pwt, True, link_gbps=100.0, pipelined=True)
    assert free <= piece["predicted_ms"] <= whole["predicted_ms"]
    assert piece["sum_link_ms_behind_the_chain"] < whole["sum_link_ms_behind_the_chain"]
    # (iii): at 100 GB/s a piece of the first panel takes (N + 256) * 128 * 8 / 1e11 s = 0.087 ms < the 0.2 ms between pieces
    p0 = (N + 128 + 128) * 128 * 8 / 100e9 * 1e3
    exposed_first = em.predict(world, [recs[0]] + recs[1:], N, pwt, True, link_gbps=100.0, pipelined=True)
    assert abs(whole["sum_link_ms"] / piece["sum_link_ms"] - 1.0) < 1e-12  # same bytes either way
    assert piece["sum_link_ms_behind_the_chain"] <= (npan + 1) * p0 + 1e-9  # at most one piece per panel is exposed
    assert exposed_first["predicted_ms"] == piece["predicted_ms"]            # deterministic
