"""Generate the golden vectors under tests/golden/  --  TEST INFRASTRUCTURE ONLY.

The reference cannot be imported in this container (ModuleNotFoundError: dask, pymc, pytensor,
arviz, netCDF4, seaborn) and ships no fixtures, so the vectors come from implementations that share
no code with oracle/gp_oracle.py:
  * mpmath at 50 digits (this file): the kernel formulas the reference composes at
    gpmcmc.py:282-307 ([3P] pymc.gp.cov: r = sqrt(r2 + 1e-12), ExpQuad, Matern52, Matern32,
    Exponential = exp(-r/2), RatQuad), K + (gv + jitter) I, LML as written at gpmcmc.py:313-318,
    its gradient by high-precision numerical differentiation, and the conditional mean / variance
    as written at gpmcmc.py:766-778 with Stationary.diag = 1 and pred_noise=True (gpmcmc.py:593-594);
  * closed forms for N = 1 and N = 2;
  * the two numbers the tutorial notebook records for the uniform / normal conversions
    (tutorial/tutorial.ipynb:366).
Run:  python oracle/gen_golden.py     (about a minute; deterministic)
"""
import json
import os

import mpmath as mp
import numpy as np

mp.mp.dps = 50
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def mp_base(name, r2, alpha):
    if name == "RBF":
        return mp.exp(-r2 / 2)
    if name == "RatQuad":
        return (1 + r2 / (2 * alpha)) ** (-alpha)
    r = mp.sqrt(r2 + mp.mpf("1e-12"))
    if name == "Matern52":
        return (1 + mp.sqrt(5) * r + mp.mpf(5) / 3 * r * r) * mp.exp(-mp.sqrt(5) * r)
    if name == "Matern32":
        return (1 + mp.sqrt(3) * r) * mp.exp(-mp.sqrt(3) * r)
    if name == "Exponential":
        return mp.exp(-r / 2)
    raise ValueError(name)


def mp_cov(Xa, Xb, kerns, ops, ls, kv, alpha):
    na, nb, d = len(Xa), len(Xb), len(Xa[0])
    K = mp.matrix(na, nb)
    for i in range(na):
        for j in range(nb):
            val = None
            for c, name in enumerate(kerns):
                r2 = mp.mpf(0)
                for m in range(d):
                    t = (Xa[i][m] - Xb[j][m]) / ls[c][m]
                    r2 += t * t
                kc = kv[c] * mp_base(name, r2, alpha[c])
                if val is None:
                    val = kc
                elif ops[c - 1] == "+":
                    val = val + kc
                else:
                    val = val * kc
            K[i, j] = val
    return K


def mp_chol(K):
    n = K.rows
    L = mp.matrix(n, n)
    for j in range(n):
        s = K[j, j] - sum(L[j, k] ** 2 for k in range(j))
        L[j, j] = mp.sqrt(s)
        for i in range(j + 1, n):
            L[i, j] = (K[i, j] - sum(L[i, k] * L[j, k] for k in range(j))) / L[j, j]
    return L


def mp_fwd(L, b):
    n = L.rows
    x = [mp.mpf(0)] * n
    for i in range(n):
        x[i] = (b[i] - sum(L[i, k] * x[k] for k in range(i))) / L[i, i]
    return x


def mp_lml(X, y, kerns, ops, ls, kv, alpha, gv, jitter):
    n = len(X)
    K = mp_cov(X, X, kerns, ops, ls, kv, alpha)
    for i in range(n):
        K[i, i] += gv + jitter
    L = mp_chol(K)
    beta = mp_fwd(L, y)
    quad = sum(b * b for b in beta)
    logdet = sum(mp.log(L[i, i]) for i in range(n))
    return -quad / 2 - logdet - mp.mpf(n) / 2 * mp.log(2 * mp.pi), L, beta


def to_mp(a):
    return [[mp.mpf(float(v)) for v in row] for row in a]


def case(name, N, d, kerns, ops, seed, M=5, with_grad=True):
    rng = np.random.default_rng(seed)
    X = rng.random((N, d))
    y = np.sin(3.0 * X.sum(1)) + 0.1 * rng.standard_normal(N)
    Xnew = rng.random((M, d))
    nk = len(kerns)
    ls = 0.4 + rng.random((nk, d))
    kv = 0.8 + rng.random(nk)
    alpha = 1.0 + rng.random(nk)
    gv, jitter = 1e-3, 1e-6
    Xm, Xn = to_mp(X), to_mp(Xnew)
    ym = [mp.mpf(float(v)) for v in y]
    lsm, kvm, alm = to_mp(ls), [mp.mpf(float(v)) for v in kv], [mp.mpf(float(v)) for v in alpha]
    gvm, jm = mp.mpf(float(gv)), mp.mpf(float(jitter))
    val, L, beta = mp_lml(Xm, ym, kerns, ops, lsm, kvm, alm, gvm, jm)
    out = {
        "name": name, "N": N, "d": d, "kerns": kerns, "ops": ops,
        "X": X.tolist(), "y": y.tolist(), "Xnew": Xnew.tolist(),
        "ls": ls.tolist(), "kv": kv.tolist(), "alpha": alpha.tolist(), "gv": gv, "jitter": jitter,
        "lml": mp.nstr(val, 30),
        "logdet": mp.nstr(sum(mp.log(L[i, i]) for i in range(N)), 30),
        "quad": mp.nstr(sum(b * b for b in beta), 30),
    }
    # a few covariance entries for elementwise checks
    K = mp_cov(Xm, Xm, kerns, ops, lsm, kvm, alm)
    out["K_samples"] = [[i, j, mp.nstr(K[i, j], 30)] for (i, j) in [(0, 0), (1, 0), (N - 1, N // 2), (N - 1, N - 1)]]
    # conditional: A = L^-1 K(X,X*), v = L^-1 y, mu = A^T v, var = kdiag - sum A^2 + gv
    Ks = mp_cov(Xm, Xn, kerns, ops, lsm, kvm, alm)
    kd = kvm[0]
    for c in range(1, nk):
        kd = kd + kvm[c] if ops[c - 1] == "+" else kd * kvm[c]
    mu, var = [], []
    for q in range(M):
        a = mp_fwd(L, [Ks[i, q] for i in range(N)])
        mu.append(mp.nstr(sum(ai * bi for ai, bi in zip(a, beta)), 30))
        var.append(mp.nstr(kd - sum(ai * ai for ai in a) + gvm, 30))
    out["mu"], out["var"] = mu, var
    if with_grad:
        def f(*p):
            p = list(p)
            l2 = [p[c * d:(c + 1) * d] for c in range(nk)]
            k2 = p[nk * d: nk * d + nk]
            a2 = p[nk * d + nk: nk * d + 2 * nk]
            g2 = p[nk * d + 2 * nk]
            return mp_lml(Xm, ym, kerns, ops, l2, k2, a2, g2, jm)[0]
        p0 = [v for row in lsm for v in row] + kvm + alm + [gvm]
        grad = []
        for i in range(len(p0)):
            order = tuple(1 if j == i else 0 for j in range(len(p0)))
            grad.append(mp.nstr(mp.diff(f, tuple(p0), order), 25))
        out["grad"] = grad  # order: ls(nk*d), kv(nk), alpha(nk), gv
    return out


def closed_forms():
    # N=1: LML = -1/2 y^2/k - 1/2 log k - 1/2 log 2pi with k = kv*k(0) + gv + jitter
    y0, kv, gv, jitter = mp.mpf("0.7"), mp.mpf("1.3"), mp.mpf("0.01"), mp.mpf("1e-6")
    k_rbf = kv + gv + jitter
    k_m52 = kv * mp_base("Matern52", mp.mpf(0), None) + gv + jitter
    n1 = {
        "y": float(y0), "kv": float(kv), "gv": float(gv), "jitter": float(jitter), "x": [0.3, 0.9], "ls": [0.5, 2.0],
        "lml_RBF": mp.nstr(-y0 ** 2 / (2 * k_rbf) - mp.log(k_rbf) / 2 - mp.log(2 * mp.pi) / 2, 30),
        "lml_Matern52": mp.nstr(-y0 ** 2 / (2 * k_m52) - mp.log(k_m52) / 2 - mp.log(2 * mp.pi) / 2, 30),
    }
    # N=2 RBF, d=1: K = [[a, b],[b, a]], a = kv+gv+jitter, b = kv exp(-(x1-x2)^2/(2 l^2))
    x1, x2, l, y1, y2 = mp.mpf("0.2"), mp.mpf("0.9"), mp.mpf("0.6"), mp.mpf("0.5"), mp.mpf("-0.4")
    a = kv + gv + jitter
    b = kv * mp.exp(-((x1 - x2) / l) ** 2 / 2)
    det = a * a - b * b
    quad = (a * y1 * y1 - 2 * b * y1 * y2 + a * y2 * y2) / det
    n2 = {
        "x": [float(x1), float(x2)], "l": float(l), "y": [float(y1), float(y2)], "kv": float(kv), "gv": float(gv),
        "jitter": float(jitter), "lml_RBF": mp.nstr(-quad / 2 - mp.log(det) / 2 - mp.log(2 * mp.pi), 30),
    }
    return {"n1": n1, "n2": n2}


def main():
    os.makedirs(OUT, exist_ok=True)
    cases = []
    i = 0
    for (N, d) in [(4, 1), (16, 2), (64, 8)]:
        for kern in ["RBF", "Matern52", "Matern32", "Exponential", "RatQuad"]:
            i += 1
            cases.append(case(f"{kern}_N{N}_d{d}", N, d, [kern], [], seed=100 + i, with_grad=(N <= 16)))
            print("done", cases[-1]["name"], flush=True)
    for (kerns, ops) in [(["RBF", "Matern52"], ["+"]), (["Matern32", "RBF"], ["*"]), (["RBF", "Matern32", "Exponential"], ["+", "*"])]:
        i += 1
        cases.append(case("_".join(kerns) + "_" + "".join("p" if o == "+" else "m" for o in ops), 16, 2, kerns, ops, seed=100 + i))
        print("done", cases[-1]["name"], flush=True)
    with open(os.path.join(OUT, "mpmath_cases.json"), "w") as f:
        json.dump(cases, f, indent=0)
    with open(os.path.join(OUT, "closed_form.json"), "w") as f:
        json.dump(closed_forms(), f, indent=1)
    # tutorial notebook pins (tutorial/tutorial.ipynb:61-68 priors U(0,2), N(1,... ) see :366)
    pins = {"x": [1.85531589, 1.24150338], "xc": [0.92765794, -0.05886629],
            "source": "tutorial/tutorial.ipynb:366 (recorded output of the uniform / normal conversions)"}
    with open(os.path.join(OUT, "tutorial_pins.json"), "w") as f:
        json.dump(pins, f, indent=1)


if __name__ == "__main__":
    main()
