"""Golden vectors for the data-side gradients -- TEST INFRASTRUCTURE ONLY.

dLML/dX, dLML/dy and the gradients of the conditional mean / variance w.r.t. a prediction point, by 50-digit
mpmath numerical differentiation of the formulas in gen_golden.py (no code shared with oracle/gp_oracle.py).
The reference obtains these quantities by PyTensor autodiff when warp parameters (gpmcmc.py:211-279) or input
points (gpmcmc.py:766-801, 1096-1165) are variables of its PyMC models.
Run:  python oracle/gen_golden_data.py      (about a minute; deterministic)
"""
import json
import os

import mpmath as mp
import numpy as np

from gen_golden import OUT, mp_cov, mp_fwd, mp_lml, to_mp

mp.mp.dps = 50


def case(name, N, d, kerns, ops, seed):
    rng = np.random.default_rng(seed)
    X = rng.random((N, d))
    y = np.sin(3.0 * X.sum(1)) + 0.1 * rng.standard_normal(N)
    xs = rng.random(d)
    nk = len(kerns)
    ls = 0.4 + rng.random((nk, d))
    kv = 0.8 + rng.random(nk)
    alpha = 1.0 + rng.random(nk)
    gv, jitter = 1e-3, 1e-6
    Xm = to_mp(X)
    ym = [mp.mpf(float(v)) for v in y]
    lsm, kvm, alm = to_mp(ls), [mp.mpf(float(v)) for v in kv], [mp.mpf(float(v)) for v in alpha]
    gvm, jm = mp.mpf(float(gv)), mp.mpf(float(jitter))

    def lml_x(i, m, v):
        X2 = [row[:] for row in Xm]
        X2[i][m] = v
        return mp_lml(X2, ym, kerns, ops, lsm, kvm, alm, gvm, jm)[0]

    def lml_y(i, v):
        y2 = ym[:]
        y2[i] = v
        return mp_lml(Xm, y2, kerns, ops, lsm, kvm, alm, gvm, jm)[0]

    gX = [[mp.nstr(mp.diff(lambda v, i=i, m=m: lml_x(i, m, v), Xm[i][m]), 25) for m in range(d)] for i in range(N)]
    gy = [mp.nstr(mp.diff(lambda v, i=i: lml_y(i, v), ym[i]), 25) for i in range(N)]
    _, L, beta = mp_lml(Xm, ym, kerns, ops, lsm, kvm, alm, gvm, jm)
    kd = kvm[0]
    for c in range(1, nk):
        kd = kd + kvm[c] if ops[c - 1] == "+" else kd * kvm[c]

    def cond(m, v, which):
        x2 = [mp.mpf(float(t)) for t in xs]
        x2[m] = v
        Ks = mp_cov(Xm, [x2], kerns, ops, lsm, kvm, alm)
        a = mp_fwd(L, [Ks[i, 0] for i in range(N)])
        if which == 0:
            return sum(ai * bi for ai, bi in zip(a, beta))
        return kd - sum(ai * ai for ai in a)

    dmu = [mp.nstr(mp.diff(lambda v, m=m: cond(m, v, 0), mp.mpf(float(xs[m]))), 25) for m in range(d)]
    dvar = [mp.nstr(mp.diff(lambda v, m=m: cond(m, v, 1), mp.mpf(float(xs[m]))), 25) for m in range(d)]
    return {"name": name, "N": N, "d": d, "kerns": kerns, "ops": ops, "X": X.tolist(), "y": y.tolist(), "xstar": xs.tolist(),
            "ls": ls.tolist(), "kv": kv.tolist(), "alpha": alpha.tolist(), "gv": gv, "jitter": jitter,
            "gX": gX, "gy": gy, "dmu": dmu, "dvar": dvar}


def main():
    cases = []
    specs = [("RBF_N4_d1", 4, 1, ["RBF"], []), ("Matern52_N16_d2", 16, 2, ["Matern52"], []),
             ("Matern32_N16_d2", 16, 2, ["Matern32"], []), ("RatQuad_N16_d2", 16, 2, ["RatQuad"], []),
             ("RBF_Matern32_Exponential_pm", 12, 2, ["RBF", "Matern32", "Exponential"], ["+", "*"])]
    for k, (name, N, d, kerns, ops) in enumerate(specs):
        cases.append(case(name, N, d, kerns, ops, seed=300 + k))
        print("done", name, flush=True)
    with open(os.path.join(OUT, "mpmath_data_grad.json"), "w") as f:
        json.dump(cases, f, indent=0)


if __name__ == "__main__":
    main()
