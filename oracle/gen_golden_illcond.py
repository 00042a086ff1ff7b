"""Golden vectors in the ILL-CONDITIONED regime -- TEST INFRASTRUCTURE ONLY.

The mpmath fixtures of gen_golden.py stop at cond(K) ~ 1e6, N <= 64; the random sweep of the GPU tests reaches 1e8 and had
its gradient tolerance widened with no independent truth to say which side (device or NumPy oracle) moved.
Three cases with cond(K + (gv + jitter) I) between 1e7 and 1e9 -- dense points in d = 1..2, small noise: RBF, Matern-5/2
and a product kernel -- with the LML by 50-digit mpmath Cholesky (gpmcmc.py:313-318 as written) and its gradient by
50-digit numerical differentiation of that LML (the same code path as gen_golden.py: nothing shared with
oracle/gp_oracle.py).  What the reference evaluates at these points: pm.find_MAP / pm.sample at gpmcmc.py:345, 351.
Run:  python oracle/gen_golden_illcond.py     (~10 minutes; deterministic)
"""
import json
import os
import sys

import mpmath as mp
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import OUT, mp_lml, to_mp  # noqa: E402

mp.mp.dps = 50


def case(name, N, d, kerns, ops, seed, ls, gv, jitter):
    rng = np.random.default_rng(seed)
    X = rng.random((N, d))
    y = np.sin(3.0 * X.sum(1)) + 3e-4 * rng.standard_normal(N)
    nk = len(kerns)
    ls = np.asarray(ls, dtype=np.float64).reshape(nk, d)
    kv = 0.9 + 0.4 * rng.random(nk)
    alpha = 1.0 + rng.random(nk)
    Xm = to_mp(X)
    ym = [mp.mpf(float(v)) for v in y]
    lsm, kvm, alm = to_mp(ls), [mp.mpf(float(v)) for v in kv], [mp.mpf(float(v)) for v in alpha]
    gvm, jm = mp.mpf(float(gv)), mp.mpf(float(jitter))
    val, L, beta = mp_lml(Xm, ym, kerns, ops, lsm, kvm, alm, gvm, jm)
    out = {"name": name, "N": N, "d": d, "kerns": kerns, "ops": ops, "X": X.tolist(), "y": y.tolist(),
           "ls": ls.tolist(), "kv": kv.tolist(), "alpha": alpha.tolist(), "gv": gv, "jitter": jitter,
           "lml": mp.nstr(val, 30),
           "logdet": mp.nstr(sum(mp.log(L[i, i]) for i in range(N)), 30),
           "quad": mp.nstr(sum(b * b for b in beta), 30)}

    def f(*p):
        p = list(p)
        l2 = [p[c * d:(c + 1) * d] for c in range(nk)]
        k2 = p[nk * d: nk * d + nk]
        g2 = p[nk * d + nk]
        return mp_lml(Xm, ym, kerns, ops, l2, k2, alm, g2, jm)[0]

    p0 = [v for row in lsm for v in row] + kvm + [gvm]
    grad = []
    for i in range(len(p0)):
        # central difference at 50 digits: step 1e-18 relative, truncation error ~1e-36 of the third derivative
        h = abs(p0[i]) * mp.mpf("1e-18")
        pp, pm_ = list(p0), list(p0)
        pp[i] += h
        pm_[i] -= h
        grad.append(mp.nstr((f(*pp) - f(*pm_)) / (2 * h), 25))
        print("  grad", i, grad[-1], flush=True)
    out["grad"] = grad  # order: ls (nk * d), kv (nk), gv
    return out


def main():
    cases = [
        case("RBF_N128_d1_cond", 128, 1, ["RBF"], [], 11, [[0.25]], 1e-7, 1e-8),
        case("Matern52_N160_d2_cond", 160, 2, ["Matern52"], [], 12, [[2.5, 3.5]], 2e-7, 1e-9),
        case("RBFxMatern32_N96_d2_cond", 96, 2, ["RBF", "Matern32"], ["*"], 13, [[0.6, 0.9], [2.0, 1.5]], 1e-8, 1e-9),
    ]
    with open(os.path.join(OUT, "mpmath_illcond.json"), "w") as fh:
        json.dump(cases, fh, indent=0)


if __name__ == "__main__":
    main()
