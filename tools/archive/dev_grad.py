"""Dev harness: gradient + predict parity vs the oracle."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from andvaranaut_amd import MiGP
from oracle import gp_oracle as orc

for (N, d, kernel) in [(100, 2, "RBF"), (300, 3, "Matern52"), (1000, 4, "Matern32+RBF"), (640, 5, "RBF*Matern52"), (700, 3, "RatQuad"), (1024, 8, "RBF"), (2500, 6, "Matern52")]:
    X, y = orc.synth_problem(N, d, seed=1)
    kerns, ops = kernel.replace("*", "+").split("+"), [c for c in kernel if c in "+*"]
    theta = orc.synth_theta(d, nkern=len(kerns), gv=1e-3)
    ref, gref = orc.lml_grad(X, y, kerns, ops, theta)
    gp = MiGP(X, y, kernel)
    val, g = gp.lml_grad(theta)
    scale = np.abs(gref).max()
    print(f"N={N} d={d} {kernel}: lml rel {abs(val-ref)/abs(ref):.2e}  grad max abs err/scale {np.abs(g-gref).max()/scale:.2e}")
    if np.abs(g-gref).max()/scale > 1e-6:
        print("  gpu ", g); print("  ref ", gref)
    Xn = np.random.default_rng(0).random((200, d))
    mu, var = gp.predict(theta, Xn)
    rmu, rvar = orc.predict(X, y, Xn, kerns, ops, theta)
    print(f"   predict: mean err {np.abs(mu-rmu).max():.2e} var err {np.abs(var-rvar).max():.2e}")
    gp.close()
if len(sys.argv) > 1:
    N, d = int(sys.argv[1]), 16
    X, y = orc.synth_problem(N, d, seed=0)
    theta = orc.synth_theta(d)
    gp = MiGP(X, y, "Matern52")
    gp.lml_grad(theta)
    gp.set_profiling(1)
    t0 = time.time(); v, g = gp.lml_grad(theta); dt = time.time() - t0
    print(f"N={N} lml+grad wall {dt*1e3:.1f} ms", gp.timers())
    gp.set_profiling(0)
    Xn = np.random.default_rng(0).random((10000, d))
    t0 = time.time(); mu, var = gp.predict(theta, Xn); dt = time.time() - t0
    print(f"predict 10000 pts wall {dt*1e3:.1f} ms; mean[:3]={mu[:3]} var[:3]={var[:3]}")
