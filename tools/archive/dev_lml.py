"""Dev harness: LML parity vs the oracle + timings."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from andvaranaut_amd import MiGP
from oracle import gp_oracle as orc
import torch

for (N, d, kernel) in [(100, 2, "RBF"), (128, 2, "RBF"), (300, 3, "Matern52"), (1024, 8, "RBF"), (1000, 4, "Matern32+RBF"), (640, 5, "RBF*Exponential"), (4096, 8, "RBF"), (2048, 4, "RatQuad")]:
    X, y = orc.synth_problem(N, d, seed=1)
    kerns, ops = kernel.replace("*", "+").split("+"), [c for c in kernel if c in "+*"]
    theta = orc.synth_theta(d, nkern=len(kerns))
    t0 = time.time(); ref = orc.lml(X, y, kerns, ops, theta); t1 = time.time()
    gp = MiGP(X, y, kernel, need_grad=False)
    val = gp.lml(theta)
    ld, qd = gp.lml_parts()
    print(f"N={N} d={d} {kernel}: gpu {val:.12e} ref {ref:.12e} rel {abs(val-ref)/abs(ref):.2e} info={gp.info} (cpu {t1-t0:.2f}s)")
    gp.close()
if len(sys.argv) > 1:
    N, d = int(sys.argv[1]), 16
    X, y = orc.synth_problem(N, d, seed=0)
    theta = orc.synth_theta(d)
    for W in (2, 4, 8):
        gp = MiGP(X, y, "Matern52", need_grad=False, panel_tiles=W)
        gp.lml(theta)
        gp.set_profiling(2)
        v = gp.lml(theta); tm = gp.timers()
        gp.set_profiling(0)
        t0 = time.time()
        for _ in range(3): gp.lml(theta)
        dt = (time.time() - t0) / 3
        print(f"N={N} W={W}: lml={v:.10e} wall {dt*1e3:.1f} ms; {tm}; chol TF={N**3/3/tm['cholesky_ms']*1e-9:.2f} gemm TF={tm['gemm_flops']/tm['gemm_ms']*1e-9:.2f}")
        gp.close()
