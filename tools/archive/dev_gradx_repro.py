import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from oracle import gp_oracle as orc
from andvaranaut_amd import _lib
if len(sys.argv) > 1: _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from andvaranaut_amd import MiGP
kernel, d, N = "Exponential*Matern32+Matern32+RBF", 2, 207
X, y = orc.synth_problem(N, d, seed=17)
kerns, ops = kernel.replace("*", "+").split("+"), [c for c in kernel if c in "+*"]
theta = orc.synth_theta(d, nkern=4, gv=1e-3)
_, _, ref = orc.lml_grad_data(X, y, kerns, ops, theta)
gp = MiGP(X, y, kernel)
val, g = gp.lml_grad(theta)
W0 = gp.W_t.clone() if hasattr(gp, "W_t") else None
outs = []
gx_t = torch.empty((N, d), dtype=torch.float64, device=gp.dev)
for i in range(8):
    gx_t.fill_(float(i))
    torch.cuda.synchronize()
    r = gp.lib.mi_gp_grad_x(gp.h, gx_t.data_ptr()); assert r == 0
    outs.append(gx_t.cpu().numpy().copy())
for i, o in enumerate(outs):
    diff = np.abs(o - outs[0]).max()
    err = np.abs(o - ref).max() / np.abs(ref).max()
    nbad = int((o != outs[0]).sum())
    print(i, "max|o - o0|", diff, "entries differing", nbad, "rel err vs oracle", err)
if W0 is not None:
    print("W unchanged:", bool(torch.equal(W0, gp.W_t)))
bad = np.argwhere(outs[1] != outs[0])
print("differing rows (first 20):", sorted(set(bad[:, 0].tolist()))[:20])
