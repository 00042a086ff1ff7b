"""Which class of stale per-CU state does the failing grad_x build read?  (round 4)
usage (inside a checkout of the failing tree): python poison_repro.py <libmi_gp variant .so> <libpoison.so>"""
import sys, os, ctypes
TREE = os.environ["GX_TREE"]
sys.path.insert(0, TREE)
import numpy as np, torch
from oracle import gp_oracle as orc
from andvaranaut_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from andvaranaut_amd import MiGP
poison = ctypes.CDLL(os.path.abspath(sys.argv[2]))
poison.poison_state.argtypes = [ctypes.c_int, ctypes.c_uint]
kernel, d, N = "Exponential*Matern32+Matern32+RBF", 2, 207
X, y = orc.synth_problem(N, d, seed=17)
kerns, ops = kernel.replace("*", "+").split("+"), [c for c in kernel if c in "+*"]
theta = orc.synth_theta(d, nkern=4, gv=1e-3)
_, _, ref = orc.lml_grad_data(X, y, kerns, ops, theta)
sc = np.maximum(np.abs(ref), 1e-3 * np.abs(ref).max())
gp = MiGP(X, y, kernel)
gp.lml_grad(theta)
gx_t = torch.empty((N, d), dtype=torch.float64, device=gp.dev)
names = {-1: "no poison", 0: "LDS", 1: "scratch (private segment)", 2: "VGPRs + AGPRs", 3: "SGPRs", 4: "VGPRs (tagged v_i = NaN | i)", 5: "AGPRs (tagged a_i = NaN | 0x1000 + i)", 6: "clean: VGPRs + AGPRs = 0"}
ORDER = [int(k) for k in os.environ.get('GX_KINDS', '-1,0,-1,1,-1,3,-1,5,-1,6,4,-1,6').split(',')]
for kind in ORDER:
    for pattern in ((0x7ff80000,) if kind < 0 or kind >= 4 else (0x7ff80000, 0x40590000)):  # NaN, then 100.0-ish doubles
        if kind >= 0:
            r = poison.poison_state(2 if kind == 6 else kind, 0 if kind == 6 else pattern)
            assert r == 0, r
        gx_t.fill_(0.0)
        torch.cuda.synchronize()
        r = gp.lib.mi_gp_grad_x(gp.h, gx_t.data_ptr()); assert r == 0
        g = gx_t.cpu().numpy()
        rel = np.abs(g - ref) / sc
        bad = np.argwhere(~(rel < 1e-6))
        print(f"{names[kind]:28s} pattern {pattern:#x}: nan entries {int(np.isnan(g).sum()):4d}  bad entries {len(bad):4d} "
              f"rowblocks {sorted(set((bad[:, 0] // 64).tolist()))} cols {sorted(set(bad[:, 1].tolist()))} "
              f"max rel {np.nanmax(np.where(np.isnan(rel), -1, rel)):.2e}", flush=True)
        if np.isnan(g).any():
            hi = (g.view(np.uint64) >> 32).astype(np.uint64)
            tags = sorted(set(int(v) & 0x1fff for v in hi[np.isnan(g)].ravel()))
            print("      NaN payload tags (register index of the high half; 0x1000 + i = AGPR i):", [hex(t) for t in tags][:40], flush=True)
