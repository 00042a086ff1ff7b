"""Every device entry point must return the same bits whatever earlier kernels left in the register file.

Round 3 shipped (for a while) a grad_x_kernel<4,1> whose accumulators the compiler had "spilled" to AGPRs with EXEC == 0
(DESIGN.md section 5.6): right in a fresh process, garbage once other kernels had run on the same SIMDs.  A repeat-and-compare
test cannot see that reliably -- the stale registers are often the previous, identical, evaluation's.  Here the whole
VGPR + AGPR file, the SGPRs, the LDS and the private segment of the chip are filled with NaN patterns
(tools/poison_state.hip) right before each call, and the result must equal, bit for bit, the one obtained after the same
state was zeroed.  The static counterpart is tests/test_isa_uninit.py.  Reference sites of the entry points:
gpmcmc.py:311-319 (LML), :345,351 (gradient), :211-279,1096-1165 (data-side gradients), :588-598,766-778 (conditional)."""
import ctypes
import os

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

NAN_HI, FINITE_HI = 0x7FF80000, 0x40590000


def _poison():
    path = os.path.join(ROOT, "tools", "libpoison.so")
    assert os.path.exists(path), "tools/libpoison.so missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(path)
    lib.poison_state.argtypes = [ctypes.c_int, ctypes.c_uint]

    def fill(kind, pattern):
        assert lib.poison_state(kind, pattern) == 0

    return fill


def _split(kernel):
    return kernel.replace("*", "+").split("+"), [c for c in kernel if c in "+*"]


CASES = [("RBF", 2), ("Matern52", 16), ("RBF+Matern32", 5), ("Matern52+RBF*RBF", 17), ("Exponential*Matern32+Matern32+RBF", 2),
         ("RBF+Matern52*Matern32+RBF", 3), ("RBF+Matern52+Matern32+RBF*Matern52", 3), ("RatQuad", 3)]


@pytest.mark.parametrize("kernel,d", CASES)
def test_entry_points_ignore_stale_registers_lds_and_scratch(kernel, d):
    import torch

    assert torch.cuda.is_available()
    from andvaranaut_amd import MiGP
    from oracle import gp_oracle as orc

    fill = _poison()
    N, M = 207, 9
    X, y = orc.synth_problem(N, d, seed=17)
    kerns, ops = _split(kernel)
    theta = orc.synth_theta(d, nkern=len(kerns), gv=1e-3)
    theta[: len(kerns) * d] *= np.sqrt(d / 2.0)
    Xn = np.random.default_rng(5).random((M, d))
    gp = MiGP(X, y, kernel)

    def evaluate(prepare):
        """every entry point once, `prepare()` right before each device call"""
        out = []
        prepare(); out.append(np.array([gp.lml(theta)]))
        prepare(); v, g = gp.lml_grad(theta); out += [np.array([v]), g]
        gx_t = torch.empty((N, d), dtype=torch.float64, device=gp.dev)
        torch.cuda.synchronize()
        prepare(); assert gp.lib.mi_gp_grad_x(gp.h, gx_t.data_ptr()) == 0; out.append(gx_t.cpu().numpy())
        prepare(); mu, var = gp.predict(theta, Xn); out += [mu, var]
        prepare(); mu2, var2, dm, dv = gp.predict_grad(theta, Xn); out += [mu2, var2, dm, dv]
        return out

    clean = evaluate(lambda: fill(2, 0))
    for kind, pattern, what in ((2, NAN_HI, "VGPRs + AGPRs = NaN"), (2, FINITE_HI, "VGPRs + AGPRs = 100.0"),
                                (3, NAN_HI, "SGPRs"), (0, NAN_HI, "LDS"), (1, NAN_HI, "private segment")):
        got = evaluate(lambda: fill(kind, pattern))
        for k, (a, b) in enumerate(zip(clean, got)):
            assert np.array_equal(a, b), (kernel, what, k, float(np.nanmax(np.abs(np.asarray(a) - np.asarray(b)))))
    # and the clean values are the right ones
    _, gy_ref, gX_ref = orc.lml_grad_data(X, y, kerns, ops, theta)
    tol = 1e-5 if "Exponential" in kernel else 1e-7
    assert np.abs(clean[3] - gX_ref).max() <= tol * np.abs(gX_ref).max()
    gp.close()
