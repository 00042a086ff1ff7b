"""a deterministic guard for the inter-wave protocol of potrf_leaf128_kernel (csrc/leaf_f64.hip).  The leaf's
eight waves meet through LDS arrival counters, no workgroup barrier inside the loop; round 5 once had waves outside a role run
that role's arriving phase, the waiting side started early and the factor was wrong now and then -- only a stress loop saw it.
tools/libleafcheck.so is the library's own leaf source compiled with -DLEAF_CHECKED: every meeting point verifies the counter
values the protocol allows there (derived from ONE constexpr wave-role table) and reports a mismatch through the info word.
Every shape of the chain's register panels runs through it once: three panels (blocks 0-3), two (4-7), the last block's
from-the-registers output, padding (identity rows below a ragged block), the folded-in y row, a batch."""
import ctypes
import os

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
LIB = os.path.join(ROOT, "tools", "libleafcheck.so")


def _run(A, with_y=None, reps=3):
    """A: (nb, 128, 128) SPD blocks; with_y: (nb, 128) right-hand sides or None.  Returns L, M, info, beta."""
    lib = ctypes.CDLL(LIB)
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
    lib.leaf_check_run.argtypes = [dp, ctypes.c_long, ctypes.c_int, ctypes.c_int, dp, ip, ctypes.c_int]
    nb, lda = A.shape[0], 144
    rows = 256 if with_y is not None else 128
    buf = np.zeros((nb, rows, lda))
    buf[:, :128, :128] = np.tril(A)
    if with_y is not None:
        buf[:, 128, :128] = with_y
    M = np.zeros((nb, 128, 128))
    info = np.zeros(nb, dtype=np.int32)
    rc = lib.leaf_check_run(buf.ctypes.data_as(dp), lda, nb, 1 if with_y is not None else 0, M.ctypes.data_as(dp), info.ctypes.data_as(ip),
                            reps)
    assert rc == 0, rc
    proto = lib.leaf_check_protocol_info()
    assert np.all(info > proto), f"protocol mismatch at site(s) {[proto - int(v) for v in info if v <= proto]}"
    L = np.tril(buf[:, :128, :128])
    return L, M, info, (buf[:, 128, :128] if with_y is not None else None)


def _spd(rng, nb, n_real=128):
    A = np.zeros((nb, 128, 128))
    for z in range(nb):
        G = rng.standard_normal((n_real, n_real))
        A[z, :n_real, :n_real] = G @ G.T / n_real + np.eye(n_real)
        A[z, n_real:, n_real:] = np.eye(128 - n_real)  # the padding of a ragged last tile column: identity
    return A


@pytest.mark.parametrize("case", ["full", "ragged-37", "ragged-113", "yrow", "batch3", "batch3-yrow"])
def test_checked_leaf_meets_its_protocol_and_factors(case):
    if not os.path.exists(LIB):
        pytest.fail("tools/libleafcheck.so is missing: run __graft_entry__.build()")
    rng = np.random.default_rng(len(case))
    nb = 3 if case.startswith("batch") else 1
    n_real = int(case.split("-")[1]) if case.startswith("ragged") else 128
    A = _spd(rng, nb, n_real)
    y = rng.standard_normal((nb, 128)) if "yrow" in case else None
    L, M, info, beta = _run(A, y)
    assert np.all(info == 0x7F7F7F7F), info  # no bad pivot either
    for z in range(nb):
        assert np.max(np.abs(L[z] @ L[z].T - A[z])) <= 5e-14 * np.max(np.abs(A[z]))
        assert np.max(np.abs(M[z] @ L[z] - np.eye(128))) <= 1e-13
        assert np.max(np.abs(np.triu(M[z], 1))) == 0.0
        if beta is not None:
            assert np.max(np.abs(beta[z] - np.linalg.solve(L[z], y[z]))) <= 1e-12 * np.max(np.abs(beta[z]))


def test_checked_leaf_reports_a_bad_pivot_as_the_library_does():
    rng = np.random.default_rng(5)
    A = _spd(rng, 1)
    A[0, 70, 70] = -1.0
    _, _, info, _ = _run(A, reps=1)
    assert info[0] == 71
