"""The fp64 MFMA GEMM behind the C-ABI (mi_gp_gemm_f64: the SYRK / trapezoid updates of the factorisation, gpmcmc.py:313's
dpotrf internals) against a plain PyTorch fp64 product, on the launch shapes the factorisation issues: the 64x64-tile
kernel, the 128x128-tile kernel, and a 128x128-tile launch whose last partial round is finished on 64x64 tiles."""
import ctypes

import pytest

pytestmark = pytest.mark.gpu


def _run(m, n, k, tri, seed):
    import torch

    from andvaranaut_amd import _lib

    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(seed)
    ld = max(m, n) + 16
    P = torch.randn(m, k + 16, dtype=torch.float64, device=dev, generator=g)[:, :k]  # row-major m x k, ld k + 16
    Q = P if tri else torch.randn(n, k + 16, dtype=torch.float64, device=dev, generator=g)[:, :k]  # the B operand, n x k
    C0 = torch.randn(m, ld, dtype=torch.float64, device=dev, generator=g)
    C = C0.clone()
    r = lib.mi_gp_gemm_f64(0, 1, m, n, k, -1.0, P.data_ptr(), k + 16, Q.data_ptr(), k + 16, 1.0, C.data_ptr(), ld, tri, 0, 1,
                           0, 0, 0, None)
    assert r == 0, lib.mi_gp_last_global_error()
    torch.cuda.synchronize()
    ref = C0[:, :n] - P @ Q[:n].T
    got = C[:, :n]
    if tri:  # element-wise lower trapezoid is guaranteed; the strict upper part of diagonal blocks is unspecified
        mask = torch.tril(torch.ones(m, n, dtype=torch.bool, device=dev))
        err = ((got - ref).abs() * mask).max().item()
        untouched = torch.triu(torch.ones(m, n, dtype=torch.bool, device=dev), diagonal=128)  # tiles above the block diagonal
        assert torch.equal(got[untouched], C0[:, :n][untouched])
    else:
        err = (got - ref).abs().max().item()
    assert torch.equal(C[:, n:], C0[:, n:])  # nothing beyond the n columns is written
    assert err <= 1e-12 * k ** 0.5 * 16, (m, n, k, tri, err)


@pytest.mark.parametrize("m,n,k,tri", [
    (1152, 384, 128, 1),     # 64x64-tile kernel, short k (in-panel update)
    (2048, 2048, 512, 1),    # 64x64-tile kernel, 136 tiles of 128^2
    (6016, 6016, 256, 1),    # 1128 tiles: 128x128 kernel, 1024 + a tail of 104 tiles finished on 64x64 tiles
    (5888, 5888, 128, 1),    # 1081 tiles: tail of 57, among them diagonal tiles (upper quadrant skipped)
    (4096, 4224, 128, 0),    # full rectangle, 1056 tiles: tail of 32
    (8192, 1024, 1024, 1),   # trapezoid with rows below the triangle, 476 tiles (64x64 kernel, several rounds)
    (4096, 4096, 512, 1),    # 528 tiles, triangle only
    (6144, 768, 256, 1),     # 273 tiles, short k
    (4096, 3072, 256, 0),    # rectangle, 768 tiles
    (9216, 2048, 256, 1),    # trapezoid, 1032 tiles: tail of 8
])
def test_gemm_matches_torch_fp64(m, n, k, tri):
    _run(m, n, k, tri, seed=m + n + k)


@pytest.mark.parametrize("m,n,k,small_below", [
    (2048, 512, 512, -1),     # 64x64-tile kernel (few tiles), four segments
    (6144, 1024, 512, 0),     # the same product forced onto the 128x128-tile kernel
    (6016, 6016, 256, -1),    # 128x128-tile kernel with a tail finished on 64x64 tiles
    (1152, 128, 128, -1),     # one segment: the jump is never taken
])
def test_k_segmented_operands_give_the_bits_of_the_plain_layout(m, n, k, small_below):
    """mi_gp_gemm_nt_kseg reads A (= B, a SYRK trapezoid) from a piece-major buffer -- tile column g of the operand in its own
    contiguous [rows][128] piece, the layout of the sharded driver's panel buffers -- and must return exactly what the
    plain row-major operand gives: same tiles, same k order."""
    import torch

    from andvaranaut_amd import _lib

    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(m + k)
    P = torch.randn(m, k + 16, dtype=torch.float64, device=dev, generator=g)[:, :k]
    ld = n + 16
    C0 = torch.randn(m, ld, dtype=torch.float64, device=dev, generator=g)
    stride = (m + 384) * 128  # pieces further apart than they are long, like a buffer sized for the tallest panel
    pieces = torch.zeros(k // 128, stride, dtype=torch.float64, device=dev)
    for s in range(k // 128):
        pieces[s, : m * 128].view(m, 128).copy_(P[:, s * 128: (s + 1) * 128])
    C1, C2 = C0.clone(), C0.clone()
    sb = 1024 if small_below < 0 else small_below
    r = lib.mi_gp_gemm_f64_tuned(0, 1, m, n, k, -1.0, P.data_ptr(), k + 16, P.data_ptr(), k + 16, 1.0, C1.data_ptr(), ld, 1, 0,
                                 sb, 1, 8, 0, None)
    assert r == 0, lib.mi_gp_last_global_error()
    r = lib.mi_gp_gemm_nt_kseg(m, n, k, -1.0, pieces.data_ptr(), 128, pieces.data_ptr(), 128, 128, stride, 1.0, C2.data_ptr(), ld,
                               1, small_below, None)
    assert r == 0, lib.mi_gp_last_global_error()
    torch.cuda.synchronize()
    assert torch.equal(C1, C2)
    assert not torch.equal(C1, C0)


def test_k_segmented_operands_reject_what_they_do_not_cover():
    import torch

    from andvaranaut_amd import _lib

    lib = _lib.load()
    t = torch.zeros(256 * 256, dtype=torch.float64, device="cuda:0")
    assert lib.mi_gp_gemm_nt_kseg(256, 128, 256, -1.0, t.data_ptr(), 128, t.data_ptr(), 128, 96, 128 * 256, 1.0, t.data_ptr(), 256,
                                  1, -1, None) == -1      # segments are whole tile columns
    assert lib.mi_gp_gemm_nt_kseg(256, 128, 384, -1.0, t.data_ptr(), 128, t.data_ptr(), 128, 256, 128 * 256, 1.0, t.data_ptr(), 256,
                                  1, -1, None) == -1      # k must be a whole number of segments
