"""Latency of the small / short-k launches that sit on the factorisation's critical path (64x64-tile GEMM kernel, leaf, strip),
each as 20 back-to-back launches timed with events (run it under rocprofv3 --kernel-trace --stats for pure kernel durations)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
REP = 20

def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REP * 1e3  # us

ld = 16384 + 16
A = torch.randn(16384 + 128, ld, dtype=torch.float64, device=dev)
# trapezoid updates as the Cholesky issues them: C[m x n] -= P[m x k] P[0:n, :]^T, lower trapezoid
for (m, n, k) in [(3584, 128, 128), (3584, 128, 256), (3584, 128, 512), (3584, 384, 512), (3584, 512, 512), (8192, 128, 128),
                  (8192, 128, 512), (8192, 384, 512), (16384, 128, 128), (16384, 128, 1024), (16384, 896, 1024), (2048, 2048, 512),
                  (4096, 4096, 512), (8192, 8192, 512)]:
    P = A[:m, 2048:2048 + k]
    C = A[:m, 4096:4096 + n]
    def f():
        r = lib.mi_gp_gemm_f64(0, 1, m, n, k, -1.0, P.data_ptr(), ld, P.data_ptr(), ld, 1.0, C.data_ptr(), ld, 1, 0, 1, 0, 0, 0, None)
        assert r == 0
    us = timeit(f)
    nt, mt = n // 128, m // 128
    tiles = nt * (nt + 1) // 2 + (mt - nt) * nt
    flops = k * (n * (n + 1.0) + 2.0 * (m - n) * n)
    print(f"trapezoid m={m:6d} n={n:5d} k={k:5d}: {us:8.1f} us  {flops / us * 1e-6:6.2f} TFLOP/s  ({tiles} tiles of 128^2, {'64x64' if tiles < 1024 else '128x128'} kernel)", flush=True)
# one panel column: leaf + strip (unfused) through mi_gp_chol_panel with w_tiles = 1
K = torch.eye(16384 + 128, ld, dtype=torch.float64, device=dev) * 4.0 + 0.01 * torch.randn(16384 + 128, ld, dtype=torch.float64, device=dev)
K = torch.tril(K) + torch.tril(K, -1).T.contiguous() if False else K
dinv = torch.zeros(16384, dtype=torch.float64, device=dev)
info = torch.zeros(4, dtype=torch.int32, device=dev)
for rows in (1, 33, 65, 129):
    W = K[: rows * 128].clone()
    W[:128, :128] = torch.eye(128, dtype=torch.float64, device=dev) * 4.0 + 0.001
    def g():
        W[:128, :128] = torch.eye(128, dtype=torch.float64, device=dev) * 4.0 + 0.001
        r = lib.mi_gp_chol_panel(W.data_ptr(), ld, rows, 1, dinv.data_ptr(), info.data_ptr(), 0, None)
        assert r == 0
    print(f"leaf + strip, {rows - 1} row tiles below: {timeit(g):8.1f} us (includes two small torch launches)", flush=True)
