"""Dev harness: correctness + timing of mi_gp_gemm_f64 on the GPU box."""
import ctypes, sys, time
import numpy as np
import torch

lib = ctypes.CDLL("andvaranaut_amd/libmi_gp.so")
lib.mi_gp_last_global_error.restype = ctypes.c_char_p
lib.mi_gp_gemm_f64.argtypes = [ctypes.c_int] * 5 + [ctypes.c_double, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long,
                               ctypes.c_double, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                               ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_void_p]

def gemm(ta, tb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, tri=0, kmode=0, batch=1, sA=0, sB=0, sC=0):
    r = lib.mi_gp_gemm_f64(ta, tb, m, n, k, alpha, A.data_ptr(), lda, B.data_ptr(), ldb, beta, C.data_ptr(), ldc, tri, kmode, batch, sA, sB, sC, None)
    assert r == 0, (r, lib.mi_gp_last_global_error())

dev = torch.device("cuda:0")
torch.manual_seed(0)
ok = True
for (ta, tb) in [(0, 1), (0, 0), (1, 0), (1, 1)]:
    m, n, k = 256, 384, 96
    A = torch.randn((k, m) if ta else (m, k), dtype=torch.float64, device=dev)
    B = torch.randn((n, k) if tb else (k, n), dtype=torch.float64, device=dev)
    C = torch.randn(m, n, dtype=torch.float64, device=dev)
    ref = 0.5 * C - 1.0 * ((A.T if ta else A) @ (B.T if tb else B))
    gemm(ta, tb, m, n, k, -1.0, A, A.shape[1], B, B.shape[1], 0.5, C, n)
    torch.cuda.synchronize()
    err = (C - ref).abs().max().item()
    print(f"ta={ta} tb={tb} max err {err:.3e}")
    ok &= err < 1e-12
# tri + kmodes
m = n = 512; k = 512
A = torch.randn(m, k, dtype=torch.float64, device=dev); B = torch.randn(n, k, dtype=torch.float64, device=dev)
C = torch.zeros(m, n, dtype=torch.float64, device=dev)
gemm(0, 1, m, n, k, 1.0, A, k, B, k, 0.0, C, n, tri=1)
ref = A @ B.T
mask = torch.ones(m, n, device=dev).tril().bool()  # tri guarantees the element-wise lower triangle
bmask = torch.ones(4, 4, device=dev).tril().bool().repeat_interleave(128, 0).repeat_interleave(128, 1)
err = ((C - ref) * mask).abs().max().item(); up = (C * (~bmask)).abs().max().item()
print(f"tri NT err {err:.3e}, untouched upper {up:.1e}"); ok &= err < 1e-11 and up == 0
# kmode 1: NN with B lower-triangular
Bl = torch.randn(k, n, dtype=torch.float64, device=dev).tril()
C.zero_(); gemm(0, 0, m, n, k, 1.0, A, k, Bl, n, 0.0, C, n, kmode=1)
err = (C - A @ Bl).abs().max().item(); print(f"kmode1 err {err:.3e}"); ok &= err < 1e-11
Al = torch.randn(m, k, dtype=torch.float64, device=dev).tril()
Bf = torch.randn(k, n, dtype=torch.float64, device=dev)
C.zero_(); gemm(0, 0, m, n, k, 1.0, Al, k, Bf, n, 0.0, C, n, kmode=2)
err = (C - Al @ Bf).abs().max().item(); print(f"kmode2 err {err:.3e}"); ok &= err < 1e-11
Z = torch.randn(k, k, dtype=torch.float64, device=dev).tril()
C.zero_(); gemm(1, 0, m, n, k, 1.0, Z, k, Z, k, 0.0, C, n, tri=1, kmode=3)
err = ((C - Z.T @ Z) * mask).abs().max().item(); print(f"kmode3 TN tri err {err:.3e}"); ok &= err < 1e-11
# batched
Ab = torch.randn(3, 128, 64, dtype=torch.float64, device=dev); Bb = torch.randn(3, 128, 64, dtype=torch.float64, device=dev)
Cb = torch.zeros(3, 128, 128, dtype=torch.float64, device=dev)
gemm(0, 1, 128, 128, 64, 1.0, Ab, 64, Bb, 64, 0.0, Cb, 128, batch=3, sA=128*64, sB=128*64, sC=128*128)
err = (Cb - Ab @ Bb.transpose(1, 2)).abs().max().item(); print(f"batched err {err:.3e}"); ok &= err < 1e-12
print("CORRECT" if ok else "WRONG")

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
N = 16384
for k in (128, 256, 512, 1024):
    P = torch.randn(N, k, dtype=torch.float64, device=dev)
    Cbig = torch.zeros(N, N, dtype=torch.float64, device=dev)
    ms = timeit(lambda: gemm(0, 1, N, N, k, -1.0, P, k, P, k, 1.0, Cbig, N, tri=1))
    ntile = (N // 128) * (N // 128 + 1) // 2
    fl = ntile * 128 * 128 * k * 2.0
    print(f"SYRK-lower N={N} k={k}: {ms:.3f} ms  {fl/ms*1e-9:.2f} TFLOP/s (tile flops), {N*N*k/ms*1e-9:.2f} TF (n^2k)")
    del P, Cbig
# full GEMM NN
for (m, n, k) in [(8192, 8192, 8192)]:
    A = torch.randn(m, k, dtype=torch.float64, device=dev); B = torch.randn(k, n, dtype=torch.float64, device=dev); C = torch.zeros(m, n, dtype=torch.float64, device=dev)
    ms = timeit(lambda: gemm(0, 0, m, n, k, 1.0, A, k, B, n, 0.0, C, n), reps=3)
    print(f"GEMM NN {m}x{n}x{k}: {ms:.3f} ms {2.0*m*n*k/ms*1e-9:.2f} TFLOP/s")
    ms = timeit(lambda: torch.matmul(A, B, out=C), reps=3)
    print(f"  (rocBLAS dgemm via torch: {ms:.3f} ms {2.0*m*n*k/ms*1e-9:.2f} TFLOP/s)")
