"""Perf-only harness for the GEMM kernel (used under rocprofv3 --pmc)."""
import ctypes, sys
import torch
import os
lib = ctypes.CDLL(os.environ.get("MIGP_LIB", "/root/repo/andvaranaut_amd/libmi_gp.so"))
lib.mi_gp_gemm_f64.argtypes = [ctypes.c_int] * 5 + [ctypes.c_double, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long,
                               ctypes.c_double, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                               ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_void_p]
dev = torch.device("cuda:0")
torch.manual_seed(0)
def gemm(ta, tb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, tri=0):
    r = lib.mi_gp_gemm_f64(ta, tb, m, n, k, alpha, A.data_ptr(), lda, B.data_ptr(), ldb, beta, C.data_ptr(), ldc, tri, 0, 1, 0, 0, 0, None)
    assert r == 0
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
if mode in ("all", "nn"):
    m = n = k = 8192
    A = torch.randn(m, k, dtype=torch.float64, device=dev); B = torch.randn(k, n, dtype=torch.float64, device=dev); C = torch.zeros(m, n, dtype=torch.float64, device=dev)
    ms = timeit(lambda: gemm(0, 0, m, n, k, 1.0, A, k, B, n, 0.0, C, n))
    print(f"GEMM NN 8192^3: {ms:.3f} ms {2.0*m*n*k/ms*1e-9:.2f} TFLOP/s")
if mode in ("all", "syrk"):
    N = 16384
    for k in (256, 1024):
        lda = N + 16
        P = torch.randn(N, k, dtype=torch.float64, device=dev)
        Cbig = torch.zeros(N, lda, dtype=torch.float64, device=dev)
        ms = timeit(lambda: gemm(0, 1, N, N, k, -1.0, P, k, P, k, 1.0, Cbig, lda, tri=1))
        print(f"SYRK-lower N={N} k={k}: {ms:.3f} ms  {N*(N+1.0)*k/ms*1e-9:.2f} TFLOP/s algorithmic")
if mode in ("all", "rocblas"):
    m = n = k = 8192
    A = torch.randn(m, k, dtype=torch.float64, device=dev); B = torch.randn(k, n, dtype=torch.float64, device=dev); C = torch.zeros(m, n, dtype=torch.float64, device=dev)
    ms = timeit(lambda: torch.matmul(A, B, out=C))
    print(f"rocBLAS dgemm 8192^3: {ms:.3f} ms {2.0*m*n*k/ms*1e-9:.2f} TFLOP/s")
