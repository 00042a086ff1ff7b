"""SYRK-shaped sweep of the GEMM kernel against rocBLAS dgemm NT of the same k (vendor reference for short-k)."""
import ctypes, sys, os
import torch
lib = ctypes.CDLL(os.environ.get("MIGP_LIB", "/root/repo/andvaranaut_amd/libmi_gp.so"))
lib.mi_gp_gemm_f64.argtypes = [ctypes.c_int] * 5 + [ctypes.c_double, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long,
                               ctypes.c_double, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                               ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_void_p]
dev = torch.device("cuda:0")
torch.manual_seed(0)
def timeit(fn, reps=4):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
NB = 16384
lda = NB + 16
M = torch.randn(NB, lda, dtype=torch.float64, device=dev)
for n in (15360, 12288, 8192, 4096, 2048):
    for k in (512, 1024, 2048):
        # in-place layout of the factorisation: P = M[r0:, 0:k], C = M[r0:, r0:] lower trapezoid
        r0 = NB - n
        if r0 < k: continue
        A_ptr = M.data_ptr() + 8 * (r0 * lda)
        C_ptr = M.data_ptr() + 8 * (r0 * lda + r0)
        def run():
            r = lib.mi_gp_gemm_f64(0, 1, n, n, k, -1.0, A_ptr, lda, A_ptr, lda, 1.0, C_ptr, lda, 1, 0, 1, 0, 0, 0, None)
            assert r == 0
        ms = timeit(run)
        tiles = (n // 128) * (n // 128 + 1) // 2
        fl = 2.0 * tiles * 128 * 128 * k
        A = M[r0:, :k]
        Cx = torch.empty(n, n, dtype=torch.float64, device=dev)
        ms_r = timeit(lambda: torch.matmul(A, A.T, out=Cx))
        print(f"n={n:6d} k={k:5d} tiles={tiles:5d} ({tiles/512:6.2f} rounds): {ms:7.3f} ms {fl/ms*1e-9:6.2f} TF issued | rocBLAS full NT {ms_r:7.3f} ms {2.0*n*n*k/ms_r*1e-9:6.2f} TF", flush=True)
