"""How much of the short-k SYRK time is the C read-modify-write?  beta = 1 vs beta = 0 on the same shapes."""
import ctypes, sys, os
import torch
lib = ctypes.CDLL(os.environ.get("MIGP_LIB", "/root/repo/andvaranaut_amd/libmi_gp.so"))
lib.mi_gp_gemm_f64.argtypes = [ctypes.c_int] * 5 + [ctypes.c_double, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long,
                               ctypes.c_double, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                               ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_void_p]
dev = torch.device("cuda:0")
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
NB = 16384
lda = NB + 16
M = torch.randn(NB, lda, dtype=torch.float64, device=dev)
for n, tri in ((15360, 1), (8192, 0), (8192, 1), (5632, 0)):
    for k in (256, 512, 1024, 2048, 4096):
        r0 = NB - n
        if r0 < k and tri: continue
        A_ptr = M.data_ptr() + 8 * (r0 * lda)
        C_ptr = M.data_ptr() + 8 * (r0 * lda + (r0 if tri else 8192))
        if not tri and k > 8192 - 16: continue
        out = []
        for beta in (1.0, 0.0):
            def run():
                r = lib.mi_gp_gemm_f64(0, 1, n, n, k, -1.0, A_ptr, lda, A_ptr, lda, beta, C_ptr, lda, tri, 0, 1, 0, 0, 0, None)
                assert r == 0
            ms = timeit(run)
            tiles = (n // 128) * (n // 128 + 1) // 2 if tri else (n // 128) ** 2
            out.append((ms, 2.0 * tiles * 128 * 128 * k / ms * 1e-9))
        print(f"n={n:6d} tri={tri} k={k:5d} tiles/256={tiles/256:6.2f}: beta=1 {out[0][0]:7.3f} ms {out[0][1]:6.2f} TF | beta=0 {out[1][0]:7.3f} ms {out[1][1]:6.2f} TF", flush=True)
