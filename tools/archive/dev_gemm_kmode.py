"""NN / NT with the triangular k-ranges of the inverse (kmode 3: k >= ti*128, kmode 4: k < (tj+1)*128)."""
import ctypes, sys, os
import torch
lib = ctypes.CDLL(os.environ.get("MIGP_LIB", "/root/repo/andvaranaut_amd/libmi_gp.so"))
lib.mi_gp_gemm_f64.argtypes = [ctypes.c_int] * 5 + [ctypes.c_double, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long,
                               ctypes.c_double, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                               ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_void_p]
dev = torch.device("cuda:0")
torch.manual_seed(0)
n = 8192
ld = n + 16
A = torch.randn(n, ld, dtype=torch.float64, device=dev)
B = torch.randn(n, ld, dtype=torch.float64, device=dev)
C = torch.zeros(n, ld, dtype=torch.float64, device=dev)
def timeit(fn, reps=4):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
for ta, tb, name in ((0, 1, "NT"), (0, 0, "NN"), (1, 0, "TN")):
    for kmode in (0, 3, 4):
        def run():
            r = lib.mi_gp_gemm_f64(ta, tb, n, n, n, 1.0, A.data_ptr(), ld, B.data_ptr(), ld, 0.0, C.data_ptr(), ld, 0, kmode, 1, 0, 0, 0, None)
            assert r == 0
        ms = timeit(run)
        fl = 2.0 * n * n * n * (1.0 if kmode == 0 else (0.5 + 64.0 / n))
        print(f"{name} kmode {kmode}: {ms:7.3f} ms  {fl/ms*1e-9:6.2f} TF", flush=True)
