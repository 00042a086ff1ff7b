"""One shape, ours and rocBLAS, for rocprofv3 --pmc passes (clock / MFMA utilisation comparison)."""
import ctypes, sys, os
import torch
lib = ctypes.CDLL(os.environ.get("MIGP_LIB", "/root/repo/andvaranaut_amd/libmi_gp.so"))
lib.mi_gp_gemm_f64.argtypes = [ctypes.c_int] * 5 + [ctypes.c_double, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long,
                               ctypes.c_double, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                               ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_void_p]
dev = torch.device("cuda:0")
torch.manual_seed(0)
n, k = 8192, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
A = torch.randn(n, k, dtype=torch.float64, device=dev)
B = torch.randn(n, k, dtype=torch.float64, device=dev)
C = torch.zeros(n, n, dtype=torch.float64, device=dev)
for it in range(4):
    r = lib.mi_gp_gemm_f64(0, 1, n, n, k, -1.0, A.data_ptr(), k, B.data_ptr(), k, 1.0, C.data_ptr(), n, 0, 0, 1, 0, 0, 0, None)
    assert r == 0
    torch.cuda.synchronize()
    torch.matmul(A, B.T, out=C)
    torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e2 = torch.cuda.Event(enable_timing=True)
e0.record()
lib.mi_gp_gemm_f64(0, 1, n, n, k, -1.0, A.data_ptr(), k, B.data_ptr(), k, 1.0, C.data_ptr(), n, 0, 0, 1, 0, 0, 0, None)
e1.record()
torch.matmul(A, B.T, out=C)
e2.record(); torch.cuda.synchronize()
print(f"ours {2.0*n*n*k/e0.elapsed_time(e1)*1e-9:.2f} TF  rocBLAS {2.0*n*n*k/e1.elapsed_time(e2)*1e-9:.2f} TF")
