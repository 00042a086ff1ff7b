"""One trapezoid update shape through mi_gp_gemm_f64, for rocprofv3 --pmc passes: python tools/dev_gemm_trap.py m n k"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import _lib
lib = _lib.load()
m, n, k = (int(a) for a in sys.argv[1:4])
dev = torch.device("cuda:0")
torch.manual_seed(0)
ld = 16384 + 16
A = torch.randn(m, ld, dtype=torch.float64, device=dev)
P, C = A[:, 2048:2048 + k], A[:, 4096:4096 + n]
for it in range(6):
    r = lib.mi_gp_gemm_f64(0, 1, m, n, k, -1.0, P.data_ptr(), ld, P.data_ptr(), ld, 1.0, C.data_ptr(), ld, 1, 0, 1, 0, 0, 0, None)
    assert r == 0
torch.cuda.synchronize()
print("done")
