"""Alone-on-the-chip rate of bulk-update shapes on both GEMM kernels at one and at two workgroups per CU, through
mi_gp_gemm_f64_tuned with HIP events:  python tools/bench_gemm_occ.py [reps]
(MIGP_LIB=<libmi_gp.so>: another build; ONLY="m,n,k,64x64|128x128,0|1": one shape, kernel and occupancy -- for counter passes)"""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import _lib
if os.environ.get("MIGP_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MIGP_LIB"])
lib = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
torch.manual_seed(0)
ld = 16384 + 16
A = torch.randn(16384 + 128, ld, dtype=torch.float64, device=dev) * 0.01
# (m, n, k): the bulk updates of an N = 8192 evaluation (48 / 40 / 32 trailing tile columns), an (a2) update, N = 16384's 1536-tile part
shapes = [(6272, 6144, 1024), (5248, 5120, 1024), (4224, 4096, 1024), (7168, 896, 1024), (3200, 3072, 1024), (3200, 3072, 512)]
if os.environ.get("ONLY"):  # ONLY="5248,5120,1024,64x64,0": one shape, one kernel, one occupancy (counter passes)
    o = os.environ["ONLY"].split(",")
    shapes = [(int(o[0]), int(o[1]), int(o[2]))]
for (m, n, k) in shapes:
    for small_below, name in ((1 << 20, "64x64"), (0, "128x128")):
        for opc in (0, 1):
            if os.environ.get("ONLY") and (name != o[3] or opc != int(o[4])):
                continue
            P, C = A[:m, 8192:8192 + k], A[:m, 0:n]
            def run():
                r = lib.mi_gp_gemm_f64_tuned(0, 1, m, n, k, -1.0, P.data_ptr(), ld, P.data_ptr(), ld, 1.0, C.data_ptr(), ld, 1, 0,
                                             small_below, 0, 8, opc, None)
                assert r == 0
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(reps):
                e0.record(); run(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            med = ts[len(ts) // 2]
            flops = k * (n * (n + 128.0) + 2.0 * (m - n) * n)
            print(f"m={m} n={n} k={k} {name:8s} {'one' if opc else 'two'} per CU: {med:8.1f} us  {flops / med * 1e-6:6.1f} TFLOP/s", flush=True)
