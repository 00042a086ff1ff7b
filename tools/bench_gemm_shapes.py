"""Stand-alone timings of the trapezoid updates an N=16384 / N=8192 factorisation issues on the 64x64-tile kernel
(next-panel update k=1024, in-panel halvings k=512/256/128), through mi_gp_gemm_f64 with HIP events.
    python tools/bench_gemm_shapes.py [reps]"""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import _lib
if os.environ.get("MIGP_LIB"):  # A/B of library builds: MIGP_LIB=tools/ab/lib_x.so
    _lib.LIB_PATH = os.path.abspath(os.environ["MIGP_LIB"])
lib = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SMALL_BELOW = int(os.environ.get("SMALL_BELOW", "1024"))  # launches with fewer 128x128 tiles run on 64x64 tiles
TAIL = int(os.environ.get("TAIL", "1"))
BAND = int(os.environ.get("BAND", "8"))
dev = torch.device("cuda:0")
torch.manual_seed(0)
ld = 16384 + 16
A = torch.randn(16384 + 128, ld, dtype=torch.float64, device=dev) * 0.01
shapes = [(3072, 2944, 32), (3072, 2944, 64), (3072, 2944, 128), (3072, 2944, 256), (3072, 2944, 512), (4096, 3968, 128), (2048, 1920, 128), (8192, 7168, 1024), (6144, 5120, 512), (4096, 3072, 512), (4096, 3072, 128), (15360, 1024, 1024), (15360, 896, 1024), (15360, 512, 512), (15360, 256, 256), (15360, 128, 128),
          (8192, 1024, 1024), (8192, 512, 512), (8192, 128, 128), (4096, 1024, 1024), (4096, 512, 512), (4096, 128, 128),
          (2048, 512, 512), (2048, 128, 128)]
out = []
for (m, n, k) in shapes:
    P, C = A[:m, 2048:2048 + k], A[:m, 4096:4096 + n]
    def run():
        r = lib.mi_gp_gemm_f64_tuned(0, 1, m, n, k, -1.0, P.data_ptr(), ld, P.data_ptr(), ld, 1.0, C.data_ptr(), ld, 1, 0,
                                     SMALL_BELOW, TAIL, BAND, int(os.environ.get("ONE_PER_CU", "0")), None)
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
    flops = k * (n * (n + 128.0) + 2.0 * (m - n) * n)  # tiles on and below the block diagonal, as issued
    rec = {"m": m, "n": n, "k": k, "us_median": med, "us_min": ts[0], "tflops": flops / med * 1e-6}
    out.append(rec)
    print(json.dumps(rec), flush=True)
json.dump(out, open("gpurun_out/gemm_shapes.json", "w"), indent=1)
