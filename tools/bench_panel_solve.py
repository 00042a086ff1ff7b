"""Stand-alone timing of the 'panel solve as ONE GEMM' shape: X (m x 1024) = A21 (m x 1024) . U11 (1024 x 1024, upper
triangular: k < (tj + 1) * 128), out of place, against the launches the recursive in-panel form issues for the same rows.
python tools/bench_panel_solve.py [reps]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import _lib
lib = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
torch.manual_seed(0)
ld = 16384 + 16
A = torch.randn(16384 + 128, ld, dtype=torch.float64, device=dev) * 0.01
Wk = torch.zeros(16384 + 128, ld, dtype=torch.float64, device=dev)
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
for m in (15360, 12288, 8192, 4096, 2048, 1024):
    A21 = A[1024:1024 + m, 0:1024]; U = A[0:1024, 2048:3072]; C = Wk[1024:1024 + m, 0:1024]
    def solve():
        # NN: A row-major [x][k], B row-major k x n -> [k][x]; kmode 4: k < (tj + 1) * 128
        r = lib.mi_gp_gemm_f64(0, 0, m, 1024, 1024, 1.0, A21.data_ptr(), ld, U.data_ptr(), ld, 0.0, C.data_ptr(), ld, 0, 4, 1, 0, 0, 0, None)
        assert r == 0
    t = timeit(solve)
    flops = m * 1024.0 * (1024 + 128)  # sum over column tiles of 2 * m * 128 * 128 (tj + 1)
    # the recursive form's in-panel updates for the same rows: k = 128 x4 (1 col), 256 x2 (2 cols), 512 x1 (4 cols), NT, beta 1
    def inpanel():
        for (nc, k, c0, k0) in ((1, 128, 1, 0), (2, 256, 2, 0), (1, 128, 3, 2), (4, 512, 4, 0), (1, 128, 5, 4), (2, 256, 6, 4), (1, 128, 7, 6)):
            P = A[1024:1024 + m, k0 * 128:k0 * 128 + k]; Pc = A[c0 * 128:(c0 + nc) * 128, k0 * 128:k0 * 128 + k]
            Cc = Wk[1024:1024 + m, c0 * 128:(c0 + nc) * 128]
            r = lib.mi_gp_gemm_f64(0, 1, m, nc * 128, k, -1.0, P.data_ptr(), ld, Pc.data_ptr(), ld, 1.0, Cc.data_ptr(), ld, 0, 0, 1, 0, 0, 0, None)
            assert r == 0
    t2 = timeit(inpanel)
    f2 = 2.0 * m * (4 * 128 * 128 + 2 * 256 * 256 + 512 * 512)
    print(f"m={m:6d}: one triangular-k GEMM {t:7.1f} us ({flops / t * 1e-6:5.1f} TFLOP/s) | seven in-panel updates {t2:7.1f} us ({f2 / t2 * 1e-6:5.1f} TFLOP/s, + 7 strips not timed)", flush=True)
