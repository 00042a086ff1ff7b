"""CPU restatement (NumPy/SciPy LAPACK on the box's host cores) beside the device, per BASELINE config shape
(SURVEY 8d 'CPU baseline beside it').  The oracle is used here as a timed baseline only."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import gp_oracle as orc
from andvaranaut_amd import MiGP
from bench import host_cores
ncpu = min(host_cores(), 16)  # the one-GPU box's CPU share, as in bench.py's cpu_baseline (OpenBLAS with more threads than the quota is slower)
try:
    import threadpoolctl
    threadpoolctl.threadpool_limits(limits=ncpu)
    nthr = min(ncpu, max([i.get("num_threads", 1) for i in threadpoolctl.threadpool_info()] + [1]))
except Exception:
    nthr = ncpu
rows = []
for name, N, d, kern in (("C1", 128, 2, "RBF"), ("-", 1024, 4, "RBF"), ("C2", 4096, 8, "RBF"), ("C5", 8192, 8, "RBF")):
    X, y = orc.synth_problem(N, d, seed=0)
    theta = orc.synth_theta(d)
    t0 = time.perf_counter(); v = orc.lml(X, y, [kern], [], theta); t_lml = time.perf_counter() - t0
    t0 = time.perf_counter(); orc.lml_grad(X, y, [kern], [], theta); t_grad = time.perf_counter() - t0
    gp = MiGP(X, y, kern)
    for _ in range(3): gp.lml(theta); gp.lml_grad(theta)
    reps = 20 if N <= 4096 else 5
    t0 = time.perf_counter()
    for _ in range(reps): vd = gp.lml(theta)
    g_lml = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps): gp.lml_grad(theta)
    g_grad = (time.perf_counter() - t0) / reps
    gp.close()
    rows.append({"config": name, "N": N, "d": d, "cpu_lml_s": t_lml, "cpu_lml_grad_s": t_grad, "gpu_lml_s": g_lml, "gpu_lml_grad_s": g_grad,
                 "rel_diff": abs(vd - v) / abs(v)})
    print(rows[-1], flush=True)
out = {"blas_threads": nthr, "host_cores_usable": ncpu, "machine_cpu_count": os.cpu_count(), "rows": rows}
json.dump(out, open("gpurun_out/cpu_baseline_table.json", "w"), indent=1)
