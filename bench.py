"""Headline benchmark: GP log-marginal-likelihood evaluations per second (BASELINE.json metric)
on the Matern-5/2, N=16384, d=16 workload, one replica (one independent hyper-parameter chain) per GPU.

A "step" is one LML evaluation at a fresh theta: covariance assembly -> blocked fp64 Cholesky with
the forward solve folded in -> log-det / quadratic-form reduction, with X and y resident in HBM.
Ranks are independent chains (SURVEY.md section 8e "replicas only": MAP restarts / MCMC chains,
gpmcmc.py:328-343,351), so there is no data-path collective and scaling is weak.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6  # AMD MI355X datasheet fp64 vector = matrix peak; measured bare-issue rate of
# v_mfma_f64_4x4x4_4b_f64 is 74.8 TFLOP/s (profiles/r01_probe_fp64_rates.txt)


def synth_problem(N, d, seed=0):
    """SURVEY.md section 8d inputs: LHS in [0,1]^d (lhc.py:42-43 recipe), standardised smooth target."""
    from scipy.stats import qmc

    X = qmc.LatinHypercube(d, seed=seed).random(N)
    rng = np.random.default_rng(seed)
    f = np.sin(3.0 * X.sum(1)) + (X ** 2).sum(1) / d
    y = f + rng.normal(0.0, 1e-2, N)
    y = (y - y.mean()) / y.std()
    return np.ascontiguousarray(X), np.ascontiguousarray(y)


def theta_sequence(d, steps, seed):
    """theta per step: the section-8d grid point jittered like successive optimiser / sampler steps."""
    rng = np.random.default_rng(1000 + seed)
    base_ls = np.exp(np.linspace(np.log(0.4), np.log(1.5), d))
    out = []
    for _ in range(steps):
        ls = base_ls * np.exp(rng.normal(0.0, 0.05, d))
        out.append(np.concatenate([ls, [1.7 * np.exp(rng.normal(0.0, 0.05))], [1.0], [1e-4, 1e-6]]))
    return out


def cpu_baseline(N, d, kernel, budget_s=25.0):
    """Oracle (NumPy/SciPy restatement of the reference's PyMC->SciPy LAPACK path) timed on this
    box's host cores on a bounded sample: the same workload at N_s < N, extrapolated to N by
    N^2 (assembly) and N^3 (dpotrf), because one N=16384 evaluation alone is ~1 min of CPU."""
    from oracle import gp_oracle as orc
    import scipy.linalg as sla

    # BLAS threads = the cores this process may actually run on (the GPU box gives a CPU share that is
    # smaller than the machine; oversubscribing OpenBLAS makes dpotrf several times slower)
    try:
        ncpu = len(os.sched_getaffinity(0))
    except Exception:
        ncpu = os.cpu_count() or 1
    try:  # cgroup v2 CPU quota, e.g. "1600000 100000" = 16 cores
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            ncpu = max(1, min(ncpu, int(int(quota) / int(period))))
    except Exception:
        pass
    ncpu = min(ncpu, 16)  # a one-GPU box's CPU share (more OpenBLAS threads than cores only slows dpotrf down)
    try:
        import threadpoolctl

        threadpoolctl.threadpool_limits(limits=ncpu)
        threads = min(ncpu, max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] + [1]))
    except Exception:
        threads = ncpu
    Ns = 4096
    X, y = orc.synth_problem(Ns, d, seed=0)
    theta = orc.synth_theta(d)
    kerns = [kernel]
    t0 = time.perf_counter()
    K = orc.noisy_cov(X, kerns, [], theta)
    t1 = time.perf_counter()
    L = sla.cholesky(K, lower=True)
    t2 = time.perf_counter()
    beta = sla.solve_triangular(L, y, lower=True)
    _ = -0.5 * beta @ beta - np.log(np.diag(L)).sum()
    t3 = time.perf_counter()
    t_asm, t_chol, t_solve = t1 - t0, t2 - t1, t3 - t2
    s = N / Ns
    t_full = t_asm * s ** 2 + t_chol * s ** 3 + t_solve * s ** 2
    return {
        "value": 1.0 / t_full,
        "unit": "evals/s",
        "cores": int(threads),
        "kind": "port",
        "sample": f"oracle LML at N={Ns} d={d} {kernel}: assembly {t_asm:.2f}s dpotrf {t_chol:.2f}s "
                  f"({Ns ** 3 / 3 / t_chol * 1e-9:.0f} GFLOP/s) solve {t_solve:.3f}s; extrapolated to N={N} "
                  f"by N^2/N^3 -> {t_full:.1f}s per eval",
    }


def sharded_main(args, X, y, rank, world, dev):
    """Strong-scaling variant: every rank owns a block-cyclic share of the 512/1024-column panels of ONE
    covariance (andvaranaut_amd/distributed.py); one broadcast per panel over RCCL."""
    import torch
    import torch.distributed as dist

    from andvaranaut_amd.distributed import DistGP

    N, d = X.shape
    gp = DistGP(X, y, args.kernel, device=dev.index)
    thetas = theta_sequence(d, args.warmup + args.steps, seed=0)  # same theta on every rank
    step = (lambda th: gp.lml_grad(th)[0]) if args.grad else gp.lml
    for i in range(args.warmup):
        step(thetas[i])
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    vals = [step(thetas[args.warmup + i]) for i in range(args.steps)]
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert all(np.isfinite(v) for v in vals)
    if rank == 0:
        print(json.dumps({
            "metric": "gp_lml_grad_evals_per_s" if args.grad else "gp_lml_evals_per_s", "value": args.steps / elapsed,
            "unit": "evals/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.kernel} GP LML eval, ONE covariance N={N} d={d} sharded over {world} GPU(s)",
                       "N": N, "d": d, "kernel": args.kernel, "parallelism": f"column-panel ({gp.pw} columns) block-cyclic x{world}"},
            "cholesky_tflops_whole_eval": (N ** 3 / 3.0) / (elapsed / args.steps) * 1e-12}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--d", type=int, default=16)
    ap.add_argument("--kernel", default="Matern52")
    ap.add_argument("--panel-tiles", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-steps", type=int, default=3)
    ap.add_argument("--sharded", action="store_true", help="ONE covariance sharded over all ranks (strong scaling, panel broadcast)")
    ap.add_argument("--grad", action="store_true", help="with --sharded: time LML + gradient (sharded K^-1) instead of the LML")
    ap.add_argument("--no-lookahead", action="store_true", help="disable the look-ahead stream everywhere (profiling aid)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from andvaranaut_amd import MiGP

    N, d = args.n, args.d
    X, y = synth_problem(N, d, seed=0)
    if args.sharded:
        return sharded_main(args, X, y, rank, world, dev)
    gp = MiGP(X, y, args.kernel, device=local_rank, panel_tiles=args.panel_tiles, need_grad=False)
    thetas = theta_sequence(d, args.warmup + args.steps, seed=rank)
    if args.no_lookahead:
        gp.set_option(0, 0)

    for i in range(args.warmup):
        gp.lml(thetas[i])
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    vals = []
    for i in range(args.steps):
        vals.append(gp.lml(thetas[args.warmup + i]))  # synchronous on return (stream-synchronised)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert all(np.isfinite(v) for v in vals), "non-finite LML in the timed region"

    # Roofline pass (rank 0): the same evaluations again with HIP events on the handle's own stream
    # around every phase and every GEMM launch.  Look-ahead is switched off for this pass so that
    # the dominant kernel runs alone on the chip and its launch durations are not inflated by the
    # panel kernels that overlap it in the timed region (graph replay is bypassed by profiling).
    acc = {"assemble_ms": 0.0, "cholesky_ms": 0.0, "gemm_ms": 0.0, "gemm_flops": 0.0, "gemm_launches": 0.0, "total_ms": 0.0,
           "gemm_b_ms": 0.0, "gemm_b_flops": 0.0, "gemm_b_launches": 0.0}
    rsteps = max(1, min(args.roofline_steps, args.steps))
    if rank == 0:
        gp.set_option(0, 0)
        gp.set_profiling(2)
        for i in range(rsteps):
            gp.lml(thetas[args.warmup + i])
            tm = gp.timers()
            for k in acc:
                acc[k] += tm[k]
        gp.set_profiling(0)
        gp.set_option(0, 0 if args.no_lookahead else 1)

    if rank == 0:
        steps = args.steps
        # dominant kernel = gemm_f64_kernel_b (128x128 tiles); the 64x64-tile kernel that serves the small
        # in-panel updates is reported next to it
        gemm_avg_ms = acc["gemm_b_ms"] / max(acc["gemm_b_launches"], 1.0)
        achieved = acc["gemm_b_flops"] / (acc["gemm_b_ms"] * 1e-3) * 1e-12 if acc["gemm_b_ms"] > 0 else 0.0
        all_gemm = acc["gemm_flops"] / (acc["gemm_ms"] * 1e-3) * 1e-12 if acc["gemm_ms"] > 0 else 0.0
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "gemm_traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "gp_lml_evals_per_s",
            "value": world * steps / elapsed,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.kernel} GP LML eval (assembly + Cholesky + solve), N={N} d={d}, LHS inputs",
                       "N": N, "d": d, "kernel": args.kernel, "parallelism": f"replicas x{world} (one chain per GPU)"},
            "cholesky_tflops_whole_eval": (N ** 3 / 3.0) / (elapsed / steps) * 1e-12,
            "roofline_pass": {"steps": rsteps, "lookahead": False,
                              "phase_ms": {"assemble": acc["assemble_ms"] / rsteps, "cholesky": acc["cholesky_ms"] / rsteps,
                                           "gemm_in_cholesky": acc["gemm_ms"] / rsteps}},
            # K1/K2: lower-triangle 64x64 tiles written once + X read once (8*N*d): HBM-side figure the
            # north star asks for next to the MFMA one
            "assembly": {"kernel": "assemble_kernel", "ms": acc["assemble_ms"] / rsteps,
                         "algorithmic_bytes": 8.0 * (N // 64) * (N // 64 + 1) / 2 * 64 * 64 + 8.0 * N * d,
                         "achieved_GBps": (8.0 * (N // 64) * (N // 64 + 1) / 2 * 64 * 64 + 8.0 * N * d)
                         / (acc["assemble_ms"] / rsteps * 1e-3) * 1e-9, "peak_GBps": 8000.0},
            "roofline": {"kernel": "gemm_f64_kernel_b (SYRK trailing/panel updates, v_mfma_f64_16x16x4_f64)",
                         "bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic,
                         "avg_launch_ms": gemm_avg_ms, "launches_per_step": acc["gemm_b_launches"] / rsteps,
                         "flop_share_of_all_gemm": acc["gemm_b_flops"] / max(acc["gemm_flops"], 1.0),
                         "all_gemm_kernels_tflops": all_gemm, "all_gemm_launches_per_step": acc["gemm_launches"] / rsteps},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(N, d, args.kernel)
        print(json.dumps(line), flush=True)
    gp.close()
    if world > 1:
        dist.barrier()  # rank 0's roofline pass runs after the timed region: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
