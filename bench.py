"""Headline benchmark: GP log-marginal-likelihood evaluations per second (BASELINE.json metric)
on the Matern-5/2, N=16384, d=16 workload, one replica (one independent hyper-parameter chain) per GPU.

A "step" is one LML evaluation at a fresh theta: covariance assembly -> blocked fp64 Cholesky with
the forward solve folded in -> log-det / quadratic-form reduction, with X and y resident in HBM.
Ranks are independent chains (SURVEY.md section 8e "replicas only": MAP restarts / MCMC chains,
gpmcmc.py:328-343,351), so there is no data-path collective in the timed region and scaling is weak.
The same JSON line carries a ``sharded`` sub-record: ONE covariance (BASELINE config 4: RBF, N=65536,
d=32) column-panel sharded over all ranks with one RCCL broadcast per panel (strong scaling).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N ...        # starts N ranks itself (torch.distributed.run) when WORLD_SIZE is unset
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6  # AMD MI355X datasheet fp64 vector = matrix peak (256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz);
# measured bare-issue rate of v_mfma_f64_16x16x4_f64 with VGPR accumulators: 77.0 TFLOP/s (profiles/r01_probe_mfma_f64_16x16x4.txt)


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


def reference_theta(d):
    """SURVEY.md section 8d grid point (== oracle.synth_theta(d)): the theta of the parity record in cpu_baseline."""
    return np.concatenate([np.exp(np.linspace(np.log(0.4), np.log(1.5), d)), [1.7], [1.0], [1e-4, 1e-6]])


def host_cores():
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota."""
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
    return ncpu


def cpu_baseline(N, d, kernel, full=True, gpu_lml=None):
    """Oracle (NumPy/SciPy restatement of the reference's PyMC -> SciPy LAPACK path) timed on this box's host
    cores.  ``full``: ONE LML evaluation at the benchmark's own N (measured, ~10 s of CPU at N=16384) whose VALUE is
    kept and compared with the device's LML on the same X, y, theta (``rel_diff_vs_gpu``: the north star's rtol 1e-10);
    the LML + gradient leg is timed at min(N, 8192) and scaled by N^3 (a full one at N=16384 is ~150 s).
    BLAS threads = the box's CPU share for one GPU (16), not the machine's core count: OpenBLAS with more threads
    than the cgroup quota allows is several times slower."""
    from oracle import gp_oracle as orc
    import scipy.linalg as sla

    ncpu = min(host_cores(), 16)
    try:
        import threadpoolctl

        threadpoolctl.threadpool_limits(limits=ncpu)
        threads = min(ncpu, max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] + [1]))
    except Exception:
        threads = ncpu

    def one(Ns):
        X, y = orc.synth_problem(Ns, d, seed=0)
        theta = orc.synth_theta(d)
        assert np.array_equal(theta, reference_theta(d))
        t0 = time.perf_counter()
        K = orc.noisy_cov(X, [kernel], [], theta)
        t1 = time.perf_counter()
        L = sla.cholesky(K, lower=True, overwrite_a=True, check_finite=False)
        t2 = time.perf_counter()
        beta = sla.solve_triangular(L, y, lower=True, check_finite=False)
        val = -0.5 * Ns * np.log(2.0 * np.pi) - 0.5 * beta @ beta - np.log(np.diag(L)).sum()
        t3 = time.perf_counter()
        return t1 - t0, t2 - t1, t3 - t2, float(val)

    Ns = N if full else min(N, 4096)
    t_asm, t_chol, t_solve, cpu_val = one(Ns)
    s = N / Ns
    t_full = t_asm * s ** 2 + t_chol * s ** 3 + t_solve * s ** 2
    # LML + analytic gradient (K^-1 via dpotri-style solves + contraction), bounded sample
    Ng = min(N, 8192 if full else 4096)
    Xg, yg = orc.synth_problem(Ng, d, seed=0)
    t0 = time.perf_counter()
    orc.lml_grad(Xg, yg, [kernel], [], orc.synth_theta(d))
    t_grad_meas = time.perf_counter() - t0
    t_grad = t_grad_meas * (N / Ng) ** 3
    how = "measured at the full size, one evaluation" if Ns == N else f"measured at N={Ns}, extrapolated by N^2/N^3"
    parity = {}
    if Ns == N and gpu_lml is not None:
        # same X, y (synth_problem seed 0) and theta (oracle.synth_theta) as the device evaluation `gpu_lml`
        parity = {"lml_cpu": cpu_val, "lml_gpu": float(gpu_lml),
                  "rel_diff_vs_gpu": abs(cpu_val - float(gpu_lml)) / abs(cpu_val), "rtol_target": 1e-10}
    return {
        **parity,
        "value": 1.0 / t_full,
        "unit": "evals/s",
        "cores": int(threads),
        "kind": "port",
        "measured_at_full_size": bool(Ns == N),
        "sample": f"oracle LML at N={Ns} d={d} {kernel} ({how}): assembly {t_asm:.2f}s dpotrf {t_chol:.2f}s "
                  f"({Ns ** 3 / 3 / t_chol * 1e-9:.0f} GFLOP/s) solve {t_solve:.3f}s -> {t_full:.1f}s per eval; "
                  f"BLAS threads {threads} of {os.cpu_count()} machine cores (the one-GPU box's CPU share)",
        "lml_grad_value": 1.0 / t_grad,
        "lml_grad_sample": f"oracle LML+grad measured at N={Ng} ({t_grad_meas:.1f}s), scaled by (N/{Ng})^3 -> {t_grad:.0f}s per eval",
    }


# ------------------------------------------------------------------------------------------------ launching ranks
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args):
    """``--gpus N`` without a launcher: start N fresh ranks (one per GPU) BEFORE this process touches HIP --
    torch.cuda.device_count() does not initialise the runtime on this image -- and exit with their code."""
    import torch

    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.exit(f"bench.py --gpus {args.gpus}: needs {args.gpus} GPUs, this box has {have}")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd).returncode)


class Deadline:
    """Safety net around the sharded sub-record (its RCCL transport cannot be rehearsed on a one-GPU box): if it
    has not finished in time, rank 0 prints the line it already has and every rank leaves."""

    def __init__(self, seconds, on_expire):
        self.timer = threading.Timer(seconds, on_expire)
        self.timer.daemon = True

    def __enter__(self):
        self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()
        return False


# ------------------------------------------------------------------------------------------------ sharded (config 4)
def sharded_record(args, rank, world, dev, N, d, kernel, steps, warmup, grad=False, exchange="bcast", lazy_sends=None,
                   single_gpu_check=True):
    """ONE covariance sharded over all ranks (andvaranaut_amd/distributed.py): block-cyclic column panels, the owner
    factors a panel and broadcasts it (RCCL), every rank updates the panels it owns.  Same theta on every rank.
    ``exchange`` / ``lazy_sends``: DistGP.set_exchange / DistGP.lazy_sends (defaults: broadcast, the owner waits for its sends)."""
    import torch
    import torch.distributed as dist

    from andvaranaut_amd.distributed import DistGP

    X, y = synth_problem(N, d, seed=0)
    gp = DistGP(X, y, kernel, device=dev.index, panel_width_tiles=args.sharded_panel_tiles or None)
    if exchange != "bcast":
        gp.set_exchange(exchange)  # (collective: every rank makes the same call)
    if lazy_sends is not None:
        gp.lazy_sends = bool(lazy_sends)
    thetas = theta_sequence(d, warmup + steps, seed=0)
    for i in range(len(thetas)):  # config 4 is RBF at d=32: keep cond(K) in the benchmark regime
        thetas[i][-2] = 1e-4
    step = (lambda th: gp.lml_grad(th)[0]) if grad else gp.lml
    for i in range(warmup):
        step(thetas[i])
    torch.cuda.synchronize(dev)
    if dist.is_initialized():
        dist.barrier()
    gp.bytes_broadcast = 0
    t0 = time.perf_counter()
    vals = [step(thetas[warmup + i]) for i in range(steps)]
    torch.cuda.synchronize(dev)
    if dist.is_initialized():
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ok = all(np.isfinite(v) for v in vals)
    flops = (N ** 3 / 3.0) * (3.0 if grad else 1.0)
    rec = {
        "metric": "gp_lml_grad_evals_per_s" if grad else "gp_lml_evals_per_s",
        "workload": f"{kernel} GP LML{'+grad' if grad else ''}, ONE covariance N={N} d={d} sharded over {world} GPU(s)",
        "scaling": "strong", "n_gpus": world, "steps": steps, "warmup": warmup,
        "value": steps / elapsed, "unit": "evals/s", "ms_per_step": elapsed / steps * 1e3,
        "tflops_whole_eval": flops / (elapsed / steps) * 1e-12,
        "frac_of_fp64_peak_all_gpus": flops / (elapsed / steps) * 1e-12 / (FP64_PEAK_TFLOPS * world),
        "parallelism": f"column-panel ({gp.pw} columns) block-cyclic x{world}, owner factors + RCCL "
                       f"{'broadcast' if exchange == 'bcast' else 'mesh exchange (scatter over the links + peer all-gather)'}, look-ahead 1",
        "exchange": exchange, "lazy_sends": bool(lazy_sends) if lazy_sends is not None else "default",
        "bytes_broadcast_per_step": gp.bytes_broadcast / max(steps, 1),
        "collectives": ("rccl" if dist.get_backend() == "nccl" else f"{dist.get_backend()} (rehearsal: host-staged, not a measurement of the links)")
                       if dist.is_initialized() else "none (single process)",
        "finite": bool(ok), "lml": float(vals[-1]),
    }
    gp_pwt = gp.pwt
    gp.close()
    del gp
    torch.cuda.empty_cache()
    # parity of the sharded result on whatever hardware this runs on: rank 0 evaluates the same covariance once on the
    # single-GPU path (34 GB at N = 65536; a second or two) and the record carries the relative difference
    if not grad and rank == 0 and single_gpu_check:
        try:
            from andvaranaut_amd import MiGP

            one = MiGP(X, y, kernel, device=dev.index, need_grad=False)
            ref = one.lml(thetas[warmup + steps - 1])
            one.close()
            del one
            torch.cuda.empty_cache()
            rec["single_gpu_lml"] = float(ref)
            rec["rel_diff_vs_single_gpu_path"] = abs(float(vals[-1]) - float(ref)) / abs(float(ref))
        except Exception as e:  # noqa: BLE001 - e.g. not enough free HBM next to other handles: the timing stands on its own
            rec["single_gpu_check"] = f"skipped: {type(e).__name__}: {e}"
    # the one-GPU rank emulation's prediction for this world size (tools/emulate_rank.py --curve), for the first hardware
    # run to be checked against: a model (measured per-rank compute and owner chain + bytes / link bandwidth), not a result
    mfile = next((os.path.join(ROOT, "profiles", f"r0{r}_sharded_model.json") for r in (6, 5, 4)
                  if os.path.exists(os.path.join(ROOT, "profiles", f"r0{r}_sharded_model.json"))), "")
    if not grad and os.path.exists(mfile):
        try:
            mj = json.load(open(mfile))
            if mj.get("N") == N and mj.get("d") == d:
                for pr in mj.get("prediction", []):
                    # the model file holds both chain placements for world > 1; the driver's default is "ahead of the bulk update"
                    # ... sending each tile column behind its strip (the first such entry; the whole-panel send follows it)
                    if (pr.get("world") == world and pr.get("panel_tiles") == gp_pwt
                            and (world == 1 or (pr.get("chain", "").startswith("on the main") and pr.get("send", "").startswith("piece")))):
                        # one entry per assumed one-way link bandwidth and exchange form (round 5): AMD's 153.6 GB/s per xGMI
                        # link is most likely a bidirectional figure, so 77 GB/s one way is modelled beside it
                        key = f"{pr.get('exchange', 'bcast')}@{pr.get('link_GBps_assumed', 153.0):g}GBps"
                        rec.setdefault("predicted_ms_per_step_by_exchange_and_link", {})[key] = pr["predicted_ms"]
                        if "predicted_ms_per_step" not in rec:
                            rec["predicted_ms_per_step"] = pr["predicted_ms"]
                            rec["prediction_source"] = f"profiles/{os.path.basename(mfile)} (one-GPU emulation of ranks + link model: a MODEL, not a measurement)"
        except Exception as e:  # noqa: BLE001
            rec["prediction_source"] = f"unreadable: {e}"
    return rec


SHARDED_FORMS = (("bcast", False), ("bcast", True), ("mesh", False))  # (exchange, lazy owner sends); the first is the default


def sharded_forms(args, rank, world, dev, N, d, kernel, steps, warmup, on_partial=None):
    """The one multi-GPU run that may come measures EVERY exchange form, not just the default: broadcast with the
    owner waiting for its sends (default), broadcast with lazy owner sends, and the mesh exchange -- back to back on the same
    thetas, each under its own deadline, each beside its column of the one-GPU emulation's model.  The three LMLs must be
    bit-equal (fixed world size and options: include/mi_gp.h); a form that fails or disagrees is reported, the others stand.
    Returns the default form's record with a ``forms`` table; ``on_partial(rec)`` sees it after every form (the deadline's
    handler prints what is there)."""
    out = None
    for exchange, lazy in SHARDED_FORMS:
        name = f"{exchange}/{'lazy' if lazy else 'eager'}"

        def expire(name=name):
            if rank == 0 and out is not None:
                out["forms"][name] = {"error": f"not finished after {args.sharded_timeout:.0f} s"}
                if on_partial is not None:
                    on_partial(out, final=True)
            os._exit(3)

        with Deadline(args.sharded_timeout, expire):
            try:
                rec = sharded_record(args, rank, world, dev, N, d, kernel, steps, warmup, exchange=exchange, lazy_sends=lazy,
                                     single_gpu_check=out is None)
            except Exception as e:  # noqa: BLE001 - one form's failure must not take the others (or the headline) down
                rec = {"error": f"{type(e).__name__}: {e}"}
        if out is None:
            if "error" in rec:
                return rec  # the default form itself failed: nothing to compare the others with
            out = rec
            out["forms"] = {}
        pred = (out.get("predicted_ms_per_step_by_exchange_and_link") or {})
        out["forms"][name] = ({"error": rec["error"]} if "error" in rec else
                              {"ms_per_step": rec["ms_per_step"], "lml": rec["lml"], "finite": rec["finite"],
                               "bytes_exchanged_per_step": rec["bytes_broadcast_per_step"],
                               "bit_equal_to_default": rec["lml"] == out["lml"],
                               "model_ms_per_step": {k: v for k, v in pred.items() if k.startswith(exchange + "@")} or None})
        if on_partial is not None:
            on_partial(out, final=False)
    ok = [f for f in out["forms"].values() if "error" not in f]
    out["forms_bit_equal"] = len(ok) == len(SHARDED_FORMS) and all(f["bit_equal_to_default"] for f in ok)
    if ok:
        best = min(out["forms"], key=lambda k: out["forms"][k].get("ms_per_step", float("inf")))
        out["fastest_form"] = best
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=None, help="default 16384 (replicas) / --sharded-n (with --sharded)")
    ap.add_argument("--d", type=int, default=None, help="default 16 / --sharded-d")
    ap.add_argument("--kernel", default=None, help="default Matern52 / --sharded-kernel")
    ap.add_argument("--panel-tiles", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-sample", action="store_true", help="time the CPU oracle at N=4096 and extrapolate (fast)")
    ap.add_argument("--roofline-steps", type=int, default=3)
    ap.add_argument("--sharded", action="store_true", help="the whole line is the sharded (strong-scaling) workload")
    ap.add_argument("--no-sharded", action="store_true", help="skip the sharded sub-record")
    ap.add_argument("--sharded-n", type=int, default=65536)
    ap.add_argument("--sharded-d", type=int, default=32)
    ap.add_argument("--sharded-kernel", default="RBF")
    ap.add_argument("--sharded-steps", type=int, default=2)
    ap.add_argument("--sharded-timeout", type=float, default=420.0)
    ap.add_argument("--sharded-panel-tiles", type=int, default=0,
                    help="sharded driver's panel width in 128-column tiles (0: its own rule: 8 up to two ranks, else 4)")
    ap.add_argument("--grad", action="store_true", help="with --sharded: time LML + gradient (sharded K^-1) instead of the LML")
    ap.add_argument("--grad-steps", type=int, default=5, help="LML + gradient evaluations timed after the LML region (0: skip)")
    ap.add_argument("--no-lookahead", action="store_true", help="disable the look-ahead stream everywhere (profiling aid)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: REHEARSAL of the multi-rank code path with every rank on GPU 0 and host-staged collectives "
                         "(tests/test_gpu_distributed.py); its timings say nothing about the links")
    ap.add_argument("--chains-per-gpu", type=int, default=3,
                    help="extra record: this many handles evaluated side by side on every GPU (independent chains; 0/1: skip)")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        spawn_ranks(args)  # does not return
    world = int(env_world or "1")
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import torch.distributed as dist

    if args.backend == "gloo":
        local_rank = 0  # rehearsal: every rank on the first GPU
    if torch.cuda.device_count() < (local_rank + 1):
        sys.exit(f"bench.py: rank {rank} needs GPU {local_rank}, this box has {torch.cuda.device_count()}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # One process group in every configuration (world 1 included), backend nccl = RCCL: the barrier / MAX reduction
    # of the timed region and the sharded sub-record's panel broadcasts go through it.
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world == 1:
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if args.backend == "gloo":
        dist.init_process_group(backend="gloo")
    else:
        dist.init_process_group(backend="nccl", device_id=dev)
    assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    from andvaranaut_amd import MiGP

    if args.sharded:
        def as_line(rec):
            return {"metric": rec["metric"], "value": rec["value"], "unit": "evals/s", "n_gpus": world, "rccl_ranks": dist.get_world_size(),
                    "steps": args.steps, "warmup": args.warmup, "ms_per_step": rec["ms_per_step"], "higher_is_better": True,
                    "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                    "config": {"workload": rec["workload"], "parallelism": rec["parallelism"]}, "sharded": rec}

        sn, sd, sk = args.n or args.sharded_n, args.d or args.sharded_d, args.kernel or args.sharded_kernel
        if world > 1 and not args.grad:
            # every exchange form, back to back; the deadline's handler prints the line with the forms measured so far
            rec = sharded_forms(args, rank, world, dev, sn, sd, sk, args.steps, args.warmup,
                                on_partial=lambda r, final: print(json.dumps(as_line(r)), flush=True) if final else None)
        else:
            rec = sharded_record(args, rank, world, dev, sn, sd, sk, args.steps, args.warmup, grad=args.grad)
        if rank == 0:
            if "error" in rec:
                print(json.dumps({"metric": "gp_lml_evals_per_s", "value": None, "n_gpus": world, "sharded": rec}), flush=True)
            else:
                print(json.dumps(as_line(rec)), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return

    N, d = args.n or 16384, args.d or 16
    args.kernel = args.kernel or "Matern52"
    X, y = synth_problem(N, d, seed=0)
    want_grad = args.grad_steps > 0
    gp = MiGP(X, y, args.kernel, device=local_rank, panel_tiles=args.panel_tiles, need_grad=want_grad)
    thetas = theta_sequence(d, args.warmup + args.steps, seed=rank)
    if args.no_lookahead:
        gp.set_option(0, 0)

    for i in range(args.warmup):
        gp.lml(thetas[i])
    torch.cuda.synchronize(dev)
    dist.barrier()
    t0 = time.perf_counter()
    vals = []
    for i in range(args.steps):
        vals.append(gp.lml(thetas[args.warmup + i]))  # synchronous on return (stream-synchronised)
    torch.cuda.synchronize(dev)
    dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    assert all(np.isfinite(v) for v in vals), "non-finite LML in the timed region"

    # LML + gradient (the MAP loop of BASELINE config 3 and every NUTS leapfrog step): same handle, same thetas
    grad_rec = None
    if want_grad:
        gp.lml_grad(thetas[0])
        torch.cuda.synchronize(dev)
        dist.barrier()
        tg = time.perf_counter()
        ng = min(args.grad_steps, len(thetas))
        for i in range(ng):
            v, g = gp.lml_grad(thetas[i])
            assert np.isfinite(v) and np.all(np.isfinite(g))
        torch.cuda.synchronize(dev)
        dist.barrier()
        tg = torch.tensor([time.perf_counter() - tg], dtype=torch.float64, device=dev)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        tg = float(tg.item()) / ng
        grad_rec = {"lml_grad_evals_per_s": world / tg, "ms_per_eval": tg * 1e3, "steps": ng,
                    "algorithmic_flops": "N^3 (factor N^3/3 + L^-T N^3/3 + K^-1 = U U^T N^3/3)",
                    "tflops_whole_eval": N ** 3 / tg * 1e-12, "frac_of_fp64_peak": N ** 3 / tg * 1e-12 / FP64_PEAK_TFLOPS}

    # Independent chains sharing a GPU (how GPMCMC.fit(method='mcmc_*') schedules more chains than GPUs): K handles
    # driven by K host threads; an evaluation is bound by the serial panel chain for part of its time, a second and third
    # one fill the idle CUs.  Extra record only: `value` above stays one chain per GPU (its ms_per_step is the latency a
    # single MAP optimisation or chain sees).
    conc_rec = None
    if args.chains_per_gpu > 1:
        import threading
        K = args.chains_per_gpu
        others = [MiGP(X, y, args.kernel, device=local_rank, panel_tiles=args.panel_tiles, need_grad=False) for _ in range(K - 1)]
        lanes = [gp] + others
        csteps = max(2, min(args.steps, 8))
        for h in lanes:
            h.lml(thetas[0])

        def lane_work(h, off):
            for i in range(csteps):
                v = h.lml(thetas[(off + i) % len(thetas)])
                assert np.isfinite(v)

        torch.cuda.synchronize(dev)
        dist.barrier()
        tc = time.perf_counter()
        ths = [threading.Thread(target=lane_work, args=(h, j)) for j, h in enumerate(lanes)]
        for th_ in ths:
            th_.start()
        for th_ in ths:
            th_.join()
        torch.cuda.synchronize(dev)
        dist.barrier()
        tc = torch.tensor([time.perf_counter() - tc], dtype=torch.float64, device=dev)
        dist.all_reduce(tc, op=dist.ReduceOp.MAX)
        tc = float(tc.item())
        conc_rec = {"chains_per_gpu": K, "evals_per_s": world * K * csteps / tc, "steps_per_chain": csteps,
                    "ms_per_round_of_K": tc / csteps * 1e3,
                    "tflops_whole_eval_per_gpu": K * csteps * (N ** 3 / 3.0) / tc * 1e-12,
                    "frac_of_fp64_peak": K * csteps * (N ** 3 / 3.0) / tc * 1e-12 / FP64_PEAK_TFLOPS}
        for h in others:
            h.close()
        del others, lanes
        torch.cuda.empty_cache()
        # the same K chains through ONE handle and ONE batched call per round (mi_gp_lml_batch: blockIdx.z = chain; how
        # GPMCMC.fit schedules the chains of a GPU since round 4).  Extra record as well.
        try:
            TK = np.array([thetas[j % len(thetas)] for j in range(K)])
            gp.lml_batch(TK)
            torch.cuda.synchronize(dev)
            dist.barrier()
            tb = time.perf_counter()
            for i in range(csteps):
                vb = gp.lml_batch(np.array([thetas[(j + i) % len(thetas)] for j in range(K)]))
                assert np.all(np.isfinite(vb))
            torch.cuda.synchronize(dev)
            dist.barrier()
            tb = torch.tensor([time.perf_counter() - tb], dtype=torch.float64, device=dev)
            dist.all_reduce(tb, op=dist.ReduceOp.MAX)
            tb = float(tb.item())
            conc_rec["batched"] = {"k": K, "evals_per_s": world * K * csteps / tb, "ms_per_batch": tb / csteps * 1e3,
                                   "entry": "mi_gp_lml_batch (one handle, blockIdx.z = chain)"}
        except Exception as e:  # noqa: BLE001 - the extra record must not take the headline down
            conc_rec["batched"] = {"error": str(e)}

    # Roofline pass (rank 0): the same evaluations again with HIP events on the handle's own stream
    # around every phase and every GEMM launch.  Look-ahead is switched off for this pass so that
    # the dominant kernel runs alone on the chip and its launch durations are not inflated by the
    # panel kernels that overlap it in the timed region.
    acc = {"assemble_ms": 0.0, "cholesky_ms": 0.0, "gemm_ms": 0.0, "gemm_flops": 0.0, "gemm_launches": 0.0, "total_ms": 0.0,
           "gemm_b_ms": 0.0, "gemm_b_flops": 0.0, "gemm_b_launches": 0.0}
    insitu = dict(acc)  # the same events with look-ahead ON: the dominant kernel next to the panel chain of the second stream
    rsteps = max(1, min(args.roofline_steps, args.steps))
    gpu_ref_lml = None
    if rank == 0:
        gp.set_option(0, 0)
        gp.set_profiling(2)
        for i in range(rsteps):
            gp.lml(thetas[args.warmup + i])
            tm = gp.timers()
            for k in acc:
                acc[k] += tm[k]
        gp.set_option(0, 0 if args.no_lookahead else 1)
        for i in range(rsteps):
            gp.lml(thetas[args.warmup + i])
            tm = gp.timers()
            for k in insitu:
                insitu[k] += tm[k]
        gp.set_profiling(0)
        # parity record: the device's LML at the theta (and seed-0 data) the CPU baseline evaluates below
        gpu_ref_lml = gp.lml(reference_theta(d))
    gp.close()
    del gp
    torch.cuda.empty_cache()

    line = None
    if rank == 0:
        steps = args.steps
        # dominant kernel = gemm_f64_kernel_b (128x128 tiles); the 64x64-tile kernel that serves the small
        # in-panel updates is reported next to it
        gemm_avg_ms = acc["gemm_b_ms"] / max(acc["gemm_b_launches"], 1.0)
        achieved = acc["gemm_b_flops"] / (acc["gemm_b_ms"] * 1e-3) * 1e-12 if acc["gemm_b_ms"] > 0 else 0.0
        all_gemm = acc["gemm_flops"] / (acc["gemm_ms"] * 1e-3) * 1e-12 if acc["gemm_ms"] > 0 else 0.0
        traffic, traffic_note = None, "profiles/gemm_traffic.json missing"
        tfile = os.path.join(ROOT, "profiles", "gemm_traffic.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                traffic = tj.get("hbm_bytes_per_launch")
                src = os.path.join(ROOT, "andvaranaut_amd", "csrc", "gemm_f64.hip")
                sha = hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]
                traffic_note = ("PMC passes of tools/prof_round.sh + tools/pmc_traffic.py on this kernel source" if tj.get("gemm_src_sha16") == sha
                                else f"STALE: measured on gemm_f64.hip {tj.get('gemm_src_sha16')}, current {sha}")
            except Exception as e:  # noqa: BLE001
                traffic_note = f"unreadable: {e}"
        asm_bytes = 8.0 * (N // 64) * (N // 64 + 1) / 2 * 64 * 64 + 8.0 * N * d
        asm_counter, asm_note = None, "no assemble record in profiles/gemm_traffic.json"
        try:
            aj = json.load(open(tfile)).get("assemble")
            if aj:
                asm_counter = aj["hbm_bytes_per_launch"]
                asha = hashlib.sha256(open(os.path.join(ROOT, "andvaranaut_amd", "csrc", "assemble.hip"), "rb").read()).hexdigest()[:16]
                asm_note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this bench (tools/prof_round.sh)" if aj.get("src_sha16") == asha
                            else f"STALE: measured on assemble.hip {aj.get('src_sha16')}, current {asha}")
        except Exception as e:  # noqa: BLE001
            asm_note = f"unreadable: {e}"
        line = {
            "metric": "gp_lml_evals_per_s",
            "value": world * steps / elapsed,
            "unit": "evals/s",
            "n_gpus": world,
            "rccl_ranks": dist.get_world_size(),
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
            "cholesky_frac_of_peak_whole_eval": (N ** 3 / 3.0) / (elapsed / steps) * 1e-12 / FP64_PEAK_TFLOPS,
            "launch_mode": "plain launches, look-ahead on a second stream (hipGraph replay removed in round 2: see DESIGN.md)",
            "lml_grad": grad_rec,
            "concurrent_chains": conc_rec,
            "roofline_pass": {"steps": rsteps, "lookahead": False,
                              "phase_ms": {"assemble": acc["assemble_ms"] / rsteps, "cholesky": acc["cholesky_ms"] / rsteps,
                                           "gemm_in_cholesky": acc["gemm_ms"] / rsteps}},
            # K1/K2: lower-triangle 64x64 tiles written once + X read once (8*N*d): HBM-side figure the
            # north star asks for next to the MFMA one
            "assembly": {"kernel": "assemble_kernel", "ms": acc["assemble_ms"] / rsteps, "algorithmic_bytes": asm_bytes,
                         "achieved_GBps": asm_bytes / (acc["assemble_ms"] / rsteps * 1e-3) * 1e-9, "peak_GBps": 8000.0,
                         # counter-derived: HBM bytes of one launch from the PMC passes / this run's launch time
                         "hbm_bytes_counters": asm_counter, "hbm_bytes_source": asm_note,
                         "counter_GBps": (asm_counter / (acc["assemble_ms"] / rsteps * 1e-3) * 1e-9) if asm_counter else None},
            "roofline": {"kernel": "gemm_f64_kernel_b (SYRK trailing/panel updates, v_mfma_f64_16x16x4_f64)",
                         "bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_note,
                         # the same kernel inside the timed two-stream configuration (HIP events on its own stream, look-ahead
                         # on: the panel chain's leaf / strip / in-panel kernels share the chip -- and the DP pipe -- with it)
                         "in_situ_frac": (insitu["gemm_b_flops"] / (insitu["gemm_b_ms"] * 1e-3) * 1e-12 / FP64_PEAK_TFLOPS
                                          if insitu["gemm_b_ms"] > 0 else None),
                         "in_situ_avg_launch_ms": insitu["gemm_b_ms"] / max(insitu["gemm_b_launches"], 1.0),
                         "in_situ_source": "HIP events around every launch on the main stream, look-ahead on, same thetas",
                         "avg_launch_ms": gemm_avg_ms, "launches_per_step": acc["gemm_b_launches"] / rsteps,
                         "flop_share_of_all_gemm": acc["gemm_b_flops"] / max(acc["gemm_flops"], 1.0),
                         "all_gemm_kernels_tflops": all_gemm, "all_gemm_launches_per_step": acc["gemm_launches"] / rsteps},
        }

    # sharded sub-record (strong scaling, BASELINE config 4) under a deadline: the replicas' line survives a stuck
    # exchange, but the PROCESS does not pretend to be healthy -- a watchdog or a caught failure ends with a non-zero
    # exit status once the line is out (nothing is retried or restarted in-process).
    exit_code = 0
    if not args.no_sharded:
        def expire():
            if rank == 0 and line is not None:
                line["sharded"] = {"error": f"not finished after {args.sharded_timeout:.0f} s (rank 0 gave up waiting)"}
                print(json.dumps(line), flush=True)
            os._exit(3)

        if world > 1:
            # the one hardware run that may come compares every exchange form (each under its own deadline inside sharded_forms);
            # a form that hangs ends the process through its deadline AFTER the line -- replicas' numbers and the forms
            # measured so far -- is out
            def partial(r, final):
                if rank == 0 and line is not None:
                    line["sharded"] = r
                    if final:
                        print(json.dumps(line), flush=True)

            try:
                rec = sharded_forms(args, rank, world, dev, args.sharded_n, args.sharded_d, args.sharded_kernel,
                                    args.sharded_steps, 1, on_partial=partial)
            except Exception as e:  # noqa: BLE001
                rec = {"error": f"{type(e).__name__}: {e}"}
            if "error" in rec:
                exit_code = 4
        else:
          with Deadline(args.sharded_timeout, expire):
            try:
                rec = sharded_record(args, rank, world, dev, args.sharded_n, args.sharded_d, args.sharded_kernel,
                                     args.sharded_steps, 1)
            except Exception as e:  # noqa: BLE001 - the replicas' numbers must still be reported
                rec = {"error": f"{type(e).__name__}: {e}"}
                exit_code = 4
        if rank == 0:
            line["sharded"] = rec

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(N, d, args.kernel, full=not args.cpu_baseline_sample, gpu_lml=gpu_ref_lml)
        print(json.dumps(line), flush=True)
    if exit_code:
        # a rank whose sharded step failed must not walk into the final barrier: its peers may be stuck inside the
        # collective it left, and they leave through their own watchdog (exit 3)
        sys.stdout.flush()
        os._exit(exit_code)
    # rank 0's roofline pass / CPU baseline run after the timed region: leave together -- but never hang on a rank that
    # died in the sharded sub-record after the line was printed
    with Deadline(1800.0 if world == 1 else 120.0, lambda: os._exit(3)):
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
