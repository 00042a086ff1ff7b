"""One-GPU emulation of rank r of a W-rank sharded LML evaluation (andvaranaut_amd/distributed.py, BASELINE config 4):
the rank owns, updates and factors exactly its block-cyclic share of the panels; the panels the other ranks would
broadcast are copied in from a complete factor computed beforehand on the same GPU.  Measured per step with HIP events:
the owner chain (update of the next panel, its factorisation, staging) and the bulk update launch.  From the per-rank
numbers a 1/2/4/8-GPU curve is PREDICTED (not measured) by replaying the per-step times on a timeline of `world` ranks
(predict() below): a rank starts step j when panel j has arrived and its previous work is done; the owner of panel j + 1
runs its chain ahead of (or beside) its bulk update; the panel arrives bytes / link bandwidth later (xGMI point-to-point,
~153 GB/s per link; the broadcast's later hops overlap the following steps).  profiles/r03_sharded_model.json keeps the
prediction so that the first hardware run can be checked against it; bench.py --sharded prints it next to the measurement.

    python tools/emulate_rank.py --world 8 --ranks 0,3,7 [--n 65536 --d 32 --kernel RBF --panel-tiles 4]
    python tools/emulate_rank.py --curve   # worlds 1,2,4,8, a few ranks each, writes the model file"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

LINK_GBPS = 153.0  # xGMI per-link, /opt/skills/guides/MI355X_MICROARCH.md


def full_factor(X, y, kernel, theta):
    import torch

    from andvaranaut_amd import MiGP

    gp = MiGP(X, y, kernel, need_grad=False)
    t0 = time.perf_counter()
    val = gp.lml(theta)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    val2 = gp.lml(theta)
    torch.cuda.synchronize()
    single_ms = (time.perf_counter() - t1) * 1e3
    logdet, quad = gp.lml_parts()
    K = gp.K_t.clone()  # L (lower) + beta^T in row np
    ld = K.stride(0)
    gp.close()
    del gp
    torch.cuda.empty_cache()
    assert val == val2
    return K, ld, val, logdet, quad, single_ms, (t1 - t0) * 1e3


def run_rank(X, y, kernel, theta, world, rank, pwt, source, reps=2, options=None):
    import torch

    from andvaranaut_amd.distributed import DistGP

    gp = DistGP(X, y, kernel, panel_width_tiles=pwt, emulate=(world, rank))
    gp.set_factor_source(*source)
    gp.set_option(1, 1)
    for k, v in (options or {}).items():
        gp.set_option(int(k), int(v))
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gp.lml(theta)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        if best is None or wall < best[0]:
            best = (wall, gp.step_times().copy(), gp.logdet, gp.quad)
    wall, times, logdet, quad = best
    rec = {"world": world, "rank": rank, "panel_tiles": gp.pwt, "npan": gp.npan, "owned": len(gp.own), "wall_ms": wall,
           "update_ms": float(times[:-1, 0].sum()), "factor_ms": float(times[:, 1].sum()), "stage_ms": float(times[:, 2].sum()),
           "bulk_ms": float(times[:-1, 3].sum()), "logdet_part": logdet, "quad_part": quad,
           "steps": times.tolist()}
    gp.close()
    del gp
    torch.cuda.empty_cache()
    return rec


def predict(world, recs, N, pwt, serial, link_gbps=LINK_GBPS):
    """Timeline of a `world`-rank evaluation from the emulated ranks' per-step times (a model, not a measurement).
    Rank r at step j starts when panel j has arrived and its own previous work is done.  The owner of panel j + 1 runs
    its chain (update + factor + stage) -- ahead of its bulk update (serial: option 3 = 1) or beside it on the side
    stream (the measured bulk time then already contains the contention) -- and panel j + 1 arrives everywhere
    bytes / link bandwidth later (xGMI point-to-point; the broadcast's later hops overlap the following steps).  Ranks that
    were not emulated take the per-step times of the nearest emulated rank; chains of panels whose owner was not
    emulated are interpolated over the panel index."""
    npan = recs[0]["npan"]
    pw = pwt * 128
    npad = (N + 127) // 128 * 128
    by_rank = {r["rank"]: np.array(r["steps"]) for r in recs}
    chain = np.full(npan + 1, np.nan)
    for r, st in by_rank.items():
        for j in range(npan):
            c = st[j, 0] + st[j, 1] + st[j, 2] if serial else st[j, 0] + st[j, 1]  # beside a bulk update the staging waits for it
            if c > 0:
                chain[j + 1] = c  # step j produced panel j + 1
        if st[npan, 1] > 0:
            chain[0] = st[npan, 1] + st[npan, 2]
    idx = np.arange(npan + 1)
    known = ~np.isnan(chain)
    chain = np.interp(idx, idx[known], chain[known])
    link = np.array([(npad + 128 - j * pw) * pw * 8 / (link_gbps * 1e9) * 1e3 for j in range(npan + 1)])  # ms
    if world == 1:
        link[:] = 0.0
    emu = sorted(by_rank)
    bulk = [by_rank[min(emu, key=lambda e: abs(e - r))][:npan, 3] for r in range(world)]
    free = np.zeros(world)
    avail = np.zeros(npan + 1)
    free[0] = chain[0]
    avail[0] = chain[0] + link[0]
    for j in range(npan):
        o = (j + 1) % world if j + 1 < npan else -1
        for r in range(world):
            t = max(free[r], avail[j])
            if r == o:
                done = t + chain[j + 1]
                avail[j + 1] = done + link[j + 1]
                free[r] = (done if serial else t) + bulk[r][j]
                if not serial:
                    free[r] = max(free[r], done)
            else:
                free[r] = t + bulk[r][j]
    return {"world": world, "chain": "on the main stream ahead of the bulk update" if serial else "on the side stream beside the bulk update",
            "predicted_ms": float(free.max()), "sum_bulk_ms_slowest_rank": float(max(b.sum() for b in bulk)),
            "sum_chain_ms": float(chain[:npan].sum()), "sum_link_ms": float(link[:npan].sum()),
            "link_GBps_assumed": link_gbps, "ranks_emulated": emu}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=65536)
    ap.add_argument("--d", type=int, default=32)
    ap.add_argument("--kernel", default="RBF")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--ranks", default="0")
    ap.add_argument("--panel-tiles", type=int, default=0)
    ap.add_argument("--curve", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="shard option id=value (repeatable)")
    ap.add_argument("--out", default="gpurun_out/sharded_model.json")
    args = ap.parse_args()

    from andvaranaut_amd.distributed import panel_tiles
    from bench import synth_problem, theta_sequence

    N, d = args.n, args.d
    X, y = synth_problem(N, d, seed=0)
    theta = theta_sequence(d, 1, seed=0)[0]
    theta[-2] = 1e-4
    K, ld, val, logdet, quad, single_ms, first_ms = full_factor(X, y, args.kernel, theta)
    print(f"single-GPU path: {single_ms:.1f} ms per LML at N={N} (first call {first_ms:.0f} ms), lml {val:.6f}", flush=True)
    opts = dict(o.split("=") for o in args.opt)
    ntc = (N + 127) // 128
    out = {"N": N, "d": d, "kernel": args.kernel, "single_gpu_ms": single_ms, "lml": val, "runs": [], "prediction": []}
    plan = [(w, None) for w in (1, 2, 4, 8)] if args.curve else [(args.world, [int(r) for r in args.ranks.split(",")])]
    for world, ranks in plan:
        pwt = args.panel_tiles or panel_tiles(ntc, world)
        if ranks is None:
            ranks = list(range(world))  # every rank: no interpolation in the timeline, and the partial sums can be checked
        for serial in ([0] if world == 1 else [1, 0]):
            recs = []
            for r in ranks:
                rec = run_rank(X, y, args.kernel, theta, world, r, pwt, (K, ld), options={**opts, "3": serial})
                rec["chain_on_main"] = serial
                recs.append(rec)
                print(json.dumps({k: v for k, v in rec.items() if k != "steps"}), flush=True)
            if len(ranks) == world:  # every rank emulated: their partial sums must add up to the single-GPU factor's
                ld_sum, q_sum = sum(r["logdet_part"] for r in recs), sum(r["quad_part"] for r in recs)
                assert abs(ld_sum - logdet) <= 1e-10 * abs(logdet) and abs(q_sum - quad) <= 1e-9 * abs(quad), (ld_sum, logdet, q_sum, quad)
                print(f"partial sums of all {world} ranks reproduce the single-GPU log-det and quadratic form", flush=True)
            pred = predict(world, recs, N, pwt, bool(serial))
            pred["panel_tiles"] = pwt
            pred["speedup_vs_single_gpu_path"] = single_ms / pred["predicted_ms"]
            pred["compute_ms_x_world_over_single"] = max(r["bulk_ms"] + r["update_ms"] + r["factor_ms"] for r in recs) * world / single_ms
            print(json.dumps(pred), flush=True)
            out["runs"] += [{k: v for k, v in r.items() if k != "steps"} for r in recs]
            out["prediction"].append(pred)
            os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
            json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
