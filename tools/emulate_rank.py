"""One-GPU emulation of rank r of a W-rank sharded LML evaluation (andvaranaut_amd/distributed.py, BASELINE config 4):
the rank owns, updates and factors exactly its block-cyclic share of the panels; the panels the other ranks would
broadcast are copied in from a complete factor computed beforehand on the same GPU.  Measured per step with HIP events:
the owner chain (update of the next panel, its factorisation, staging) and the bulk update launch.  From the per-rank
numbers a 1/2/4/8-GPU curve is PREDICTED (not measured) by replaying the per-step times on a timeline of `world` ranks
(predict() below): a rank starts step j when panel j has arrived and its previous work is done; the owner of panel j + 1
runs its chain ahead of (or beside) its bulk update; the panel arrives bytes / link bandwidth later (xGMI point-to-point,
~153 GB/s per link; the broadcast's later hops overlap the following steps).  Round 4: the panel is sent piece by piece
(one tile column each) as the owner's chain completes the columns; the per-piece staging times are measured too and the
timeline sends piece c at max(its staging, the end of piece c - 1's transfer).  profiles/r04_sharded_model.json keeps the
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
            best = (wall, gp.step_times().copy(), gp.logdet, gp.quad, gp.piece_times().copy())
    wall, times, logdet, quad, pieces = best
    rec = {"world": world, "rank": rank, "panel_tiles": gp.pwt, "npan": gp.npan, "owned": len(gp.own), "wall_ms": wall,
           "update_ms": float(times[:-1, 0].sum()), "factor_ms": float(times[:, 1].sum()), "stage_ms": float(times[:, 2].sum()),
           "bulk_ms": float(times[:-1, 3].sum()), "logdet_part": logdet, "quad_part": quad,
           "steps": times.tolist(), "pieces": pieces.tolist()}
    gp.close()
    del gp
    torch.cuda.empty_cache()
    return rec


def predict(world, recs, N, pwt, serial, link_gbps=LINK_GBPS, pipelined=True, exchange="bcast"):
    """Timeline of a `world`-rank evaluation from the emulated ranks' per-step times (a model, not a measurement).
    Rank r at step j starts when panel j has arrived and its own previous work is done.  The owner of panel j + 1 runs
    its chain (update + factor + stage) -- ahead of its bulk update (serial: option 3 = 1) or beside it on the side
    stream (the measured bulk time then already contains the contention).  Panel j + 1 is sent piece by piece over one
    link (xGMI point-to-point; the broadcast's later hops overlap the following steps): piece c leaves at
    max(its staging, the end of piece c - 1's transfer) and takes bytes / link bandwidth; the panel has arrived with its
    last piece.  pipelined=False (and every chain beside a bulk update, which stages at its end): all pieces leave when
    the chain is done.  Ranks that were not emulated take the per-step times of the nearest emulated rank; chains of
    panels whose owner was not emulated are interpolated over the panel index."""
    npan = recs[0]["npan"]
    pw = pwt * 128
    npad = (N + 127) // 128 * 128
    by_rank = {r["rank"]: np.array(r["steps"]) for r in recs}
    pc_rank = {r["rank"]: np.array(r.get("pieces", np.zeros((npan + 1, pwt)))) for r in recs}
    chain = np.full(npan + 1, np.nan)
    ready = np.full((npan + 1, pwt), np.nan)  # piece c of panel p staged this long after the chain's start, as a fraction of the chain
    for r, st in by_rank.items():
        for j in range(npan):
            c = st[j, 0] + st[j, 1] + st[j, 2] if serial else st[j, 0] + st[j, 1]  # beside a bulk update the staging waits for it
            if c > 0:
                chain[j + 1] = c  # step j produced panel j + 1
                if pc_rank[r][j].max() > 0:
                    ready[j + 1] = np.where(pc_rank[r][j] > 0, pc_rank[r][j], c) / c
        if st[npan, 1] > 0:
            chain[0] = st[npan, 1] + st[npan, 2]
            if pc_rank[r][npan].max() > 0:
                ready[0] = np.where(pc_rank[r][npan] > 0, pc_rank[r][npan], chain[0]) / chain[0]
    idx = np.arange(npan + 1)
    known = ~np.isnan(chain)
    chain = np.interp(idx, idx[known], chain[known])
    for c in range(pwt):
        k2 = ~np.isnan(ready[:, c])
        ready[:, c] = np.interp(idx, idx[k2], ready[k2, c]) if k2.any() else 1.0
    ready = np.minimum(ready, 1.0)
    if not (pipelined and serial):
        ready[:] = 1.0
    wj = np.array([min(pwt, (npad // 128) - j * pwt) for j in range(npan + 1)])
    piece_ms = np.array([(npad + 128 - j * pw + 128) * 128 * 8 / (link_gbps * 1e9) * 1e3 for j in range(npan + 1)])
    if exchange == "mesh" and world > 2:
        # DistGP.set_exchange("mesh"): the owner sends 1 / (W - 1) of a piece to every peer over that peer's own link, then the
        # peers all-gather among themselves over theirs: two transfers of bytes / (W - 1) per link instead of one of `bytes`
        piece_ms *= 2.0 / (world - 1)
    link = piece_ms * np.maximum(wj, 0)
    if world == 1:
        link[:] = 0.0
        piece_ms[:] = 0.0

    def arrival(p):  # ms after the chain's start at which panel p has arrived everywhere
        end = 0.0
        for c in range(max(int(wj[p]), 1)):
            end = max(end, ready[p, min(c, pwt - 1)] * chain[p]) + piece_ms[p]
        return max(end, chain[p])

    emu = sorted(by_rank)
    bulk = [by_rank[min(emu, key=lambda e: abs(e - r))][:npan, 3] for r in range(world)]
    free = np.zeros(world)
    avail = np.zeros(npan + 1)
    free[0] = chain[0]
    avail[0] = arrival(0)
    exposed = 0.0
    for j in range(npan):
        o = (j + 1) % world if j + 1 < npan else -1
        for r in range(world):
            t = max(free[r], avail[j])
            if r == o:
                done = t + chain[j + 1]
                avail[j + 1] = t + arrival(j + 1)
                exposed += avail[j + 1] - done
                free[r] = (done if serial else t) + bulk[r][j]
                if not serial:
                    free[r] = max(free[r], done)
            else:
                free[r] = t + bulk[r][j]
    return {"world": world, "chain": "on the main stream ahead of the bulk update" if serial else "on the side stream beside the bulk update",
            "send": "piece by piece behind each tile column's strip" if (pipelined and serial) else "behind the whole panel",
            "predicted_ms": float(free.max()), "sum_bulk_ms_slowest_rank": float(max(b.sum() for b in bulk)),
            "sum_chain_ms": float(chain[:npan].sum()), "sum_link_ms": float(link[:npan].sum()),
            "sum_link_ms_behind_the_chain": float(exposed),
            "link_GBps_assumed": link_gbps, "exchange": exchange, "ranks_emulated": emu}


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
    ap.add_argument("--links", default="153,77", help="one-way GB/s per xGMI link to model (AMD quotes 153.6 GB/s per link, "
                    "most likely bidirectional: 77 one way)")
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
                print(json.dumps({k: v for k, v in rec.items() if k not in ("steps", "pieces")}), flush=True)
            if len(ranks) == world:  # every rank emulated: their partial sums must add up to the single-GPU factor's
                ld_sum, q_sum = sum(r["logdet_part"] for r in recs), sum(r["quad_part"] for r in recs)
                assert abs(ld_sum - logdet) <= 1e-10 * abs(logdet) and abs(q_sum - quad) <= 1e-9 * abs(quad), (ld_sum, logdet, q_sum, quad)
                print(f"partial sums of all {world} ranks reproduce the single-GPU log-det and quadratic form", flush=True)
            for pipelined in ([True, False] if (serial and world > 1) else [False]):
                for link in ([float(v) for v in args.links.split(",")] if world > 1 else [LINK_GBPS]):
                    for exchange in (["bcast", "mesh"] if world > 2 else ["bcast"]):
                        pred = predict(world, recs, N, pwt, bool(serial), link_gbps=link, pipelined=pipelined, exchange=exchange)
                        pred["panel_tiles"] = pwt
                        pred["speedup_vs_single_gpu_path"] = single_ms / pred["predicted_ms"]
                        pred["compute_ms_x_world_over_single"] = max(r["bulk_ms"] + r["update_ms"] + r["factor_ms"] for r in recs) * world / single_ms
                        print(json.dumps(pred), flush=True)
                        out["prediction"].append(pred)
            out["runs"] += [{k: v for k, v in r.items() if k not in ("steps", "pieces")} for r in recs]
            os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
            json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
