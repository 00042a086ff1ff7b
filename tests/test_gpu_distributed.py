"""Column-panel sharded Cholesky / LML (andvaranaut_amd/distributed.py): single process, and two ranks
that share the one GPU of the test box and exchange panels through gloo (the RCCL path needs one GPU
per rank; the sharding logic and every kernel call are the same)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

# The sharded driver and the single-GPU path are two ALGORITHMS since round 5 (block-cyclic panels with recursive in-panel
# updates here; super-panels, extended panels and a column-by-column tail there): they round differently.  Until then the two
# shared every launch of the chain and agreed to 1e-11.  Measured at cond(K) ~ 1e7 (tools, N = 900 RatQuad d = 2): the
# single-GPU path is 4.9e-12 from the oracle, its round-4 arithmetic (option 37 = 0) 7.9e-12 on the other side, the sharded
# path 1.1e-11 -- so two correct results may differ by 2e-11.  The bound is the contract's 1e-10, which each path also has to
# keep against the oracle.
SHARD_VS_SINGLE = 1e-10

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,d,kernel", [(100, 2, "RBF"), (700, 3, "Matern52"), (1500, 4, "RBF"), (2100, 5, "Matern32+RBF")])
def test_single_rank_matches_oracle(N, d, kernel):
    from andvaranaut_amd.distributed import DistGP
    from oracle import gp_oracle as orc

    X, y = orc.synth_problem(N, d, seed=N)
    kerns, ops = kernel.replace("*", "+").split("+"), [c for c in kernel if c in "+*"]
    theta = orc.synth_theta(d, nkern=len(kerns))
    gp = DistGP(X, y, kernel)
    val = gp.lml(theta)
    ref = orc.lml(X, y, kerns, ops, theta)
    assert abs(val - ref) <= 1e-10 * abs(ref), (val, ref)
    assert gp.lml(theta) == val
    bad = theta.copy()
    bad[-1] = -10.0  # negative jitter -> not positive definite
    assert gp.lml(bad) == -np.inf
    assert abs(gp.lml(theta) - ref) <= 1e-10 * abs(ref)


def test_single_rank_wide_panels_match_single_gpu_path():
    """N large enough for the 1024-column panels the driver picks on its own (ragged last panel of one tile)."""
    from andvaranaut_amd import MiGP
    from andvaranaut_amd.distributed import DistGP, panel_tiles
    from oracle import gp_oracle as orc

    N, d = 4200, 4
    assert panel_tiles((N + 127) // 128, 1) == 8 and panel_tiles(24, 2) == 4
    X, y = orc.synth_problem(N, d, seed=3)
    theta = orc.synth_theta(d, nkern=1)
    gp = DistGP(X, y, "Matern52")
    assert gp.pwt == 8 and gp.npan == 5 and gp._w(4) == 1
    val, g = gp.lml_grad(theta)
    one = MiGP(X, y, "Matern52")
    v1, g1 = one.lml_grad(theta)
    assert abs(val - v1) <= SHARD_VS_SINGLE * abs(v1)
    assert _grad_close(g, g1, rtol=1e-8), (g, g1)
    assert abs(gp.lml(theta) - orc.lml(X, y, ["Matern52"], [], theta)) <= 1e-10 * abs(v1)
    one.close()


def _grad_close(g, ref, rtol=1e-7):
    scale = np.maximum(np.abs(ref), 1e-3 * np.max(np.abs(ref)))
    return np.max(np.abs(g - ref) / scale) <= rtol


@pytest.mark.parametrize("N,d,kernel", [(100, 2, "RBF"), (700, 3, "Matern52"), (1500, 4, "RBF"), (2100, 5, "Matern32+RBF"),
                                        (1100, 3, "RBF*Matern52"), (900, 2, "RatQuad")])
def test_single_rank_gradient_matches_oracle_and_single_gpu_path(N, d, kernel):
    """slab-by-slab K^-1 and trace contraction (mi_gp_trsm_block / mi_gp_grad_contract_block) against the oracle's
    analytic gradient and against mi_gp_lml_grad on the same data."""
    from andvaranaut_amd import MiGP
    from andvaranaut_amd.distributed import DistGP
    from oracle import gp_oracle as orc

    X, y = orc.synth_problem(N, d, seed=N)
    ops = [c for c in kernel if c in "+*"]
    kerns = kernel.replace("*", "+").split("+")
    theta = orc.synth_theta(d, nkern=len(kerns))
    gp = DistGP(X, y, kernel)
    val, g = gp.lml_grad(theta)
    ref, gref = orc.lml_grad(X, y, kerns, ops, theta)
    assert abs(val - ref) <= 1e-10 * abs(ref)
    assert _grad_close(g, gref), (g, gref)
    one = MiGP(X, y, kernel)
    v1, g1 = one.lml_grad(theta)
    assert abs(val - v1) <= SHARD_VS_SINGLE * abs(v1)
    assert _grad_close(g, g1, rtol=1e-8), (g, g1)
    one.close()
    bad = theta.copy()
    bad[-1] = -10.0
    vb, gb = gp.lml_grad(bad)
    assert vb == -np.inf and not gb.any()
    v2, g2 = gp.lml_grad(theta)  # buffers are reusable after a failed factorisation
    assert v2 == val and np.array_equal(g2, g)


WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from andvaranaut_amd import parallel
from andvaranaut_amd.distributed import DistGP
from oracle import gp_oracle as orc
rank, world, _ = parallel.init_distributed(backend="gloo")
torch.cuda.set_device(0)
out = {}
for (N, d, kernel, pwt) in [(1500, 4, "RBF", None), (2100, 5, "Matern52", None), (3000, 3, "RBF+Matern32", None),
                            (2900, 3, "Matern52", 8)]:
    X, y = orc.synth_problem(N, d, seed=N)
    kerns, ops = kernel.split("+"), ["+"] * (kernel.count("+"))
    theta = orc.synth_theta(d, nkern=len(kerns))
    gp = DistGP(X, y, kernel, device=0, panel_width_tiles=pwt)
    assert gp.npan >= 3 and (len(gp.own) >= 1 or pwt)
    val = gp.lml(theta)
    ref = orc.lml(X, y, kerns, ops, theta)
    assert abs(val - ref) <= 1e-10 * abs(ref), (rank, N, val, ref)
    v2, g = gp.lml_grad(theta)
    _, gref = orc.lml_grad(X, y, kerns, ops, theta)
    scale = np.maximum(np.abs(gref), 1e-3 * np.max(np.abs(gref)))
    assert v2 == val and np.max(np.abs(g - gref) / scale) <= 1e-7, (rank, N, g, gref)
    out[(N, pwt)] = [val] + g.tolist()
    # the owner chain beside the bulk update on the side stream (one-rank default) instead of ahead of it on the main
    # stream (several-rank default): same arithmetic per panel up to the tile size of one update -> agreement to rounding
    gp.set_option(3, 0)
    v3 = gp.lml(theta)
    assert abs(v3 - val) <= 1e-11 * abs(val), (rank, N, v3, val)
    v4, g4 = gp.lml_grad(theta)
    assert v4 == v3 and np.max(np.abs(g4 - g) / scale) <= 1e-9, (rank, N)
    gp.set_option(3, 1)
    assert gp.lml(theta) == val  # and back: bit-identical to the first evaluation
    # the pipelined send: tile columns staged and broadcast one by one behind their strips (1) or the panel behind its last
    # column (0) -- the same launches, the same bits; the default (2) also takes the first column's update first and alone
    gp.set_option(5, 1)
    v5 = gp.lml(theta)
    gp.set_option(5, 0)
    v6, g6 = gp.lml_grad(theta)
    assert v5 == v6 and abs(v5 - val) <= 1e-11 * abs(val), (rank, N, v5, v6, val)
    assert np.max(np.abs(g6 - g) / scale) <= 1e-9, (rank, N)
    gp.set_option(5, 2)
    assert gp.lml(theta) == val
    # the mesh form of the exchange (owner scatters 1 / (W - 1) of every piece to each peer, the peers all-gather; under
    # gloo through host staging): the same bytes arrive, so the same bits come out
    gp.set_exchange("mesh")
    vm = gp.lml(theta)
    vm2, gm = gp.lml_grad(theta)
    assert vm == val and vm2 == val and np.array_equal(gm, g), (rank, N, vm, val)
    gp.set_option(5, 0)
    assert abs(gp.lml(theta) - val) <= 1e-11 * abs(val)
    gp.set_option(5, 2)
    gp.set_exchange("bcast")
    assert gp.lml(theta) == val
vals = parallel.gather_objects(out)
assert all(v == vals[0] for v in vals), vals  # every rank holds the same all-reduced LML
if rank == 0:
    print(json.dumps({"ok": True}))
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharing_one_gpu_via_gloo(tmp_path, world):
    """world = 3: panel counts (3, 5, 6) do not divide evenly -- ranks own different numbers of panels, and the
    owner of the look-ahead panel changes every step."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="4")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    assert '"ok": true' in outs[0][0]


@pytest.mark.parametrize("world", [2, 3])
def test_bench_measures_every_exchange_form_end_to_end(world):
    """at world > 1 bench.py's sharded record runs the broadcast with eager and with lazy owner sends and the
    mesh exchange back to back, each under its own deadline, and compares their LMLs bit for bit.  Rehearsed here through the
    bench's own code path with gloo ranks that share the one GPU (--backend gloo: host-staged collectives, so lazy sends fall
    back to eager and the timings say nothing about links -- what is tested is that every form runs, agrees, and lands in the
    line)."""
    import json

    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="4")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--sharded", "--backend", "gloo",
                                       "--n", "3000", "--d", "4", "--kernel", "Matern52", "--steps", "2", "--warmup", "1",
                                       "--sharded-panel-tiles", "2", "--sharded-timeout", "300"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    line = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    rec = line["sharded"]
    assert line["n_gpus"] == world and line["scaling"] == "strong" and rec["finite"]
    assert set(rec["forms"]) == {"bcast/eager", "bcast/lazy", "mesh/eager"}, rec["forms"]
    assert all("error" not in f and f["finite"] and f["bit_equal_to_default"] for f in rec["forms"].values()), rec["forms"]
    assert rec["forms_bit_equal"] is True and rec["fastest_form"] in rec["forms"]
    assert rec["forms"]["mesh/eager"]["bytes_exchanged_per_step"] >= rec["forms"]["bcast/eager"]["bytes_exchanged_per_step"]
    assert "gloo" in rec["collectives"]
    # the sharded LML against the single-GPU path of the same covariance (rank 0 evaluates it once)
    assert rec["rel_diff_vs_single_gpu_path"] <= SHARD_VS_SINGLE


def test_bench_default_line_carries_the_exchange_forms_at_world_2():
    """The DEFAULT bench line (replicas headline + sharded sub-record, what the driver's scaling run calls) at world 2, rehearsed with
    gloo ranks on one GPU and small sizes: the headline survives, the sub-record holds the three forms and their bit comparison."""
    import json

    world, port = 2, _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="4")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--backend", "gloo",
                                       "--n", "2500", "--d", "4", "--steps", "3", "--warmup", "1", "--grad-steps", "1", "--chains-per-gpu", "0",
                                       "--roofline-steps", "1", "--sharded-n", "3000", "--sharded-d", "4", "--sharded-kernel", "Matern52",
                                       "--sharded-steps", "2", "--sharded-panel-tiles", "2", "--sharded-timeout", "300"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    line = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    assert line["metric"] == "gp_lml_evals_per_s" and line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["roofline"]["unit"] == "TFLOP/s" and line["lml_grad"]["ms_per_eval"] > 0 and "cpu_baseline" not in line
    rec = line["sharded"]
    assert set(rec["forms"]) == {"bcast/eager", "bcast/lazy", "mesh/eager"} and rec["forms_bit_equal"] is True, rec.get("forms")
    assert rec["scaling"] == "strong" and rec["n_gpus"] == 2 and rec["rel_diff_vs_single_gpu_path"] <= SHARD_VS_SINGLE


@pytest.mark.parametrize("world,pwt,N", [(3, 2, 3000), (4, 1, 1700), (2, 4, 4200), (8, 2, 9000)])
def test_emulated_ranks_partition_the_factorisation(world, pwt, N):
    """Every rank of a `world`-rank job played in turn by one process (DistGP(emulate=...), the mode tools/emulate_rank.py
    times): the panel-list GEMM launch then covers NON-contiguous panels (the gaps are other ranks' columns).  Each
    rank's panels must equal the single-GPU factor's columns, and the ranks' partial log-det / quadratic-form sums must
    add up to the single-GPU values."""
    import torch

    from andvaranaut_amd import MiGP
    from andvaranaut_amd.distributed import DistGP
    from oracle import gp_oracle as orc

    d = 4
    X, y = orc.synth_problem(N, d, seed=N)
    theta = orc.synth_theta(d)
    one = MiGP(X, y, "Matern52", need_grad=False)
    ref = one.lml(theta)
    logdet, quad = one.lml_parts()
    K = one.K_t.clone()
    one.close()
    assert abs(ref - orc.lml(X, y, ["Matern52"], [], theta)) <= 1e-10 * abs(ref)
    ld_sum = q_sum = 0.0
    for rank in range(world):
        gp = DistGP(X, y, "Matern52", panel_width_tiles=pwt, emulate=(world, rank))
        gp.set_factor_source(K, K.stride(0))
        # option 2 only reorders launches of the same arithmetic per panel; option 3 moves the owner's chain between the
        # streams; option 5 stages (and sends) each tile column behind its strip (1), with the first column updated first and alone (2), or
        # the panel behind its last column (0)
        for early, on_main, piecewise in ((1, 1, 2), (0, 1, 2), (1, 1, 1), (1, 1, 0), (1, 0, 2), (0, 0, 2)):
            gp.set_option(2, early)
            gp.set_option(3, on_main)
            gp.set_option(5, piecewise)
            gp.lml(theta)
            torch.cuda.synchronize()
            for li, j in enumerate(gp.own):
                w, r0 = gp._w(j), j * gp.pw
                mine = torch.tril(gp.K[r0: gp.np_ + 1, li * gp.pw: li * gp.pw + w * 128], diagonal=0)
                want = torch.tril(K[r0: gp.np_ + 1, r0: r0 + w * 128], diagonal=0)
                err = (mine - want).abs().max().item()
                assert err <= 1e-9, (world, rank, j, err)
        ld_sum += gp.logdet
        q_sum += gp.quad
        gp.close()
    assert abs(ld_sum - logdet) <= 1e-10 * abs(logdet), (ld_sum, logdet)
    assert abs(q_sum - quad) <= 1e-9 * abs(quad), (q_sum, quad)


@pytest.mark.parametrize("chain_on_main", [0, 1])
def test_kept_factor_of_the_sharded_gradient_equals_the_single_gpu_factor(chain_on_main):
    """with the owner's chain on the SIDE stream the copy that keeps panel j for the gradient raced with
    the staging of panel j + 2 into the same buffer.  The kept factor (lower triangle of Lf, the leaf inverses through
    the gradient, beta) must equal the single-GPU factor of the same covariance, in both stream placements, over many
    narrow panels and repeated evaluations."""
    import torch

    from andvaranaut_amd import MiGP
    from andvaranaut_amd.distributed import DistGP
    from oracle import gp_oracle as orc

    N, d = 3000, 4
    X, y = orc.synth_problem(N, d, seed=21)
    theta = orc.synth_theta(d, nkern=1)
    one = MiGP(X, y, "Matern52")
    v1, g1 = one.lml_grad(theta)
    torch.cuda.synchronize()
    L1 = torch.tril(one.K_t[:N, :N]).cpu().numpy()   # mi_gp_lml_grad leaves L (marginal form) in K_dev
    gp = DistGP(X, y, "Matern52", panel_width_tiles=2)   # 12 panels: many buffer reuses
    gp.set_option(3, chain_on_main)
    for _ in range(3):
        val, g = gp.lml_grad(theta)
        torch.cuda.synchronize()
        L = torch.tril(gp.Lf[:N, :N]).cpu().numpy()
        assert np.abs(L - L1).max() <= 1e-11 * np.abs(L1).max()
        assert abs(val - v1) <= SHARD_VS_SINGLE * abs(v1)
        assert _grad_close(g, g1, rtol=1e-8), (g, g1)
    one.close()


RCCL_WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from andvaranaut_amd import MiGP
from andvaranaut_amd.distributed import DistGP
from oracle import gp_oracle as orc
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
N, d = 5000, 6
X, y = orc.synth_problem(N, d, seed=3)
theta = orc.synth_theta(d)
one = MiGP(X, y, "Matern52")
ref, gref = one.lml_grad(theta)
one.close()
gp = DistGP(X, y, "Matern52", device=0, panel_width_tiles=4)
vals = {}
for on_main, piecewise in ((0, 2), (1, 2), (1, 1), (1, 0)):
    gp.set_option(3, on_main)
    gp.set_option(5, piecewise)
    v = gp.lml(theta)
    v2, g = gp.lml_grad(theta)
    assert v == v2 and abs(v - ref) <= 1e-11 * abs(ref), (on_main, piecewise, v, ref)
    assert np.max(np.abs(g - gref)) <= 1e-9 * np.max(np.abs(gref)), (on_main, piecewise)
    vals[(on_main, piecewise)] = v
assert vals[(1, 1)] == vals[(1, 0)]  # the same launches, staged piece by piece or at the end
assert gp.bytes_broadcast > 0
dist.destroy_process_group()
print(json.dumps({"ok": True}))
'''


def test_pipelined_exchange_over_a_one_rank_rccl_group(tmp_path):
    """The exchange code path of a multi-GPU run -- one RCCL broadcast per tile column behind mi_gp_shard_wait_piece, sends
    that the owner waits for only before it re-uses their buffer -- through the real backend ("nccl" = RCCL), as far as one
    GPU allows: a group of one rank, with the owner's chain on the main stream (the several-rank default) and every
    staging mode, against the single-GPU path."""
    script = tmp_path / "worker.py"
    script.write_text(RCCL_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert '"ok": true' in p.stdout
