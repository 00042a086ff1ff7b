"""BASELINE.json configs at their full sizes through the C-ABI: config 3's LML + gradient leg (Matern-5/2, N=16384,
d=16), config 5 (RBF, N=8192, d=8: LML against the oracle, then NUTS draws on the device), and config 4's sharded
driver at N=65536 on one rank.  Size-independent properties where the oracle would take minutes."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _mods():
    import torch

    assert torch.cuda.is_available()
    from andvaranaut_amd import MiGP
    from oracle import gp_oracle as orc

    return MiGP, orc


def test_config3_lml_grad_n16384():
    """MAP-loop evaluation of config 3: (a) bit-identical on re-evaluation, (b) the directional derivative along the
    gradient equals a central difference of the LML, (c) the sharded driver on one rank returns the same LML and
    gradient (different launch sequence: panel-by-panel, slab-by-slab K^-1), (d) the leading block of K^-1 ... is
    covered by (b); the LML itself is pinned at this size by tests/test_gpu_lml.py::test_full_size_properties_n16384."""
    MiGP, orc = _mods()
    from andvaranaut_amd.distributed import DistGP

    N, d = 16384, 16
    X, y = orc.synth_problem(N, d, seed=0)
    theta = orc.synth_theta(d)
    gp = MiGP(X, y, "Matern52")
    v1, g1 = gp.lml_grad(theta)
    v2, g2 = gp.lml_grad(theta)
    assert v1 == v2 and np.array_equal(g1, g2) and np.isfinite(v1) and np.all(np.isfinite(g1))
    assert gp.lml(theta) == v1  # the LML entry point and the LML + gradient entry point factor identically
    # (b) central difference along the gradient direction in log-parameters (jitter excluded: it is not a free parameter)
    u = g1[:-1] * theta[:-1]
    u /= np.linalg.norm(u)
    h = 1e-4
    tp, tm = theta.copy(), theta.copy()
    tp[:-1] *= np.exp(h * u)
    tm[:-1] *= np.exp(-h * u)
    fd = (gp.lml(tp) - gp.lml(tm)) / (2 * h)
    an = float(np.dot(g1[:-1] * theta[:-1], u))
    assert abs(fd - an) <= 1e-6 * abs(an), (fd, an)
    gp.close()
    del gp
    # (c) sharded driver, one rank
    dgp = DistGP(X, y, "Matern52")
    vd, gd = dgp.lml_grad(theta)
    assert abs(vd - v1) <= 1e-10 * abs(v1), (vd, v1)  # (two algorithms since round 5: see test_gpu_distributed.SHARD_VS_SINGLE)
    scale = np.maximum(np.abs(g1), 1e-3 * np.max(np.abs(g1)))
    assert np.max(np.abs(gd - g1) / scale) <= 1e-8, (gd, g1)


def test_config5_n8192_lml_vs_oracle_then_nuts_on_device():
    """Config 5 per chain: LML at rtol 1e-10 against the oracle (8 s of CPU), then a few NUTS transitions whose every
    leapfrog step is one device LML + gradient evaluation: finite log-posterior, no divergences."""
    MiGP, orc = _mods()
    from andvaranaut_amd.nuts import sample_chain
    from andvaranaut_amd.optimize import find_MAP
    from andvaranaut_amd.priors import HyperModel

    N, d = 8192, 8
    X, y = orc.synth_problem(N, d, seed=1)
    theta = orc.synth_theta(d)
    gp = MiGP(X, y, "RBF")
    val = gp.lml(theta)
    ref = orc.lml(X, y, ["RBF"], [], theta)
    assert abs(val - ref) <= 1e-10 * abs(ref), (val, ref)
    v2, g = gp.lml_grad(theta)
    assert v2 == val and np.all(np.isfinite(g))
    model = HyperModel(d, ["RBF"], noise=True, jitter=1e-6)
    qmap, info = find_MAP(lambda q: model.logp_dlogp(q, gp.lml_grad, jacobian=False), model.initial_point(), maxeval=25)
    assert np.isfinite(info["logp"])
    r = sample_chain(lambda q: model.logp_dlogp(q, gp.lml_grad), qmap, draws=5, tune=5, seed=0)
    assert r["q"].shape == (5, model.initial_point().size)
    assert np.all(np.isfinite(r["lp"])) and r["diverging"] == 0 and r["n_leapfrog"] >= 10
    gp.close()


def test_config4_sharded_driver_n65536_one_rank_equals_single_gpu_path():
    """Config 4's shape through the sharded driver (64 column panels of 1024, recursive in-panel updates, one panel-list launch
    per step) against the single-GPU path (super-panels with look-ahead, extended panels, a column-by-column tail) on the same
    data: 1e-10."""
    MiGP, _ = _mods()
    from andvaranaut_amd.distributed import DistGP
    from bench import synth_problem

    N, d = 65536, 32
    X, y = synth_problem(N, d, seed=0)
    theta = np.concatenate([np.exp(np.linspace(np.log(0.8), np.log(3.0), d)), [1.7], [1.0], [1e-4, 1e-6]])
    gp = MiGP(X, y, "RBF", need_grad=False)
    v1 = gp.lml(theta)
    assert gp.info == 0 and np.isfinite(v1)
    gp.close()
    del gp
    dgp = DistGP(X, y, "RBF")
    assert dgp.npan == 64 and dgp.pw == 1024
    vd = dgp.lml(theta)
    assert abs(vd - v1) <= 1e-10 * abs(v1), (vd, v1)  # (two algorithms since round 5: see test_gpu_distributed.SHARD_VS_SINGLE)


NCCL_WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))   # RCCL, one rank
from andvaranaut_amd.distributed import DistGP
from oracle import gp_oracle as orc
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
for (N, d, kernel) in [(1500, 4, "RBF"), (2900, 3, "Matern52")]:
    X, y = orc.synth_problem(N, d, seed=N)
    theta = orc.synth_theta(d, nkern=1)
    gp = DistGP(X, y, kernel, device=0)
    assert gp.collective and gp.npan >= 3
    val = gp.lml(theta)
    ref = orc.lml(X, y, [kernel], [], theta)
    assert abs(val - ref) <= 1e-10 * abs(ref), (N, val, ref)
    assert gp.bytes_broadcast > 0            # every panel went through dist.broadcast on the RCCL communicator
    v2, g = gp.lml_grad(theta)
    _, gref = orc.lml_grad(X, y, [kernel], [], theta)
    scale = np.maximum(np.abs(gref), 1e-3 * np.max(np.abs(gref)))
    assert v2 == val and np.max(np.abs(g - gref) / scale) <= 1e-7, (N, g, gref)
dist.barrier()
dist.destroy_process_group()
print(json.dumps({"ok": True}))
'''


def _env(port):
    return dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                HSA_ENABLE_IPC_MODE_LEGACY="0")


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_sharded_driver_over_rccl_one_rank(tmp_path):
    """backend="nccl" (RCCL) with a one-rank group: communicator initialisation, the per-panel broadcast, the U-panel
    exchange of the sharded gradient and the scalar all-reduces all execute on the transport a multi-GPU run uses."""
    script = tmp_path / "nccl_worker.py"
    script.write_text(NCCL_WORKER)
    p = subprocess.run([sys.executable, str(script), ROOT], env=_env(_free_port()), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert '"ok": true' in p.stdout


def test_bench_sharded_line_over_rccl_one_rank():
    """bench.py --sharded end to end (own process: it creates the process group)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--sharded", "--n", "3000", "--d", "4", "--kernel", "Matern52",
                        "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and line["scaling"] == "strong"
    assert line["sharded"]["collectives"] == "rccl" and line["sharded"]["bytes_broadcast_per_step"] > 0
    assert line["sharded"]["finite"] and line["value"] > 0


def test_bench_gpus_2_on_a_one_gpu_box_fails_loudly():
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("box has two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "needs 2 GPUs" in (p.stderr + p.stdout)
