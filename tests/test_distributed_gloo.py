"""N>1 path on CPU: two gloo ranks run their share of independent chains and gather the draws."""
import os
import socket
import subprocess
import sys

from conftest import ROOT

WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1])
from andvaranaut_amd import parallel
from oracle import gp_oracle as orc
from andvaranaut_amd.priors import HyperModel
rank, world, local = parallel.init_distributed(backend="gloo")
assert world == 2
X, y = orc.synth_problem(24, 1, seed=0)
model = HyperModel(1, ["RBF"], noise=True)
f = lambda q: model.logp_dlogp(q, lambda th: orc.lml_grad(X, y, ["RBF"], [], th))
assert parallel.shard_units(5, rank, world) == ([0, 2, 4] if rank == 0 else [1, 3])
parallel.barrier()
q, lp = parallel.sample_chains_distributed(f, model.initial_point(), n_chains=3, seed=11, draws=40, tune=40)
t = parallel.max_over_ranks(1.0 + rank)
assert t == 2.0
assert q.shape == (3, 40, model.nq) and lp.shape == (3, 40) and np.isfinite(lp).all()
# every rank holds identical gathered results, in global chain order
digest = float(np.sum(q * np.arange(1, 4)[:, None, None]))
allv = parallel.gather_objects(digest)
assert abs(allv[0] - allv[1]) == 0.0
if rank == 0:
    print(json.dumps({"ok": True, "digest": digest}))
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gloo_chain_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    assert '"ok": true' in outs[0][0]
