"""Rehearse the sharded driver with more ranks than the test suite uses (gloo, every rank on the one GPU of the box):
python tools/rehearse_ranks.py WORLD   (WORLD <= 6: the box allows six GPU processes).  The parent never touches the GPU."""
import os, socket, subprocess, sys, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
WORKER = r'''
import os, sys, json, time
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from andvaranaut_amd import parallel
from andvaranaut_amd.distributed import DistGP
from oracle import gp_oracle as orc
rank, world, _ = parallel.init_distributed(backend="gloo")
torch.cuda.set_device(0)
out = {}
for (N, d, kernel, pwt) in [(6000, 4, "RBF", 4), (7300, 6, "Matern52", 4), (5000, 3, "RBF+Matern32", 2), (9000, 8, "RBF", None)]:
    X, y = orc.synth_problem(N, d, seed=N)
    kerns, ops = kernel.split("+"), ["+"] * (kernel.count("+"))
    theta = orc.synth_theta(d, nkern=len(kerns))
    gp = DistGP(X, y, kernel, device=0, panel_width_tiles=pwt)
    t0 = time.perf_counter()
    val = gp.lml(theta)
    ref = orc.lml(X, y, kerns, ops, theta)
    assert abs(val - ref) <= 1e-10 * abs(ref), (rank, N, val, ref)
    v2, g = gp.lml_grad(theta)
    _, gref = orc.lml_grad(X, y, kerns, ops, theta)
    scale = np.maximum(np.abs(gref), 1e-3 * np.max(np.abs(gref)))
    assert v2 == val and np.max(np.abs(g - gref) / scale) <= 1e-7, (rank, N, g, gref)
    out[str((N, pwt))] = [val, gp.npan, len(gp.own)] + g.tolist()
    del gp
    torch.cuda.empty_cache()
vals = parallel.gather_objects({k: [v[0]] + v[3:] for k, v in out.items()})
assert all(v == vals[0] for v in vals), vals
print(json.dumps({"rank": rank, "panels_owned": {k: v[2] for k, v in out.items()}, "npan": {k: v[1] for k, v in out.items()}}), flush=True)
'''
world = int(sys.argv[1])
assert 1 <= world <= 6
import tempfile
fd, path = tempfile.mkstemp(prefix="rehearse_worker_", suffix=".py")  # concurrent rehearsals must not share a worker file
with os.fdopen(fd, "w") as fh:
    fh.write(WORKER)
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
procs = []
for rank in range(world):
    env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    procs.append(subprocess.Popen([sys.executable, path, ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
ok = True
for p in procs:
    o, e = p.communicate(timeout=900)
    print(o.strip() or e[-2000:])
    ok = ok and p.returncode == 0
print("rehearsal", "ok" if ok else "FAILED")
sys.exit(0 if ok else 1)
