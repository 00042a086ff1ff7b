"""Multi-GPU plumbing.  The path shards by independent units -- MAP restarts,
MCMC chains, train/test refits (gpmcmc.py:328-343,351,947) -- so ranks hold full replicas and there is
no data-path collective: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm,
"gloo" on CPU), a barrier to bracket timed regions, a MAX reduction of the elapsed time and an
object gather of the (tiny) per-chain draws."""
import os

import numpy as np


def init_distributed(backend=None):
    """(rank, world, local_rank); initialises torch.distributed when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if not dist.is_initialized():
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            if backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend=backend)
    return rank, world, local_rank


def barrier():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device=None):
    """MAX of a Python float over all ranks (elapsed time of a timed region)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_units(n_units, rank, world):
    """Indices of the independent units (chains / restarts / theta evaluations) this rank owns."""
    return list(range(rank, n_units, world))


def gather_objects(obj):
    """All ranks receive the list of every rank's object (per-chain draws are a few kB)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def sample_chains_distributed(logp_dlogp, q0, n_chains, seed=0, **nuts_kwargs):
    """Run this rank's share of ``n_chains`` NUTS chains and gather all draws on every rank.
    Returns (q [chain, draw, nq], lp [chain, draw]) in global chain order."""
    import torch.distributed as dist

    from .nuts import sample_chain

    rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    seeds = np.random.SeedSequence(seed).spawn(n_chains)
    mine = {}
    for c in shard_units(n_chains, rank, world):
        r = sample_chain(logp_dlogp, q0, seed=seeds[c], **nuts_kwargs)
        mine[c] = (r["q"], r["lp"])
    merged = {}
    for part in gather_objects(mine):
        merged.update(part)
    q = np.stack([merged[c][0] for c in range(n_chains)])
    lp = np.stack([merged[c][1] for c in range(n_chains)])
    return q, lp
