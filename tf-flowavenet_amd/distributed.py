"""Multi-GPU plumbing for the flow path: one process per GPU, batch sharded across ranks.

The path shards naturally by batch: clips are independent in
``forward`` / ``reverse``, so inference needs no data-path collective at all.  The only
exchange is the scalar NLL reduction: the reference's ``log_p`` / ``logdet`` are means over
the whole batch (model.py:135,343), so the global values are the clip-count-weighted mean of
the per-rank values - one all-reduce of three floats (RCCL over xGMI on GPUs, gloo on CPU).
This replaces the tower loop + gather-to-one-device of train.py:43-57,75-77 for this path.
"""
from __future__ import annotations

import os


def init_from_env(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torchrun contract)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 or dist.is_initialized():
        return world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group(backend, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return world


def shard_bounds(n_items: int, rank: int, world: int):
    """Contiguous [lo, hi) slice of a global batch for ``rank``; remainders go to the low ranks."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world %d/%d" % (rank, world))
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_sum_(t, group=None):
    """In-place sum of ``t`` over the ranks of ``group`` in the order of the current stream (the ActNorm moments of
    a data-parallel init, ``FloWaveNet._forward_init_dp``).  RCCL reduces device tensors directly; a gloo group (CPU
    tests, or several ranks sharing one GPU) goes through a host copy."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t
    if t.is_cuda and dist.get_backend(group) != "nccl":
        host = t.cpu()
        dist.all_reduce(host, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, group=group)
    return t


def global_nll(log_p, logdet, local_clips: int, group=None):
    """Clip-weighted global mean of the per-rank (log_p, logdet); returns a 2-element tensor.

    Ranks that hold zero clips contribute nothing (empty shards are legal)."""
    import torch
    import torch.distributed as dist
    vals = torch.stack([log_p.reshape(()).float(), logdet.reshape(()).float()]) if local_clips > 0 \
        else torch.zeros(2, device=log_p.device)
    buf = torch.cat([vals * float(local_clips), vals.new_tensor([float(local_clips)])])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf, group=group)
    return buf[:2] / buf[2]


def sharded_forward(model, x, c, group=None):
    """Global-batch ``FloWaveNet.forward``: each rank runs its shard, NLL is all-reduced."""
    import torch.distributed as dist
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    lo, hi = shard_bounds(int(x.shape[0]), rank, world)
    if hi > lo:
        log_p, logdet = model.forward(x[lo:hi], c[lo:hi])
    else:
        import torch
        log_p = logdet = torch.zeros((), device=x.device)
    return global_nll(log_p, logdet, hi - lo, group)


def sharded_reverse(model, z, c, group=None):
    """Global-batch ``FloWaveNet.reverse``: returns this rank's slice and its [lo, hi) bounds."""
    import torch.distributed as dist
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    lo, hi = shard_bounds(int(z.shape[0]), rank, world)
    return (model.reverse(z[lo:hi], c[lo:hi]) if hi > lo else None), (lo, hi)
