"""world_size-2 gloo tests (CPU) of the batch-shard path: shard bounds and the clip-weighted
NLL all-reduce, with a stand-in model whose per-clip values are known."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tf_flowavenet_amd import distributed as D


def test_shard_bounds_cover_batch_exactly():
    for n in (0, 1, 7, 8, 64):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                lo, hi = D.shard_bounds(n, r, world)
                assert 0 <= lo <= hi <= n
                got += list(range(lo, hi))
            assert got == list(range(n))
    with pytest.raises(ValueError):
        D.shard_bounds(4, 2, 2)


class FakeModel:
    """forward returns the means of per-clip scalars stored in x[:,0,0] / c[:,0,0]."""
    def forward(self, x, c):
        return x[:, 0, 0].mean(), c[:, 0, 0].mean()

    def reverse(self, z, c):
        return z * 2.0


def _worker(rank, world, port, n_clips, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    assert D.init_from_env("gloo") == world
    x = torch.arange(n_clips, dtype=torch.float32).reshape(n_clips, 1, 1) + 1.0
    c = -2.0 * x
    mom = torch.tensor([1.0 + rank, 10.0 * (rank + 1), 3.0], dtype=torch.float64)     # (sum x, sum x^2, rows) of a shard
    assert D.allreduce_sum_(mom) is mom and mom.tolist() == [3.0, 30.0, 6.0]            # the ActNorm moment exchange
    nll = D.sharded_forward(FakeModel(), x, c)
    out, (lo, hi) = D.sharded_reverse(FakeModel(), x, c)
    q.put((rank, nll.tolist(), lo, hi, None if out is None else out.reshape(-1).tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [5, 8, 1])
def test_global_nll_matches_single_process(n_clips):
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clips, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x = torch.arange(n_clips, dtype=torch.float32) + 1.0
    for rank, nll, lo, hi, out in res:
        assert abs(nll[0] - float(x.mean())) < 1e-6 and abs(nll[1] + 2 * float(x.mean())) < 1e-6
        assert (lo, hi) == D.shard_bounds(n_clips, rank, world)
        if hi > lo:
            assert out == (2.0 * x[lo:hi]).tolist()
        else:
            assert out is None
