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


# ---- lock-step checks of the data-parallel training job -----------------------------
class _FakeOpt:
    """The pieces of optim.DataParallelAdam that restore_checkpoint / weights_identical touch, on CPU tensors."""
    def __init__(self, n=4, group=None):
        self.w, self.m, self.v, self.global_step, self.group = torch.zeros(n), torch.zeros(n), torch.zeros(n), 0, group

    def master_views(self):
        return {"p": self.w}

    from tf_flowavenet_amd.optim import DataParallelAdam as _D
    weights_checksum, weights_identical = _D.weights_checksum, _D.weights_identical


class _FakeTrainer:
    def __init__(self, n=4):
        self.opt = _FakeOpt(n)


def _lockstep_worker(rank, world, port, tmp, q):
    import numpy as np
    from tf_flowavenet_amd import train as T
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    D.init_from_env("gloo")
    res = {}
    # identical masters -> identical; one differing last bit on one rank -> not identical
    tr = _FakeTrainer()
    tr.opt.w[:] = torch.tensor([1.0, -2.0, 3.5, 0.0])
    res["same"] = tr.opt.weights_identical()
    if rank == 1:
        tr.opt.w.view(torch.int32)[2] += 1
    res["one_bit"] = tr.opt.weights_identical()
    # every rank restores the step rank 0 sees
    if rank == 0:
        src = _FakeTrainer()
        src.opt.w[:] = 5.0
        src.opt.global_step = 100
        T.save_checkpoint(os.path.join(tmp, "flowavenet_model.ckpt-100.npz"), src)
    dist.barrier()
    tr = _FakeTrainer()
    res["restored"] = (T.restore_checkpoint(tmp, tr), float(tr.opt.w[0]))
    # a newer file that only rank 1 can see (rank 0's directory listing lags): rank 0's choice wins, no divergence
    tmp1 = os.path.join(tmp, "r%d" % rank)
    os.makedirs(tmp1)
    src = _FakeTrainer()
    src.opt.w[:] = 1.0
    src.opt.global_step = 10
    T.save_checkpoint(os.path.join(tmp1, "flowavenet_model.ckpt-10.npz"), src)
    if rank == 1:
        src.opt.global_step = 20
        T.save_checkpoint(os.path.join(tmp1, "flowavenet_model.ckpt-20.npz"), src)
    tr = _FakeTrainer()
    res["lagging"] = T.restore_checkpoint(tmp1, tr)
    # a checkpoint only one rank can read at all: abort, never resume from different steps
    tmp2 = os.path.join(tmp, "q%d" % rank)
    os.makedirs(tmp2)
    if rank == 0:
        T.save_checkpoint(os.path.join(tmp2, "flowavenet_model.ckpt-10.npz"), src)
    else:
        with open(os.path.join(tmp2, "flowavenet_model.ckpt-10.npz"), "wb") as f:
            f.write(b"PK\x03\x04 torn")
    try:
        T.restore_checkpoint(tmp2, _FakeTrainer())
        res["torn"] = "no error"
    except RuntimeError as e:
        res["torn"] = "disagree" if "disagree" in str(e) else str(e)
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_restore_the_same_step_and_notice_diverged_weights(tmp_path):
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_lockstep_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        r = res[rank]
        assert r["same"] is True and r["one_bit"] is False
        assert r["restored"] == (100, 5.0)
        assert r["lagging"] == 10
        assert r["torn"] == "disagree"


# ---- the bf16 gradient exchange (round 5: optim.DataParallelAdam(grad_reduce_dtype="bf16")) ----
def _bf16_exchange_worker(rank, world, port, q):
    import numpy as np
    from tf_flowavenet_amd import optim
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    D.init_from_env("gloo")
    n = 100003
    rng = np.random.default_rng(7 + rank)
    # heavy-tailed like a real gradient: a few large entries, most tiny, some exactly zero
    g_np = (rng.standard_normal(n) * np.exp(rng.standard_normal(n) * 2.0)).astype(np.float32)
    g_np[rng.integers(0, n, 500)] = 0.0
    res = {}
    for dt in ("fp32", "bf16"):
        g = torch.from_numpy(g_np.copy())
        stage = torch.empty(n, dtype=torch.bfloat16) if dt == "bf16" else None
        works = optim.allreduce_flat(g, None, 30000, True, dt, stage)        # four buckets, the last one short
        assert len(works) == 4
        for w in works:
            w.wait()
        res[dt] = g.numpy().copy()
    # one range started early (the per-block exchange of the training step), then the rest
    g = torch.from_numpy(g_np.copy())
    stage = torch.empty(n, dtype=torch.bfloat16)
    w1 = optim.allreduce_slice(g, 50000, n, None, "bf16", stage)
    w0 = optim.allreduce_slice(g, 0, 50000, None, "bf16", stage)
    w1.wait(); w0.wait()
    res["ranges"] = g.numpy().copy()
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_gradient_exchange_is_lockstep_and_bounds_the_step():
    """grad_reduce_dtype = "bf16": every rank receives the SAME reduced gradient (lock-step of the masters is untouched), it
    is the fp32 sum up to two bf16 roundings per element, and the clip + Adam step built from it (oracle/optim_np.py, the
    restatement of train.py:15-32) moves the weights by what the fp32 exchange would up to that noise."""
    import numpy as np
    from oracle import optim_np
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bf16_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for key in ("fp32", "bf16", "ranges"):
        assert np.array_equal(res[0][key], res[1][key]), key                  # identical on both ranks, bit for bit
    assert np.array_equal(res[0]["bf16"], res[0]["ranges"])                    # bucket bounds do not matter
    g32, g16 = res[0]["fp32"].astype(np.float64), res[0]["bf16"].astype(np.float64)
    # each rank's term rounded to bf16 (2^-9 relative each), the sum rounded once more - to nearest by RCCL, by truncation in
    # gloo's bf16 reduction (measured here: 1.9 x 2^-8): |error| <= 3 x 2^-8 (|a| + |b|) per element (relative to the SUM it
    # can be anything where the two ranks' terms cancel)
    terms = []
    for rank in range(world):
        rng = np.random.default_rng(7 + rank)
        t = (rng.standard_normal(g32.size) * np.exp(rng.standard_normal(g32.size) * 2.0)).astype(np.float32)
        t[rng.integers(0, g32.size, 500)] = 0.0
        terms.append(np.abs(t.astype(np.float64)))
    err = np.abs(g16 - g32)
    assert (err <= 3 * 2.0 ** -8 * (terms[0] + terms[1]) + 1e-30).all(), float((err / (terms[0] + terms[1] + 1e-30)).max())
    cos = float((g16 * g32).sum() / np.sqrt((g16 * g16).sum() * (g32 * g32).sum()))
    assert cos > 1.0 - 1e-5, cos
    assert abs(np.sqrt((g16 * g16).sum()) / np.sqrt((g32 * g32).sum()) - 1.0) < 1e-3    # the global norm the clip sees
    # the step: same masters, same slots, the two reduced gradients
    rng = np.random.default_rng(3)
    w0 = rng.standard_normal(g32.size)
    outs = []
    for g in (g32, g16):
        w, m, v = w0.copy(), np.zeros_like(w0), np.zeros_like(w0)
        for step in range(1, 4):
            (gc,), _ = optim_np.clip_by_global_norm([g / world], 1.0)
            w, m, v = optim_np.adam_step(w, gc, m, v, step, 1e-3)
        outs.append(w)
    # Adam normalises the step (every element moves by ~lr per step whatever its gradient's size), so where the ranks' terms
    # cancel and the bf16 sum differs in relative terms - or in sign - an element can end up to 2 x 3 lr away; those are few:
    d = np.abs(outs[1] - outs[0])
    assert d.max() <= 6.1e-3 and np.median(d) < 3e-6 and (d > 3e-4).mean() < 0.01, (d.max(), np.median(d), (d > 3e-4).mean())


# ---- the sharded-optimiser exchange (round 6: optim.DataParallelAdam(exchange="zero1"), SURVEY section 8e's alternative) ----
def _zero1_worker(rank, world, port, q):
    import numpy as np
    from oracle import optim_np
    from tf_flowavenet_amd import optim
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    D.init_from_env("gloo")
    n = 100003                      # not a multiple of world * 4: the last shard is short
    rng = np.random.default_rng(11 + rank)
    g_np = (rng.standard_normal(n) * np.exp(rng.standard_normal(n))).astype(np.float32)
    w0 = np.random.default_rng(5).standard_normal(n).astype(np.float32)         # the same masters on every rank
    # all-reduce path: every rank reduces everything and updates everything
    g = torch.from_numpy(g_np.copy())
    for wk in optim.allreduce_flat(g, None, 30000, True):
        wk.wait()
    g_all = g.numpy().astype(np.float64) / world
    (gc,), gn_all = optim_np.clip_by_global_norm([g_all], 1.0)
    w_all, _, _ = optim_np.adam_step(w0.astype(np.float64), gc, np.zeros(n), np.zeros(n), 1, 1e-3)
    # sharded path: reduce onto the owners, shard norms all-reduced, update of the own shard, gather
    g = torch.from_numpy(g_np.copy())
    (lo, hi), _ = optim.reduce_to_owners(g)
    bounds = optim.shard_bounds(n, world)
    assert (lo, hi) == bounds[rank] and bounds[0][0] == 0 and bounds[-1][1] == n and all(b[0] % 4 == 0 for b in bounds)
    gs = g[lo:hi].numpy().astype(np.float64) / world
    sq = torch.tensor([float((gs * gs).sum())], dtype=torch.float64)
    dist.all_reduce(sq)
    gn = float(sq.sqrt())
    w = torch.from_numpy(w0.copy())
    ws, _, _ = optim_np.adam_step(w0[lo:hi].astype(np.float64), gs * (1.0 / max(gn, 1.0)), np.zeros(hi - lo), np.zeros(hi - lo), 1, 1e-3)
    w[lo:hi] = torch.from_numpy(ws.astype(np.float32))
    optim.gather_from_owners(w)
    q.put((rank, dict(shard=(lo, hi), g_shard=g[lo:hi].numpy().copy(), g_all=(g_all * world).astype(np.float32), gn=gn, gn_all=gn_all,
                      w=w.numpy().copy(), w_all=w_all.astype(np.float32))))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_optimiser_exchange_reproduces_the_all_reduced_update():
    """exchange = "zero1" on two gloo ranks: a shard's owner holds exactly the all-reduced gradient of its shard, the global
    norm from the all-reduced shard norms is the norm of the whole gradient, and after clip + Adam on the shards and the
    gather EVERY rank holds the same masters - the ones the all-reduce path's full update (oracle/optim_np.py, the restatement
    of train.py:15-32 / utils.py:34-60) produces."""
    import numpy as np
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_zero1_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        lo, hi = res[r]["shard"]
        assert np.array_equal(res[r]["g_shard"], res[r]["g_all"][lo:hi])           # the owner's shard = the all-reduced gradient
        assert abs(res[r]["gn"] - res[r]["gn_all"]) <= 1e-12 * res[r]["gn_all"]
    assert np.array_equal(res[0]["w"], res[1]["w"])                                # lock-step: identical masters everywhere
    assert np.array_equal(res[0]["w"], res[0]["w_all"])                            # ... and the all-reduce path's
    assert res[0]["shard"][1] == res[1]["shard"][0] and res[1]["shard"][1] == 100003
