"""Data-parallel optimiser step: oracle pins on CPU, the bucketed
all-reduce over gloo (world size 2), and - on the GPU - the fused norm/clip/Adam kernels against
the fp64 restatement of utils.py:34-60 + train.py:15-32,75-81."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import optim_np as O
from tf_flowavenet_amd import optim
from tf_flowavenet_amd import weights as W

from conftest import small_hparams


def test_learning_rate_schedule_matches_reference():
    for step, lr in [(0, 1e-3), (199999, 1e-3), (200000, 5e-4), (399999, 5e-4), (400000, 2.5e-4), (600000, 1e-3 / 6),
                     (10 ** 7, 1e-3 / 6)]:
        assert optim.learning_rate(step) == pytest.approx(lr) == pytest.approx(O.learning_rate(step))


def test_tf_adam_equals_torch_adam_when_eps_vanishes():
    """The TF form (eps outside the bias correction) coincides with torch.optim.Adam for eps -> 0,
    which pins the restated update rule against an independent implementation."""
    rng = np.random.default_rng(0)
    theta = rng.standard_normal(50)
    p = torch.nn.Parameter(torch.tensor(theta))
    opt = torch.optim.Adam([p], lr=1e-3, eps=1e-30)
    m = v = np.zeros(50)
    for step in range(1, 6):
        g = rng.standard_normal(50)
        theta, m, v = O.adam_step(theta, g, m, v, step, 1e-3, eps=1e-30)
        p.grad = torch.tensor(g)
        opt.step()
    np.testing.assert_allclose(p.detach().numpy(), theta, rtol=1e-10)


def test_clip_and_average_semantics():
    g1, g2 = [np.array([3.0, 0.0])], [np.array([0.0, 4.0 * 3])]
    avg = O.average_gradients([g1, g2])
    np.testing.assert_allclose(avg[0], [1.5, 6.0])
    clipped, gn = O.clip_by_global_norm(avg, 1.0)
    assert gn == pytest.approx(np.hypot(1.5, 6.0))
    np.testing.assert_allclose(np.linalg.norm(clipped[0]), 1.0)
    same, gn2 = O.clip_by_global_norm([np.array([0.3, 0.4])], 1.0)     # below the threshold: untouched
    np.testing.assert_allclose(same[0], [0.3, 0.4]) and gn2 == 0.5


def test_flat_layout_round_trip_and_buckets():
    hp = small_hparams()
    params = W.synthetic_params(hp, 3, actnorm="random")
    lay = optim.FlatLayout(hp)
    flat = lay.flatten(params)
    assert flat.size == lay.size and lay.size >= W.count_params(hp)
    for name, v in lay.views(flat).items():
        np.testing.assert_array_equal(v, params[name])
    for off, shape, n in lay.slots.values():
        assert off % 4 == 0
    assert optim.bucket_bounds(10, 4) == [(0, 4), (4, 8), (8, 10)]
    with pytest.raises(ValueError):
        optim.bucket_bounds(10, 0)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo")
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    for w in optim.allreduce_flat(g, bucket_elems=300):
        w.wait()
    q.put((rank, (g / world).tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_is_the_tower_mean():
    """utils.py:34-60 average_gradients == all-reduce(sum) / world, bucket by bucket (gloo, 2 ranks)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    towers = [[np.arange(1000.0) * (r + 1)] for r in range(2)]
    expect = O.average_gradients(towers)[0]
    for _, got in res:
        np.testing.assert_allclose(got, expect, rtol=1e-6)


@pytest.mark.gpu
def test_fused_clip_adam_matches_oracle_and_repacks():
    from tf_flowavenet_amd.model import FloWaveNet
    hp = small_hparams(n_block=2, n_flow=2)
    params = W.synthetic_params(hp, 11, actnorm="random")
    opt = optim.DataParallelAdam(hp, params)
    lay = opt.layout
    rng = np.random.default_rng(5)
    theta = lay.flatten(params).astype(np.float64)
    m = np.zeros_like(theta)
    v = np.zeros_like(theta)
    for step, amp in enumerate([1e-4, 3.0, 0.05], start=1):          # below / above / below the clip threshold
        g = (amp * rng.standard_normal(lay.size)).astype(np.float32)
        opt.g.copy_(torch.from_numpy(g))
        gn = float(opt.step(loss_scale=64.0))
        theta, m, v, gn0 = O.data_parallel_update(theta, [g.astype(np.float64)], m, v, step, scale=64.0)
        assert gn == pytest.approx(gn0, rel=1e-5)
        np.testing.assert_allclose(opt.w.cpu().numpy(), theta, rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(opt.m.cpu().numpy(), m, rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(opt.v.cpu().numpy(), v, rtol=1e-5, atol=1e-12)
    # fp32 masters -> bf16 MFMA layouts straight from the device buffer (utils.py:3-31 analogue)
    inp = W.synthetic_inputs(hp, 2, 128)
    x, c = torch.from_numpy(inp["x"]).cuda(), torch.from_numpy(inp["c"]).cuda()
    m_dev = FloWaveNet(hp).load_params(opt.master_views())
    m_host = FloWaveNet(hp).load_params({k: t.cpu().numpy() for k, t in opt.master_views().items()})
    a, b = m_dev.forward(x, c), m_host.forward(x, c)
    assert float(a[0]) == float(b[0]) and float(a[1]) == float(b[1])


def test_block_ranges_tile_the_flat_buffer_in_backward_friendly_order():
    """The per-block all-reduce ranges of the training step: contiguous, exhaustive, one per block plus
    the up-sampling convs, and every parameter of a block inside its range."""
    hp = small_hparams(n_block=3, n_flow=2)
    lay = optim.FlatLayout(hp)
    fake = type("F", (), {"layout": lay, "block_ranges": optim.DataParallelAdam.block_ranges})()
    rngs = fake.block_ranges()
    assert [k for k, _, _ in rngs] == ["upsample", "Block_0", "Block_1", "Block_2"]
    assert rngs[0][1] == 0 and rngs[-1][2] == lay.size
    assert all(a[2] == b[1] for a, b in zip(rngs, rngs[1:]))
    for name, (off, _, n) in lay.slots.items():
        key = name.split("/")[0] if name.startswith("Block_") else "upsample"
        lo, hi = next((lo, hi) for k, lo, hi in rngs if k == key)
        assert lo <= off and off + n <= hi
