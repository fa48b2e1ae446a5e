"""RCCL (torch.distributed backend "nccl") on the GPU box.  The box the driver gives the tests has one GPU, so the
collective path is exercised with a ONE-rank RCCL group: process-group init, the ActNorm moment all-reduce, the
per-block gradient all-reduces issued between the hipGraph segments of the recorded training step (with the RCCL
watchdog thread alive next to the capture), the NLL all-reduce of bench.py.  With two or more GPUs the same checks
run with one rank per GPU through ``bench.py --gpus 2`` (which launches the ranks itself)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, timeout=900, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env.update(extra_env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                          "--train-steps", "4"] + list(flags), capture_output=True, text=True, timeout=timeout, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_with_a_one_rank_rccl_group_runs_every_collective():
    rec = _bench("--force-collectives")
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and "note" not in rec
    tr = rec["train"]
    assert "error" not in tr, tr
    assert tr["recorded_step"] is True and np.isfinite(tr["loss"]) and np.isfinite(tr["grad_norm"])
    assert tr["allreduce_ms"] is not None and tr["allreduce_ms"] > 0 and tr["compute_ms"] > 0
    assert tr["gradient_bytes"] > 700e6
    assert "error" not in rec["rtf_10s"] and rec["rtf_10s"]["rtf_per_gpu"] > 100
    assert rec["roofline"]["frac"] > 0.2 and rec["roofline"]["bound"] == "mfma"


def _one_rank_worker(q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from conftest import small_hparams
    from tf_flowavenet_amd import weights as W
    from tf_flowavenet_amd.training import Trainer
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    hp = small_hparams(n_block=3, n_flow=2, n_layer=2, hop_size=16, upsample_scales=[4, 4], num_mels=16)
    inp = W.synthetic_inputs(hp, 4, 256)
    x, c = torch.from_numpy(inp["x"]).reshape(4, 256).cuda(), torch.from_numpy(inp["c"]).cuda()
    res = {}
    for graph in (False, True):
        tr = Trainer(hp, W.synthetic_params(hp, 11), graph=graph)
        tr.opt.force_collectives = True
        tr.ddi(x, c)
        outs = [tuple(float(v) for v in tr.step(x, c)) for _ in range(4)]
        if graph:
            segs = tr._recorded[(tuple(x.shape), tuple(c.shape))]["segs"]
            # five hand-over points (3 blocks, the up-sampling convs, the optimiser); the up-sampling one carries no graph: block 0
            # and the up-sampling convs are reported back to back, an empty segment is not recorded
            assert sum(1 for g_, _ in segs if g_ is None) == 1 and [i_ for g_, i_ in segs if g_ is None] == [-1]
            res["segments"] = len(segs)
        res[graph] = (outs, tr.opt.w.cpu())
    q.put((res[True][0] == res[False][0], bool(torch.equal(res[True][1], res[False][1])), res["segments"]))
    dist.destroy_process_group()


def test_recorded_step_with_rccl_all_reduces_between_graph_segments_equals_the_eager_step():
    """Trainer over a one-rank RCCL group with the collectives forced on: the recording is cut at every block (4
    hipGraph segments + the optimiser for a 3-block model), each block's all-reduce is issued on RCCL's stream between
    two replays (the up-sampling range's right behind block 0's: nothing runs between the two hooks, so no empty segment is
    recorded for it), and loss / gradient norm / weights equal the eager trainer's bit for bit."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_worker, args=(q,), daemon=True)
    p.start()
    try:
        same_outs, same_w, segments = q.get(timeout=300)
    finally:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
    assert same_outs and same_w and segments == 5, (same_outs, same_w, segments)


def test_bench_two_ranks_sharing_the_gpu_run_every_leg_in_step():
    """`bench.py --gpus 2` (its own launcher -> torch.distributed.run -> two ranks) with FWN_BENCH_SHARE_GPU=1: both ranks on
    cuda:0, exchanges over gloo.  The multi-rank control flow of every leg - global ActNorm init, NLL all-reduce, the
    rank-0-only fp8 leg beside the other rank's training collectives (a collective mismatch here once aborted rank 0),
    per-block gradient all-reduces, the timing reductions - on the one GPU the test box has."""
    rec = _bench("--gpus", "2", extra_env={"FWN_BENCH_SHARE_GPU": "1"})
    assert rec["n_gpus"] == 2 and "note" not in rec
    for leg in ("train", "rtf_10s", "fp8"):
        assert rec[leg] is not None and "error" not in rec[leg], (leg, rec[leg])
    assert rec["train"]["n_gpus"] == 2 and rec["train"]["global_batch"] == 16 and rec["train"]["allreduce_ms"] > 0
    assert np.isfinite(rec["train"]["loss"]) and rec["rtf_10s"]["n_gpus"] == 2


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_bench_launches_two_rccl_ranks_itself():
    rec = _bench("--gpus", "2")
    assert rec["n_gpus"] == 2 and rec["train"]["n_gpus"] == 2 and "error" not in rec["train"]
    assert rec["train"]["allreduce_ms"] > 0 and 0.0 <= rec["train"]["overlap"] <= 1.0


def _zero1_one_rank_worker(q):
    """DataParallelAdam(exchange="zero1") over a one-rank RCCL group with the collectives forced on, next to the all-reduce
    exchange on the same gradients: three steps (below / above / below the clip threshold)."""
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import small_hparams
    from tf_flowavenet_amd import optim, weights as W
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    hp = small_hparams(n_block=2, n_flow=2)
    params = W.synthetic_params(hp, 11, actnorm="random")
    opts = {ex: optim.DataParallelAdam(hp, params, exchange=ex) for ex in optim.EXCHANGES}
    for o in opts.values():
        o.force_collectives = True
    rng = np.random.default_rng(5)
    norms = {ex: [] for ex in opts}
    for amp in (1e-4, 3.0, 0.05):
        g = torch.from_numpy((amp * rng.standard_normal(opts["zero1"].layout.size)).astype(np.float32)).cuda()
        for ex, o in opts.items():
            o.g[:g.numel()].copy_(g)
            norms[ex].append(float(o.step(loss_scale=64.0)))
    torch.cuda.synchronize()
    a, b = opts["allreduce"], opts["zero1"]
    q.put((norms, float((a.w - b.w).abs().max()), float(a.w.abs().max()), bool(torch.equal(a.m, b.m)), float((a.v - b.v).abs().max())))
    dist.destroy_process_group()


def test_sharded_optimiser_step_on_rccl_equals_the_all_reduce_step():
    """exchange="zero1" through RCCL (reduce onto the owner, all-reduced shard norms, fwn_clip_adam on the shard, broadcast of
    the masters): with one rank the shard is the whole buffer, so the step equals the all-reduce exchange's up to the one
    fp32 rounding of the norm (sqrt of the all-reduced square against fwn_grad_norm's own root)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_zero1_one_rank_worker, args=(q,), daemon=True)
    p.start()
    try:
        norms, dw, wmax, same_m, dv = q.get(timeout=300)
    finally:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
    for x, y in zip(norms["allreduce"], norms["zero1"]):
        assert abs(x - y) <= 1e-6 * abs(x), (x, y)
    assert dw <= 1e-6 * max(1.0, wmax) and dv <= 1e-9, (dw, wmax, same_m, dv)
