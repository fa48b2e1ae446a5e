"""Training-step benchmark (BASELINE configs[2]: data-parallel training, batch 8 per GPU, 6400-sample
crops, full n_block=8 n_flow=6 model).  One process per GPU:

    python tools/bench_train.py --steps 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        tools/bench_train.py --gpus N --steps 10

Each step = gradients of -(log_p + logdet) on this rank's batch (HIP stage kernels), RCCL all-reduce
of the flat fp32 gradient, global-norm clip, Adam.  Prints one JSON line (whole-job samples/s)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--samples", type=int, default=6400)
    a = ap.parse_args()
    import torch.distributed as dist
    from tf_flowavenet_amd.hparams import default_hparams
    from tf_flowavenet_amd import weights as W
    from tf_flowavenet_amd.training import Trainer
    world, rank = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0))
    # FWN_BENCH_SHARE_GPU=1: plumbing check on a one-GPU box (every rank on cuda:0, exchanges over gloo)
    share = os.environ.get("FWN_BENCH_SHARE_GPU") == "1"
    torch.cuda.set_device(0 if share else int(os.environ.get("LOCAL_RANK", 0)))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if share else "nccl")
    hp = default_hparams()
    inp = W.synthetic_inputs(hp, a.batch, a.samples)
    x = torch.from_numpy(np.roll(inp["x"], 997 * rank, axis=1)).reshape(a.batch, a.samples).cuda()
    c = torch.from_numpy(np.roll(inp["c"], rank, axis=1)).cuda()
    tr = Trainer(hp, W.synthetic_params(hp, 1234))
    tr.ddi(x, c)
    for _ in range(a.warmup):
        tr.step(x, c)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss, lp, ld, gn = tr.step(x, c)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        tm = torch.tensor([el], device="cuda", dtype=torch.float64)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        el = float(tm)
    if rank == 0:
        print(json.dumps({"metric": "training audio samples/sec (forward + backward + all-reduce + clip/Adam), n_block=8 bf16",
                          "value": a.batch * a.samples * world * a.steps / el, "unit": "samples/s", "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3,
                          "config": {"workload": "configs[2]: data-parallel training step", "clips_per_gpu": a.batch,
                                     "samples_per_clip": a.samples}, "loss": float(loss), "grad_norm": float(gn)}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
