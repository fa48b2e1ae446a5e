#!/usr/bin/env python3
"""Headline benchmark: FloWaveNet flow forward (NLL) + inverse (synthesis) throughput.

    python bench.py --gpus N --steps K --warmup W          (N>1 via torch.distributed.run)

A *step* is one pass of the hot path over one batch of synthetic input per GPU:
``FloWaveNet.forward`` (log_p, logdet) on B clips followed by ``FloWaveNet.reverse`` on
B latent clips, n_block=8 / n_flow=6 / n_layer=2, bf16 compute with fp32 accumulation,
T = 16128 samples per clip (63 frames x hop 256; 16000 itself is not a legal length,
SURVEY section 0).  Inputs and packed weights are resident in HBM before the timed
region.  ``value`` = audio samples pushed through a flow pass per second, whole job:
(B*T forward + B*T inverse) * n_gpus / step time.  The batch is sharded across ranks
(weak scaling); the only exchange on the path is the 2-scalar NLL all-reduce.

One JSON line on stdout (rank 0) with ``roofline`` and ``cpu_baseline`` objects as
described in DESIGN.md "Measurement".
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_SAMPLE = 16527360       # per direction, n_block=8 n_flow=6 n_layer=2
MFMA_PEAK_TFLOPS = 2500.0        # dense bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def flop_per_sample(hp):
    """Algorithmic FLOP per audio sample per direction (SURVEY 8d formula, dead res_conv excluded)."""
    total = 0.0
    half = hp.num_mels // 2
    for i in range(hp.n_block):
        c = 2 << i
        cin = half * c
        mac = 3 * (c // 2) * 256 + hp.n_layer * (2 * 3 * 256 * 256 + 2 * cin * 256) \
            + (2 * hp.n_layer - 1) * 256 * 256 + 256 * 256 + 256 * c
        total += 2.0 * mac * hp.n_flow / c
    return total


def cpu_baseline(hp, params, t, budget_s=25.0):
    """torch-CPU fp32 restatement (oracle/flowavenet_torch.py) timed on a bounded sample."""
    import torch
    from oracle import flowavenet_torch as ot
    from tf_flowavenet_amd import weights as W
    # The GPU box exposes 256 hardware threads; torch's conv kernels stop scaling (and with
    # 256 threads collapse) well before that, so the baseline uses a fixed 16 threads and says so.
    ncore = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(ncore)
    fp = ot.fold(params, hp, dtype=torch.float32)
    inp = W.synthetic_inputs(hp, 1, t)
    x, c, z = (torch.from_numpy(inp[k]) for k in ("x", "c", "z"))
    times = []
    t_start = time.perf_counter()
    with torch.no_grad():
        for it in range(4):
            t0 = time.perf_counter()
            ot.forward(fp, x, c, hp)
            ot.reverse(fp, z, c, hp)
            dt = time.perf_counter() - t0
            if it > 0:
                times.append(dt)
            if time.perf_counter() - t_start > budget_s and times:
                break
    med = float(np.median(times))
    return {"value": 2.0 * t / med, "unit": "samples/s", "cores": ncore, "kind": "port",
            "sample": "torch-CPU fp32 restatement (TF 1.12 unavailable), full n_block=%d model, B=1, T=%d, "
                      "forward+inverse, median of %d timed passes after 1 warm-up" % (hp.n_block, t, len(times))}


def gate_roofline(model, hp, b, t, iters=30):
    """Time the dominant kernel (block-0 gated dilated layer) alone with HIP events on the
    launch stream and price it against the dense bf16 MFMA peak."""
    import ctypes as C
    import torch
    from tf_flowavenet_amd import _lib
    lib = _lib.load()
    d = model._packed.flow_descs[0]
    ti = t // 2
    m = b * ti
    dev = torch.device("cuda")
    h = (torch.randn(m, 256, device=dev) * 0.5).to(torch.bfloat16)
    ca = torch.rand(m, d.cin, device=dev).to(torch.bfloat16)
    o = torch.empty(m, 256, device=dev, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream

    def launch():
        _lib.check(lib.fwn_gate(C.byref(d), 0, h.data_ptr(), ca.data_ptr(), None, o.data_ptr(), m, ti, st), "fwn_gate")

    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / iters
    flops = 2.0 * m * (768 + d.cin) * 512
    ach = flops / sec / 1e12
    traffic = None      # HBM bytes per launch from the committed PMC profile of this exact launch shape
    tj = os.path.join(ROOT, "profiles", "r01_gate_traffic.json")
    if os.path.exists(tj):
        with open(tj) as f:
            rec = json.load(f)
        if rec.get("rows") == m:
            traffic = rec["traffic_bytes"]
    return {"bound": "mfma", "kernel": "gate_halo_kernel<256,256,GateProb> (block 0 gated dilated layer, fwn_gate)",
            "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS,
            "traffic": traffic, "launch_us": sec * 1e6, "flop_per_launch": flops, "rows": m}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU (hparams.batch_size)")
    ap.add_argument("--samples", type=int, default=16128, help="samples per clip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cond-mode", type=int, default=0)
    ap.add_argument("--serial", action="store_true", help="forward and inverse on one stream (no overlap)")
    ap.add_argument("--lanes", type=int, default=3,
                    help="HIP streams per direction; successive (independent) steps rotate over them")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from tf_flowavenet_amd.hparams import default_hparams
    from tf_flowavenet_amd import weights as W
    from tf_flowavenet_amd.model import FloWaveNet

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # FWN_BENCH_SHARE_GPU=1 is a plumbing test for boxes with one GPU: every rank uses cuda:0 and the
    # scalar exchanges go over gloo.  The measured configuration is always one GPU per rank + RCCL.
    share_gpu = os.environ.get("FWN_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    def allreduce(tensor, op=dist.ReduceOp.SUM):
        if share_gpu:
            host = tensor.cpu()
            dist.all_reduce(host, op=op)
            tensor.copy_(host)
        else:
            dist.all_reduce(tensor, op=op)

    hp = default_hparams()
    b, t = args.batch, args.samples
    params = W.synthetic_params(hp, 1234)
    model = FloWaveNet(hp, init=True, device=dev, cond_mode=args.cond_mode).load_params(params)
    inp = W.synthetic_inputs(hp, b, t)
    # each rank works on its own shard of the global batch (different clips per rank)
    roll = rank * 997
    x = torch.from_numpy(np.roll(inp["x"], roll, axis=1)).to(dev)
    c = torch.from_numpy(np.roll(inp["c"], rank, axis=1)).to(dev)
    z = torch.from_numpy(np.roll(inp["z"], roll, axis=1)).to(dev)
    model.forward(x, c)          # ActNorm data-dependent init on the first batch
    torch.cuda.synchronize()

    # The forward (NLL) and inverse (synthesis) passes of a step are independent, and so are
    # successive steps: each direction gets its own HIP stream and the K steps are enqueued back to
    # back (joined once, before the clock stops).  The small-M kernels of one pass (late blocks
    # leave most CUs idle) then overlap the MFMA-bound kernels of the other.  --serial puts
    # everything on one stream.  --lanes L gives each direction L streams, step k on lane k % L:
    # with 2L chains in flight the overlap no longer depends on the two passes drifting out of
    # phase (measured: 1 lane 7.3 ms/step, 2 lanes 7.4, 3 lanes 6.6, 8 lanes 6.6).
    lanes_f = [torch.cuda.Stream(dev) for _ in range(args.lanes)]
    lanes_i = [torch.cuda.Stream(dev) for _ in range(args.lanes)]
    step_no = [0]

    def enqueue_step():
        s_fwd, s_inv = lanes_f[step_no[0] % args.lanes], lanes_i[step_no[0] % args.lanes]
        step_no[0] += 1
        if args.serial:
            log_p, logdet = model.forward(x, c)
            nll = torch.stack([log_p, logdet])
            if world > 1:
                allreduce(nll)
                nll = nll / world
            return nll, model.reverse(z, c)
        with torch.cuda.stream(s_fwd):
            log_p, logdet = model.forward(x, c)
            nll = torch.stack([log_p, logdet])
            if world > 1:                    # global-batch NLL: the path's only exchange
                allreduce(nll)               # (every rank holds the same number of clips)
                nll = nll / world
        with torch.cuda.stream(s_inv):
            wav = model.reverse(z, c)
        return nll, wav

    def run_steps(n):
        cur = torch.cuda.current_stream(dev)
        for st in lanes_f + lanes_i:
            st.wait_stream(cur)
        for _ in range(n):
            nll, wav = enqueue_step()
        for st in lanes_f + lanes_i:
            cur.wait_stream(st)
        return nll, wav

    # reference for the result check below: one pass of each direction alone on the current stream
    ref_nll = torch.stack(model.forward(x, c)).clone()
    ref_wav = model.reverse(z, c).clone()
    run_steps(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nll, wav = run_steps(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        allreduce(tmax, dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert bool(torch.isfinite(nll).all()) and bool(torch.isfinite(wav).all())
    # the overlapped passes of the timed region must reproduce the single-stream result bit for bit
    same = bool(torch.equal(wav, ref_wav)) and (world > 1 or bool(torch.equal(nll, ref_nll)))
    if not same:
        raise SystemExit("bench.py: the overlapped passes differ from the single-stream result")

    # per-direction timings (HIP events on the launch stream), rank 0, for the breakdown fields
    def timed(fn, n=5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / n

    if rank == 0:
        fwd_s = timed(lambda: model.forward(x, c))
        inv_s = timed(lambda: model.reverse(z, c))
        fps = flop_per_sample(hp)
        ms = elapsed / args.steps * 1e3
        value = 2.0 * b * t * world / (elapsed / args.steps)
        out = {
            "metric": "audio samples/sec: forward NLL + inverse synth, n_block=8 bf16",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "configs[1]: full model n_block=8 n_flow=6 n_layer=2, forward NLL + inverse "
                                   "synthesis, 16128-sample (63-frame) clips @22.05 kHz",
                       "clips_per_gpu": b, "samples_per_clip": t, "samples_per_step_per_gpu": 2 * b * t,
                       "weights": "synthetic seed 1234, ActNorm DDI on first batch",
                       "parallelism": "batch shard x%d, no data-path collective (2-scalar NLL all-reduce)" % world,
                       "streams": "serial" if args.serial else "%d per direction" % args.lanes,
                       "results": "last step bit-identical to the single-stream pass"},
            "fwd_samples_per_s": b * t / fwd_s, "inv_samples_per_s": b * t / inv_s,
            "fwd_ms": fwd_s * 1e3, "inv_ms": inv_s * 1e3,
            "model_tflops": value / world * fps / 1e12,
            "model_mfma_frac": value / world * fps / 1e12 / MFMA_PEAK_TFLOPS,
            "realtime_factor_inverse": b * t / inv_s / hp.sample_rate,
        }
        out["roofline"] = gate_roofline(model, hp, b, t)
        if args.no_cpu_baseline or world > 1:
            out["cpu_baseline"] = None
        else:
            out["cpu_baseline"] = cpu_baseline(hp, params, t)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
