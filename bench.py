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

``--gpus N`` (N > 1) without a torchrun environment makes this process a pure launcher: it never touches the GPU,
starts ``python -m torch.distributed.run --nproc-per-node N bench.py ...`` as a CHILD process (one rank per GPU over
RCCL), relays rank 0's JSON line and exits non-zero if fewer than N GPUs exist or the line does not say ``n_gpus: N``.

One JSON line on stdout (rank 0) with ``roofline`` and ``cpu_baseline`` objects as described in DESIGN.md
"Measurement", plus ``rtf_10s`` (BASELINE configs[3]: 10 s clip inverse, one clip per GPU) and ``train``
(configs[2]: the data-parallel training step, B=8 x 6400 samples per GPU, RCCL gradient all-reduce).
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


def cpu_baseline(hp, params, t, budget_s=20.0, max_passes=16):
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
        for it in range(max_passes + 1):
            t0 = time.perf_counter()
            ot.forward(fp, x, c, hp)
            ot.reverse(fp, z, c, hp)
            dt = time.perf_counter() - t0
            if it > 0:
                times.append(dt)
            if time.perf_counter() - t_start > budget_s and len(times) >= 3:
                break
    med = float(np.median(times))
    # SURVEY section 8(d) asks for configs[0] ("the reference's own CPU-runnable case") always: forward log-p of the
    # n_block = 2, n_flow = 2 model on one clip
    import copy
    hp0 = copy.copy(hp)
    hp0.n_block, hp0.n_flow = 2, 2
    fp0 = ot.fold(W.synthetic_params(hp0, 1234), hp0, dtype=torch.float32)
    t0s = []
    with torch.no_grad():
        for it in range(6):
            t0 = time.perf_counter()
            ot.forward(fp0, x, c, hp0)
            if it > 0:
                t0s.append(time.perf_counter() - t0)
    med0 = float(np.median(t0s))
    return {"value": 2.0 * t / med, "unit": "samples/s", "cores": ncore, "kind": "port",
            "config0": {"workload": "configs[0]: forward log-p, n_block=2 n_flow=2, one %d-sample clip" % t,
                        "value": t / med0, "unit": "samples/s", "ms": med0 * 1e3, "passes": len(t0s)},
            "sample": "torch-CPU fp32 restatement (TF 1.12 unavailable), full n_block=%d model, B=1, T=%d, "
                      "forward+inverse, median of %d timed passes after 1 warm-up (spread %.0f%%)"
                      % (hp.n_block, t, len(times), 100.0 * (max(times) - min(times)) / med)}


def gate_roofline(model, hp, b, t, iters=200):
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

    for _ in range(10):
        launch()
    torch.cuda.synchronize()
    # `iters` launches as groups of 20 back to back, every group between its own pair of HIP events on the launch stream
    # (inside a pass the launches of a flow follow each other without event packets in between: this is the duration a
    # chain of launches sees, and what rocprofv3's per-dispatch duration agrees with); the MEDIAN of the group means is the
    # figure - a clock dip or a neighbour on the box moves a mean -, min and max beside it
    group = 20
    ngroups = max(1, iters // group)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(ngroups)]
    for e0, e1 in ev:
        e0.record()
        for _ in range(group):
            launch()
        e1.record()
    torch.cuda.synchronize()
    us = np.sort(np.array([e0.elapsed_time(e1) for e0, e1 in ev]) * 1e3 / group)
    # and every launch between its own event pair (adds the event packets' gap to every launch: an upper bound)
    ev1 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(group * 2)]
    for e0, e1 in ev1:
        e0.record()
        launch()
        e1.record()
    torch.cuda.synchronize()
    single_us = float(np.median([e0.elapsed_time(e1) for e0, e1 in ev1])) * 1e3
    sec = float(np.median(us)) * 1e-6
    flops = 2.0 * m * (768 + d.cin) * 512
    ach = flops / sec / 1e12
    # the clock the chip holds inside this kernel: a diagnostic instantiation of the same kernel (fwn_gate_clock) stamps
    # s_memtime / s_memrealtime at the start and end of every wave; launched right behind the timed launches (the chip is
    # warm), median over waves.  The product kernel executes no stamp.
    clock_ghz = None
    if bool(d.Wgs[0]) and m >= 24576:
        nwg = 2 * ((m + 255) // 256)
        stamps = torch.zeros(nwg * 8 * 4, dtype=torch.int64, device=dev)
        for _ in range(group):
            launch()
        n = lib.fwn_gate_clock(C.byref(d), 0, h.data_ptr(), ca.data_ptr(), o.data_ptr(), m, ti, stamps.data_ptr(), st)
        torch.cuda.synchronize()
        if n == nwg:
            sv = stamps.cpu().numpy().reshape(-1, 4).astype(np.float64)
            ok = (sv[:, 3] > sv[:, 2]) & (sv[:, 1] > sv[:, 0])
            if ok.any():
                clock_ghz = float(np.median((sv[ok, 1] - sv[ok, 0]) / (sv[ok, 3] - sv[ok, 2]))) * 0.1     # reference ticks at 100 MHz
    traffic, traffic_source = gate_traffic(m)
    streamed = bool(d.Wgs[0]) and m >= lib.fwn_gate_stream_rows()
    kernel = ("gate_rs_kernel<5> (register-streamed weights, csrc/gate_rs.h)" if streamed
              else "gate_halo_kernel<256,256,GateProb> (tap-sharing tile, csrc/gate_halo.h)")
    return {"bound": "mfma", "kernel": kernel + ": block 0 gated dilated layer, fwn_gate",
            "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS,
            "traffic": traffic, "traffic_source": traffic_source, "launch_us": sec * 1e6, "flop_per_launch": flops,
            "rows": m, "timing": "median over %d groups of %d back-to-back launches, one HIP event pair per group" % (ngroups, group),
            "launch_us_min": float(us[0]), "launch_us_max": float(us[-1]), "launch_us_single_event_pair": single_us,
            # in-kernel shader clock (fwn_gate_clock) and the fraction of the peak AT THAT CLOCK (datasheet peak x clock / 2.4 GHz)
            "clock_ghz": clock_ghz, "frac_at_clock": (ach / (MFMA_PEAK_TFLOPS * clock_ghz / 2.4)) if clock_ghz else None}


GATE_SOURCES = ("gate_rs.h", "gate_halo.h", "gemm_ring.h", "common.h", "flow_kernels.hip")
# per-block lower bound sum_i max(FLOP_i / 2.5 PF, bytes_i / 8 TB/s) of one pass at datasheet peaks
BOUND_US = {"B8_T16128": 853.0, "B1_T16128": 130.8, "B1_T220672": 1459.0}


def kernel_source_hash():
    """sha256 over every kernel source of the library (what a per-block / whole-pass profile must have been taken on)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "tf-flowavenet_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(base, "*.hip")) + glob.glob(os.path.join(base, "*.h"))):
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def committed_profile(pattern):
    """Newest profiles/<pattern> taken on the current kernel sources (its "source_sha" field), else (None, reason)."""
    import glob
    sha = kernel_source_hash()
    for pj in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        with open(pj) as f:
            rec = json.load(f)
        if rec.get("source_sha") == sha:
            return rec, os.path.relpath(pj, ROOT)
    return None, "no profile of the current kernel sources (hash %s) under profiles/%s" % (sha, pattern)


BLOCK_PASSES = 9


def block_table(model, hp, b, t, x, c, z):
    """us per block of the one-stream forward and inverse pass, measured live: fwn_model_desc.block_events makes the
    whole-model calls record a HIP event in front of every block (and behind the last); median over BLOCK_PASSES passes."""
    import ctypes as C
    import torch
    md = model._packed.model_desc
    nb = hp.n_block
    half = hp.num_mels // 2
    out = {}
    for dname, fn in (("fwd", lambda: model.forward(x, c)), ("inv", lambda: model.reverse(z, c))):
        samples = []
        for _ in range(BLOCK_PASSES + 1):
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(nb + 1)]
            for e in evs:
                e.record()                      # creates the handle
            handles = (C.c_void_p * (nb + 1))(*[e.cuda_event for e in evs])
            md.block_events = C.cast(handles, C.POINTER(C.c_void_p))
            try:
                fn()
            finally:
                md.block_events = None
            torch.cuda.synchronize()
            samples.append([evs[k].elapsed_time(evs[k + 1]) * 1e3 for k in range(nb)])
        med = np.median(np.array(samples[1:]), axis=0)
        rows = []
        for k in range(nb):
            i = k if dname == "fwd" else nb - 1 - k
            cc = 2 << i
            cin = half * cc
            mac = 3 * (cc // 2) * 256 + hp.n_layer * (2 * 3 * 256 * 256 + 2 * cin * 256) + (2 * hp.n_layer - 1) * 256 * 256 + 256 * 256 + 256 * cc
            gflop = 2.0 * mac * hp.n_flow / cc * b * t / 1e9
            rows.append({"block": i, "rows": b * t // cc, "us": round(float(med[k]), 1), "gflop": round(gflop, 1),
                         "mfma_frac": round(gflop / float(med[k]) * 1e3 / MFMA_PEAK_TFLOPS, 4)})
        out[dname] = sorted(rows, key=lambda r: r["block"])
    return out


def path_roofline(hp, b, t, fwd_s, inv_s, live_blocks=None):
    """The PATH against its rooflines (SURVEY section 8d asks for both): whole-pass MFMA fraction from the live timings,
    the per-block table and the whole-pass HBM bytes from the committed rocprofv3 profiles of these kernel sources."""
    flop = flop_per_sample(hp) * b * t
    bound = BOUND_US.get("B%d_T%d" % (b, t))
    out = {"flop_per_pass": flop, "mfma_peak_tflops": MFMA_PEAK_TFLOPS, "hbm_peak_gbs": HBM_PEAK_GBS,
           "survey_bound_us": bound, "serial_pair_ms": (fwd_s + inv_s) * 1e3}
    for name, sec in (("fwd", fwd_s), ("inv", inv_s)):
        out[name] = {"ms": sec * 1e3, "mfma_frac": flop / sec / 1e12 / MFMA_PEAK_TFLOPS,
                     "frac_of_survey_bound": (bound * 1e-6 / sec) if bound else None}
    # per block: measured in THIS run (HIP events the whole-model calls record at the block boundaries,
    # fwn_model_desc.block_events); the launch counts are those of the committed rocprofv3 pass table when it was taken on
    # these kernel sources
    tab, src = committed_profile("r*_pass_table.json")
    out["blocks_source"] = "live: HIP events at the block boundaries of one-stream passes (median of %d)" % BLOCK_PASSES
    out["blocks_launch_counts_source"] = src
    out["blocks"] = live_blocks
    if tab and live_blocks:
        for dname in ("fwd", "inv"):
            if dname in tab and dname in live_blocks:
                by_block = {r["block"]: r["launches"] for r in tab[dname]["blocks"]}
                for r in live_blocks[dname]:
                    r["launches"] = by_block.get(r["block"])
    tr, src = committed_profile("r*_pass_traffic.json")
    out["hbm_source"] = src
    out["hbm"] = None
    if tr:
        out["hbm"] = {d: {"traffic_bytes": tr[d]["traffic_bytes"], "ratio_to_algorithmic": tr[d]["ratio_to_algorithmic"],
                          "gbs_at_this_pass": tr[d]["traffic_bytes"] / (fwd_s if d == "fwd" else inv_s) / 1e9,
                          "hbm_frac": tr[d]["traffic_bytes"] / (fwd_s if d == "fwd" else inv_s) / 1e9 / HBM_PEAK_GBS}
                      for d in ("fwd", "inv") if d in tr}
        out["hbm"]["algorithmic_bytes"] = tr["algorithmic_bytes"]
    return out


def _median_pass_s(fn, iters):
    """Median duration (s) of `iters` calls of fn, each between its own pair of HIP events on the launch stream (a secondary
    leg of a few passes: one slow pass - a clock dip - must not move the figure the way it moves a mean)."""
    import torch
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for e0, e1 in ev:
        e0.record()
        fn()
        e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    return ts[len(ts) // 2] * 1e-3


def one_launch_twin(model, hp, params, dev, tail_stream=True):
    """The same parameters (the data-dependent init's tables exported from `model`) packed twice more: with
    fwn_model_desc.persist_mode = 2 (csrc/flow_persist.h: ONE launch per flow wherever the form exists) and 1 (a launch per
    stage everywhere; the default, 0, takes the one-launch form up to 512 rows) - the pair whose results must be equal bit for bit (the device init and the host packing of `model` itself differ
    in last bits of exp(3 logs))."""
    import numpy as np
    from tf_flowavenet_amd.model import FloWaveNet
    p2 = dict(params)
    for k, v in model.export_actnorm().items():
        p2[k] = np.asarray(v, dtype=np.float32).reshape(np.asarray(p2[k]).shape)
    plain = FloWaveNet(hp, device=dev, persist_mode=1, tail_stream=tail_stream).load_params(p2)
    one = FloWaveNet(hp, device=dev, persist_mode=2, tail_stream=tail_stream).load_params(p2)
    return one, plain


def latency_b1(model, hp, t, dev, iters=10, params=None):
    """configs[1] at the latency shape: ONE 16128-sample clip, forward and inverse, HIP events on the launch stream."""
    import torch
    from tf_flowavenet_amd import weights as W
    inp = W.synthetic_inputs(hp, 1, t)
    x, c, z = (torch.from_numpy(inp[k]).to(dev) for k in ("x", "c", "z"))

    def timed(fn):
        for _ in range(3):
            fn()
        return _median_pass_s(fn, iters)

    fwd, inv = timed(lambda: model.forward(x, c)), timed(lambda: model.reverse(z, c))
    flop = flop_per_sample(hp) * t
    bound = BOUND_US.get("B1_T%d" % t)
    one_launch = None
    if params is not None:       # round 5: the same clip with one launch per flow wherever the form exists, and with none
        try:
            one, plain = one_launch_twin(model, hp, params, dev)
            f1, i1 = timed(lambda: one.forward(x, c)), timed(lambda: one.reverse(z, c))
            f0, i0 = timed(lambda: plain.forward(x, c)), timed(lambda: plain.reverse(z, c))
            del one, plain
            # the identity check on a pair WITHOUT the tail's fragment stream: the one-launch flow reproduces the N-split tail's
            # arithmetic, which the launch-per-stage path runs below 4 097 rows only when csrc/tail_rs.h does not
            one, plain = one_launch_twin(model, hp, params, dev, tail_stream=False)
            a, b_ = one.forward(x, c, return_z=True), plain.forward(x, c, return_z=True)
            same = bool(torch.equal(a[2], b_[2])) and float(a[0]) == float(b_[0]) and float(a[1]) == float(b_[1]) and \
                bool(torch.equal(one.reverse(z, c), plain.reverse(z, c)))
            one_launch = {"fwd_ms": f1 * 1e3, "inv_ms": i1 * 1e3, "launch_per_stage_fwd_ms": f0 * 1e3, "launch_per_stage_inv_ms": i0 * 1e3,
                          "bit_identical_to_launch_per_stage": same,
                          "bit_identity_pair": "both twins packed with tail_stream=False (the N-split tail below 4 097 rows, whose arithmetic "
                                               "the one-launch flow reproduces); the timed twins carry the product's operands",
                          "what": "fwn_model_desc.persist_mode = 2: blocks 2 - 7 of this clip as one launch per flow "
                                  "(csrc/flow_persist.h; DESIGN.md section 3.7), and = 1: a launch per stage everywhere; the "
                                  "line's fwd_ms / inv_ms are the default (0): one launch per flow up to 512 rows (blocks 4 - 7)"}
            del one, plain
        except Exception as e:   # a diagnostic: never instead of the line
            one_launch = {"error": "%s: %s" % (type(e).__name__, e)}
    return {"workload": "configs[1] latency shape: B=1, T=%d" % t, "fwd_ms": fwd * 1e3, "inv_ms": inv * 1e3,
            "one_launch_flows": one_launch,
            "fwd_mfma_frac": flop / fwd / 1e12 / MFMA_PEAK_TFLOPS, "inv_mfma_frac": flop / inv / 1e12 / MFMA_PEAK_TFLOPS,
            "survey_bound_us": bound, "fwd_frac_of_survey_bound": bound * 1e-6 / fwd if bound else None,
            "inv_frac_of_survey_bound": bound * 1e-6 / inv if bound else None,
            "realtime_factor_inverse": t / inv / hp.sample_rate, "timing": "median of %d passes, one HIP event pair each" % iters}


def gate_source_hash():
    """sha256 over the sources the dominant kernel is compiled from (what a PMC profile must have been taken on)."""
    import hashlib
    h = hashlib.sha256()
    for name in GATE_SOURCES:
        with open(os.path.join(ROOT, "tf-flowavenet_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def gate_traffic(rows):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC profile
    (``profiles/r*_gate_traffic.json``, written by tools/gate_pmc.py from separate rocprofv3 --pmc passes).  The
    counters cannot be read inside this process, so the number is only reported when the profile was taken at this
    launch shape AND on the current kernel sources (hash recorded in the profile); otherwise null."""
    import glob
    for tj in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gate_traffic.json")), reverse=True):
        with open(tj) as f:
            rec = json.load(f)
        if rec.get("rows") == rows and rec.get("source_sha") == gate_source_hash():
            return rec["traffic_bytes"], "%s (source_sha %s)" % (os.path.relpath(tj, ROOT), rec["source_sha"])
    return None, "no PMC profile of the current kernel sources (hash %s) under profiles/" % gate_source_hash()


def rtf_10s(model, hp, dev, world, iters=7):
    """BASELINE configs[3]: inverse synthesis of a 10 s clip @ 22.05 kHz (T = 220672 = 862 frames), one clip per GPU
    (the batch shard of the 8-clip job), HIP events on the launch stream."""
    import torch
    from tf_flowavenet_amd import weights as W
    t = 220672 if hp.hop_size == 256 else (10 * hp.sample_rate // np.lcm(hp.hop_size, 1 << hp.n_block)) * np.lcm(hp.hop_size, 1 << hp.n_block)
    inp = W.synthetic_inputs(hp, 1, int(t), want=("c", "z"))
    z, c = torch.from_numpy(inp["z"]).to(dev), torch.from_numpy(inp["c"]).to(dev)
    for _ in range(2):
        wav = model.reverse(z, c)
    torch.cuda.synchronize()
    out = []
    sec = _median_pass_s(lambda: out.__setitem__(slice(None), [model.reverse(z, c)]), iters)
    wav = out[0]
    assert bool(torch.isfinite(wav).all())
    audio_s = t / hp.sample_rate
    bound = BOUND_US.get("B1_T%d" % t)
    return {"workload": "configs[3]: inverse synthesis, one %.3f s clip (T=%d) per GPU, B=1" % (audio_s, t),
            "inverse_ms": sec * 1e3, "timing": "median of %d passes, one HIP event pair each" % iters, "rtf_per_gpu": audio_s / sec, "rtf_whole_job": world * audio_s / sec,
            "samples_per_s_whole_job": world * t / sec, "n_gpus": world,
            "mfma_frac": flop_per_sample(hp) * t / sec / 1e12 / MFMA_PEAK_TFLOPS,
            "survey_bound_us": bound, "frac_of_survey_bound": bound * 1e-6 / sec if bound else None}


def fp8_leg(hp, params, model_bf16, x, c, z, b, t):
    """BASELINE configs[4]: the same workload with the dilated taps of the MFMA-bound gates (blocks 0-2 here) on the fp8
    (e4m3, v_mfma_scale_f32_32x32x64_f8f6f4) path.  Secondary numbers: the headline stays bf16."""
    import ctypes as C
    import torch
    from tf_flowavenet_amd import _lib
    from tf_flowavenet_amd.model import FloWaveNet
    lib = _lib.load()
    m8 = FloWaveNet(hp, init=True, device=x.device, gate_fp8=True, group=False).load_params(params)     # rank 0 only: local init
    lp8, ld8 = m8.forward(x, c)
    lpb, ldb = model_bf16.forward(x, c)

    def timed(fn, n=8):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / n

    fwd, inv = timed(lambda: m8.forward(x, c)), timed(lambda: m8.reverse(z, c))
    d = m8._packed.flow_descs[0]
    ti = t // 2
    m = b * ti
    h8 = torch.randint(0, 120, (m, 256), dtype=torch.uint8, device=x.device)       # e4m3 bytes of values in [0, 448)
    ca = torch.rand(m, d.cin, device=x.device).to(torch.bfloat16)
    o = torch.empty(m, 256, device=x.device, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    gate_s = timed(lambda: _lib.check(lib.fwn_gate_fp8(C.byref(d), 0, h8.data_ptr(), ca.data_ptr(), o.data_ptr(), m, ti, st)), n=20)
    flops = 2.0 * m * (768 + d.cin) * 512
    return {"workload": "configs[4]: fp8 (e4m3) dilated taps in the gates of blocks 0-2, otherwise the configs[1] workload",
            "fwd_ms": fwd * 1e3, "inv_ms": inv * 1e3, "fwd_samples_per_s": b * t / fwd, "inv_samples_per_s": b * t / inv,
            "log_p": float(lp8), "log_p_bf16": float(lpb), "log_p_rel_diff_vs_bf16": abs(float(lp8) - float(lpb)) / abs(float(lpb)),
            "logdet": float(ld8), "logdet_bf16": float(ldb),
            "gate_launch_us": gate_s * 1e6, "gate_tflops": flops / gate_s / 1e12,
            "gate_kernel": "gate_halo_kernel<256,256,GateProb,FP8> (block 0)"}


def train_leg(hp, params, rank, world, dev, steps=10, batch=8, samples=6400, force_collectives=False):
    """BASELINE configs[2]: the data-parallel training step, `batch` crops of `samples` samples per GPU (global batch
    64 on 8 GPUs): gradients of -(log_p + logdet), RCCL all-reduce of the flat fp32 gradient started block by block
    under the backward pass, global-norm clip, Adam.  Also times the step without the exchange and the exchange
    alone, so the overlap can be read off."""
    import torch
    import torch.distributed as dist
    from tf_flowavenet_amd import weights as W
    from tf_flowavenet_amd.training import Trainer
    inp = W.synthetic_inputs(hp, batch, samples, want=("x", "c"))
    x = torch.from_numpy(np.roll(inp["x"], 997 * rank, axis=1)).reshape(batch, samples).to(dev)
    c = torch.from_numpy(np.roll(inp["c"], rank, axis=1)).to(dev)
    tr = Trainer(hp, params, device=dev)
    tr.opt.force_collectives = bool(force_collectives)
    tr.ddi(x, c)
    exchanging = world > 1 or force_collectives

    def timed(n):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            out = tr.step(x, c)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            tm = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            el = float(tm.item())
        return el / n, out

    for _ in range(3):                   # eager step, recording, first replay
        tr.step(x, c)
    step_s, (loss, _, _, gnorm) = timed(steps)
    rec = {"workload": "configs[2]: data-parallel training step, %d x %d samples per GPU, full n_block=%d model"
                       % (batch, samples, hp.n_block),
           "ms_per_step": step_s * 1e3, "samples_per_s": batch * samples * world / step_s, "n_gpus": world,
           "global_batch": batch * world, "loss": float(loss), "grad_norm": float(gnorm),
           "recorded_step": bool(tr.graph), "allreduce_ms": None, "compute_ms": None, "overlap": None,
           "gradient_bytes": int(tr.opt.g.numel()) * 4,
           # forward + backward = 3 x the forward FLOP of the samples (the recompute by inversion is not counted)
           "mfma_frac": 3.0 * flop_per_sample(hp) * batch * samples / step_s / 1e12 / MFMA_PEAK_TFLOPS,
           # lock-step: every rank holds bit-identical master weights after the timed steps (all-reduce of a 64-bit hash)
           "weights_identical": bool(tr.opt.weights_identical()),
           # what the process group itself reports (N > 1: RCCL over xGMI; one rank with --force-collectives: a one-rank RCCL group)
           "collective": ({"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rank": dist.get_rank()}
                          if dist.is_available() and dist.is_initialized() else None)}
    if exchanging:
        tr.exchange = False              # same step without the gradient exchange (weights drift apart: timing only)
        compute_s, _ = timed(max(3, steps // 2))
        tr.exchange = True
        ranges = [(lo, hi) for _, lo, hi in tr.opt.block_ranges()]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            works = [tr.opt.allreduce_range(lo, hi) for lo, hi in reversed(ranges)]
            for w in works:
                if w is not None:
                    w.wait()
            torch.cuda.synchronize()
        ar_s = (time.perf_counter() - t0) / 3
        rec.update(allreduce_ms=ar_s * 1e3, compute_ms=compute_s * 1e3,
                   overlap=max(0.0, min(1.0, (compute_s + ar_s - step_s) / ar_s)) if ar_s > 0 else None)
    return rec


def launch_ranks(n):
    """``--gpus N`` outside torchrun: this process stays off the GPU and runs the N ranks as a child job."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        print("bench.py: the %d-rank job failed (exit code %d)" % (n, proc.returncode), file=sys.stderr)
        raise SystemExit(proc.returncode or 1)
    if json.loads(line).get("n_gpus") != n:
        print("bench.py: the job reported n_gpus=%r, expected %d" % (json.loads(line).get("n_gpus"), n), file=sys.stderr)
        raise SystemExit(1)
    print(line, flush=True)
    raise SystemExit(0)


class Deadline:
    """The optional legs (10 s clip, training step) must never cost the headline line: if one of them is still
    running `seconds` after it started (a stuck collective on an untested topology), rank 0 prints the line with what
    it has and every rank leaves with exit code 3 (the line carries a "note"; a caller must not mistake the run for a complete one)."""

    def __init__(self, emit):
        import threading
        self._emit, self._timer, self._threading = emit, None, threading

    def arm(self, seconds, what):
        def fire():
            self._emit("%s did not finish within %d s" % (what, seconds))
            sys.stdout.flush()
            os._exit(3)                  # the partial line is out; the run did not complete
        self._timer = self._threading.Timer(seconds, fire)
        self._timer.daemon = True
        self._timer.start()

    def disarm(self):
        if self._timer is not None:
            self._timer.cancel()
            self._timer = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU (hparams.batch_size)")
    ap.add_argument("--samples", type=int, default=16128, help="samples per clip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cond-mode", type=int, default=0)
    ap.add_argument("--chain-mode", type=int, default=0, help="fwn_model_desc.chain_mode (developer A/B: 1 = every flow on its own)")
    ap.add_argument("--persist-mode", type=int, default=0, help="fwn_model_desc.persist_mode (0 default, 1 no one-launch flows, 2 wherever they exist)")
    ap.add_argument("--serial", action="store_true", help="forward and inverse on one stream (no overlap)")
    ap.add_argument("--repeats", type=int, default=3, help="how many times the K-step timed region is run (the line reports the median)")
    ap.add_argument("--lanes", type=int, default=7,
                    help="HIP streams per direction; successive (independent) steps rotate over them")
    ap.add_argument("--no-train", action="store_true", help="skip the configs[2] training-step leg")
    ap.add_argument("--no-rtf", action="store_true", help="skip the configs[3] 10 s clip leg")
    ap.add_argument("--no-fp8", action="store_true", help="skip the configs[4] fp8 gate leg")
    ap.add_argument("--no-latency", action="store_true", help="skip the B=1 latency leg (profile runs: only passes of the bench workload)")
    ap.add_argument("--train-steps", type=int, default=10)
    ap.add_argument("--leg-timeout", type=int, default=240, help="seconds an optional leg may take")
    ap.add_argument("--force-collectives", action="store_true",
                    help="issue the gradient all-reduces even with one rank (exercises the RCCL path on one GPU)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args.gpus)          # never returns; this process has not touched the GPU

    import torch
    import torch.distributed as dist
    from tf_flowavenet_amd.hparams import default_hparams
    from tf_flowavenet_amd import weights as W
    from tf_flowavenet_amd.model import FloWaveNet

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (args.gpus, world))
    # FWN_BENCH_SHARE_GPU=1 is a plumbing test for boxes with one GPU: every rank uses cuda:0 and the
    # scalar exchanges go over gloo.  The measured configuration is always one GPU per rank + RCCL.
    share_gpu = os.environ.get("FWN_BENCH_SHARE_GPU") == "1"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if share_gpu:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit("bench.py: --gpus %d needs %d GPUs on this node, found %d" % (world, world, torch.cuda.device_count()))
    if world > 1 or args.force_collectives:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
        if world == 1:                   # --force-collectives: a one-rank RCCL group
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
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
    model = FloWaveNet(hp, init=True, device=dev, cond_mode=args.cond_mode, chain_mode=args.chain_mode,
                       persist_mode=args.persist_mode).load_params(params)
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
    # phase (round 1: 1 lane 7.3 ms/step, 2 lanes 7.4, 3 lanes 6.6, 8 lanes 6.6; round 5, three runs each on one box:
    # 3 lanes 5.68 ms, 5 lanes 5.63, 7 lanes 5.59 at 30 steps - and 4 / 6 / 8 lanes 5.92 / 5.76 / 5.69: odd counts win whatever
    # the order the streams are created in; the margin depends on how the timed steps fall onto the lanes: +0.5 % at 20 steps,
    # +-0 at 60, -3 % at 10 against 3 lanes; profiles/r05_lanes.txt).
    lanes_f = [torch.cuda.Stream(dev) for _ in range(args.lanes)]
    lanes_i = [torch.cuda.Stream(dev) for _ in range(args.lanes)]
    step_no = [0]
    if not args.serial:
        # resources, not steps: every lane's scratch (one workspace per stream, model._workspace) exists before the warm-up, so
        # that with fewer warm-up steps than lanes no allocation falls into the timed region
        for st in lanes_f + lanes_i:
            with torch.cuda.stream(st):
                model._workspace(b, t)

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
    # The timed region - EXACTLY args.steps steps between barrier + synchronize brackets, MAX over the ranks - is run
    # args.repeats times (default 3); the line reports the MEDIAN region and lists all of them (at 0.1 - 0.2 s a single
    # region moves by about as much from run to run as a round's gain; the dominant kernel's own launch groups spread 52 - 63 us).
    regions = []
    for _ in range(max(1, args.repeats)):
        t0 = time.perf_counter()
        nll, wav = run_steps(args.steps)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([el], device=dev, dtype=torch.float64)
            allreduce(tmax, dist.ReduceOp.MAX)
            el = float(tmax.item())
        regions.append(el)
    elapsed = sorted(regions)[len(regions) // 2]
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

    out = {}
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
                       "weights": "synthetic seed 1234, ActNorm DDI on the first (global) batch",
                       "parallelism": "batch shard x%d, no data-path collective (2-scalar NLL all-reduce)" % world,
                       "streams": "serial" if args.serial else "%d per direction" % args.lanes,
                       "results": "last step bit-identical to the single-stream pass",
                       "timed_regions": {"repeats": len(regions), "steps_each": args.steps, "reported": "median",
                                         "ms_per_step_each": [r / args.steps * 1e3 for r in regions],
                                         "ms_per_step_min": min(regions) / args.steps * 1e3,
                                         "ms_per_step_max": max(regions) / args.steps * 1e3}},
            "fwd_samples_per_s": b * t / fwd_s, "inv_samples_per_s": b * t / inv_s,
            "fwd_ms": fwd_s * 1e3, "inv_ms": inv_s * 1e3,
            "model_tflops": value / world * fps / 1e12,
            "model_mfma_frac": value / world * fps / 1e12 / MFMA_PEAK_TFLOPS,
            "realtime_factor_inverse": b * t / inv_s / hp.sample_rate,
        }
        out["roofline"] = gate_roofline(model, hp, b, t)
        out["path"] = path_roofline(hp, b, t, fwd_s, inv_s, block_table(model, hp, b, t, x, c, z))
        out["latency_b1"] = None if args.no_latency else latency_b1(model, hp, t, dev, params=params)
        if args.no_cpu_baseline or world > 1:
            out["cpu_baseline"] = None
        else:
            out["cpu_baseline"] = cpu_baseline(hp, params, t)
    out["rtf_10s"] = out["train"] = out["fp8"] = None

    def emit(note=None):
        if rank == 0:
            if note:
                out["note"] = note
            print(json.dumps(out), flush=True)

    def leg_failed(name, e):
        import traceback
        out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        print("bench.py rank %d: the %s leg failed:\n%s" % (rank, name, traceback.format_exc()), file=sys.stderr, flush=True)

    # optional legs (all ranks take part; a failure or a stall is reported inside the line, never instead of it)
    deadline = Deadline(emit)
    if not args.no_rtf:
        deadline.arm(args.leg_timeout, "the configs[3] 10 s clip leg")
        try:
            out["rtf_10s"] = rtf_10s(model, hp, dev, world)
        except Exception as e:
            leg_failed("rtf_10s", e)
        deadline.disarm()
    if not args.no_fp8 and rank == 0:
        deadline.arm(args.leg_timeout, "the configs[4] fp8 leg")
        try:
            out["fp8"] = fp8_leg(hp, params, model, x, c, z, b, t)
        except Exception as e:
            leg_failed("fp8", e)
        deadline.disarm()
    if not args.no_train:
        del model
        torch.cuda.empty_cache()
        deadline.arm(args.leg_timeout, "the configs[2] training-step leg")
        try:
            out["train"] = train_leg(hp, params, rank, world, dev, steps=args.train_steps,
                                     force_collectives=args.force_collectives)
        except Exception as e:
            leg_failed("train", e)
        deadline.disarm()
    emit()
    if world > 1 or args.force_collectives:
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
