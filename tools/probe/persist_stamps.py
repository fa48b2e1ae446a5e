"""GPU box: where the time of a one-launch flow goes (csrc/flow_persist.h built with -DFWN_PS_STAMP: make -C tf-flowavenet_amd/csrc stamp).

    FWN_LIB=tf-flowavenet_amd/csrc/libfwn_ps.so python tools/probe/persist_stamps.py [block] [clips] [inverse]

Per stage: when its first / last ticket started, got its producers, had its rows in LDS, finished the K loop, issued and drained
its stores (microseconds from the first stamp of the launch; 100 MHz reference clock = 10 ns resolution), and the median
duration of each segment over the stage's tickets."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from tf_flowavenet_amd import _lib, weights as W
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd.model import FloWaveNet

blk = int(sys.argv[1]) if len(sys.argv) > 1 else 7
b = int(sys.argv[2]) if len(sys.argv) > 2 else 1
inverse = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hp = default_hparams()
model = FloWaveNet(hp).load_params(W.synthetic_params(hp, 1234, actnorm="random"))
lib = _lib.load()
d = model._packed.flow_descs[blk * hp.n_flow + 2]
T, ch = 16128, 1 << blk
ti = T // (2 * ch)
m = b * ti
rt = (m + 63) // 64
rng = np.random.default_rng(0)
st = torch.cuda.current_stream().cuda_stream
ca = torch.from_numpy(rng.random((m, d.cin)).astype(np.float32)).cuda().to(torch.bfloat16)
P = torch.empty(hp.n_layer, m, 512, device="cuda", dtype=torch.float32)
_lib.check(lib.fwn_cond(ca.data_ptr(), d.Wc[0], P.data_ptr(), 512 * d.kcpad, m * 512, 0, 1, 1, hp.n_layer, m, d.cin, d.kcpad, st), "fwn_cond")
xa0 = torch.from_numpy(rng.standard_normal((m, ch)).astype(np.float32) * 0.3).cuda()
xb0 = torch.from_numpy(rng.standard_normal((m, ch)).astype(np.float32) * 0.3).cuda()
h0 = torch.empty(m, 256, device="cuda", dtype=torch.bfloat16)
h1 = torch.empty_like(h0)
o = torch.empty(hp.n_layer, m, 256, device="cuda", dtype=torch.bfloat16)
part = torch.zeros(lib.fwn_tail_partials(m), device="cuda", dtype=torch.float32)
words = lib.fwn_flow_persist_sync_bytes(m, hp.n_layer) // 4
base = (8 + (2 * hp.n_layer + 3) * rt + 3) & ~3
assert words > base, "not the stamp build: FWN_LIB=.../libfwn_ps.so"
big = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
flush = os.environ.get("PS_FLUSH", "1") == "1"
for rep in range(4):
    if flush: big.fill_(rep)                                   # the flow's weights leave the caches (512 MiB > Infinity Cache)
    xa, xb = xa0.clone(), xb0.clone()
    sync = torch.zeros(words, device="cuda", dtype=torch.int32)
    torch.cuda.synchronize()
    e0.record()
    _lib.check(lib.fwn_flow_run_persist(C.byref(d), b, T, xa.data_ptr(), xb.data_ptr(), h0.data_ptr(), h1.data_ptr(), o.data_ptr(),
                                        P.data_ptr(), part.data_ptr(), inverse, sync.data_ptr(), st), "fwn_flow_run_persist")
    e1.record()
    torch.cuda.synchronize()
assert lib.fwn_flow_persist_status(sync.data_ptr(), st) == 0
raw = sync[base:].cpu().numpy().view(np.uint64).reshape(-1, 8)
raw = raw[raw[:, 0] != 0]
t0 = raw[:, 0].min()
us = (raw[:, :7].astype(np.int64) - np.int64(t0)) / 100.0
info = raw[:, 7]
stage, wg, xcc = (info & 0xff).astype(int), ((info >> 32) & 0xffff).astype(int), ((info >> 48) & 15).astype(int)
print("block %d, %d clip(s), M = %d rows (%d row tiles), %s: %d tickets on %d workgroups, events %.1f us, stamps span %.1f us"
      % (blk, b, m, rt, "inverse" if inverse else "forward", len(raw), len(set(wg)), e0.elapsed_time(e1) * 1e3, us[:, 6].max()))
names = ["start", "w-issued", "deps", "rows", "kloop", "st-issued", "drained"]
print("stage tickets | first..last of: " + " | ".join("%9s" % n for n in names) + " || median segment us: wait  rows  kloop  epi  drain")
for s_ in sorted(set(stage)):
    sel = stage == s_
    u = us[sel]
    u4 = np.where(u[:, 4] > 0, u[:, 4], u[:, 3])      # stages without a reduction stamp
    seg = np.stack([u[:, 2] - u[:, 1], u[:, 3] - u[:, 2], u4 - u[:, 3], u[:, 5] - u4, u[:, 6] - u[:, 5]], 1)
    cols = " | ".join("%4.1f-%4.1f" % (u[:, k].min(), u[:, k].max()) if k != 4 else "%4.1f-%4.1f" % (u4.min(), u4.max()) for k in range(7))
    print("%5d %7d | %s || %s   xcc %s" % (s_, sel.sum(), cols, "  ".join("%5.2f" % v for v in np.median(seg, 0)), sorted(set(xcc[sel]))))
