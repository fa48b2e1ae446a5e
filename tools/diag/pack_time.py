"""Diagnostic: the step's weight re-packing alone (PackPlan.run_kernels of the full training model): time and bytes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tf_flowavenet_amd import weights as W
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd.training import Trainer

hp = default_hparams()
tr = Trainer(hp, W.synthetic_params(hp, 5), device="cuda", graph=False)
inp = W.synthetic_inputs(hp, 2, 1024)
x, c = torch.from_numpy(inp["x"]).reshape(2, 1024).cuda(), torch.from_numpy(inp["c"]).cuda()
tr.ddi(x, c)
tr.step(x, c)
plan = tr.engine._tp.plan
rd = wr = 0
for (v, sk, sn, out, ld, n_src, kd, nd, slot, trn, mul) in plan.jobs:
    rd += kd * nd * 4
    wr += kd * nd * 2
print("jobs", len(plan.jobs), "scale jobs", len(plan.sjobs), "read MB %.0f write MB %.0f" % (rd / 1e6, wr / 1e6))
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t = timed(plan.run_kernels)
print("run_kernels %.3f ms  -> %.2f TB/s" % (t, (rd + wr) / t / 1e9))
# which half is slow: the transposing jobs (inference packing, LDS tiles + 2-byte stores) or the straight ones (training copies)
import ctypes as C
from tf_flowavenet_amd import _lib
lib = _lib.load()
if not plan._built:
    plan._build()
order = sorted(range(len(plan.jobs)), key=lambda i: plan.jobs[i][0])
for name, sel in (("transposing (inference layouts)", [i for i in order if not plan.jobs[i][9]]), ("straight (training copies)", [i for i in order if plan.jobs[i][9]])):
    pj = (_lib.PackJob * len(sel))()
    rd = wr = 0
    for n, i in enumerate(sel):
        (v, sk, sn, out, ld, n_src, kd, nd, slot, trn, mul) = plan.jobs[i]
        j = pj[n]
        j.v, j.src_k, j.src_n, j.out, j.ld_dst = v, sk, sn, out, ld
        j.n_src, j.k_dst, j.n_dst, j.scale_slot, j.transposed, j.mul = n_src, kd, nd, slot, trn, mul
        rd += kd * nd * 4; wr += kd * nd * 2
    tab = torch.frombuffer(bytearray(bytes(pj)), dtype=torch.uint8).cuda()
    st = torch.cuda.current_stream().cuda_stream
    t = timed(lambda: lib.fwn_pack_jobs(None, 0, tab.data_ptr(), len(sel), plan._scales.data_ptr(), 512, st))
    print("%-34s %4d jobs  %.3f ms  (nominal read %.0f MB, write %.0f MB)" % (name, len(sel), t, rd / 1e6, wr / 1e6))
t = timed(lambda: lib.fwn_pack_jobs(plan._sj.data_ptr(), len(plan.sjobs), None, 0, plan._scales.data_ptr(), 512, torch.cuda.current_stream().cuda_stream))
print("scale jobs alone %.3f ms" % t)
