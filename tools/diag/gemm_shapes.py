"""Developer probe (GPU box): time of fwn_gemm at the shapes the training step uses for its small-M blocks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_flowavenet_amd import training as TR

def timeit(fn, n=100):
    """n launches recorded into a hipGraph and replayed: kernel time without the Python launch cost."""
    for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        TR._STREAMS.clear()
        g.capture_begin()
        for _ in range(n): fn()
        g.capture_end()
    torch.cuda.current_stream().wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * n) * 1e3

bf = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
for m in (6400, 400):
    ti = m // 8
    for name, k, n, kw in [("1x1 K=256", 256, 256, {}), ("1x1 K=256 +res", 256, 256, dict(res=True)), ("1x1 K=64", 64, 256, {}), ("1x1 K=1024", 1024, 256, {}), ("1x1 K=256 +mask", 256, 256, dict(mask=True)),
                           ("skip K=512 +bias+relu", 512, 256, dict(bias=True, relu=True)),
                           ("dil^T K=1536 3 taps +res", 1536, 256, dict(taps=True, res=True)),
                           ("cond^T K=512 N=640 acc", 512, 640, dict(acc=True)), ("zero K=256 N=16 f32", 256, 16, dict(f32=True))]:
        w = bf(n, k)
        if kw.get("taps"):
            x = bf(m, 512)
            segs = [(x, 512, -(tap - 1), tap * 512) for tap in range(3)]
        else:
            x = bf(m, k)
            segs = [(x, k, 0, 0)]
        args = {}
        if kw.get("mask"): args["mask"] = bf(m, n)
        if kw.get("res"): args.update(res=bf(m, n), rscale=0.7)
        if kw.get("bias"): args["bias"] = torch.randn(n, device="cuda")
        if kw.get("relu"): args["relu"] = True
        if kw.get("acc"): args.update(out=torch.zeros(m, n, device="cuda"), accumulate=True)
        if kw.get("f32"): args["out_f32"] = True
        us = timeit(lambda: TR.gemm(segs, w, n, m, ti=ti if kw.get("taps") else 0, **args))
        print("M %5d  %-26s %7.1f us   (%.1f TFLOP/s)" % (m, name, us, 2.0 * m * k * n / us * 1e-6))
