"""Developer probe: host enqueue cost per pass and hipGraph replay of overlapped passes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd import weights as W
from tf_flowavenet_amd.model import FloWaveNet

hp = default_hparams()
B, T = 8, 16128
m = FloWaveNet(hp, init=True).load_params(W.synthetic_params(hp, 1234))
inp = W.synthetic_inputs(hp, B, T)
x, c, z = (torch.from_numpy(inp[k]).cuda() for k in ("x", "c", "z"))
m.forward(x, c); torch.cuda.synchronize()
for _ in range(3): m.forward(x, c); m.reverse(z, c)
torch.cuda.synchronize()
t0 = time.perf_counter(); m.forward(x, c); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("forward: host enqueue %.3f ms, total %.3f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))

def run_split(nsplit, iters=20, graph=False):
    sb = B // nsplit
    xs = [x[i*sb:(i+1)*sb].contiguous() for i in range(nsplit)]
    cs = [c[i*sb:(i+1)*sb].contiguous() for i in range(nsplit)]
    zs = [z[i*sb:(i+1)*sb].contiguous() for i in range(nsplit)]
    streams = [torch.cuda.Stream() for _ in range(2 * nsplit)]
    def step():
        cur = torch.cuda.current_stream()
        outs = []
        for s_ in streams: s_.wait_stream(cur)
        for i in range(nsplit):
            with torch.cuda.stream(streams[2*i]): outs.append(m.forward(xs[i], cs[i]))
            with torch.cuda.stream(streams[2*i+1]): outs.append(m.reverse(zs[i], cs[i]))
        for s_ in streams: cur.wait_stream(s_)
        return outs
    for _ in range(3): step()
    torch.cuda.synchronize()
    if graph:
        g = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream()
        with torch.cuda.stream(cap):
            step(); torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=cap):
                outs = step()
        torch.cuda.synchronize()
        fn = g.replay
    else:
        fn = step
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print("split %d graph %d: %.3f ms/step  %.2f M samples/s" % (nsplit, graph, dt * 1e3, 2 * B * T / dt / 1e6), flush=True)

for ns in (1, 2, 4):
    run_split(ns, graph=False)
for ns in (1, 2, 4):
    try:
        run_split(ns, graph=True)
    except Exception as e:
        print("graph split", ns, "failed:", repr(e)[:300])
