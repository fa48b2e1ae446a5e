"""Experiment (GPU box): single-pass latency (B = 1) launched from the host vs replayed from a hipGraph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_flowavenet_amd import weights as W
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd.model import FloWaveNet
b, t = int(sys.argv[1]), int(sys.argv[2])
hp = default_hparams()
dev = torch.device("cuda", 0)
model = FloWaveNet(hp, init=True, device=dev).load_params(W.synthetic_params(hp, 1234))
inp = W.synthetic_inputs(hp, b, t)
x, c, z = (torch.from_numpy(inp[k]).to(dev) for k in ("x", "c", "z"))
model.forward(x, c)
def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
host = timed(lambda: model.reverse(z, c))
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    for _ in range(2): model.reverse(z, c)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    g.capture_begin()
    out = model.reverse(z, c)
    g.capture_end()
torch.cuda.synchronize()
with torch.cuda.stream(s):
    graph = timed(lambda: g.replay())
print("B %d T %d inverse: host launches %.3f ms, graph replay %.3f ms" % (b, t, host, graph))
