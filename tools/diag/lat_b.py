import sys, json
sys.path.insert(0, '/root/repo')
import torch, numpy as np
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd import weights as W
from tf_flowavenet_amd.model import FloWaveNet
hp = default_hparams()
m = FloWaveNet(hp, init=True).load_params(W.synthetic_params(hp, 1234))
inp8 = W.synthetic_inputs(hp, 8, 16128)
m.forward(torch.from_numpy(inp8["x"]).cuda(), torch.from_numpy(inp8["c"]).cuda())
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts=[]
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
for b in (1, 2, 4):
    inp = W.synthetic_inputs(hp, b, 16128)
    x, c, z = (torch.from_numpy(inp[k]).cuda() for k in ("x", "c", "z"))
    print("B=%d fwd %.3f ms inv %.3f ms" % (b, timed(lambda: m.forward(x, c)), timed(lambda: m.reverse(z, c))))
