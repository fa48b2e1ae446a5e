"""Developer diagnostic: step-by-step timing of the full-size path with progress prints."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd import weights as W
from tf_flowavenet_amd.model import FloWaveNet

def log(*a):
    print("[%7.2f]" % (time.time() - T0), *a, flush=True)

T0 = time.time()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16128
hp = default_hparams()
params = W.synthetic_params(hp, 1234); log("params generated")
m = FloWaveNet(hp, init=True).load_params(params); log("packed, weight MB", m.weight_bytes / 1e6)
inp = W.synthetic_inputs(hp, B, T)
x, c, z = (torch.from_numpy(inp[k]).cuda() for k in ("x", "c", "z"))
torch.cuda.synchronize(); log("inputs on device")
lp, ld = m.forward(x, c); torch.cuda.synchronize(); log("ddi forward", float(lp), float(ld))
for i in range(3):
    t0 = time.time(); lp, ld = m.forward(x, c); torch.cuda.synchronize(); log("forward %.3f ms" % ((time.time() - t0) * 1e3), float(lp), float(ld))
for i in range(3):
    t0 = time.time(); w = m.reverse(z, c); torch.cuda.synchronize(); log("reverse %.3f ms" % ((time.time() - t0) * 1e3), float(w.abs().max()))
