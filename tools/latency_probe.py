"""Developer probe (GPU box): B=1 single-pass latency, wall vs summed kernel time (run under rocprofv3 --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd import weights as W
from tf_flowavenet_amd.model import FloWaveNet
hp = default_hparams()
b, t, n = int(sys.argv[1]) if len(sys.argv) > 1 else 1, 16128, 50
m = FloWaveNet(hp, init=True).load_params(W.synthetic_params(hp, 1234))
inp = W.synthetic_inputs(hp, b, t)
x, c, z = (torch.from_numpy(inp[k]).cuda() for k in ("x", "c", "z"))
m.forward(x, c)
for _ in range(5): m.reverse(z, c)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): m.reverse(z, c)
e1.record(); torch.cuda.synchronize()
print("B=%d inverse wall %.3f ms per pass (%d passes timed, 5 warm-up, 1 forward)" % (b, e0.elapsed_time(e1) / n, n))
