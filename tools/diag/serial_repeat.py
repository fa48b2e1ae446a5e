import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_flowavenet_amd import weights as W
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd.model import FloWaveNet
hp = default_hparams()
m = FloWaveNet(hp, init=True).load_params(W.synthetic_params(hp, 1234))
inp = W.synthetic_inputs(hp, 8, 16128)
x, c, z = (torch.from_numpy(inp[k]).cuda() for k in ("x", "c", "z"))
m.forward(x, c)
outs = [torch.stack(m.forward(x, c)).clone() for _ in range(6)]
wavs = [m.reverse(z, c).clone() for _ in range(6)]
torch.cuda.synchronize()
print("serial forward identical:", all(torch.equal(outs[0], o) for o in outs), [float(o[0]) for o in outs[:3]])
print("serial inverse identical:", all(torch.equal(wavs[0], w) for w in wavs), "max diff", max(float((wavs[0] - w).abs().max()) for w in wavs))
