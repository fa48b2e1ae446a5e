"""Diagnostic: one process; does a pass on a non-default stream (alone, serial) equal the default-stream result?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
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
ref_wav = model.reverse(z, c).clone()
ref2 = model.reverse(z, c).clone()
print("default stream twice equal:", torch.equal(ref_wav, ref2))
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
torch.cuda.synchronize()
for name, s in (("s1", s1), ("s1 again", s1), ("s2", s2)):
    with torch.cuda.stream(s):
        w = model.reverse(z, c).clone()
    torch.cuda.synchronize()
    print(name, "alone equals default:", torch.equal(w, ref_wav), "max abs", float((w - ref_wav).abs().max()))
# two streams concurrently, inverse only
outs = []
for k in range(8):
    with torch.cuda.stream(s1 if k % 2 == 0 else s2):
        outs.append(model.reverse(z, c).clone())
torch.cuda.synchronize()
print("two lanes concurrent (inverse only): mismatches", sum(int(not torch.equal(w, ref_wav)) for w in outs), "of 8; max abs",
      max(float((w - ref_wav).abs().max()) for w in outs))
outs = []
for k in range(8):
    with torch.cuda.stream(s1):
        outs.append(model.reverse(z, c).clone())
torch.cuda.synchronize()
print("one lane back-to-back (no sync): mismatches", sum(int(not torch.equal(w, ref_wav)) for w in outs), "of 8")
