"""Diagnostic: ONE process, bench.py's stream lanes: is every step's result bit-identical to the serial result?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tf_flowavenet_amd import weights as W
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd.model import FloWaveNet

b, t, lanes, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
hp = default_hparams()
dev = torch.device("cuda", 0)
model = FloWaveNet(hp, init=True, device=dev).load_params(W.synthetic_params(hp, 1234))
inp = W.synthetic_inputs(hp, b, t)
x, c, z = (torch.from_numpy(inp[k]).to(dev) for k in ("x", "c", "z"))
model.forward(x, c)
lp0, ld0 = model.forward(x, c)
ref_nll = torch.stack([lp0, ld0]).clone()
ref_wav = model.reverse(z, c).clone()
torch.cuda.synchronize()
lf = [torch.cuda.Stream(dev) for _ in range(lanes)]
li = [torch.cuda.Stream(dev) for _ in range(lanes)]
bad_f = bad_i = 0
for rnd in range(steps // 12):
    outs = []
    cur = torch.cuda.current_stream(dev)
    for s in lf + li:
        s.wait_stream(cur)
    for k in range(12):
        with torch.cuda.stream(lf[k % lanes]):
            lp, ld = model.forward(x, c)
            nll = torch.stack([lp, ld])
        with torch.cuda.stream(li[k % lanes]):
            wav = model.reverse(z, c).clone()
        outs.append((nll, wav))
    for s in lf + li:
        cur.wait_stream(s)
    torch.cuda.synchronize()
    for nll, wav in outs:
        bad_f += int(not torch.equal(nll, ref_nll))
        if not torch.equal(wav, ref_wav):
            bad_i += 1
            if bad_i <= 3:
                d = (wav != ref_wav)
                print("inverse mismatch: ndiff", int(d.sum()), "max abs", float((wav - ref_wav).abs().max()), flush=True)
print("B", b, "T", t, "lanes", lanes, "steps", steps // 12 * 12, "forward mismatches", bad_f, "inverse mismatches", bad_i)
