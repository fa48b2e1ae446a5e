import sys, os, traceback, collections
sys.path.insert(0, os.getcwd())
import torch
from tf_flowavenet_amd import _lib, weights as W
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd.training import Trainer
hp = default_hparams()
lib = _lib.load()
orig = lib.fwn_wn_scale
sites = collections.Counter()
phase = ["init"]
def wrapped(*a):
    st = traceback.extract_stack(limit=6)
    sites[(phase[0],) + tuple("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in st[:-1])] += 1
    return orig(*a)
lib.fwn_wn_scale = wrapped
tr = Trainer(hp, W.synthetic_params(hp, 1234), device="cuda")
inp = W.synthetic_inputs(hp, 8, 6400)
x, c = torch.from_numpy(inp["x"]).cuda(), torch.from_numpy(inp["c"]).cuda()
tr.ddi(x, c)
for k in range(3):
    phase[0] = "step%d" % k
    tr.step(x, c)
torch.cuda.synchronize()
for k, v in sorted(sites.items(), key=lambda kv: -kv[1])[:12]:
    print(v, k)
