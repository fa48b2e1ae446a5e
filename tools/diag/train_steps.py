"""GPU box: loss / log_p / logdet / gradient norm of the first steps of the training benchmark's run (for comparing two trees)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd import weights as W
from tf_flowavenet_amd.training import Trainer
hp = default_hparams()
inp = W.synthetic_inputs(hp, 8, 6400)
x = torch.from_numpy(inp["x"]).reshape(8, 6400).cuda()
c = torch.from_numpy(inp["c"]).cuda()
tr = Trainer(hp, W.synthetic_params(hp, 1234))
tr.ddi(x, c)
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    loss, lp, ld, gn = tr.step(x, c)
    print("step %d: loss %.6f log_p %.6f logdet %.6f gnorm %.6f" % (k, float(loss), float(lp), float(ld), float(gn)))
