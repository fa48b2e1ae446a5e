"""Developer check (GPU box): GradEngine against the autograd oracle on a tiny model; prints the
relative error per parameter tensor."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from conftest import small_hparams
from oracle import grad_torch as G
from tf_flowavenet_amd import weights as W
from tf_flowavenet_amd.training import GradEngine

cfg = dict(n_block=2, n_flow=2, n_layer=2, hop_size=16, upsample_scales=[4, 4], num_mels=8)
b, t = 2, 128
if len(sys.argv) > 1:
    cfg = dict(n_block=3, n_flow=3, n_layer=2, hop_size=16, upsample_scales=[4, 4], num_mels=16); b, t = 3, 256
hp = small_hparams(**cfg)
p = W.synthetic_params(hp, 5)
inp = W.synthetic_inputs(hp, b, t)
loss0, lp0, ld0, g0 = G.loss_and_grads(p, inp["x"], inp["c"], hp)
eng = GradEngine(hp)
loss, lp, ld, g = eng.loss_and_grads(p, torch.from_numpy(inp["x"]).reshape(b, t), torch.from_numpy(inp["c"]))
torch.cuda.synchronize()
print("loss %.6f / %.6f   log_p %.6f / %.6f   logdet %.6f / %.6f" % (float(loss), loss0, float(lp), lp0, float(ld), ld0))
worst = []
for k in sorted(g0):
    a, r = g[k].detach().cpu().numpy().astype(np.float64).reshape(-1), g0[k].reshape(-1)
    nr = np.linalg.norm(r)
    err = np.linalg.norm(a - r) / max(nr, 1e-12)
    worst.append((err, k, nr))
worst.sort(reverse=True)
for err, k, nr in worst[:40]:
    print("%-60s rel %.3e  |ref| %.3e" % (k, err, nr))
print("median rel err %.3e" % np.median([w[0] for w in worst]))
