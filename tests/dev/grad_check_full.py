"""Developer check (GPU box): full-size model (n_block=8, n_flow=6) gradients against the autograd
oracle on a short clip.  Slow on the CPU side (fp64 autograd of 181 M parameters)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
torch.set_num_threads(32)
from oracle import grad_torch as G
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd import weights as W
from tf_flowavenet_amd.training import GradEngine

hp = default_hparams()
b, t = 2, 1024
p = W.synthetic_params(hp, 1234, actnorm="random")
inp = W.synthetic_inputs(hp, b, t)
t0 = time.time()
loss0, lp0, ld0, g0 = G.loss_and_grads(p, inp["x"], inp["c"], hp)
print("oracle %.1f s" % (time.time() - t0), flush=True)
loss, lp, ld, g = GradEngine(hp).loss_and_grads(p, torch.from_numpy(inp["x"]).reshape(b, t), torch.from_numpy(inp["c"]))
torch.cuda.synchronize()
print("loss %.6f / %.6f   log_p %.6f / %.6f   logdet %.6f / %.6f" % (float(loss), loss0, float(lp), lp0, float(ld), ld0))
worst, dot, na, nb = [], 0.0, 0.0, 0.0
for k in sorted(g0):
    a, r = g[k].detach().cpu().numpy().astype(np.float64).reshape(-1), g0[k].reshape(-1)
    nr = np.linalg.norm(r)
    if nr == 0:
        assert not a.any(), k
        continue
    worst.append((np.linalg.norm(a - r) / nr, k, nr))
    dot += float(a @ r); na += float(a @ a); nb += float(r @ r)
worst.sort(reverse=True)
for err, k, nr in worst[:25]:
    print("%-60s rel %.3e  |ref| %.3e" % (k, err, nr))
print("tensors %d  median rel err %.3e  cosine %.6f" % (len(worst), np.median([w[0] for w in worst]), dot / np.sqrt(na * nb)))
by_block = {}
for err, k, nr in worst:
    by_block.setdefault(k.split("/")[0], []).append(err)
for kb in sorted(by_block):
    print("%-10s median %.3e  max %.3e" % (kb, np.median(by_block[kb]), max(by_block[kb])))
