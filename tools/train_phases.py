"""Developer probe (GPU box): where a full-size training step spends its time (host vs GPU)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd import weights as W, training as TR

hp = default_hparams()
b, t = 8, 6400
p = W.synthetic_params(hp, 1234)
inp = W.synthetic_inputs(hp, b, t)
x, c = torch.from_numpy(inp["x"]).reshape(b, t).cuda(), torch.from_numpy(inp["c"]).cuda()
tr = TR.Trainer(hp, p)
tr.ddi(x, c)
tr.step(x, c)
torch.cuda.synchronize()
params = tr.opt.master_views()
t0 = time.perf_counter(); tp = TR._TrainPack(params, hp, "cuda"); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("pack: host %.1f ms, +gpu drain %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
t0 = time.perf_counter(); out = tr.engine.loss_and_grads(params, x, c); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("loss_and_grads (incl. pack): host %.1f ms, +gpu drain %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); tr.engine.loss_and_grads(params, x, c); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
