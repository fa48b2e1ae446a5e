"""Developer probe (GPU box): phases of the recorded (hipGraph) full-size training step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd import weights as W, training as TR

hp = default_hparams()
b, t = 8, 6400
inp = W.synthetic_inputs(hp, b, t)
x, c = torch.from_numpy(inp["x"]).reshape(b, t).cuda(), torch.from_numpy(inp["c"]).cuda()
tr = TR.Trainer(hp, W.synthetic_params(hp, 1234), graph=True)
tr.ddi(x, c)
for _ in range(3):
    tr.step(x, c)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
t_tab = t_rep = 0.0
t00 = time.perf_counter()
for _ in range(n):
    t0 = time.perf_counter()
    tr.engine.refresh_host_tables()
    t1 = time.perf_counter()
    tr.opt.advance()
    for g, i in tr._recorded[next(iter(tr._recorded))]["segs"]:
        g.replay()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    t_tab += t1 - t0
    t_rep += t2 - t1
print("per step: host tables %.2f ms, replay + drain %.2f ms, total %.2f ms" % (t_tab / n * 1e3, t_rep / n * 1e3, (time.perf_counter() - t00) / n * 1e3))
