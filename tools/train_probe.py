"""Developer probe (GPU box): a few full-size training steps (configs[2]: batch 8 per GPU, 6400-sample
crops, n_block=8 n_flow=6) on one GPU: step time, memory, loss trajectory on a fixed batch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tf_flowavenet_amd.hparams import default_hparams, hparams8000
from tf_flowavenet_amd import weights as W
from tf_flowavenet_amd.training import Trainer

hp = hparams8000() if os.environ.get("FWN_8K") else default_hparams()
b, t = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 6400
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
p = W.synthetic_params(hp, 1234)
inp = W.synthetic_inputs(hp, b, t)
x, c = torch.from_numpy(inp["x"]).reshape(b, t).cuda(), torch.from_numpy(inp["c"]).cuda()
tr = Trainer(hp, p)
tr.ddi(x, c)
torch.cuda.synchronize()
for k in range(steps):
    t0 = time.perf_counter()
    loss, lp, ld, gn = tr.step(x, c)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("step %d  loss %.5f  log_p %.5f  logdet %.5f  |g| %.4f   %.1f ms   peak mem %.2f GB" % (
        k, float(loss), float(lp), float(ld), float(gn), dt * 1e3, torch.cuda.max_memory_allocated() / 2**30), flush=True)
