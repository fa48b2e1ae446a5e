"""Experiment (GPU box): bench.py's overlapped passes replayed from hipGraphs (one graph per lane and direction)
instead of ~600 launches per pass from the host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_flowavenet_amd import weights as W
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd.model import FloWaveNet

b, t, lanes, steps = 8, 16128, int(sys.argv[1]) if len(sys.argv) > 1 else 3, 30
hp = default_hparams()
dev = torch.device("cuda", 0)
model = FloWaveNet(hp, init=True, device=dev).load_params(W.synthetic_params(hp, 1234))
inp = W.synthetic_inputs(hp, b, t)
x, c, z = (torch.from_numpy(inp[k]).to(dev) for k in ("x", "c", "z"))
model.forward(x, c)
ref_wav = model.reverse(z, c).clone()
torch.cuda.synchronize()
graphs = []
for k in range(lanes):
    for direction in ("f", "i"):
        s = torch.cuda.Stream(dev)
        with torch.cuda.stream(s):
            for _ in range(2):
                out = model.forward(x, c) if direction == "f" else model.reverse(z, c)     # allocates this stream's workspace
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            g.capture_begin()
            out = model.forward(x, c) if direction == "f" else model.reverse(z, c)
            g.capture_end()
        graphs.append((s, g, out, direction))
torch.cuda.synchronize()

def run(n):
    cur = torch.cuda.current_stream(dev)
    for s, _, _, _ in graphs:
        s.wait_stream(cur)
    for k in range(n):
        for s, g, _, _ in graphs[2 * (k % lanes):2 * (k % lanes) + 2]:
            with torch.cuda.stream(s):
                g.replay()
    for s, _, _, _ in graphs:
        cur.wait_stream(s)

run(3); torch.cuda.synchronize()
t0 = time.perf_counter(); run(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
ok = all(torch.equal(o, ref_wav) for _, _, o, d in graphs if d == "i")
print("lanes %d graph replay: %.3f ms per step, %.2f M samples/s, inverse outputs bit-identical to serial: %s" % (lanes, dt / steps * 1e3, b * t * steps / dt / 1e6, ok))
