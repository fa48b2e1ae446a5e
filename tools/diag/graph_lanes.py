"""Experiment (GPU box): bench.py's overlapped passes replayed from hipGraphs (one graph per lane and direction)
instead of ~600 launches per pass from the host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_flowavenet_amd import weights as W
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd.model import FloWaveNet

b, t, lanes, steps = 8, 16128, int(sys.argv[1]) if len(sys.argv) > 1 else 3, 30
split = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # each pass as `split` chains of b / split clips
hp = default_hparams()
dev = torch.device("cuda", 0)
model = FloWaveNet(hp, init=True, device=dev).load_params(W.synthetic_params(hp, 1234))
inp = W.synthetic_inputs(hp, b, t)
x, c, z = (torch.from_numpy(inp[k]).to(dev) for k in ("x", "c", "z"))
model.forward(x, c)
ref_wav = model.reverse(z, c).clone()
torch.cuda.synchronize()
graphs = []
bs = b // split
for k in range(lanes):
    for direction in ("f", "i"):
        for part in range(split):
            xs, cs, zs = x[part * bs:(part + 1) * bs], c[part * bs:(part + 1) * bs], z[part * bs:(part + 1) * bs]
            s = torch.cuda.Stream(dev)
            with torch.cuda.stream(s):
                for _ in range(2):
                    out = model.forward(xs, cs) if direction == "f" else model.reverse(zs, cs)     # allocates this stream's workspace
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                g.capture_begin()
                out = model.forward(xs, cs) if direction == "f" else model.reverse(zs, cs)
                g.capture_end()
            graphs.append((s, g, out, direction, part))
torch.cuda.synchronize()

def run(n):
    cur = torch.cuda.current_stream(dev)
    for s, *_ in graphs:
        s.wait_stream(cur)
    per = 2 * split
    for k in range(n):
        for s, g, *_ in graphs[per * (k % lanes):per * (k % lanes) + per]:
            with torch.cuda.stream(s):
                g.replay()
    for s, *_ in graphs:
        cur.wait_stream(s)

run(3); torch.cuda.synchronize()
t0 = time.perf_counter(); run(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
ok = all(torch.equal(o, ref_wav[part * bs:(part + 1) * bs]) for _, _, o, d, part in graphs if d == "i")
print("lanes %d split %d graph replay: %.3f ms per step, %.2f M samples/s (forward + inverse), inverse outputs bit-identical to serial: %s" % (
    lanes, split, dt / steps * 1e3, 2 * b * t * steps / dt / 1e6, ok))
