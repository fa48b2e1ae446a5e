"""Per-stage micro-benchmark through the C-ABI (GPU box): every stage kernel of a flow (front, gate 0, res 0, gate 1,
tail) at every block's shape, timed alone with HIP events, the launches rotating over the block's 6 flows so each one
meets weights it has not just used.  FWN_LIB selects the build; several libraries can be A/B'd in one call, each in a
child process, interleaved:

    python tools/stage_bench.py [--batch 8] [--samples 16128] [--blocks 0,1,2,3] [--libs a.so,b.so] [--rounds 2]

Prints us per launch per (block, stage) - a developer tool for same-box A/B; in-situ numbers come from tools/pass_table.py.
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, os, json, ctypes as C
sys.path.insert(0, %(root)r)
import torch
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd import weights as W, _lib
from tf_flowavenet_amd.model import FloWaveNet
b, t, blocks, iters = %(b)d, %(t)d, %(blocks)r, %(iters)d
hp = default_hparams()
m = FloWaveNet(hp, device="cuda").load_params(W.synthetic_params(hp, 1234, actnorm="random"))
lib = _lib.load()
descs = m._packed.flow_descs
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
out = {}
def timed(fn, n):
    for k in range(6): fn(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(n): fn(k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for i in blocks:
    ch = 1 << i
    ti = t // (2 * ch)
    M = b * ti
    cin = 40 * 2 * ch
    g = torch.Generator(device="cuda").manual_seed(i)
    xa = torch.randn(M, ch, device=dev, generator=g) * 0.3
    xb = torch.randn(M, ch, device=dev, generator=g) * 0.3
    h0 = (torch.randn(M, 256, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    h1 = torch.empty_like(h0)
    o = (torch.randn(2, M, 256, device=dev, generator=g) * 0.3).to(torch.bfloat16)
    ca = torch.rand(M, cin, device=dev, generator=g).to(torch.bfloat16)
    scratch = torch.empty(2, M, 256, device=dev, dtype=torch.bfloat16)
    part = torch.zeros(int(lib.fwn_tail_partials(M)) + 8, device=dev)
    fd = [descs[i * hp.n_flow + j] for j in range(hp.n_flow)]
    ck = lambda rc, what: _lib.check(rc, what)
    r = {}
    r["front"] = timed(lambda k: ck(lib.fwn_front(C.byref(fd[k %% 6]), xa.data_ptr(), h0.data_ptr(), h1.data_ptr(), M, ti, 1, st), "front"), iters)
    r["gate0"] = timed(lambda k: ck(lib.fwn_gate(C.byref(fd[k %% 6]), 0, h0.data_ptr(), ca.data_ptr(), None, o[0].data_ptr(), M, ti, st), "gate"), iters)
    r["res0"] = timed(lambda k: ck(lib.fwn_res(C.byref(fd[k %% 6]), 0, o[0].data_ptr(), h0.data_ptr(), h1.data_ptr(), M, st), "res"), iters)
    r["gate1"] = timed(lambda k: ck(lib.fwn_gate(C.byref(fd[k %% 6]), 1, h1.data_ptr(), ca.data_ptr(), None, o[1].data_ptr(), M, ti, st), "gate"), iters)
    xa2, xb2 = xa.clone(), xb.clone()
    def tail(k):
        ck(lib.fwn_tail(C.byref(fd[k %% 6]), o.data_ptr(), xa2.data_ptr(), xb2.data_ptr(), part.data_ptr(), M, 0, scratch.data_ptr(), st), "tail")
    r["tail"] = timed(tail, iters)
    def flow(k):
        ck(lib.fwn_flow_run(C.byref(fd[k %% 6]), b, t, xa2.data_ptr(), xb2.data_ptr(), ca.data_ptr(), h0.data_ptr(), h1.data_ptr(), o.data_ptr(), None, part.data_ptr(), 0, 0, st), "flow")
    r["flow"] = timed(flow, iters)
    out[i] = r
print(json.dumps(out))
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--samples", type=int, default=16128)
    ap.add_argument("--blocks", default="0,1,2,3,4,5,6,7")
    ap.add_argument("--libs", default="")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--iters", type=int, default=60)
    a = ap.parse_args()
    import json
    blocks = [int(s) for s in a.blocks.split(",")]
    libs = [s for s in a.libs.split(",") if s] or [os.path.join(ROOT, "tf-flowavenet_amd", "csrc", "libfwn.so")]
    child = CHILD % dict(root=ROOT, b=a.batch, t=a.samples, blocks=blocks, iters=a.iters)
    res = {lib: [] for lib in libs}
    for _ in range(a.rounds):
        for lib in libs:
            r = subprocess.run([sys.executable, "-c", child], env=dict(os.environ, FWN_LIB=os.path.abspath(lib)), capture_output=True, text=True)
            if r.returncode != 0:
                print(lib, "FAILED:", r.stderr[-1500:])
                continue
            res[lib].append(json.loads(r.stdout.strip().splitlines()[-1]))
    stages = ("front", "gate0", "res0", "gate1", "tail", "flow")
    print("us per launch (min over %d rounds), B=%d T=%d" % (a.rounds, a.batch, a.samples))
    print("%-28s block " % "lib" + " ".join("%8s" % s for s in stages))
    for i in blocks:
        for lib in libs:
            if not res[lib]:
                continue
            vals = [min(rr[str(i)][s] for rr in res[lib]) for s in stages]
            print("%-28s %5d " % (os.path.basename(lib)[-28:], i) + " ".join("%8.2f" % v for v in vals))


if __name__ == "__main__":
    main()
