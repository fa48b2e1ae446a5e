"""Diagnostic: ONE process.  Stream A repeats fwn_flow_run of one block on fixed inputs; stream B runs an aggressor
(whole-model inverse passes, or the same flow).  Which intermediate of A first differs from its solo result?"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tf_flowavenet_amd import _lib, weights as W
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd.model import FloWaveNet
b, t, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
aggr = sys.argv[4] if len(sys.argv) > 4 else "model"
hp = default_hparams()
dev = torch.device("cuda", 0)
model = FloWaveNet(hp, init=True, device=dev).load_params(W.synthetic_params(hp, 1234))
inp = W.synthetic_inputs(hp, b, t)
x, c, z = (torch.from_numpy(inp[k]).to(dev) for k in ("x", "c", "z"))
model.forward(x, c)
lib = _lib.load()
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
L = hp.n_layer


def run_flow(blk, st, bufs):
    d = model._packed.flow_descs[blk * hp.n_flow]
    ch = 1 << blk
    ti = t // (2 * ch)
    m = b * ti
    xa, xb = bufs["xa0"].clone(), bufs["xb0"].clone()
    if m < 4096:
        _lib.check(lib.fwn_cond(bufs["ca"].data_ptr(), d.Wc[0], bufs["P"].data_ptr(), 512 * d.kcpad, m * 512, 0, 1, 1, L, m, d.cin, d.kcpad, st), "cond")
    _lib.check(lib.fwn_flow_run(C.byref(d), b, t, xa.data_ptr(), xb.data_ptr(), None if m < 4096 else bufs["ca"].data_ptr(),
                                bufs["h0"].data_ptr(), bufs["h1"].data_ptr(), bufs["o"].data_ptr(), bufs["P"].data_ptr() if m < 4096 else None,
                                bufs["partial"].data_ptr(), 0, 0, st), "flow_run")
    out = dict(h0=bufs["h0"].clone(), o0=bufs["o"][0].clone(), h1=bufs["h1"].clone(), o1=bufs["o"][1].clone(), xb=xb, partial=bufs["partial"].clone())
    if m < 4096:
        out["P"] = bufs["P"].clone()
    return out


def make_bufs(blk, seed):
    d = model._packed.flow_descs[blk * hp.n_flow]
    ch = 1 << blk
    m = b * (t // (2 * ch))
    g = torch.Generator(device="cpu").manual_seed(seed)
    return dict(xa0=torch.randn(m, ch, generator=g).to(dev), xb0=torch.randn(m, ch, generator=g).to(dev),
                ca=torch.rand(m, d.cin, generator=g).to(dev).to(torch.bfloat16),
                h0=torch.empty(m, 256, device=dev, dtype=torch.bfloat16), h1=torch.empty(m, 256, device=dev, dtype=torch.bfloat16),
                o=torch.empty(L, m, 256, device=dev, dtype=torch.bfloat16), P=torch.empty(L, m, 512, device=dev, dtype=torch.float32),
                partial=torch.zeros(lib.fwn_tail_partials(m), device=dev, dtype=torch.float32))


for blk in range(hp.n_block):
    bufs = make_bufs(blk, blk)
    bufs_b = make_bufs(blk, 100 + blk)
    with torch.cuda.stream(sa):
        ref = run_flow(blk, sa.cuda_stream, bufs)
    torch.cuda.synchronize()
    bad = {}
    for it in range(iters):
        with torch.cuda.stream(sb):
            if aggr == "model":
                model.reverse(z, c)
            else:
                for _ in range(4):
                    run_flow(blk, sb.cuda_stream, bufs_b)
        with torch.cuda.stream(sa):
            cur = run_flow(blk, sa.cuda_stream, bufs)
        torch.cuda.synchronize()
        for k in cur:
            if not torch.equal(cur[k], ref[k]):
                bad.setdefault(k, []).append(int((cur[k] != ref[k]).sum()))
    torch.cuda.synchronize()
    print("block", blk, "M", b * (t // (2 << blk)), {k: (len(v), v[:3]) for k, v in bad.items()}, flush=True)
