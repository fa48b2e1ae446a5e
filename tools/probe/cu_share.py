"""GPU box: how many CUs does an HBM-bound / MFMA-bound stage kernel need?  The block-0 residual layer (fwn_res, 99 MB per
launch) and the block-0 gate (fwn_gate) on streams confined to a CU subset (hipExtStreamCreateWithCUMask), device time per
launch from events on that stream.  A launch on a masked stream is slow to START, so 20 launches are
timed back to back.   python tools/probe/cu_share.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tf_flowavenet_amd import _lib, weights as W
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd.model import FloWaveNet

hip = C.CDLL("libamdhip64.so")
def masked(word):
    words = (C.c_uint32 * 8)(*(word if isinstance(word, (list, tuple)) else [word] * 8))
    st = C.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words) == 0
    return torch.cuda.ExternalStream(st.value)

hp = default_hparams()
model = FloWaveNet(hp).load_params(W.synthetic_params(hp, 1234))
lib = _lib.load()
d = model._packed.flow_descs[1]
M, Ti = 64512, 8064
h = torch.randn(M, 256, device="cuda").to(torch.bfloat16)
o = torch.randn(M, 256, device="cuda").to(torch.bfloat16)
ca = torch.rand(M, d.cin, device="cuda").to(torch.bfloat16)
out = torch.empty_like(h)
F, Z = 0xffffffff, 0
for name, word in (("all 256 CUs (plain stream)", None), ("0000ffff x8", 0x0000ffff), ("ffff0000 x8", 0xffff0000), ("00ff00ff x8", 0x00ff00ff), ("000000ff x8", 0x000000ff),
                   ("0000000f x8", 0x0000000f), ("words F,Z alternating", [F, Z] * 4), ("words FFFF ZZZZ", [F] * 4 + [Z] * 4), ("words F Z Z Z x2", [F, Z, Z, Z] * 2),
                   ("00ffffff x8", 0x00ffffff), ("0fffffff x8", 0x0fffffff), ("0000ffff,ffffffff alternating", [0x0000ffff, F] * 4)):
    st = torch.cuda.Stream() if word is None else masked(word)
    res = {}
    for kname, call in (("res", lambda s: lib.fwn_res(C.byref(d), 0, o.data_ptr(), h.data_ptr(), out.data_ptr(), M, s)),
                        ("gate", lambda s: lib.fwn_gate(C.byref(d), 0, h.data_ptr(), ca.data_ptr(), None, out.data_ptr(), M, Ti, s))):
        with torch.cuda.stream(st):
            for _ in range(3):
                _lib.check(call(st.cuda_stream), kname)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(20):
                call(st.cuda_stream)
            e1.record(st)
        torch.cuda.synchronize()
        res[kname] = e0.elapsed_time(e1) * 1e3 / 20
    print("%-30s res %.1f us  gate %.1f us" % (name, res["res"], res["gate"]), flush=True)
