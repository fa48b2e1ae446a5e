"""Diagnostic: fwn_front alone per block (B = 8, T = 16128), replayed back to back."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, ctypes as C
from tf_flowavenet_amd import weights as W, _lib
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd.model import FloWaveNet
hp = default_hparams()
m = FloWaveNet(hp, init=False).load_params(W.synthetic_params(hp, 1234, actnorm="random"))
lib = _lib.load()
B, T = 8, 16128
st = torch.cuda.current_stream().cuda_stream
for i in range(hp.n_block):
    ch = 1 << i
    M, Ti = B * T // (2 * ch), T // (2 * ch)
    d = m._packed.flow_descs[i * hp.n_flow]
    xa = torch.randn(M, ch, device="cuda")
    h = torch.empty(M, 256, dtype=torch.bfloat16, device="cuda")
    scr = torch.empty(M * 2 * ch, dtype=torch.bfloat16, device="cuda")
    def run():
        _lib.check(lib.fwn_front(C.byref(d), xa.data_ptr(), h.data_ptr(), scr.data_ptr(), M, Ti, 1, st), "fwn_front")
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): run()
    e1.record(); torch.cuda.synchronize()
    print("block %d  M %6d Ch %3d  fwn_front %.2f us" % (i, M, ch, e0.elapsed_time(e1) * 1000 / 200))
