"""GPU box: device time per launch of the residual layer (fwn_res) at blocks 0 / 1 of the 8-clip pass.
Round 4 sweep: two 64 KB workgroups per CU (128 x 128 tiles, ring depth 2; the product from
24 576 rows on) against one 147 KB workgroup (256 x 128, depth 3): 17.6 vs 19.4 us at block 0, 9.3 vs 9.4 at block 1.
With the tunable build the old tile is one switch away:
   FWN_LIB=tf-flowavenet_amd/csrc/libfwn_tune.so FWN_RES_TWO_PER_CU=0 python tools/probe/res_tiles.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_flowavenet_amd import _lib, weights as W
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd.model import FloWaveNet

hp = default_hparams()
model = FloWaveNet(hp).load_params(W.synthetic_params(hp, 1234))
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
for blk, M in ((0, 64512), (1, 32256)):
    d = model._packed.flow_descs[blk * hp.n_flow + 1]
    h = torch.randn(M, 256, device="cuda").to(torch.bfloat16)
    o = torch.randn(M, 256, device="cuda").to(torch.bfloat16)
    out = torch.empty_like(h)
    for rnd in range(3):
        for _ in range(5):
            _lib.check(lib.fwn_res(C.byref(d), 0, o.data_ptr(), h.data_ptr(), out.data_ptr(), M, st), "fwn_res")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            lib.fwn_res(C.byref(d), 0, o.data_ptr(), h.data_ptr(), out.data_ptr(), M, st)
        e1.record()
        torch.cuda.synchronize()
        print("block %d (M = %d): %.2f us per launch" % (blk, M, e0.elapsed_time(e1) * 1e3 / 40), flush=True)
