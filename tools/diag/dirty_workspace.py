import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from conftest import small_hparams
from tf_flowavenet_amd import weights as W, _lib
from tf_flowavenet_amd.training import GradEngine
hp = small_hparams(n_block=2, n_flow=2, n_layer=2, hop_size=16, upsample_scales=[4, 4], num_mels=8)
p = W.synthetic_params(hp, 3, actnorm="random")
inp = W.synthetic_inputs(hp, 2, 128)
x, c = torch.from_numpy(inp["x"]).reshape(2, 128).cuda(), torch.from_numpy(inp["c"]).cuda()
eng = GradEngine(hp)
l0 = float(eng.loss_and_grads(p, x, c)[0])
lib, td = eng.lib, eng._desc
need = int(lib.fwn_train_workspace_bytes(C.byref(td), 2, 128))
out3 = torch.zeros(3, device="cuda")
cb = _lib.BLOCK_DONE_FN(lambda user, blk: 0)
for fill in (0, 255, 0x7f, 0x3c):
    ws = torch.full((need + 512,), fill, dtype=torch.uint8, device="cuda")
    base = ws.data_ptr() + (-ws.data_ptr()) % 256
    rc = lib.fwn_train_loss_and_grads(C.byref(td), 2, 128, x.data_ptr(), c.data_ptr(), base, need, out3.data_ptr(), cb, None, None)
    torch.cuda.synchronize()
    print("fill", fill, "rc", rc, "loss", out3.tolist(), "expected", l0)
# the engine's own second call (reused workspace)
print("engine again:", float(eng.loss_and_grads(p, x, c)[0]))
