import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd import weights as W
from tf_flowavenet_amd.model import FloWaveNet
from oracle import flowavenet_np as onp
hp = default_hparams().replace(n_block=3, n_flow=2, n_layer=2, hop_size=16, upsample_scales=[4,4], num_mels=16)
params = W.synthetic_params(hp, 1234, actnorm="random")
inp = W.synthetic_inputs(hp, 1, 256)
m = FloWaveNet(hp).load_params(params)
lp, ld, zp = m.forward(torch.from_numpy(inp["x"]).cuda(), torch.from_numpy(inp["c"]).cuda(), return_z=True)
p64 = onp.to_f64(params)
lp0, ld0, z0 = onp.forward(p64, inp["x"].astype(np.float64), inp["c"].astype(np.float64), hp)
from tf_flowavenet_amd.model import z_planes_to_squeezed
z = z_planes_to_squeezed(zp, 3, 2).cpu().numpy()
err = np.abs(z - z0)[0]
print("lp", float(lp), lp0, "ld", float(ld), ld0)
print("max err per channel", err.max(0))
print("rows with err>0.01:", np.where(err.max(1) > 0.01)[0])
