import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd import weights as W
from tf_flowavenet_amd.model import FloWaveNet, z_planes_to_squeezed
hp = default_hparams()
for name, b, t in (("full_b8f6_B1_T16128", 1, 16128), ("full_b8f6_B8_T16128", 8, 16128)):
    g = np.load("tests/golden/%s.npz" % name)
    m = FloWaveNet(hp, init=True).load_params(W.synthetic_params(hp, 1234))
    inp = W.synthetic_inputs(hp, b, t)
    lp, ld, zp = m.forward(torch.from_numpy(inp["x"]).cuda(), torch.from_numpy(inp["c"]).cuda(), return_z=True)
    z = z_planes_to_squeezed(zp, hp.n_block, hp.n_flow).cpu().numpy()
    d = np.abs(z - g["z"].astype(np.float32))
    w = m.reverse(torch.from_numpy(inp["z"]).cuda(), torch.from_numpy(inp["c"]).cuda()).cpu().numpy()
    dw = np.abs(w - g["x_rev"].astype(np.float32))
    print(os.environ.get("FWN_TAIL_SPLIT_MAX"), name, "z max %.4f mean %.5f | wav max %.4f mean %.5f | lp %.6f/%.6f ld %.6f/%.6f" % (d.max(), d.mean(), dw.max(), dw.mean(), float(lp), float(g["log_p"]), float(ld), float(g["logdet"])))
