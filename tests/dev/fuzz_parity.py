"""Developer fuzz (GPU box): random small configurations, forward + inverse through the model surface
against the NumPy oracle.  usage: python tests/dev/fuzz_parity.py [n_cases] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import flowavenet_np as onp
from tf_flowavenet_amd import weights as W
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd.model import FloWaveNet, z_planes_to_squeezed

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    n_block = int(rng.integers(1, 6)); n_flow = int(rng.integers(1, 5)); n_layer = int(rng.integers(1, 4))
    s0, s1 = int(rng.choice([2, 4])), int(rng.choice([2, 4, 8]))
    hop = s0 * s1
    num_mels = int(rng.choice([8, 16, 24, 80]))
    hp = default_hparams().replace(n_block=n_block, n_flow=n_flow, n_layer=n_layer, hop_size=hop,
                                   upsample_scales=[s0, s1], num_mels=num_mels)
    unit = int(np.lcm(hop, 1 << n_block))
    t = unit * int(rng.integers(1, 9)); b = int(rng.integers(1, 6))
    if t < 8:
        continue
    p = W.synthetic_params(hp, int(rng.integers(1 << 30)), actnorm="random")
    inp = W.synthetic_inputs(hp, b, t)
    p64 = onp.to_f64(p)
    lp0, ld0, z0 = onp.forward(p64, inp["x"].astype(np.float64), inp["c"].astype(np.float64), hp)
    m = FloWaveNet(hp, cond_mode=int(rng.integers(0, 3))).load_params(p)
    lp, ld, zp = m.forward(torch.from_numpy(inp["x"]), torch.from_numpy(inp["c"]), return_z=True)
    zs = z_planes_to_squeezed(zp, n_block, n_flow).cpu().numpy()
    e_lp, e_ld = abs(float(lp) - lp0) / abs(lp0), abs(float(ld) - ld0) / max(1.0, abs(ld0))
    e_z = float(np.abs(zs - z0).max())
    msg = "case %2d n_block %d n_flow %d L %d mels %2d hop %2d B %d T %4d | log_p %.1e logdet %.1e z %.1e" % (
        case, n_block, n_flow, n_layer, num_mels, hop, b, t, e_lp, e_ld, e_z)
    ok = e_lp < 1e-3 and e_ld < 1e-3 and e_z < 3e-2
    if (n_block * n_flow) % 2 == 0:
        x0 = onp.reverse(p64, inp["z"].astype(np.float64), inp["c"].astype(np.float64), hp)
        xr = m.reverse(torch.from_numpy(inp["z"]), torch.from_numpy(inp["c"])).cpu().numpy()
        e_x = float(np.abs(xr - x0).max()) / max(1.0, float(np.abs(x0).max()))
        msg += " inv %.1e" % e_x
        ok = ok and e_x < 5e-2
    print(msg + ("" if ok else "   <-- FAIL"), flush=True)
    bad += not ok
print("failures: %d" % bad)
