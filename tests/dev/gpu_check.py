"""Developer diagnostic (runs on the GPU box): HIP path vs the fp64 oracle on small configs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd import weights as W
from tf_flowavenet_amd.model import FloWaveNet, z_planes_to_squeezed
from oracle import flowavenet_np as onp


def run(n_block, n_flow, n_layer, B, T, hop_scales, mels, cond_mode, ddi=False, seed=1234):
    hop = int(np.prod(hop_scales))
    hp = default_hparams().replace(n_block=n_block, n_flow=n_flow, n_layer=n_layer, hop_size=hop,
                                   upsample_scales=list(hop_scales), num_mels=mels)
    params = W.synthetic_params(hp, seed, actnorm="zeros" if ddi else "random")
    inp = W.synthetic_inputs(hp, B, T)
    m = FloWaveNet(hp, init=ddi, cond_mode=cond_mode).load_params(params)
    x = torch.from_numpy(inp["x"]).cuda(); c = torch.from_numpy(inp["c"]).cuda(); z = torch.from_numpy(inp["z"]).cuda()
    lp, ld, zp = m.forward(x, c, return_z=True)
    torch.cuda.synchronize()
    p64 = onp.to_f64(params)
    lp0, ld0, z0 = onp.forward(p64, inp["x"].astype(np.float64), inp["c"].astype(np.float64), hp, init=ddi)
    zs = z_planes_to_squeezed(zp, n_block).cpu().numpy()
    ez = np.abs(zs - z0).max()
    msg = "nb=%d nf=%d nl=%d B=%d T=%d cm=%d ddi=%d | log_p %.6f vs %.6f (rel %.2e) logdet %.6f vs %.6f (rel %.2e) | z maxerr %.3e (rms z %.3f)" % (
        n_block, n_flow, n_layer, B, T, cond_mode, ddi, float(lp), lp0, abs(float(lp) - lp0) / abs(lp0),
        float(ld), ld0, abs(float(ld) - ld0) / max(abs(ld0), 1e-9), ez, np.sqrt((z0 ** 2).mean()))
    if (n_block * n_flow) % 2 == 0:
        xr = m.reverse(z, c).cpu().numpy()
        xr0 = onp.reverse(p64, inp["z"].astype(np.float64), inp["c"].astype(np.float64), hp)
        msg += " | rev maxerr %.3e (rms %.3f)" % (np.abs(xr - xr0).max(), np.sqrt((xr0 ** 2).mean()))
        # round trip on device: reverse(forward z) == x
        zflat = torch.empty(B, T, 1, device="cuda")
        zflat[:, 0::2, 0] = zp[0]; zflat[:, 1::2, 0] = zp[1]
        xrt = m.reverse(zflat, c)
        msg += " | roundtrip %.3e" % float((xrt - x).abs().max())
    print(msg, flush=True)


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    run(1, 2, 1, 1, 64, (4, 4), 8, 1)
    run(1, 2, 2, 2, 64, (4, 4), 8, 1)
    run(2, 2, 2, 2, 128, (4, 4), 8, 1)
    run(2, 2, 2, 2, 128, (4, 4), 8, 2)
    run(3, 3, 2, 1, 256, (4, 4), 16, 0)
    run(4, 2, 3, 3, 512, (4, 8), 16, 1)
    run(4, 2, 3, 3, 512, (4, 8), 16, 2)
    run(2, 2, 2, 2, 128, (4, 4), 8, 1, ddi=True)
    run(6, 2, 2, 2, 2048, (16, 16), 80, 0)
    run(8, 6, 2, 1, 2048, (16, 16), 80, 0)
