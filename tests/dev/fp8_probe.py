"""Developer probe (CPU, uses the oracle - test infrastructure): what would an fp8 (e4m3) dilated-conv GEMM cost in
accuracy?  BASELINE configs[4] names an "fp8 MFMA dilated-conv im2col-GEMM path"; north_star wants forward log-p within
1e-3 relative of the reference.  The oracle's Conv_filter / Conv_gate (the dilated convs only; everything else
stays fp64) are run with their inputs and weight-normed kernels rounded to e4m3 (per-output-channel weight scale,
per-tensor activation scale, fp32-like accumulation = exact here), and log_p / logdet are compared with the
unquantised oracle and with the same rounding to bf16 (what the HIP path does).

    python tests/dev/fp8_probe.py [n_block n_flow T]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import flowavenet_np as onp
from tf_flowavenet_amd import weights as W
from tf_flowavenet_amd.hparams import default_hparams


def round_mantissa(x, mbits, emin, vmax):
    """Round to a float format with `mbits` explicit mantissa bits, minimum normal exponent `emin`, max `vmax`."""
    x = np.clip(x, -vmax, vmax)
    ax = np.abs(x)
    e = np.floor(np.log2(np.maximum(ax, 2.0 ** emin)))
    step = 2.0 ** (e - mbits)
    return np.round(x / step) * step


def e4m3(x):
    return round_mantissa(x, 3, -6, 448.0)


def bf16(x):
    return round_mantissa(x, 7, -126, 3.38e38)


def run(hp, p, x, c, quant):
    orig = onp.conv_layer
    def conv_layer(pp, prefix, xx, kernel_size=3, dilation=1, causal=False):
        if quant is None or not (prefix.endswith("Conv_filter") or prefix.endswith("Conv_gate")):
            return orig(pp, prefix, xx, kernel_size, dilation, causal)
        w = onp.wn_kernel_1d(pp, prefix)                       # [k][Cin][Cout]
        if quant in ("e4m3", "e4m3-static"):
            if quant == "e4m3":       # per-output-channel weight scale, per-tensor activation scale (needs a max pass)
                ws = np.abs(w).max(axis=(0, 1), keepdims=True) / 448.0
                xs = max(np.abs(xx).max() / 448.0, 1e-30)
            else:                     # one weight scale per matrix, activations stored as they are (no reduction)
                ws, xs = np.abs(w).max() / 448.0, 1.0
            wq, xq = e4m3(w / ws) * ws, e4m3(xx / xs) * xs
        else:
            wq, xq = quant(w), quant(xx)
        pad = dilation * (kernel_size - 1) // 2
        xp = np.pad(xq, ((0, 0), (pad, pad), (0, 0)))
        return onp.conv1d_valid(xp, wq, pp[prefix + "/bias"], dilation)
    onp.conv_layer = conv_layer
    try:
        return onp.forward(p, x, c, hp)
    finally:
        onp.conv_layer = orig


if __name__ == "__main__":
    nb, nf, t = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 6, 2048)
    hp = default_hparams().replace(n_block=nb, n_flow=nf)
    p = onp.to_f64(W.synthetic_params(hp, 1234))
    inp = W.synthetic_inputs(hp, 1, t)
    x, c = inp["x"].astype(np.float64), inp["c"].astype(np.float64)
    onp.forward(p, x, c, hp, init=True)                       # ActNorm DDI, as every benchmark does
    ref = run(hp, p, x, c, None)
    for name, q in (("bf16", bf16), ("fp8 e4m3 (scaled)", "e4m3"), ("fp8 e4m3 (static)", "e4m3-static")):
        lp, ld = run(hp, p, x, c, q)[:2]
        print("%-18s dilated convs: log_p %.6f (ref %.6f, rel err %.2e)   logdet %.6f (ref %.6f, abs err %.2e, rel %.2e)" % (
            name, lp, ref[0], abs(lp - ref[0]) / abs(ref[0]), ld, ref[1], abs(ld - ref[1]), abs(ld - ref[1]) / max(abs(ref[1]), 1e-12)))
