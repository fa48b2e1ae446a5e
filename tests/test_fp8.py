"""fp8 (OCP e4m3) dilated-conv path, BASELINE configs[4] (-m gpu): the cast / packing kernels bit for bit against
torch's float8_e4m3fn, the fp8 gate kernel against an fp64 reference computed from the SAME quantised operands (so the
test isolates the kernel: fragment layout, scale operand, tap shifts, clip edges), and the whole model with
``gate_fp8=True`` against the fp64 oracle's golden outputs at north_star's tolerance (log_p within 1e-3 relative)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import flowavenet_np as onp
from tf_flowavenet_amd import _lib, packing
from tf_flowavenet_amd import weights as W
from tf_flowavenet_amd.hparams import default_hparams, hparams8000
from tf_flowavenet_amd.model import FloWaveNet, z_planes_to_squeezed

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def e4m3_bytes(t):
    return t.to(torch.float8_e4m3fn).view(torch.uint8)


def e4m3_value(u8):
    return u8.view(torch.float8_e4m3fn).to(torch.float64)


def test_cast_kernel_matches_torch_float8_bit_for_bit():
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    # magnitudes from subnormal (2^-9) to the format's maximum, both signs, exact ties included
    x = torch.cat([torch.randn(40000, generator=g) * s for s in (0.002, 0.05, 1.0, 20.0, 150.0)] +
                  [torch.arange(-448, 449, dtype=torch.float32) * 0.5, torch.tensor([0.0, -0.0, 2.0 ** -9, 2.0 ** -10, 3 * 2.0 ** -10])])
    x = x.clamp(-448, 448).to(torch.bfloat16).cuda()
    out = torch.empty(x.numel(), dtype=torch.uint8, device="cuda")
    _lib.check(lib.fwn_cast_e4m3(x.data_ptr(), out.data_ptr(), x.numel(), torch.cuda.current_stream().cuda_stream))
    assert torch.equal(out, e4m3_bytes(x.float()))
    # saturation: beyond +-448 the kernel clamps (an overflow would be NaN in e4m3fn, which has no infinity)
    big = torch.tensor([500.0, -1e4, 449.0, 3e38], dtype=torch.bfloat16, device="cuda")
    o2 = torch.empty(4, dtype=torch.uint8, device="cuda")
    _lib.check(lib.fwn_cast_e4m3(big.data_ptr(), o2.data_ptr(), 4, torch.cuda.current_stream().cuda_stream))
    assert e4m3_value(o2).tolist() == [448.0, -448.0, 448.0, 448.0]


def test_pack_e4m3_chooses_the_power_of_two_scale_and_rounds_like_torch():
    lib = _lib.load()
    rng = np.random.default_rng(3)
    st = torch.cuda.current_stream().cuda_stream
    for amp in (0.03, 1.0, 700.0):
        v = (rng.standard_normal((72, 64)) * amp).astype(np.float32)
        g = rng.uniform(0.5, 1.5, 64).astype(np.float32)
        src_k = np.concatenate([rng.permutation(72), [-1] * 8]).astype(np.int32)
        src_n = rng.permutation(64).astype(np.int32)
        src_n[3] = -1
        vd, gd = torch.from_numpy(v).cuda(), torch.from_numpy(g).cuda()
        kd, nd = torch.from_numpy(src_k).cuda(), torch.from_numpy(src_n).cuda()
        scale = torch.empty(64, device="cuda")
        amax = torch.zeros(1, device="cuda")
        exp = torch.zeros(1, dtype=torch.int32, device="cuda")
        out = torch.full((64, 80), 7, dtype=torch.uint8, device="cuda")
        mul = -2.885
        _lib.check(lib.fwn_wn_scale(vd.data_ptr(), gd.data_ptr(), 72, 64, scale.data_ptr(), st))
        _lib.check(lib.fwn_wn_absmax(vd.data_ptr(), scale.data_ptr(), 72, 64, mul, amax.data_ptr(), st))
        _lib.check(lib.fwn_pack_e4m3(vd.data_ptr(), scale.data_ptr(), kd.data_ptr(), nd.data_ptr(), 64, 80, 64, 80, mul,
                                     amax.data_ptr(), out.data_ptr(), exp.data_ptr(), st))
        w = (vd * np.float32(mul)) * scale[None, :]               # the kernel's fp32 order: v * mul, then * scale
        assert float(amax) == float(w.abs().max())
        e = int(exp)
        assert float(amax) * 2.0 ** e <= 448.0 < float(amax) * 2.0 ** (e + 1)
        want = torch.full((64, 80), 7, dtype=torch.uint8, device="cuda")
        for n in range(64):
            if src_n[n] < 0:
                continue
            row = torch.zeros(80, device="cuda")
            row[:72] = w[torch.from_numpy(src_k[:72].astype(np.int64)).cuda(), int(src_n[n])] * 2.0 ** e
            want[n] = e4m3_bytes(row)
        assert torch.equal(out, want)


@pytest.fixture(scope="module")
def fp8_model():
    hp = default_hparams()
    model = FloWaveNet(hp, init=True, gate_fp8=True).load_params(W.synthetic_params(hp, 1234))
    inp = W.synthetic_inputs(hp, 8, 16128)
    x, c, z = (torch.from_numpy(inp[k]).cuda() for k in ("x", "c", "z"))
    lp, ld, zp = model.forward(x, c, return_z=True)          # DDI on the first batch
    return hp, model, x, c, z, (float(lp), float(ld), zp)


@pytest.mark.parametrize("layer", [0, 1])
@pytest.mark.parametrize("b,ti", [(13, 1000), (26, 1000)])
def test_fp8_gate_kernel_matches_a_reference_on_the_same_quantised_operands(fp8_model, b, ti, layer):
    """fwn_gate_fp8 (block 0, flow 0; dilation 1 and 3; M = 13000 -> 256 x 128 tiles, 26000 -> 256 x 256 tiles, clip
    edges inside every tile): tanh(f) sigmoid(g) with f, g computed in fp64 from the e4m3 bytes the kernel reads
    (h8 and the packed Wd8 with its exponent) and the bf16 conditioning weights."""
    hp, model, *_ = fp8_model
    lib = _lib.load()
    pm = model._packed
    d = pm.flow_descs[0]
    m, half = b * ti, hp.num_mels // 2
    assert lib.fwn_gate_fp8_supported(m, layer) == 1 and lib.fwn_gate_fp8_supported(4000, layer) == 0
    rng = np.random.default_rng(b * 10 + layer)
    h = torch.from_numpy(rng.standard_normal((m, 256)).astype(np.float32) * 0.7).cuda().to(torch.bfloat16)
    ca = torch.from_numpy(rng.random((m, d.cin)).astype(np.float32)).cuda().to(torch.bfloat16)
    h8 = torch.empty(m, 256, dtype=torch.uint8, device="cuda")
    o = torch.empty(m, 256, device="cuda", dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.fwn_cast_e4m3(h.data_ptr(), h8.data_ptr(), m * 256, st))
    _lib.check(lib.fwn_gate_fp8(C.byref(d), layer, h8.data_ptr(), ca.data_ptr(), o.data_ptr(), m, ti, st), "fwn_gate_fp8")
    # reference pre-activations in the kernel's packed-N order: u[n'] = sum_tap h8[t + (tap-1) dil] . W8[n'][tap] / 2^e + cond + bias
    dil = 3 ** layer
    w8 = e4m3_value(pm.wd8[layer]).cpu().numpy() * 2.0 ** (-int(d.wd8_exp[layer]))          # [512][768]
    hq = e4m3_value(h8).cpu().numpy().reshape(b, ti, 256)
    u = np.zeros((b, ti, 512))
    for tap in range(3):
        sh = (tap - 1) * dil
        src = np.zeros_like(hq)
        lo, hi = max(0, -sh), min(ti, ti - sh)
        src[:, lo:hi] = hq[:, lo + sh:hi + sh]
        u += src @ w8[:, tap * 256:(tap + 1) * 256].T
    # conditioning + bias from the oracle's weights, scaled / ordered like the packed rows (packing.GATE_MUL, gate_row_channel)
    p64 = onp.to_f64(W.synthetic_params(hp, 1234))
    fg, gch = packing.gate_row_channel()
    src_k = packing.cond_src_k(0, half)[:d.cin]
    c_log = np.empty((b, ti, d.cin))
    c_log[:, :, src_k] = ca.float().cpu().numpy().astype(np.float64).reshape(b, ti, d.cin)
    rp = W.flow_prefix(0, 0) + "/WaveNet/ResBlock_%d" % layer
    zero = np.zeros((b, ti, 256))
    cf = onp.conv1x1(p64, rp + "/filter_conv_c", c_log) + onp.conv_layer(p64, rp + "/Conv_filter", zero, 3, dil)   # cond + both biases
    cg = onp.conv1x1(p64, rp + "/gate_conv_c", c_log) + onp.conv_layer(p64, rp + "/Conv_gate", zero, 3, dil)
    f = u[:, :, fg == 0][:, :, np.argsort(gch[fg == 0])] / packing.GATE_MUL[0] + cf
    g = u[:, :, fg == 1][:, :, np.argsort(gch[fg == 1])] / packing.GATE_MUL[1] + cg
    want = (np.tanh(f) * onp.sigmoid(g)).reshape(m, 256)
    err = np.abs(o.float().cpu().numpy() - want)
    # remaining error: bf16 conditioning weights, fp32 accumulation, bf16 output (half an ulp = 2e-3 near 1)
    assert err.max() < 2e-2 and err.mean() < 1.5e-3, (err.max(), err.mean(), np.unravel_index(err.argmax(), err.shape))


def test_full_model_with_fp8_gates_stays_inside_the_log_p_tolerance(fp8_model):
    """BASELINE configs[4] on the 22.05 kHz model at the bench workload (B=8, T=16128: the gates of blocks 0-2 run in
    fp8): log_p within 1e-3 relative and logdet within 1e-3 of the fp64 oracle (north_star), latent and waveform within
    looser per-sample bounds than the bf16 path (e4m3 carries 3 mantissa bits), and the path really is the fp8 one."""
    hp, model, x, c, z, (lp, ld, zp) = fp8_model
    g = np.load(os.path.join(GOLDEN, "full_b8f6_B8_T16128.npz"))
    assert abs(lp - float(g["log_p"])) <= 1e-3 * abs(float(g["log_p"])), (lp, float(g["log_p"]))
    assert abs(ld - float(g["logdet"])) <= 1e-3 * max(1.0, abs(float(g["logdet"]))), (ld, float(g["logdet"]))
    zz = z_planes_to_squeezed(zp, hp.n_block, hp.n_flow).cpu().numpy()
    dz = np.abs(zz - g["z"].astype(np.float32))
    assert dz.max() < 0.15 and dz.mean() < 1.2e-2, (dz.max(), dz.mean())        # measured 0.092 / 7.8e-3
    wav = model.reverse(z, c).cpu().numpy()
    dw = np.abs(wav - g["x_rev"].astype(np.float32))
    assert dw.max() < 5e-2 and dw.mean() < 4e-3, (dw.max(), dw.mean())
    # the bf16 model on the same inputs differs in the bits (the fp8 kernels ran) and is reproduced run to run
    ref = FloWaveNet(hp, init=True).load_params(W.synthetic_params(hp, 1234))
    lp_b, ld_b, zp_b = ref.forward(x, c, return_z=True)
    assert not torch.equal(zp_b, zp)
    model2 = FloWaveNet(hp, init=True, gate_fp8=True).load_params(W.synthetic_params(hp, 1234))
    _, _, zp2 = model2.forward(x, c, return_z=True)
    assert torch.equal(zp2, zp)
    # round trip through the fp8 model: decode(encode(x)) ~ x
    zf = torch.empty_like(x)
    zf[:, 0::2, 0], zf[:, 1::2, 0] = zp[0], zp[1]
    xr = model.reverse(zf, c)
    assert float((xr - x).abs().max()) < 8e-2 and float((xr - x).abs().mean()) < 4e-3


def test_8khz_model_with_fp8_gates_matches_golden():
    """BASELINE configs[4]: hparams8000 (n_block=5, hop 96, up-sampling [8, 12]) with the fp8 gate path
    (hparams.gate_fp8), B=2, T=16128: block 0 (M = 16128) takes the 256 x 128 fp8 tiles, the rest bf16."""
    hp = hparams8000().replace(gate_fp8=True)
    g = np.load(os.path.join(GOLDEN, "hp8000_b5f6_B2_T16128.npz"))
    model = FloWaveNet(hp, init=True).load_params(W.synthetic_params(hp, 1234))
    assert model._packed.model_desc.gate_fp8 == 1
    inp = W.synthetic_inputs(hp, 2, 16128)
    c = torch.from_numpy(inp["c"]).cuda()
    lp, ld, zp = model.forward(torch.from_numpy(inp["x"]).cuda(), c, return_z=True)
    assert abs(float(lp) - float(g["log_p"])) <= 1e-3 * abs(float(g["log_p"]))
    assert abs(float(ld) - float(g["logdet"])) <= 1e-3 * max(1.0, abs(float(g["logdet"])))
    dz = np.abs(z_planes_to_squeezed(zp, hp.n_block, hp.n_flow).cpu().numpy() - g["z"].astype(np.float32))
    assert dz.max() < 0.15 and dz.mean() < 1.2e-2, (dz.max(), dz.mean())
    wav = model.reverse(torch.from_numpy(inp["z"]).cuda(), c).cpu().numpy()
    dw = np.abs(wav - g["x_rev"].astype(np.float32))
    assert dw.max() < 5e-2 and dw.mean() < 4e-3, (dw.max(), dw.mean())


def test_fp8_passes_on_concurrent_streams_reproduce_the_serial_result(fp8_model):
    """The fp8 kernels beside each other and beside the bf16 ones on four HIP streams (bench.py's lanes): every
    overlapped pass equals the one-stream pass bit for bit."""
    hp, model, x, c, z, _ = fp8_model
    ref_wav = model.reverse(z, c).clone()
    ref_nll = torch.stack(model.forward(x, c)).clone()
    torch.cuda.synchronize()
    lanes = [torch.cuda.Stream() for _ in range(4)]
    outs = []
    for k in range(12):
        with torch.cuda.stream(lanes[k % 4]):
            outs.append(("inv", model.reverse(z, c).clone()) if k % 2 else ("fwd", torch.stack(model.forward(x, c)).clone()))
    torch.cuda.synchronize()
    for kind, got in outs:
        assert torch.equal(got, ref_wav if kind == "inv" else ref_nll), kind
