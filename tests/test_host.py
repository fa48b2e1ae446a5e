"""CPU tests of the host logic: the packing index tables and the plane algebra (squeeze /
change_order as index bookkeeping) are checked by emulating, in fp64 NumPy, exactly what the
HIP kernels compute from the packed layouts, and comparing with the oracle on the reference's
logical layouts.  No GPU, no libfwn compute calls."""
import numpy as np
import pytest

from oracle import flowavenet_np as onp
from tf_flowavenet_amd import packing as P
from tf_flowavenet_amd import weights as W
from tf_flowavenet_amd.hparams import HParams, default_hparams, hparams8000

from conftest import small_hparams


def wn_pack(params, name, src_k, src_n, weight_norm=True):
    """NumPy twin of fwn_wn_scale + fwn_pack_bf16 (without the bf16 cast)."""
    v = np.asarray(params[name + "/kernel"], np.float64)
    v2 = v.reshape(v.shape[0] * v.shape[1], v.shape[2])
    if weight_norm:
        sc = np.asarray(params[name + "/g"], np.float64) / np.sqrt(np.maximum((v2 ** 2).sum(0), 1e-12))
    else:
        sc = np.ones(v2.shape[1])
    out = np.zeros((len(src_n), len(src_k)))
    for nd, sn in enumerate(src_n):
        if sn < 0:
            continue
        for kd, sk in enumerate(src_k):
            if sk >= 0:
                out[nd, kd] = v2[sk, sn] * sc[sn]
    return out


def to_planes(x):
    """[B,T,1] -> planes[2][B][T/2] (fwn_split_planes)."""
    return np.stack([x[:, 0::2, 0], x[:, 1::2, 0]])


def cplanes_of(cu):
    """upsampled c [B,T,mels] -> cplanes[2][B][T][mels/2] (fwn_upsample_stage output)."""
    half = cu.shape[2] // 2
    return np.stack([cu[:, :, :half], cu[:, :, half:]])


def taps(h, ti, d):
    """rows shifted by -d, 0, +d with zero padding at clip edges; h: [M,256] with M = B*ti."""
    m = h.shape[0]
    out = []
    t = np.arange(m) % ti
    for s in (-d, 0, d):
        sh = np.zeros_like(h)
        ok = (t + s >= 0) & (t + s < ti)
        idx = np.arange(m)[ok]
        sh[idx] = h[idx + s]
        out.append(sh)
    return np.concatenate(out, 1)


def run_emulated_forward(params, hp, x, c):
    """Whole forward on device layouts with packed weights (fp64)."""
    b, t, _ = x.shape
    half = hp.num_mels // 2
    planes = to_planes(x)
    cpl = cplanes_of(onp.upsample(params, c, hp))
    fg, gch = P.gate_row_channel()
    logdet_sum = 0.0
    p = 0
    for i in range(hp.n_block):
        ch = 1 << i
        ti = t // (2 * ch)
        m = b * ti
        cin = half * 2 * ch
        br = P.bitrev_table(i).astype(np.int64)
        for j in range(hp.n_flow):
            fp = W.flow_prefix(i, j)
            wp = fp + "/WaveNet"
            xa = planes[p].reshape(m, ch)
            xb = planes[1 - p].reshape(m, ch)
            ca = cpl[p].reshape(m, cin)
            b_ = np.asarray(params[fp + "/ActNorm/b"], np.float64).reshape(-1)
            l3 = 3.0 * np.asarray(params[fp + "/ActNorm/logs"], np.float64).reshape(-1)
            sh = [b_[r * ch + br] for r in range(2)]
            sc = [np.exp(l3[r * ch + br]) for r in range(2)]
            l3n = [l3[r * ch + br] for r in range(2)]
            ya = (xa + sh[0]) * sc[0]
            # front conv: K = tap*Ch + tau'
            wf = wn_pack(params, wp + "/Conv_front", P.front_src_k(i), np.arange(256))
            a_front = np.zeros((m, wf.shape[1]))
            tt = np.arange(m) % ti
            for tap in range(3):
                ok = (tt + tap - 1 >= 0) & (tt + tap - 1 < ti)
                idx = np.arange(m)[ok]
                a_front[idx, tap * ch:(tap + 1) * ch] = ya[idx + tap - 1]
            h = np.maximum(a_front @ wf.T + np.asarray(params[wp + "/Conv_front/bias"], np.float64), 0.0)
            o_list = []
            for l in range(hp.n_layer):
                rp = "%s/ResBlock_%d" % (wp, l)
                rows = [np.where(fg == s, gch, -1) for s in (0, 1)]
                wd = wn_pack(params, rp + "/Conv_filter", np.arange(768), rows[0]) + \
                    wn_pack(params, rp + "/Conv_gate", np.arange(768), rows[1])
                wc = wn_pack(params, rp + "/filter_conv_c", P.cond_src_k(i, half), rows[0]) + \
                    wn_pack(params, rp + "/gate_conv_c", P.cond_src_k(i, half), rows[1])
                bsum = [np.asarray(params[rp + "/Conv_filter/bias"], np.float64) + np.asarray(params[rp + "/filter_conv_c/bias"], np.float64),
                        np.asarray(params[rp + "/Conv_gate/bias"], np.float64) + np.asarray(params[rp + "/gate_conv_c/bias"], np.float64)]
                bg = np.where(fg == 0, bsum[0][gch], bsum[1][gch])
                ca_pad = np.zeros((m, wc.shape[1]))
                ca_pad[:, :cin] = ca
                pre = taps(h, ti, 3 ** l) @ wd.T + ca_pad @ wc.T + bg
                o = np.zeros((m, 256))
                f_rows = np.where(fg == 0)[0]
                o[:, gch[f_rows]] = np.tanh(pre[:, f_rows]) * onp.sigmoid(pre[:, f_rows + 32])
                o_list.append(o)
                if l + 1 < hp.n_layer:
                    wr = wn_pack(params, rp + "/res_conv", np.arange(256), np.arange(256))
                    h = (h + o @ wr.T + np.asarray(params[rp + "/res_conv/bias"], np.float64)) * np.sqrt(0.5)
            ws = np.concatenate([wn_pack(params, "%s/ResBlock_%d/skip_conv" % (wp, l), np.arange(256), np.arange(256))
                                 for l in range(hp.n_layer)], 1)
            bs = sum(np.asarray(params["%s/ResBlock_%d/skip_conv/bias" % (wp, l)], np.float64) for l in range(hp.n_layer))
            s = np.maximum(np.concatenate(o_list, 1) @ ws.T + bs, 0.0)
            wfin = wn_pack(params, wp + "/Conv_final", np.arange(256), np.arange(256))
            u = np.maximum(s @ wfin.T + np.asarray(params[wp + "/Conv_final/bias"], np.float64), 0.0)
            zsn = P.zero_src_n(i)
            wz = wn_pack(params, wp + "/ZeroConv1d", np.arange(256), zsn, weight_norm=False)
            zb = np.asarray(params[wp + "/ZeroConv1d/bias"], np.float64).reshape(-1)
            zs = np.asarray(params[wp + "/ZeroConv1d/scale"], np.float64).reshape(-1)
            net = u @ wz.T
            yb = (xb + sh[1]) * sc[1]
            new_b = np.empty_like(xb)
            for tau in range(ch):
                pt, jj = divmod(tau, 32)
                nls, nt = pt * 64 + jj, pt * 64 + 32 + jj
                ls = (net[:, nls] + zb[zsn[nls]]) * np.exp(3.0 * zs[zsn[nls]])
                tv = (net[:, nt] + zb[zsn[nt]]) * np.exp(3.0 * zs[zsn[nt]])
                new_b[:, tau] = (yb[:, tau] - tv) * np.exp(-ls)
                logdet_sum += np.sum(l3n[0][tau] + l3n[1][tau] - ls)
            planes[p] = ya.reshape(b, -1)
            planes[1 - p] = new_b.reshape(b, -1)
            p ^= 1
    log_p = np.mean(0.5 * (-np.log(2 * np.pi) - planes ** 2))
    return log_p, logdet_sum / (b * t), planes


@pytest.mark.parametrize("cfg,b,t", [
    (dict(n_block=2, n_flow=2), 2, 64),
    (dict(n_block=3, n_flow=3, num_mels=16), 1, 128),
    (dict(n_block=4, n_flow=2, n_layer=1, num_mels=16), 2, 128),
])
def test_packed_layout_emulation_matches_oracle(cfg, b, t):
    hp = small_hparams(**cfg)
    params = onp.to_f64(W.synthetic_params(hp, 7, actnorm="random"))
    inp = W.synthetic_inputs(hp, b, t)
    x, c = inp["x"].astype(np.float64), inp["c"].astype(np.float64)
    lp0, ld0, z0 = onp.forward(params, x, c, hp)
    lp, ld, planes = run_emulated_forward(params, hp, x, c)
    assert abs(lp - lp0) < 1e-12 and abs(ld - ld0) < 1e-12
    # planes -> canonical squeezed layout of the last block (q*Ch + bitrev(tau'))
    n = hp.n_block
    ch = 1 << (n - 1)
    rows = t // (2 * ch)
    br = P.bitrev_table(n - 1).astype(np.int64)
    z = np.empty((b, rows, 2 * ch))
    for q in range(2):
        z[:, :, q * ch + br] = planes[q].reshape(b, rows, ch)
    if (hp.n_block * hp.n_flow) % 2:   # odd number of swaps: logical = halves exchanged
        z = np.concatenate([z[:, :, ch:], z[:, :, :ch]], 2)
    np.testing.assert_allclose(z, z0, atol=1e-12)


def test_actnorm_table_round_trip():
    rng = np.random.default_rng(3)
    for i in range(5):
        c = 2 << i
        b = rng.standard_normal((1, 1, c)).astype(np.float32)
        logs = (0.1 * rng.standard_normal((1, 1, c))).astype(np.float32)
        an = P.actnorm_table(b, logs, i)
        assert an.shape == (2, 4, c // 2)
        b2, l2 = P.actnorm_from_table(an, i)
        np.testing.assert_allclose(b2, b, rtol=1e-6)
        np.testing.assert_allclose(l2, logs, rtol=1e-5, atol=1e-7)


def test_index_tables_are_permutations():
    for i in range(8):
        ch = 1 << i
        f = P.front_src_k(i)
        assert sorted(f[f >= 0]) == list(range(3 * ch)) and len(f) % 64 == 0
        ck = P.cond_src_k(i, 40)
        assert sorted(ck[ck >= 0]) == list(range(40 * 2 * ch)) and len(ck) % 64 == 0
        z = P.zero_src_n(i)
        assert sorted(z[z >= 0]) == list(range(2 * ch))
    fg, gch = P.gate_row_channel()
    assert sorted(gch[fg == 0]) == list(range(256)) and sorted(gch[fg == 1]) == list(range(256))
    # filter row n' and gate row n'+32 address the same channel (the epilogue pairs them)
    fr = np.where(fg == 0)[0]
    assert np.all(gch[fr] == gch[fr + 32]) and np.all(fg[fr + 32] == 1)


def test_hparams_surface():
    hp = default_hparams()
    # names / defaults of reference hparams.py:6-50
    assert (hp.n_block, hp.n_flow, hp.n_layer, hp.num_mels, hp.hop_size, hp.sample_rate) == (8, 6, 2, 80, 256, 22050)
    assert hp.upsample_scales == [16, 16] and hp.temp == 0.7 and hp.batch_size == 8 and hp.affine and not hp.causality
    assert hp.gin_channels == -1 and hp.max_time_steps == 6400 and hp.tf_random_seed == 75 and hp.scale == 64.0
    h8 = hparams8000()   # reference hparams8000.py
    assert (h8.n_block, h8.hop_size, h8.sample_rate, h8.upsample_scales, h8.max_time_steps, h8.n_fft, h8.fmax) == \
        (5, 96, 8000, [8, 12], 2320, 512, 4000)
    with pytest.raises(AttributeError):
        hp.replace(not_a_param=1)
    with pytest.raises(AttributeError):
        _ = hp.not_a_param
    assert isinstance(hp, HParams) and "n_block" in hp.values()


def test_synthetic_inputs_contract():
    hp = default_hparams()
    with pytest.raises(ValueError):
        W.synthetic_inputs(hp, 1, 16000)        # not a multiple of hop 256
    inp = W.synthetic_inputs(hp, 2, 512)
    assert inp["x"].shape == (2, 512, 1) and inp["c"].shape == (2, 2, 80) and inp["z"].shape == (2, 512, 1)
    assert np.abs(inp["x"]).max() <= 0.999 and inp["c"].min() >= 0 and inp["c"].max() < 1


def test_wav_writer_and_checkpoint_loader(tmp_path):
    import wave
    from tf_flowavenet_amd import synthesize as S
    audio = np.array([0.0, 0.5, -0.5, 1.5, -1.5, 0.25])
    S.write_wav(str(tmp_path / "x.wav"), audio, 8000)
    with wave.open(str(tmp_path / "x.wav")) as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 8000, 6)
        pcm = np.frombuffer(w.readframes(6), dtype="<i2")
    np.testing.assert_array_equal(pcm, [0, 16384, -16384, 32767, -32767, 8192])   # clipped to +-1
    with pytest.raises(FileNotFoundError):
        S.load_checkpoint(str(tmp_path))
    np.savez(tmp_path / "m.npz", a=np.arange(3))
    assert list(S.load_checkpoint(str(tmp_path))) == ["a"]


# The reference's variable names for n_block=1, n_flow=1, n_layer=2, written out by hand from its scopes:
# train.py:53 'vocoder' / model.py:283 'FloWaveNet' / :297 'Block_%d' / :218 'Flow_%d' / :13 'ActNorm' (b, logs :57,71) /
# :110 'AffineCoupling' / modules.py:136 'WaveNet' / :144 'Conv_front' / :152 'ResBlock_%d_%d' / :73-74 'Conv_filter',
# 'Conv_gate' / :158 'Conv_final' / :41 'ZeroConv1d' (+ 'scale' :49); every un-named tf.layers layer opens
# variable_scope(default_name='conv1d' | 'conv2d_transpose') at its FIRST CALL (modules.py:113-127: filter_conv_c,
# gate_conv_c, res_conv, skip_conv; model.py:401-402: the two up-sampling convs), variables 'kernel', 'wn/g', 'bias'
# (convolutional.py:64-87,167-192).
_P = "vocoder/FloWaveNet/"
_W = _P + "Block_0/Flow_0/AffineCoupling/WaveNet/"
_O = "Block_0/Flow_0/WaveNet/"
REFERENCE_NAMES = {
    _P + "conv2d_transpose/kernel:0": "upsample_0/kernel",
    _P + "conv2d_transpose/wn/g:0": "upsample_0/g",
    _P + "conv2d_transpose/bias:0": "upsample_0/bias",
    _P + "conv2d_transpose_1/kernel:0": "upsample_1/kernel",
    _P + "conv2d_transpose_1/wn/g:0": "upsample_1/g",
    _P + "conv2d_transpose_1/bias:0": "upsample_1/bias",
    _P + "Block_0/Flow_0/ActNorm/b:0": "Block_0/Flow_0/ActNorm/b",
    _P + "Block_0/Flow_0/ActNorm/logs:0": "Block_0/Flow_0/ActNorm/logs",
    _W + "Conv_front/conv1d/kernel:0": _O + "Conv_front/kernel",
    _W + "Conv_front/conv1d/wn/g:0": _O + "Conv_front/g",
    _W + "Conv_front/conv1d/bias:0": _O + "Conv_front/bias",
    _W + "ResBlock_0_0/Conv_filter/conv1d/kernel:0": _O + "ResBlock_0/Conv_filter/kernel",
    _W + "ResBlock_0_0/Conv_filter/conv1d/wn/g:0": _O + "ResBlock_0/Conv_filter/g",
    _W + "ResBlock_0_0/Conv_filter/conv1d/bias:0": _O + "ResBlock_0/Conv_filter/bias",
    _W + "ResBlock_0_0/Conv_gate/conv1d/kernel:0": _O + "ResBlock_0/Conv_gate/kernel",
    _W + "ResBlock_0_0/Conv_gate/conv1d/wn/g:0": _O + "ResBlock_0/Conv_gate/g",
    _W + "ResBlock_0_0/Conv_gate/conv1d/bias:0": _O + "ResBlock_0/Conv_gate/bias",
    _W + "ResBlock_0_0/conv1d/kernel:0": _O + "ResBlock_0/filter_conv_c/kernel",
    _W + "ResBlock_0_0/conv1d/wn/g:0": _O + "ResBlock_0/filter_conv_c/g",
    _W + "ResBlock_0_0/conv1d/bias:0": _O + "ResBlock_0/filter_conv_c/bias",
    _W + "ResBlock_0_0/conv1d_1/kernel:0": _O + "ResBlock_0/gate_conv_c/kernel",
    _W + "ResBlock_0_0/conv1d_1/wn/g:0": _O + "ResBlock_0/gate_conv_c/g",
    _W + "ResBlock_0_0/conv1d_1/bias:0": _O + "ResBlock_0/gate_conv_c/bias",
    _W + "ResBlock_0_0/conv1d_2/kernel:0": _O + "ResBlock_0/res_conv/kernel",
    _W + "ResBlock_0_0/conv1d_2/wn/g:0": _O + "ResBlock_0/res_conv/g",
    _W + "ResBlock_0_0/conv1d_2/bias:0": _O + "ResBlock_0/res_conv/bias",
    _W + "ResBlock_0_0/conv1d_3/kernel:0": _O + "ResBlock_0/skip_conv/kernel",
    _W + "ResBlock_0_0/conv1d_3/wn/g:0": _O + "ResBlock_0/skip_conv/g",
    _W + "ResBlock_0_0/conv1d_3/bias:0": _O + "ResBlock_0/skip_conv/bias",
    _W + "ResBlock_0_1/Conv_filter/conv1d/kernel:0": _O + "ResBlock_1/Conv_filter/kernel",
    _W + "ResBlock_0_1/Conv_filter/conv1d/wn/g:0": _O + "ResBlock_1/Conv_filter/g",
    _W + "ResBlock_0_1/Conv_filter/conv1d/bias:0": _O + "ResBlock_1/Conv_filter/bias",
    _W + "ResBlock_0_1/Conv_gate/conv1d/kernel:0": _O + "ResBlock_1/Conv_gate/kernel",
    _W + "ResBlock_0_1/Conv_gate/conv1d/wn/g:0": _O + "ResBlock_1/Conv_gate/g",
    _W + "ResBlock_0_1/Conv_gate/conv1d/bias:0": _O + "ResBlock_1/Conv_gate/bias",
    _W + "ResBlock_0_1/conv1d/kernel:0": _O + "ResBlock_1/filter_conv_c/kernel",
    _W + "ResBlock_0_1/conv1d/wn/g:0": _O + "ResBlock_1/filter_conv_c/g",
    _W + "ResBlock_0_1/conv1d/bias:0": _O + "ResBlock_1/filter_conv_c/bias",
    _W + "ResBlock_0_1/conv1d_1/kernel:0": _O + "ResBlock_1/gate_conv_c/kernel",
    _W + "ResBlock_0_1/conv1d_1/wn/g:0": _O + "ResBlock_1/gate_conv_c/g",
    _W + "ResBlock_0_1/conv1d_1/bias:0": _O + "ResBlock_1/gate_conv_c/bias",
    _W + "ResBlock_0_1/conv1d_2/kernel:0": _O + "ResBlock_1/res_conv/kernel",
    _W + "ResBlock_0_1/conv1d_2/wn/g:0": _O + "ResBlock_1/res_conv/g",
    _W + "ResBlock_0_1/conv1d_2/bias:0": _O + "ResBlock_1/res_conv/bias",
    _W + "ResBlock_0_1/conv1d_3/kernel:0": _O + "ResBlock_1/skip_conv/kernel",
    _W + "ResBlock_0_1/conv1d_3/wn/g:0": _O + "ResBlock_1/skip_conv/g",
    _W + "ResBlock_0_1/conv1d_3/bias:0": _O + "ResBlock_1/skip_conv/bias",
    _W + "Conv_final/conv1d/kernel:0": _O + "Conv_final/kernel",
    _W + "Conv_final/conv1d/wn/g:0": _O + "Conv_final/g",
    _W + "Conv_final/conv1d/bias:0": _O + "Conv_final/bias",
    _W + "ZeroConv1d/conv1d/kernel:0": _O + "ZeroConv1d/kernel",
    _W + "ZeroConv1d/conv1d/bias:0": _O + "ZeroConv1d/bias",
    _W + "ZeroConv1d/scale:0": _O + "ZeroConv1d/scale",
}


def test_reference_variable_names_map_onto_parameter_names():
    """A dump of a reference checkpoint (names written out literally above, NOT derived from this package) maps onto
    exactly the parameters load_params wants - together with the slots a tf.train.Saver(global_variables) file also
    holds (train.py:190)."""
    from tf_flowavenet_amd.weights import from_reference_names, param_shapes, synthetic_params, to_reference_names
    hp = small_hparams(n_block=1, n_flow=1, n_layer=2)
    p = synthetic_params(hp, 3)
    assert sorted(REFERENCE_NAMES.values()) == sorted(param_shapes(hp))
    dumped = {tf_name: p[ours] for tf_name, ours in REFERENCE_NAMES.items()}
    dumped[_P + "Block_0/Flow_0/ActNorm/b/Adam:0"] = np.zeros(1)
    dumped[_W + "ResBlock_0_0/conv1d_2/wn/g/Adam_1:0"] = np.zeros(1)
    dumped["global_step:0"] = np.zeros(())
    dumped["beta1_power:0"] = np.zeros(())
    dumped["beta2_power:0"] = np.zeros(())
    dumped[_W + "Conv_front/conv1d/kernel/fp16_cast:0"] = np.zeros(1)
    dumped[_P + "speaker_embeddings:0"] = np.zeros((2, 4))
    got = from_reference_names(dumped)
    assert sorted(got) == sorted(param_shapes(hp))
    for tf_name, ours in REFERENCE_NAMES.items():
        assert got[ours] is p[ours], (tf_name, ours)
    # the inverse map regenerates the literal list
    assert sorted(k + ":0" for k in to_reference_names(p)) == sorted(REFERENCE_NAMES)
    # synthesize.py builds its graph under the same 'vocoder' scope (synthesize.py:11); a dump without it, or one
    # already in this package's names, also loads
    assert sorted(from_reference_names({k[len("vocoder/"):] if k.startswith("vocoder/") else k: v for k, v in dumped.items()})) == sorted(param_shapes(hp))
    assert from_reference_names(p) == dict(p)


def test_reference_name_map_survives_construction_order_numbering():
    """If the four bare 1x1 convs of a ResBlock were numbered in construction order instead (res, skip, filter_c,
    gate_c: modules.py:75-94), kernel shapes + order inside each pair still identify them."""
    from tf_flowavenet_amd.weights import from_reference_names, synthetic_params
    hp = small_hparams(n_block=1, n_flow=1, n_layer=2)
    p = synthetic_params(hp, 3)
    alt = {"conv1d": "conv1d_2", "conv1d_1": "conv1d_3", "conv1d_2": "conv1d", "conv1d_3": "conv1d_1"}
    dumped = {}
    for tf_name, ours in REFERENCE_NAMES.items():
        parts = tf_name.split("/")
        if parts[-3].startswith("ResBlock_0_") and parts[-2] in alt:            # .../ResBlock_0_n/conv1d_k/kernel
            parts[-2] = alt[parts[-2]]
        elif parts[-4].startswith("ResBlock_0_") and parts[-3] in alt and parts[-2] == "wn":
            parts[-3] = alt[parts[-3]]
        dumped["/".join(parts)] = p[ours]
    got = from_reference_names(dumped)
    assert all(got[k] is p[k] for k in p)
    with pytest.raises(KeyError):
        from_reference_names({_W + "ResBlock_1_0/conv1d/kernel:0": np.zeros((1, 8, 256))})


def test_load_params_accepts_a_reference_named_dump(tmp_path):
    """synthesize.load_checkpoint on an .npz holding the reference's variable names."""
    from tf_flowavenet_amd import synthesize as S
    from tf_flowavenet_amd.weights import param_shapes, synthetic_params, to_reference_names
    hp = small_hparams(n_block=2, n_flow=2)
    p = synthetic_params(hp, 4)
    np.savez(tmp_path / "dump.npz", **{k + ":0": v for k, v in to_reference_names(p).items()})
    got = S.load_checkpoint(str(tmp_path))
    assert sorted(got) == sorted(param_shapes(hp)) and all(np.array_equal(got[k], p[k]) for k in p)


def test_checkpoints_survive_a_crash_mid_write_and_order_by_step(tmp_path):
    import os
    """ADVICE r1: the temp file of save_checkpoint must not match the restore globs; restore takes the highest
    STEP (not mtime) and skips a truncated file (train.py:190,214-218,251-252 Saver semantics: latest checkpoint)."""
    import torch
    from tf_flowavenet_amd import synthesize as S
    from tf_flowavenet_amd import train as T

    class Opt:
        def __init__(self, fill):
            self.w = torch.full((6,), float(fill))
            self.m, self.v, self.global_step = torch.zeros(6), torch.ones(6), int(fill)

        def master_views(self):
            return {"a/kernel": self.w[:4].view(2, 2), "a/bias": self.w[4:]}

    tr = lambda fill: type("Tr", (), {"opt": Opt(fill)})()
    base = str(tmp_path / "flowavenet_model.ckpt")
    T.save_checkpoint(base + "-200.npz", tr(200))
    T.save_checkpoint(base + "-1000.npz", tr(1000))
    os.utime(base + "-1000.npz", (1, 1))                       # older mtime than step 200: the step number decides
    assert [os.path.basename(p) for p in T.checkpoint_files(str(tmp_path))] == ["flowavenet_model.ckpt-200.npz",
                                                                                "flowavenet_model.ckpt-1000.npz"]
    assert not [f for f in os.listdir(tmp_path) if "tmp" in f]
    fresh = tr(0)
    assert T.restore_checkpoint(str(tmp_path), fresh) == 1000 and float(fresh.opt.w[0]) == 1000.0
    # a crash while writing step 3000 leaves either a '.tmp' (ignored) or, with the old naming, a truncated .npz
    with open(base + "-3000.npz.tmp", "wb") as f:
        f.write(b"PK\x03\x04 truncated")
    with open(base + "-3000.npz", "wb") as f:
        f.write(b"PK\x03\x04 truncated")
    fresh = tr(0)
    assert T.restore_checkpoint(str(tmp_path), fresh) == 1000
    assert float(S.load_checkpoint(str(tmp_path))["a/bias"][0]) == 1000.0
    assert T.restore_checkpoint(str(tmp_path / "nothing"), fresh) is None


def test_synthesize_splits_batches_under_the_per_call_addressing_limit():
    """B = 40 x 10 s is rejected by the C layer (2 GiB per activation buffer); the CLI must pick the batch
    itself.  n_layer * (B T / 2) * 512 < 2^31 with the default hparams -> 4 194 303 samples per call."""
    from tf_flowavenet_amd import synthesize as S
    from tf_flowavenet_amd.hparams import default_hparams
    hp = default_hparams()
    assert S.max_clips_per_call(hp, 220672) == 19          # 10 s clips: 19 per call, 20 would exceed the limit
    assert hp.n_layer * (19 * 220672 // 2) * 512 < 1 << 31 <= hp.n_layer * (20 * 220672 // 2) * 512
    assert S.max_clips_per_call(hp, 16128) == 260 and S.max_clips_per_call(hp, 5_000_000) == 0


def test_restore_raises_on_a_checkpoint_of_another_model_instead_of_starting_over(tmp_path):
    """a checkpoint that READS but does not fit the model (other hparams: missing entry, other size) must
    raise - skipping it as 'unreadable' would restart at step 0 and overwrite the run's files.  Only I/O / container
    errors are skipped."""
    import os
    import torch
    from tf_flowavenet_amd import train as T

    class _Opt:
        def __init__(self, n=2):
            self.w, self.m, self.v, self.global_step, self.group = torch.zeros(n), torch.zeros(n), torch.zeros(n), 0, None

        def master_views(self):
            return {"p": self.w}

    class _Tr:
        def __init__(self, n=2):
            self.opt = _Opt(n)

    tr = _Tr()
    tr.opt.w[:] = 7.0
    tr.opt.global_step = 300
    T.save_checkpoint(str(tmp_path / "flowavenet_model.ckpt-300.npz"), tr)
    with pytest.raises(ValueError, match="shape"):
        T.restore_checkpoint(str(tmp_path), _Tr(n=3))            # same names, another size
    np.savez(str(tmp_path / "flowavenet_model.ckpt-400.npz"), q=np.zeros(2))
    with pytest.raises(KeyError, match="does not belong"):
        T.restore_checkpoint(str(tmp_path), _Tr())               # a readable .npz without this model's entries
    os.remove(str(tmp_path / "flowavenet_model.ckpt-400.npz"))
    with open(str(tmp_path / "flowavenet_model.ckpt-500.npz"), "wb") as f:
        f.write(b"PK\x03\x04 torn by a crash")
    fresh = _Tr()
    assert T.restore_checkpoint(str(tmp_path), fresh) == 300 and float(fresh.opt.w[0]) == 7.0     # the torn file is skipped


def test_dataset_keeps_a_bounded_number_of_file_descriptors_open(tmp_path):
    """every numpy memmap pins a descriptor; the LRU of open utterances must stay far below the usual soft
    RLIMIT_NOFILE of 1024 and release descriptors when it evicts."""
    import os
    from tf_flowavenet_amd import train as T
    from tf_flowavenet_amd.hparams import default_hparams
    hp = default_hparams().replace(max_time_steps=512, hop_size=256, batch_size=4, test_size=2)
    os.makedirs(str(tmp_path / "audios"))
    os.makedirs(str(tmp_path / "mels"))
    n = 40
    with open(str(tmp_path / "train.txt"), "w") as f:
        for k in range(n):
            np.save(str(tmp_path / "audios" / ("a%d.npy" % k)), np.zeros(4 * 256, np.float32))
            np.save(str(tmp_path / "mels" / ("m%d.npy" % k)), np.zeros((4, hp.num_mels), np.float32))
            f.write("a%d.npy|m%d.npy|%d|0|x\n" % (k, k, 4 * 256))
    assert 2 * T.Dataset.MAX_OPEN <= 256
    old = T.Dataset.MAX_OPEN
    T.Dataset.MAX_OPEN = 4
    try:
        ds = T.Dataset(str(tmp_path / "train.txt"), hp, seed=1)
        nfd = lambda: len(os.listdir("/proc/self/fd"))
        for _ in range(3):
            ds.next_train()
        base = nfd()
        for _ in range(40):
            mels, audios = ds.next_train()
            ds.eval_sample()
        assert mels.shape == (4, 2, hp.num_mels) and audios.shape == (4, 512)
        assert nfd() <= base + 2 * 4 + 4, (base, nfd())          # bounded by the LRU, not by the utterances touched
        assert len(ds._cache) <= 4 + 1
    finally:
        T.Dataset.MAX_OPEN = old
