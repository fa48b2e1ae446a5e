"""fp64 NumPy restatement of the reference's flow forward / inverse path.

TEST INFRASTRUCTURE ONLY.  This module is the *checker* for the HIP product
path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it; nothing under ``tf-flowavenet_amd/`` does.

PARITY UNPINNED: the reference (ryhorv/tf-flowavenet) ships no tests, no golden
vectors and no checkpoint, and TensorFlow 1.12 (where all of its arithmetic
lives, requirements.txt:3) cannot be imported in this image.  This restatement
is therefore pinned by (1) agreement to ~1e-12 with an independently written
formulation (``oracle/flowavenet_torch.py``), and (2) flow invariants that hold
for any correct implementation (tests/test_oracle.py): exact invertibility,
logdet == log|det J| / T, zero-init known answer, DDI known answer, squeeze
closed form.

Every function cites the reference file:line it follows.  Layout is the
reference's channels-last ``[B, T, C]`` (convolutional.py:17).  Parameters are a
flat ``dict[str, np.ndarray]`` whose keys follow the reference's variable
scopes (model.py:284,297,218,181-183):

  upsample_{n}/{kernel,g,bias}                         kernel (2s,3,1,1)
  Block_{i}/Flow_{j}/ActNorm/{b,logs}                  (1,1,C)
  Block_{i}/Flow_{j}/WaveNet/Conv_front/{kernel,g,bias}        kernel (3,C/2,256)
  Block_{i}/Flow_{j}/WaveNet/ResBlock_{n}/{Conv_filter,Conv_gate,
        filter_conv_c,gate_conv_c,res_conv,skip_conv}/{kernel,g,bias}
  Block_{i}/Flow_{j}/WaveNet/Conv_final/{kernel,g,bias}        kernel (1,256,256)
  Block_{i}/Flow_{j}/WaveNet/ZeroConv1d/{kernel,bias,scale}    kernel (1,256,C)
"""
from __future__ import annotations

import math

import numpy as np

FILTER_SIZE = 256  # model.py:217 hard-codes filter_size=256


# --------------------------------------------------------------------------
# TF-1.12 op semantics
# --------------------------------------------------------------------------
def l2_normalize(x, axis, eps=1e-12):
    """nn_impl.l2_normalize: x * rsqrt(max(sum(x^2, axis), eps)) (convolutional.py:80,186)."""
    ss = np.sum(np.square(x), axis=axis, keepdims=True)
    return x / np.sqrt(np.maximum(ss, eps))


def conv1d_valid(x, kernel, bias, dilation=1):
    """keras Conv1D.call, padding='valid', channels_last (convolutional.py:102-108).

    Cross-correlation, no flip: y[b,t,o] = sum_k sum_i x[b,t+k*d,i] W[k,i,o] + bias[o].
    """
    k = kernel.shape[0]
    t_out = x.shape[1] - dilation * (k - 1)
    y = np.zeros((x.shape[0], t_out, kernel.shape[2]), dtype=x.dtype)
    for kk in range(k):
        y += x[:, kk * dilation: kk * dilation + t_out, :] @ kernel[kk]
    if bias is not None:
        y = y + bias
    return y


def wn_kernel_1d(p, prefix, weight_norm=True):
    """Conv1D.build weight-norm: l2_normalize(V, axis=[0,1]) * g (convolutional.py:73-80)."""
    v = p[prefix + "/kernel"]
    if not weight_norm:
        return v
    return l2_normalize(v, axis=(0, 1)) * p[prefix + "/g"]


def conv_layer(p, prefix, x, kernel_size=3, dilation=1, causal=False):
    """modules.py:6-33 ``Conv``: symmetric (or causal) zero pad + valid dilated conv."""
    if causal:
        pad = dilation * (kernel_size - 1)  # modules.py:12-13
    else:
        pad = dilation * (kernel_size - 1) // 2  # modules.py:15
    xp = np.pad(x, ((0, 0), (pad, pad), (0, 0)))  # modules.py:27
    out = conv1d_valid(xp, wn_kernel_1d(p, prefix), p[prefix + "/bias"], dilation)
    if causal and pad != 0:
        out = out[:, :-pad]  # modules.py:30-31
    return out


def conv1x1(p, prefix, x, weight_norm=True):
    """Plain keras Conv1D(kernel_size=1) used for res/skip/cond (modules.py:74-95)."""
    return conv1d_valid(x, wn_kernel_1d(p, prefix, weight_norm), p[prefix + "/bias"], 1)


def zero_conv1d(p, prefix, x):
    """modules.py:39-56: 1x1 conv without weight-norm, then * exp(3*scale)."""
    out = conv1x1(p, prefix, x, weight_norm=False)
    return out * np.exp(p[prefix + "/scale"] * 3.0)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


# --------------------------------------------------------------------------
# modules.py: ResBlock, WaveNet
# --------------------------------------------------------------------------
def res_block(p, prefix, h, c, kernel_size, dilation, causal=False):
    """modules.py:110-128 (global conditioning is dead: WaveNet.__call__ drops g, :188-189)."""
    h_filter = conv_layer(p, prefix + "/Conv_filter", h, kernel_size, dilation, causal)
    h_gate = conv_layer(p, prefix + "/Conv_gate", h, kernel_size, dilation, causal)
    h_filter = h_filter + conv1x1(p, prefix + "/filter_conv_c", c)
    h_gate = h_gate + conv1x1(p, prefix + "/gate_conv_c", c)
    out = np.tanh(h_filter) * sigmoid(h_gate)
    res = conv1x1(p, prefix + "/res_conv", out)
    skip = conv1x1(p, prefix + "/skip_conv", out)
    return (h + res) * math.sqrt(0.5), skip


def wavenet(p, prefix, x, c, n_layer, causal=False):
    """modules.py:161-186 with skip connections on (skip_channels=256)."""
    h = conv_layer(p, prefix + "/Conv_front", x, 3, 1, causal)
    h = np.maximum(h, 0.0)
    skip = 0.0
    for n in range(n_layer):
        h, s = res_block(p, prefix + "/ResBlock_%d" % n, h, c, 3, 3 ** n, causal)
        skip = skip + s
    out = np.maximum(skip, 0.0)
    out = conv_layer(p, prefix + "/Conv_final", out, 1, 1, causal)
    out = np.maximum(out, 0.0)
    return zero_conv1d(p, prefix + "/ZeroConv1d", out)


# --------------------------------------------------------------------------
# model.py: ActNorm, AffineCoupling, change_order, Flow, Block
# --------------------------------------------------------------------------
def actnorm_ddi(p, prefix, x):
    """model.py:30-41,55-56,65-71: data-dependent init of b and logs from batch x."""
    mean = np.mean(x, axis=(0, 1), keepdims=True)
    p[prefix + "/b"] = (-mean).astype(p[prefix + "/b"].dtype)
    xc = x + p[prefix + "/b"]
    var = np.mean(np.square(xc), axis=(0, 1), keepdims=True)
    logs = np.log(1.0 / (np.sqrt(var) + 1e-7)) / 3.0
    p[prefix + "/logs"] = logs.astype(p[prefix + "/logs"].dtype)


def actnorm_forward(p, prefix, x, init=False):
    """model.py:86-94: center (x + b) then scale (* exp(3*logs)); dlogdet = mean_C(3*logs)."""
    if init:
        actnorm_ddi(p, prefix, x)
    x = x + p[prefix + "/b"]
    logs = p[prefix + "/logs"] * 3.0
    x = x * np.exp(logs)
    return x, np.mean(logs)


def actnorm_reverse(p, prefix, y):
    """model.py:97-102: scale^-1 then center^-1."""
    logs = p[prefix + "/logs"] * 3.0
    x = y * np.exp(-logs)
    return x - p[prefix + "/b"]


def split2(x):
    h = x.shape[2] // 2
    return x[:, :, :h], x[:, :, h:]


def coupling_forward(p, prefix, x, c, n_layer, affine=True, causal=False):
    """model.py:121-141."""
    in_a, in_b = split2(x)
    c_a, _ = split2(c)
    net = wavenet(p, prefix + "/WaveNet", in_a, c_a, n_layer, causal)
    if affine:
        log_s, t = split2(net)
        out_b = (in_b - t) * np.exp(-log_s)
        logdet = np.mean(-log_s) / 2.0
    else:
        out_b = in_b + net
        logdet = None
    return np.concatenate([in_a, out_b], 2), logdet


def coupling_reverse(p, prefix, y, c, n_layer, affine=True, causal=False):
    """model.py:143-161."""
    out_a, out_b = split2(y)
    c_a, _ = split2(c)
    net = wavenet(p, prefix + "/WaveNet", out_a, c_a, n_layer, causal)
    if affine:
        log_s, t = split2(net)
        in_b = out_b * np.exp(log_s) + t
    else:
        in_b = out_b - net
    return np.concatenate([out_a, in_b], 2)


def change_order(x, c):
    """model.py:166-174."""
    x_a, x_b = split2(x)
    c_a, c_b = split2(c)
    return np.concatenate([x_b, x_a], 2), np.concatenate([c_b, c_a], 2)


def flow_forward(p, prefix, x, c, hp, init=False):
    """model.py:185-194: ActNorm -> coupling -> change_order."""
    out, logdet = actnorm_forward(p, prefix + "/ActNorm", x, init)
    out, det = coupling_forward(p, prefix, out, c, hp.n_layer, hp.affine, hp.causality)
    out, c = change_order(out, c)
    if det is not None:
        logdet = logdet + det
    return out, c, logdet


def flow_reverse(p, prefix, y, c, hp):
    """model.py:196-202: change_order -> coupling^-1 -> ActNorm^-1."""
    y, c = change_order(y, c)
    x = coupling_reverse(p, prefix, y, c, hp.n_layer, hp.affine, hp.causality)
    x = actnorm_reverse(p, prefix + "/ActNorm", x)
    return x, c


def squeeze(x):
    """model.py:226-228: reshape -> transpose[0,1,3,2] -> reshape; out[b,t,2c+j]=in[b,2t+j,c]."""
    b, t, ch = x.shape
    x = x.reshape(b, t // 2, 2, ch)
    x = x.transpose(0, 1, 3, 2)
    return x.reshape(b, t // 2, 2 * ch)


def unsqueeze(x):
    """model.py:260-263."""
    b, t, ch = x.shape
    x = x.reshape(b, t, ch // 2, 2)
    x = x.transpose(0, 1, 3, 2)
    return x.reshape(b, t * 2, ch // 2)


def block_forward(p, prefix, x, c, hp, init=False):
    """model.py:221-247."""
    out = squeeze(x)
    c = squeeze(c)
    logdet = 0.0
    for j in range(hp.n_flow):
        out, c, det = flow_forward(p, "%s/Flow_%d" % (prefix, j), out, c, hp, init)
        logdet = logdet + det
    return out, c, logdet


def block_reverse(p, prefix, y, c, hp):
    """model.py:249-275."""
    x = y
    for j in reversed(range(hp.n_flow)):
        x, c = flow_reverse(p, "%s/Flow_%d" % (prefix, j), x, c, hp)
    return unsqueeze(x), unsqueeze(c)


# --------------------------------------------------------------------------
# model.py: FloWaveNet.upsample / forward / reverse
# --------------------------------------------------------------------------
def conv2d_transpose_same(x, kernel, bias, s):
    """keras Conv2DTranspose, padding='same', strides (s,1), kernel (2s,3), filters=1.

    Scatter form of SURVEY Appendix A: y[i*s + k - s//2, w + kw - 1] += x[i,w] W[k,kw].
    x: [B,H,W] (the trailing channel axis of size 1 is dropped).
    """
    b, h, w = x.shape
    kh, kw_n = kernel.shape[0], kernel.shape[1]
    y = np.zeros((b, h * s, w), dtype=x.dtype)
    for i in range(h):
        for k in range(kh):
            tau = i * s + k - s // 2
            if tau < 0 or tau >= h * s:
                continue
            for kw in range(kw_n):
                lo = max(0, 1 - kw)          # source w range with 0 <= w+kw-1 < W
                hi = min(w, w + 1 - kw)
                y[:, tau, lo + kw - 1: hi + kw - 1] += x[:, i, lo:hi] * kernel[k, kw, 0, 0]
    return y + bias


def upsample(p, c, hp):
    """model.py:398-404 + Conv2DTranspose weight-norm over axis [0,2] (convolutional.py:186)."""
    out = c
    for n, s in enumerate(hp.upsample_scales):
        pre = "upsample_%d" % n
        kern = l2_normalize(p[pre + "/kernel"], axis=(0, 2)) * p[pre + "/g"]
        out = conv2d_transpose_same(out, kern, p[pre + "/bias"][0], s)
        out = np.maximum(out, 0.4 * out)  # leaky_relu(x, 0.4), model.py:307
    return out


def forward(p, x, c, hp, init=False):
    """FloWaveNet.forward (model.py:317-347): returns (log_p, logdet) scalars and z.

    ``init=True`` performs the ActNorm data-dependent init in place on ``p``.
    """
    out = x
    c = upsample(p, c, hp)
    logdet = 0.0
    for i in range(hp.n_block):
        out, c, det = block_forward(p, "Block_%d" % i, out, c, hp, init)
        logdet = logdet + det
    log_p = np.mean(0.5 * (-math.log(2.0 * math.pi) - np.square(out)))
    return float(log_p), float(logdet), out


def reverse(p, z, c, hp):
    """FloWaveNet.reverse (model.py:350-396)."""
    c = upsample(p, c, hp)
    x = z
    for _ in range(hp.n_block):  # model.py:374-392 pre-squeeze
        x = squeeze(x)
        c = squeeze(c)
    for i in reversed(range(hp.n_block)):
        x, c = block_reverse(p, "Block_%d" % i, x, c, hp)
    return x


def to_f64(params):
    return {k: np.asarray(v, dtype=np.float64) for k, v in params.items()}
