"""Independent torch-CPU formulation of the same path (second oracle + CPU baseline).

TEST INFRASTRUCTURE ONLY (same rule as flowavenet_np.py): imported by tests/,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg only.

Written differently on purpose from ``flowavenet_np.py`` so that agreement
between the two means something:
  * channels-first tensors, ``F.conv1d(padding=, dilation=)`` instead of pad +
    shifted matmul; ``F.conv_transpose2d`` instead of the scatter loop
    (SURVEY Appendix A states the equivalence);
  * weight-norm is folded once up front (``fold``), not per call;
  * squeeze is the closed-form bit-reversal gather of SURVEY Appendix C applied
    n times at once, change_order is index bookkeeping (no concat of c): the
    conditioning half is selected by a swap-parity bit.

Reference lines restated: model.py:317-347 (forward), :350-396 (reverse),
:398-404 (upsample), :86-102 (ActNorm), :121-161 (coupling), modules.py:110-128,
:161-186, convolutional.py:73-80,179-186.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def bitrev(r: int, n: int) -> int:
    out = 0
    for _ in range(n):
        out = (out << 1) | (r & 1)
        r >>= 1
    return out


def squeeze_n(x: torch.Tensor, n: int) -> torch.Tensor:
    """n squeezes at once, channels-first.  x: [B, C0, T] -> [B, C0*2^n, T/2^n].

    s_n[b, m*2^n + r, t] = x[b, m, t*2^n + bitrev_n(r)]   (SURVEY Appendix C).
    """
    b, c0, t = x.shape
    p = 1 << n
    xv = x.reshape(b, c0, t // p, p)                    # [.., t, phase]
    idx = torch.tensor([bitrev(r, n) for r in range(p)], dtype=torch.long)
    xv = xv.index_select(3, idx)                        # [.., t, r]
    return xv.permute(0, 1, 3, 2).reshape(b, c0 * p, t // p)


def unsqueeze_n(x: torch.Tensor, n: int, c0: int) -> torch.Tensor:
    b, c, t = x.shape
    p = 1 << n
    xv = x.reshape(b, c0, p, t).permute(0, 1, 3, 2)     # [b, m, t, r]
    inv = torch.empty(p, dtype=torch.long)
    for r in range(p):
        inv[bitrev(r, n)] = r
    xv = xv.index_select(3, inv)
    return xv.reshape(b, c0, t * p)


def fold(params, hp, dtype=torch.float64):
    """Fold weight-norm and lay kernels out channels-first ([out, in, k])."""
    out = {}

    def t(a):
        return torch.as_tensor(a, dtype=torch.float64)

    def wn(pre, weight_norm=True):
        v = t(params[pre + "/kernel"])                  # [k, in, out]
        if weight_norm:
            nrm = torch.sqrt(torch.clamp((v * v).sum(dim=(0, 1), keepdim=True), min=1e-12))
            v = v / nrm * t(params[pre + "/g"])
        out[pre + "/w"] = v.permute(2, 1, 0).contiguous().to(dtype)
        out[pre + "/bias"] = t(params[pre + "/bias"]).to(dtype)

    for n, _ in enumerate(hp.upsample_scales):
        pre = "upsample_%d" % n
        v = t(params[pre + "/kernel"])                  # [kh, kw, out=1, in=1]
        nrm = torch.sqrt(torch.clamp((v * v).sum(dim=(0, 2), keepdim=True), min=1e-12))
        v = v / nrm * t(params[pre + "/g"])
        out[pre + "/w"] = v.permute(3, 2, 0, 1).contiguous().to(dtype)   # [in, out, kh, kw]
        out[pre + "/bias"] = t(params[pre + "/bias"]).to(dtype)
    for i in range(hp.n_block):
        for j in range(hp.n_flow):
            fp = "Block_%d/Flow_%d" % (i, j)
            out[fp + "/ActNorm/b"] = t(params[fp + "/ActNorm/b"]).reshape(-1).to(dtype)
            out[fp + "/ActNorm/logs"] = t(params[fp + "/ActNorm/logs"]).reshape(-1).to(dtype)
            wp = fp + "/WaveNet"
            wn(wp + "/Conv_front")
            for n in range(hp.n_layer):
                for nm in ("Conv_filter", "Conv_gate", "filter_conv_c", "gate_conv_c",
                           "res_conv", "skip_conv"):
                    wn("%s/ResBlock_%d/%s" % (wp, n, nm))
            wn(wp + "/Conv_final")
            wn(wp + "/ZeroConv1d", weight_norm=False)
            out[wp + "/ZeroConv1d/escale"] = torch.exp(
                3.0 * t(params[wp + "/ZeroConv1d/scale"]).reshape(-1)).to(dtype)
    return out


def upsample(fp, c, hp):
    """c: [B, F, mels] -> [B, mels, T] channels-first."""
    x = c.unsqueeze(1)                                   # [B, 1, F, mels]
    for n, s in enumerate(hp.upsample_scales):
        pre = "upsample_%d" % n
        x = F.conv_transpose2d(x, fp[pre + "/w"], fp[pre + "/bias"], stride=(s, 1),
                               padding=(s // 2, 1))
        x = F.leaky_relu(x, 0.4)
    return x.squeeze(1).transpose(1, 2).contiguous()


def wavenet(fp, pre, x, c, n_layer):
    h = F.relu(F.conv1d(x, fp[pre + "/Conv_front/w"], fp[pre + "/Conv_front/bias"], padding=1))
    skip = None
    for n in range(n_layer):
        rp = "%s/ResBlock_%d" % (pre, n)
        d = 3 ** n
        f = F.conv1d(h, fp[rp + "/Conv_filter/w"], fp[rp + "/Conv_filter/bias"], padding=d, dilation=d)
        g = F.conv1d(h, fp[rp + "/Conv_gate/w"], fp[rp + "/Conv_gate/bias"], padding=d, dilation=d)
        f = f + F.conv1d(c, fp[rp + "/filter_conv_c/w"], fp[rp + "/filter_conv_c/bias"])
        g = g + F.conv1d(c, fp[rp + "/gate_conv_c/w"], fp[rp + "/gate_conv_c/bias"])
        o = torch.tanh(f) * torch.sigmoid(g)
        s = F.conv1d(o, fp[rp + "/skip_conv/w"], fp[rp + "/skip_conv/bias"])
        skip = s if skip is None else skip + s
        if n + 1 < n_layer:   # last res_conv output is unused
            h = (h + F.conv1d(o, fp[rp + "/res_conv/w"], fp[rp + "/res_conv/bias"])) * math.sqrt(0.5)
    u = F.relu(F.conv1d(F.relu(skip), fp[pre + "/Conv_final/w"], fp[pre + "/Conv_final/bias"]))
    z = F.conv1d(u, fp[pre + "/ZeroConv1d/w"], fp[pre + "/ZeroConv1d/bias"])
    return z * fp[pre + "/ZeroConv1d/escale"].view(1, -1, 1)


def _halves(x, parity):
    """Logical (first, second) halves of a tensor whose physical halves are swapped iff parity."""
    h = x.shape[1] // 2
    lo, hi = x[:, :h], x[:, h:]
    return (hi, lo) if parity else (lo, hi)


def forward(fp, x, c, hp, as_tensors=False):
    """x: [B,T,1], c: [B,F,mels] -> (log_p, logdet, z[B,T_last,C_last] channels-last).
    as_tensors: keep log_p / logdet as 0-dim tensors (for autograd, oracle/grad_torch.py)."""
    b, t, _ = x.shape
    cu = upsample(fp, c, hp)                             # [B, mels, T]
    xs = x.transpose(1, 2)                               # [B, 1, T]
    logdet = xs.new_zeros(())
    parity = 0
    for i in range(hp.n_block):
        n = i + 1
        cur = squeeze_n(unsqueeze_n(xs, i, 1), n) if i else squeeze_n(xs, 1)
        # xs is kept in *canonical* (un-swapped) channel order; parity tracks change_order.
        cs = squeeze_n(cu, n)
        for j in range(hp.n_flow):
            pre = "Block_%d/Flow_%d" % (i, j)
            a, bb = _halves(cur, parity)
            ca, _ = _halves(cs, parity)
            cdim = cur.shape[1]
            ab = fp[pre + "/ActNorm/b"]
            al = fp[pre + "/ActNorm/logs"] * 3.0
            h = cdim // 2
            a = (a + ab[:h].view(1, -1, 1)) * torch.exp(al[:h]).view(1, -1, 1)
            bb = (bb + ab[h:].view(1, -1, 1)) * torch.exp(al[h:]).view(1, -1, 1)
            logdet = logdet + al.mean()
            net = wavenet(fp, pre + "/WaveNet", a, ca, hp.n_layer)
            log_s, tt = net[:, :h], net[:, h:]
            bb = (bb - tt) * torch.exp(-log_s)
            logdet = logdet + (-log_s).mean() / 2.0
            cur = torch.cat([bb, a], 1) if parity else torch.cat([a, bb], 1)
            parity ^= 1
        xs = cur
    if parity:  # leave the tensor in logical order for the caller (odd n_block*n_flow)
        h = xs.shape[1] // 2
        xs = torch.cat([xs[:, h:], xs[:, :h]], 1)
    log_p = (0.5 * (-math.log(2.0 * math.pi) - xs * xs)).mean()
    if as_tensors:
        return log_p, logdet, xs.transpose(1, 2).contiguous()
    return float(log_p), float(logdet), xs.transpose(1, 2).contiguous()


def reverse(fp, z, c, hp):
    """z: [B,T,1], c: [B,F,mels] -> x [B,T,1]."""
    cu = upsample(fp, c, hp)
    cur = squeeze_n(z.transpose(1, 2), hp.n_block)       # logical order == canonical, parity 0
    parity = 0
    for i in reversed(range(hp.n_block)):
        n = i + 1
        cs = squeeze_n(cu, n)
        for j in reversed(range(hp.n_flow)):
            pre = "Block_%d/Flow_%d" % (i, j)
            parity ^= 1                                   # change_order first (model.py:199)
            a, bb = _halves(cur, parity)
            ca, _ = _halves(cs, parity)
            h = cur.shape[1] // 2
            net = wavenet(fp, pre + "/WaveNet", a, ca, hp.n_layer)
            log_s, tt = net[:, :h], net[:, h:]
            bb = bb * torch.exp(log_s) + tt
            ab = fp[pre + "/ActNorm/b"]
            al = fp[pre + "/ActNorm/logs"] * 3.0
            a = a * torch.exp(-al[:h]).view(1, -1, 1) - ab[:h].view(1, -1, 1)
            bb = bb * torch.exp(-al[h:]).view(1, -1, 1) - ab[h:].view(1, -1, 1)
            cur = torch.cat([bb, a], 1) if parity else torch.cat([a, bb], 1)
        if parity:
            # unsqueeze acts on logical order in the reference; our tensor is canonical-with-
            # parity, and squeeze commutes with the half swap, so the
            # parity bit simply carries to the next level.
            pass
        cur = squeeze_n(unsqueeze_n(cur, n, 1), n - 1) if n > 1 else unsqueeze_n(cur, 1, 1)
    if parity:
        raise NotImplementedError("odd n_block*n_flow leaves a swapped tensor; not a BASELINE config")
    return cur.transpose(1, 2).contiguous()


def forward_z(fp, x, c, hp):
    """Differentiable z (channels-last, squeezed) for Jacobian tests."""
    return forward(fp, x, c, hp)[2]
