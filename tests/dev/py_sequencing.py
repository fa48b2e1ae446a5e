"""Developer reference: the training forward / backward sequenced from Python over the stage entry points (what
tf-flowavenet_amd/training.py did before csrc/train_api.hip took the sequencing over).  tests/test_train.py checks that
``fwn_train_loss_and_grads`` reproduces it bit for bit.  Call ``loss_and_grads(engine, tp, params, x, c)`` with a
GradEngine whose packing (engine._tp) is current."""
import ctypes as C

import numpy as np
import torch

from tf_flowavenet_amd import weights
from tf_flowavenet_amd.training import (SQH, gemm, tn_block_splits, tn_weight_grad_group, transpose_shift, weight_grad_partials,
                                        wn_backward_group)


def _flush(self, grads, prefix):
    """Copy the gradients under `prefix` that were not produced in place into grad_out."""
    go = self._gout
    if go is None:
        return
    for k in [k for k in grads if k.startswith(prefix)]:
        if grads[k].data_ptr() != go[k].data_ptr():
            go[k].copy_(grads[k].reshape(go[k].shape))
            grads[k] = go[k]

def _wn_group(self, grads, items):
    """items: ``(name, part, k, col0, shape, scale, row_src)`` - split-K partials of a weight-gradient GEMM (fp32
    [S][rows + 1][ncols], bias gradient in the last row) -> gradients of a weight-normed conv's kernel, g and bias,
    written straight into grad_out where that is contiguous."""
    import torch
    go, jobs = self._gout, []
    for name, part, k, col0, shape, scale, row_src, col_src in items:
        dev = part.device
        normed = (name + "/g") in self._tp.params                  # ZeroConv1d carries no weight norm
        v = self._tp._f32(name + "/kernel")
        g = self._tp._f32(name + "/g") if normed else None
        n = int(v.shape[-1])
        if go is not None and go[name + "/kernel"].is_contiguous():
            dv, dg, dbo = go[name + "/kernel"].view(k, n), go[name + "/g"] if normed else None, go[name + "/bias"].view(-1)
        else:
            dv = torch.empty(k, n, dtype=torch.float32, device=dev)
            dg = torch.empty(n, dtype=torch.float32, device=dev) if normed else None
            dbo = torch.empty(n, dtype=torch.float32, device=dev)
        jobs.append(dict(part=part, k=k, n=n, col0=col0, bias_row=int(part.shape[1]) - 1, scale=scale, row_src=row_src,
                         col_src=col_src, v=v if normed else None, g=g, dv=dv, dg=dg, db=dbo))
        grads[name + "/kernel"] = dv.view(shape)
        if normed:
            grads[name + "/g"] = dg
        grads[name + "/bias"] = dbo.view(self._shapes[name + "/bias"])
    wn_backward_group(jobs)




def loss_and_grads(self, tp, params, x, c):
    self._zeroed = getattr(self, "_zeroed", set())
    self._consts = getattr(self, "_consts", {})
    hp, lib = self.hp, self.lib
    dev = x.device
    st = torch.cuda.current_stream(dev).cuda_stream
    shp = self._shapes
    pm, md = tp.pm, tp.pm.model_desc
    L, half = hp.n_layer, hp.num_mels // 2
    B, T = int(x.shape[0]), int(x.shape[1])
    if T % (1 << hp.n_block) or int(c.shape[1]) * hp.hop_size != T:
        raise ValueError("bad shapes")
    f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    b16 = lambda *s: torch.empty(*s, dtype=torch.bfloat16, device=dev)

    # ---------------- forward, keeping what the backward needs ----------------
    cplanes = b16(2, B, T, half)
    cur, ups = c, []
    for n, s_ in enumerate(hp.upsample_scales):
        last = n == len(hp.upsample_scales) - 1
        bb, hh, ww = cur.shape
        out = None if last else f32(bb, hh * s_, ww)
        ub = tp._f32("upsample_%d/bias" % n)          # read on the device: nothing of the parameters rides in launch arguments
        self._call("fwn_upsample_stage_dev", cur.data_ptr(), bb, hh, ww, md.up_w[n], ub.data_ptr(), int(s_),
                   None if last else out.data_ptr(), cplanes.data_ptr() if last else None, st)
        ups.append(cur)
        cur = out
    planes = f32(2, B * T // 2)
    self._call("fwn_split_planes", x.data_ptr(), B, T, planes.data_ptr(), st)
    saved, partials = [], []
    p = 0
    for i in range(hp.n_block):
        ch = 1 << i
        ti = T // (2 * ch)
        m = B * ti
        cin = half * (2 << i)
        # Small-M blocks: the conditioning projections P = c_a Wc of ALL flows and layers of the block in two
        # batched launches (one per conditioning parity), as in inference (api.hip hoist_cond) - fused into every
        # gate they would put K = cin (up to 10240) on the latency chain of a handful of workgroups.
        P = None
        if m < 4096:
            d0 = pm.flow_descs[i * hp.n_flow]
            P = f32(hp.n_flow, L, m, 512)
            ns = int(self.lib.fwn_cond_splits(m, ((hp.n_flow + 1) // 2) * L, d0.kcpad))     # few rows: K dealt over workgroups
            part = f32(max(1, ns - 1), hp.n_flow, L, m, 512)
            for g_ in range(min(2, hp.n_flow)):
                self._call("fwn_cond_split", cplanes[p ^ g_].data_ptr(), d0.Wc[0], P.data_ptr(), 512 * d0.kcpad, m * 512, g_, 2,
                           (hp.n_flow - g_ + 1) // 2, L, m, cin, d0.kcpad, part.data_ptr(), P.numel(), ns, st)
            self._call("fwn_cond_reduce", P.data_ptr(), part.data_ptr(), P.numel(), ns, P.numel(), st)
        for j in range(hp.n_flow):
            d = pm.flow_descs[i * hp.n_flow + j]
            t = tp.flows[(i, j)]
            an = pm.an[(i, j)]                                   # [2][4][Ch]
            xa, xb = planes[p].view(m, ch), planes[p ^ 1].view(m, ch)
            ca = cplanes[p].view(m, cin)
            h = [b16(m, 256) for _ in range(L)]
            o_all = b16(L, m, 256)
            o = [o_all[l] for l in range(L)]
            aux = [b16(m, 512) for _ in range(L)]
            scr = torch.empty(m * 2 * ch, dtype=torch.bfloat16, device=dev) if ch >= 32 else None      # (hi | lo) image: ring-GEMM front
            # the inference kernels: the front conv applies the flow's ActNorm on the fly, the tail normalises both planes
            self._call("fwn_front", C.byref(d), xa.data_ptr(), h[0].data_ptr(), scr.data_ptr() if scr is not None else None, m, ti, 1, st)
            for l in range(L):
                self._call("fwn_gate_train", C.byref(d), l, h[l].data_ptr(), ca.data_ptr() if P is None else None,
                           P[j, l].data_ptr() if P is not None else None, o[l].data_ptr(), aux[l].data_ptr(), m, ti, st)
                if l + 1 < L:
                    self._call("fwn_res", C.byref(d), l, o[l].data_ptr(), h[l].data_ptr(), h[l + 1].data_ptr(), m, st)
            s_act, u_act, z = b16(m, 256), b16(m, 256), f32(m, 2 * ch)
            part = f32(int(self.lib.fwn_tail_partials_desc(C.byref(d), m, -1)))      # the exact count (fwn_tail_partials is an upper bound)
            self._call("fwn_tail_train", C.byref(d), o_all.data_ptr(), m * 256, xa.data_ptr(), xb.data_ptr(), part.data_ptr(), m,
                       s_act.data_ptr(), u_act.data_ptr(), z.data_ptr(), st)
            partials.append(part)
            saved.append((i, j, p, h, o, aux, s_act, u_act, z))
            p ^= 1
    partial_all = torch.cat(partials)
    out2 = f32(2)
    self._call("fwn_prior_logp", planes.data_ptr(), B * T, partial_all.data_ptr(), partial_all.numel(), out2.data_ptr(), st)
    log_p, logdet = out2[0], out2[1]          # (the tail's log-det partials carry the ActNorm terms)
    loss = -(log_p + logdet)

    # ---------------- backward ----------------
    grads = {}
    # d loss / d z = z / (B T)   (log_p = mean 0.5(-log 2pi - z^2)): a copy, scaled by the ActNorm kernel
    gplanes = planes.clone()
    if (dev, B * T) not in self._consts:
        self._consts[(dev, B * T)] = torch.tensor([0.0, 1.0 / (B * T), 0.0, 0.0], dtype=torch.float32, device=dev)
    inv = self._consts[(dev, B * T)]
    self._call("fwn_actnorm_apply", gplanes.data_ptr(), inv.data_ptr(), B * T, 1, st)
    dcplanes = torch.zeros(2, B * T * half, dtype=torch.float32, device=dev)
    for (i, j, p, h, o, aux, s_act, u_act, z) in reversed(saved):
        ch = 1 << i
        ti = T // (2 * ch)
        m = B * ti
        cin = half * (2 << i)
        t = tp.flows[(i, j)]
        an = pm.an[(i, j)]
        fp = weights.flow_prefix(i, j)
        wp = fp + "/WaveNet"
        xa, xb = planes[p].view(m, ch), planes[p ^ 1].view(m, ch)          # y_a, out_b
        ga, gb = gplanes[p].view(m, ch), gplanes[p ^ 1].view(m, ch)
        ca = cplanes[p].view(m, cin)
        dca = dcplanes[p].view(m, cin)
        br = tp.br[i]
        # coupling
        ldz = t["ldz"]
        dz = (torch.zeros if ldz > 2 * ch else torch.empty)(m, ldz, dtype=torch.bfloat16, device=dev)   # padding columns must be 0
        dzz = f32(m, 2 * ch)
        self._call("fwn_coupling_bwd", gb.data_ptr(), xb.data_ptr(), z.data_ptr(), t["ez"].data_ptr(), m, ch,
                   1.0 / (2.0 * m * ch), dz.data_ptr(), ldz, dzz.data_ptr(), st)
        du = gemm([(dz, ldz, 0, 0)], t["WzT"], 256, m, mask=u_act)
        # the weight gradients of the flow are collected and run as ONE grouped GEMM + ONE grouped weight-norm
        # backward at the end of the flow (their operands stay alive until then)
        tnj, wnj = [], []

        def wgrad(x_, dy_, kx, n, shifts=(0,)):
            tnj.append((x_, dy_, kx, n, shifts))
            return len(tnj) - 1

        def wn(name, job, k, col0, shape, scale=1.0, row_src=None, col_src=None):
            wnj.append((name, job, k, col0, shape, scale, row_src, col_src))

        # ZeroConv1d (no weight norm): columns back to the reference's channel order through col_src
        wn(wp + "/ZeroConv1d", wgrad(u_act, dz, 256, ldz), 256, 0, (1, 256, 2 * ch), col_src=t["zinv32"])
        wn(wp + "/Conv_final", wgrad(s_act, du, 256, 256), 256, 0, (1, 256, 256))
        ds = gemm([(du, 256, 0, 0)], t["WfinT"], 256, m, mask=s_act)
        d_all = gemm([(ds, 256, 0, 0)], t["WskipT_all"], L * 256, m)       # do_l = dS Wskip_l for every layer at once
        d_o = [d_all[:, l * 256:(l + 1) * 256] for l in range(L)]
        for l in range(L):
            rp = "%s/ResBlock_%d" % (wp, l)
            wn(rp + "/skip_conv", wgrad(o[l], ds, 256, 256), 256, 0, (1, 256, 256))
        dh_next = None
        dpres = [None] * L
        for l in range(L - 1, -1, -1):
            rp = "%s/ResBlock_%d" % (wp, l)
            dil = 3 ** l
            if dh_next is not None:      # h_{l+1} = (h_l + res(o_l)) sqrt(1/2)
                wn(rp + "/res_conv", wgrad(o[l], dh_next, 256, 256), 256, 0, (1, 256, 256), scale=SQH)
                d_o[l] = gemm([(dh_next, 256, 0, 0)], t["WresT"][l], 256, m, res=d_o[l], rscale=1.0 / SQH, oscale=SQH)
            else:
                for nm in ("kernel", "g", "bias"):      # dead res_conv of the last layer (modules.py:126-128)
                    key = "%s/res_conv/%s" % (rp, nm)
                    go_ = self._gout
                    if go_ is None:
                        grads[key] = torch.zeros(shp[key], dtype=torch.float32, device=dev)
                    else:       # nothing else ever writes there: zeroed the first time this buffer is seen
                        if (key, go_[key].data_ptr()) not in self._zeroed:
                            go_[key].zero_()
                            self._zeroed.add((key, go_[key].data_ptr()))
                        grads[key] = go_[key]
            dpre = b16(m, 512)
            self._call("fwn_gate_bwd", d_o[l].data_ptr(), int(d_o[l].stride(0)), aux[l].data_ptr(), m, dpre.data_ptr(), st)
            jd = wgrad(h[l], dpre, 256, 512, (-dil, 0, dil))
            wn(rp + "/Conv_filter", jd, 768, 0, (3, 256, 256))
            wn(rp + "/Conv_gate", jd, 768, 256, (3, 256, 256))
            jc = wgrad(ca, dpre, cin, 512)
            wn(rp + "/filter_conv_c", jc, cin, 0, (1, cin, 256), row_src=tp.cond_rows[i])
            wn(rp + "/gate_conv_c", jc, cin, 256, (1, cin, 256), row_src=tp.cond_rows[i])
            dpres[l] = dpre          # the conditioning gradient of the flow is one GEMM over all layers (K = L * 512), below
            segs = [(dpre, 512, -(tap - 1) * dil, tap * 512) for tap in range(3)]
            dh = gemm(segs, t["WdT"][l], 256, m, ti=ti, res=dh_next, rscale=SQH if dh_next is not None else 0.0,
                      mask=h[0] if l == 0 else None)
            dh_next = dh
        gemm([(dpres[l], 512, 0, l * 512) for l in range(L)], t["WcT_all"], cin, m, out=dca, accumulate=True)
        # front conv
        ya_bf = xa.to(torch.bfloat16)
        kxp = max(ch, 8)        # rows of fewer than 8 channels are padded to 8 (16 bytes): the same TN job for every block
        if kxp != ch:
            ya8 = torch.zeros(m, kxp, dtype=torch.bfloat16, device=dev)
            ya8[:, :ch] = ya_bf
            ya_bf = ya8
        wn(wp + "/Conv_front", wgrad(ya_bf, dh_next, kxp, 256, (-1, 0, 1)), 3 * ch, 0, (3, ch, 256), row_src=tp.front_rows[i])
        # the split count the C sequencing plans for the flow's block (one launch for the weight gradients of all its flows)
        parts = tn_weight_grad_group(tnj, m, ti, nsplit=tn_block_splits([(kx, n, len(sh)) for _, _, kx, n, sh in tnj], hp.n_flow, m))
        _wn_group(self, grads, [(nm, parts[jb] if isinstance(jb, int) else jb, k_, c0_, shp_, sc_, rs_, cs_)
                               for nm, jb, k_, c0_, shp_, sc_, rs_, cs_ in wnj])
        segs = [(dh_next, 256, -(tap - 1), tap * 256) for tap in range(3)]
        gemm(segs, t["WfT"], ch, m, ti=ti, out=ga, accumulate=True)
        # ActNorm of both planes back to the flow's inputs, with its b / logs gradients and the ZeroConv scale
        # gradient (three parameter-sized reductions) in the same two launches
        go = self._gout
        names = (fp + "/ActNorm/b", fp + "/ActNorm/logs", wp + "/ZeroConv1d/scale")
        outs = [go[nm].view(-1) if go is not None and go[nm].is_contiguous() else f32(2 * ch) for nm in names]
        scr = torch.empty(int(lib.fwn_flow_small_grads_partials(m, ch)), dtype=torch.float64, device=dev)
        self._call("fwn_flow_small_grads", ga.data_ptr(), xa.data_ptr(), gb.data_ptr(), xb.data_ptr(), dzz.data_ptr(),
                   an.data_ptr(), m, ch, br.data_ptr(), t["zcol"].data_ptr(), scr.data_ptr(), outs[0].data_ptr(),
                   outs[1].data_ptr(), outs[2].data_ptr(), st)
        for nm, o_ in zip(names, outs):
            grads[nm] = o_.view(1, 1, -1)
        _flush(self, grads, fp + "/")
        if j == 0 and self._on_block is not None:
            self._on_block(i)
    # up-sampling transposed convolutions (model.py:301-311), last stage first
    nmel = 2 * half
    dy = dcplanes.view(2, B, T, half).permute(1, 2, 0, 3).reshape(B, T, nmel).contiguous()
    y = cplanes.permute(1, 2, 0, 3).reshape(B, T, nmel).float().contiguous()
    for n in range(len(hp.upsample_scales) - 1, -1, -1):
        xin, s_ = ups[n], int(hp.upsample_scales[n])
        hh = int(xin.shape[1])
        dx = f32(B, hh, nmel) if n > 0 else None
        dwb = f32(6 * s_ + 1)
        dwk, dbias = dwb[:6 * s_].view(2 * s_, 3), dwb[6 * s_:]
        scr = f32(lib.fwn_upsample_bwd_partials(B, hh, s_))
        self._call("fwn_upsample_bwd", dy.data_ptr(), y.data_ptr(), xin.data_ptr(), B, hh, nmel, s_, md.up_w[n],
                   dx.data_ptr() if dx is not None else None, dwb.data_ptr(), scr.data_ptr(), st)
        v = torch.as_tensor(params["upsample_%d/kernel" % n]).to(device=dev, dtype=torch.float32).reshape(2 * s_, 3).contiguous()
        g3 = torch.as_tensor(params["upsample_%d/g" % n]).to(device=dev, dtype=torch.float32).reshape(1).expand(3).contiguous()
        dv, dg3 = f32(2 * s_, 3), f32(3)
        wn_backward_group([dict(part=dwk.view(1, 2 * s_, 3), k=2 * s_, n=3, v=v, g=g3, dv=dv, dg=dg3)])
        grads["upsample_%d/kernel" % n] = dv.view(2 * s_, 3, 1, 1)
        grads["upsample_%d/g" % n] = dg3.sum().view(1)      # the three kw columns share one scalar g (convolutional.py:186)
        grads["upsample_%d/bias" % n] = dbias
        dy, y = dx, xin
    _flush(self, grads, "upsample_")
    if self._on_block is not None:
        self._on_block(-1)
    self.last_dcplanes = dcplanes
    return loss, log_p, logdet, grads

