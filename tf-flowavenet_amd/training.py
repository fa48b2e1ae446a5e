"""Training side of the path (train.py:35-83).

``GradEngine`` computes loss = -(log_p + logdet) and its gradient with respect to every trainable
tensor of the reference, ``Trainer`` adds the data-parallel clip / Adam step of ``optim.py``.  The
building blocks are thin wrappers over ``csrc/train_kernels.hip``: ``fwn_gemm`` (generic multi-segment
GEMM on the LDS-DMA ring core: data gradients on transposed packed weights), ``fwn_transpose_shift`` +
split-K ``fwn_tn_gemm_group`` + ``fwn_wn_backward_group`` (weight gradients: K = rows, partials summed in a fixed
order), and the element-wise ``fwn_*_bwd`` kernels.  The sequencing is Python; every activation-sized
arithmetic step runs in ``libfwn.so`` and there is no fallback path.
"""
from __future__ import annotations

import ctypes as C
import os

from . import _lib


_STREAMS = {}     # device -> stream handle, only while a GradEngine call is running (it never switches streams)


def _stream(t):
    import torch
    h = _STREAMS.get(t.device)
    return h if h is not None else torch.cuda.current_stream(t.device).cuda_stream


def gemm(segs, w, n, m, *, ti=0, bias=None, res=None, rscale=1.0, mask=None, relu=False, oscale=1.0,
         out=None, out_f32=False, accumulate=False, nsplit=1, split_stride=0, gate=None):
    """``out[m, n] = oscale * relu?(mask?(sum_s shift(x_s)[:, :k_s] @ w[:, koff_s:koff_s+k_s].T + bias + rscale*res))``.

    segs: list of ``(x, k, shift, koff)`` with ``x`` a bf16 2-D tensor (rows, ld); ``w`` bf16 (n, ldw).
    gate: ``(aux, dpre, col0)`` - the 256 output columns from ``col0`` go through the gate derivative into ``dpre``
    (bf16 [m, 512]) instead of ``out`` (fwn.h ``fwn_gemm_desc.gate_aux``)."""
    import torch
    lib = _lib.load()
    d = _lib.GemmDesc()
    if not 1 <= len(segs) <= _lib.FWN_GEMM_MAXSEG:
        raise ValueError("1..%d segments" % _lib.FWN_GEMM_MAXSEG)
    for i, (x, k, shift, koff) in enumerate(segs):
        if x.dtype != torch.bfloat16 or x.dim() != 2 or x.stride(1) != 1:
            raise ValueError("segment %d must be a row-major bf16 matrix" % i)
        s = d.seg[i]
        s.x, s.rows, s.ld, s.k, s.shift, s.koff = x.data_ptr(), int(x.shape[0]), int(x.stride(0)), int(k), int(shift), int(koff)
    d.nseg, d.M, d.N, d.Ti = len(segs), int(m), int(n), int(ti)
    d.W, d.ldw = w.data_ptr(), int(w.stride(0))
    d.bias = bias.data_ptr() if bias is not None else None
    if res is not None:
        d.R, d.ldr, d.rscale = res.data_ptr(), int(res.stride(0)), float(rscale)
    if mask is not None:
        d.mask, d.ldmask = mask.data_ptr(), int(mask.stride(0))
    d.relu, d.oscale = int(bool(relu)), float(oscale)
    if out is None:
        shape = (nsplit, m, n) if nsplit > 1 else (m, n)
        out = torch.empty(*shape, dtype=torch.float32 if out_f32 else torch.bfloat16, device=w.device)
    d.Y, d.ldy = out.data_ptr(), int(out.stride(-2))
    d.out_f32, d.accumulate, d.nsplit = int(out.dtype == torch.float32), int(bool(accumulate)), int(nsplit)
    d.split_stride = int(split_stride or (out.stride(0) if nsplit > 1 else 0))
    if gate is not None:
        aux, dpre, col0 = gate
        if aux.dtype != torch.bfloat16 or dpre.dtype != torch.bfloat16 or tuple(aux.shape) != (m, 512) or tuple(dpre.shape) != (m, 512) \
                or not (aux.is_contiguous() and dpre.is_contiguous()):
            raise ValueError("gate: aux and dpre must be contiguous bf16 (m, 512)")
        d.gate_aux, d.gate_out, d.gate_col0 = aux.data_ptr(), dpre.data_ptr(), int(col0)
    _lib.check(lib.fwn_gemm(C.byref(d), _stream(w)), "fwn_gemm")
    return out


def transpose_shift(x, m, c, *, shift=0, dshift=0, ntap=1, ti=0, ones_row=False, ld_dst=None):
    """``dst[z*c + c', m'] = x[m' + shift + z*dshift, c']`` for tap z < ntap (zero where the tap leaves its
    clip), bf16 ``[ntap*c (+1), ld_dst]``; ones_row appends a row of ones (for m' < m)."""
    import torch
    lib = _lib.load()
    ld_dst = ld_dst or (m + 63) // 64 * 64
    dst = torch.empty(ntap * c + (1 if ones_row else 0), ld_dst, dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.fwn_transpose_shift(x.data_ptr(), int(m), int(c), int(x.stride(0)), int(shift), int(dshift), int(ntap),
                                       int(ti), dst.data_ptr(), int(ld_dst), int(bool(ones_row)), _stream(x)), "fwn_transpose_shift")
    return dst


def reduce_splits(partial, scale=1.0):
    """``partial [S, ...] fp32 -> sum over S`` in a fixed order (the deterministic second pass of split-K)."""
    import torch
    lib = _lib.load()
    s = int(partial.shape[0])
    out = torch.empty(partial.shape[1:], dtype=torch.float32, device=partial.device)
    _lib.check(lib.fwn_reduce_splits(partial.data_ptr(), s, int(partial.stride(0)), out.numel(), float(scale),
                                     out.data_ptr(), _stream(partial)), "fwn_reduce_splits")
    return out


def weight_grad_partials(x, m, kx, dyt, n, *, shifts=(0,), ti=0, nsplit=None):
    """Split-K partials of ``dW[tap*kx + i, j] = sum_r x[r + shift_tap, i] * dy[r, j]`` with the bias
    gradient ``sum_r dy[r, j]`` as one more row: fp32 ``[S, len(shifts)*kx + 1, n]``.

    x: bf16 [m, >=kx]; dyt: ``transpose_shift(dy, m, n)`` (shared by every weight gradient that
    contracts with the same dy); shifts: an arithmetic progression."""
    mp = int(dyt.shape[1])
    rows = len(shifts) * kx
    dshift = shifts[1] - shifts[0] if len(shifts) > 1 else 0
    xt = transpose_shift(x, m, kx, shift=shifts[0], dshift=dshift, ntap=len(shifts), ti=ti, ones_row=True, ld_dst=mp)
    if nsplit is None:
        nchunks, n128 = mp // 64, (n + 127) // 128
        t256 = ((rows + 1 + 255) // 256) * n128
        if rows + 1 >= 512 and nchunks >= 64:      # long contraction: enough splits for the 256 x 128 tile to fill the chip
            nsplit = max(1, min(nchunks // 8, -(-200 // t256)))
        else:
            nsplit = max(1, min(nchunks, -(-256 // (((rows + 1 + 63) // 64) * n128))))
    part = gemm([(xt, mp, 0, 0)], dyt, n, rows + 1, out_f32=True, nsplit=nsplit)
    return part if nsplit > 1 else part[None]


def tn_weight_grad_partials(x, dy, m, kx, n, *, shifts=(0,), ti=0, nsplit=None, bias_row=True):
    """The same partials (bias row = column sums of dy, last) straight from x and dy as they lie in memory
    (``fwn_tn_gemm``: operands read transposed out of LDS): fp32 ``[S, len(shifts)*kx (+ 1), n]``.
    Needs 16-byte aligned rows: kx, n and both row strides multiples of 8."""
    import torch
    lib = _lib.load()
    rows = len(shifts) * kx
    dshift = shifts[1] - shifts[0] if len(shifts) > 1 else 0
    if nsplit is None:
        tiles = len(shifts) * ((kx + 127) // 128) * ((n + 127) // 128)
        nsplit = max(1, min((m + 63) // 64 // 4 or 1, -(-256 // tiles)))
    part = torch.empty(nsplit, rows + (1 if bias_row else 0), n, dtype=torch.float32, device=x.device)
    _lib.check(lib.fwn_tn_gemm(x.data_ptr(), int(x.stride(0)), int(kx), len(shifts), int(shifts[0]), int(dshift),
                               dy.data_ptr(), int(dy.stride(0)), int(n), int(m), int(ti), int(nsplit), part.data_ptr(),
                               int(part.stride(0)), int(bool(bias_row)), _stream(x)), "fwn_tn_gemm")
    return part


def tn_group_splits(specs, m):
    """Splits of the row range for a group of weight-gradient GEMMs ``(kx, n, ntap)``: as many as keep the whole
    group within ONE round of workgroups (a workgroup per CU), so every GEMM writes few partials."""
    e = int(_lib.load().fwn_tn_gemm_tile(int(m)))             # 128 x 128 or 256 x 256 output tiles at this m
    tiles = sum(ntap * ((kx + e - 1) // e) * ((n + e - 1) // e) for kx, n, ntap in specs)
    return max(1, min((m + 63) // 64, 256 // max(1, tiles)))


def tn_block_splits(specs, n_flow, m):
    """The split count ``fwn_train_loss_and_grads`` uses for the weight-gradient GEMMs ``specs`` of ONE flow: planned for the
    flow's block (csrc/train_api.hip ``tn_block_splits``) - the groups of all ``n_flow`` flows of a block run as one launch
    on the side stream, so a handful of splits fills the chip."""
    e = int(_lib.load().fwn_tn_gemm_tile(int(m)))
    tiles = n_flow * sum(ntap * ((kx + e - 1) // e) * ((n + e - 1) // e) for kx, n, ntap in specs)
    if m < 4096:                            # few rows: bound by writing the output tiles - one split
        return 1
    a = min((m + 63) // 64, 32)
    for n in range(1, a + 1):               # the fewest splits that fill the launch's rounds of 256 workgroups to 85 %
        w = tiles * n
        if 100 * w >= 85 * 256 * ((w + 255) // 256):
            return n
    return max(1, a)


def tn_weight_grad_group(jobs, m, ti=0, nsplit=None):
    """``jobs``: list of ``(x, dy, kx, n, shifts)`` sharing m and ti.  One launch (``fwn_tn_gemm_group``); returns the
    fp32 partials ``[S, len(shifts)*kx + 1, n]`` of every job (views of one buffer; last row = bias gradient)."""
    import torch
    lib = _lib.load()
    if len(jobs) > _lib.FWN_MAX_GROUP:
        return (tn_weight_grad_group(jobs[:_lib.FWN_MAX_GROUP], m, ti, nsplit) +
                tn_weight_grad_group(jobs[_lib.FWN_MAX_GROUP:], m, ti, nsplit))
    if nsplit is None:
        nsplit = tn_group_splits([(kx, n, len(sh)) for _, _, kx, n, sh in jobs], m)
    sizes = [(len(sh) * kx + 1) * n for _, _, kx, n, sh in jobs]
    buf = torch.empty(nsplit * sum(sizes), dtype=torch.float32, device=jobs[0][0].device)
    arr = (_lib.TnJob * len(jobs))()
    parts, off = [], 0
    for q, (x, dy, kx, n, sh), size in zip(arr, jobs, sizes):
        part = buf[off:off + nsplit * size].view(nsplit, len(sh) * kx + 1, n)
        off += nsplit * size
        parts.append(part)
        q.x, q.dy, q.part, q.split_stride = x.data_ptr(), dy.data_ptr(), part.data_ptr(), size
        q.ldx, q.Kx, q.ntap, q.shift0 = int(x.stride(0)), int(kx), len(sh), int(sh[0])
        q.dshift = int(sh[1] - sh[0]) if len(sh) > 1 else 0
        q.ldy, q.N, q.nsplit, q.bias_row = int(dy.stride(0)), int(n), int(nsplit), 1
    _lib.check(lib.fwn_tn_gemm_group(arr, len(jobs), int(m), int(ti), _stream(jobs[0][0])), "fwn_tn_gemm_group")
    return parts


def wn_backward_group(jobs):
    """``jobs``: list of dicts(part [S, rows, ldp], k, n, col0, bias_row (-1: none), scale, row_src, v, g, dv, dg, db):
    split reduction + weight-norm backward of all of them in two launches (``fwn_wn_backward_group``)."""
    import torch
    lib = _lib.load()
    for i0 in range(0, len(jobs), _lib.FWN_MAX_GROUP):
        chunk = jobs[i0:i0 + _lib.FWN_MAX_GROUP]
        arr = (_lib.WnJob * len(chunk))()
        for q, j in zip(arr, chunk):
            part = j["part"]
            q.part, q.split_stride, q.nsplit, q.ldp = part.data_ptr(), int(part.stride(0)), int(part.shape[0]), int(part.shape[2])
            q.row_src = j["row_src"].data_ptr() if j.get("row_src") is not None else None
            q.col_src = j["col_src"].data_ptr() if j.get("col_src") is not None else None
            q.col0, q.bias_row, q.K, q.N, q.scale = int(j.get("col0", 0)), int(j.get("bias_row", -1)), int(j["k"]), int(j["n"]), float(j.get("scale", 1.0))
            q.V = j["v"].data_ptr() if j.get("v") is not None else None
            q.g = j["g"].data_ptr() if j.get("g") is not None else None
            q.dV, q.dg = j["dv"].data_ptr(), j["dg"].data_ptr() if j.get("dg") is not None else None
            q.db = j["db"].data_ptr() if j.get("db") is not None else None
        scr = torch.empty(int(lib.fwn_wn_group_scratch(arr, len(chunk))), dtype=torch.float64, device=chunk[0]["part"].device)
        _lib.check(lib.fwn_wn_backward_group(arr, len(chunk), scr.data_ptr(), _stream(chunk[0]["part"])), "fwn_wn_backward_group")


def colsum_bf16(dy, m, n, scale=1.0, out=None):
    """``out[j] = scale * sum_r dy[r, j]`` (bias gradient), fp32 [n], fixed summation order."""
    import torch
    lib = _lib.load()
    out = torch.empty(n, dtype=torch.float32, device=dy.device) if out is None else out
    scr = torch.empty(lib.fwn_colsum_partials(m, n), dtype=torch.float32, device=dy.device)
    _lib.check(lib.fwn_colsum_bf16(dy.data_ptr(), int(m), int(n), int(dy.stride(0)), float(scale), scr.data_ptr(),
                                   out.data_ptr(), _stream(dy)), "fwn_colsum_bf16")
    return out


def weight_grad(x, dy, m, kx, n, *, shifts=(0,), ti=0, nsplit=None):
    """``(dW [len(shifts)*kx, n], db [n])`` fp32: the partials of ``weight_grad_partials`` summed in a fixed order."""
    dyt = transpose_shift(dy, m, n)
    part = weight_grad_partials(x, m, kx, dyt, n, shifts=shifts, ti=ti, nsplit=nsplit)
    full = reduce_splits(part) if part.shape[0] > 1 else part[0]
    rows = len(shifts) * kx
    return full[:rows], full[rows]


# =============================================================================================
# Loss and gradients of the whole model (train.py:56-66: loss = -(log_p + logdet)): ONE C call per batch,
# fwn_train_loss_and_grads (csrc/train_api.hip sequences the training forward and the backward over the stage
# kernels).  What stays here is parameter-side: the packed bf16 copies the GEMMs read (_TrainPack, refreshed from the
# fp32 masters every step by two grouped packing launches) and the descriptors that tell the C side where masters,
# copies and gradients live.
# =============================================================================================
import numpy as np

from . import packing, weights

SQH = float(np.sqrt(0.5))


class _TrainPack:
    """Inference packing plus the natural-order / transposed bf16 copies the backward GEMMs read."""

    def __init__(self, params, hp, device):
        import torch
        self.hp, self.dev = hp, torch.device(device)
        if self.dev.type == "cuda" and self.dev.index is None:      # "cuda" != "cuda:0" for torch: make it concrete
            self.dev = torch.device("cuda", torch.cuda.current_device())
        self.lib = _lib.load()
        self.params = params
        # parameters that are contiguous fp32 device tensors (the masters of a training run) have
        # stable addresses: record the packing once (packing.PackPlan) and refresh() it every step
        stable = all(isinstance(v, torch.Tensor) and v.device == self.dev and v.dtype == torch.float32 and v.is_contiguous()
                     for v in params.values())
        self.plan = packing.PackPlan(self.dev) if stable else None
        self.pm = packing.pack_model(params, hp, device, cond_mode=1, plan=self.plan)
        self.flows = {}
        self._idx = {}
        self._scale = torch.empty(512, dtype=torch.float32, device=self.dev)
        self.br = [self._i64(("br", i), packing.bitrev_table(i)) for i in range(hp.n_block)]
        half_ = hp.num_mels // 2
        # logical row of a weight gradient -> row of the GEMM that computed it in device channel order
        self.cond_rows = [self._i32(("cond_rows", i), np.argsort(packing.cond_src_k(i, half_)[:half_ * (2 << i)]))
                          for i in range(hp.n_block)]
        # logical row tap Ch + c of a front weight gradient -> row of the TN GEMM over y_a rows padded to max(Ch, 8) channels
        self.front_rows = [self._i32(("front_rows8", i), np.concatenate([tap * max(1 << i, 8) + packing.bitrev_table(i) for tap in range(3)]))
                           for i in range(hp.n_block)]
        self.csrc64 = [self._i64(("csrc", i), packing.cond_src_k(i, hp.num_mels // 2)[:(hp.num_mels // 2) * (2 << i)])
                       for i in range(hp.n_block)]
        for i in range(hp.n_block):
            for j in range(hp.n_flow):
                self.flows[(i, j)] = self._pack_flow_plan(i, j) if stable else self._pack_flow(i, j)
        if stable:
            self.plan.run_kernels()
            self.plan.enable_device_tables(params)      # masters in one flat vector: the small tables refresh on the device
        self._small_tables()

    def refresh(self):
        """The masters changed (an optimiser step): re-pack everything into the same buffers."""
        self.plan.refresh()
        self._small_tables()

    def _small_tables(self):
        """Per-flow bias / scale vectors in device channel order (parameter-sized; the backward reads ez).  With the
        masters in one flat vector all flows are done by a handful of batched gathers (``_batched_tables``)."""
        import torch
        hp = self.hp
        if self.plan is not None and self.plan._dev_ready:
            return self._batched_tables()
        for (i, j), t in self.flows.items():
            wp = weights.flow_prefix(i, j) + "/WaveNet"
            t["bskip"] = sum(self._f32("%s/ResBlock_%d/skip_conv/bias" % (wp, l)) for l in range(hp.n_layer))
            t["bfin"] = self._f32(wp + "/Conv_final/bias")
            t["bz"] = self._f32(wp + "/ZeroConv1d/bias").reshape(-1)[t["zcol"]].contiguous()
            t["ez"] = torch.exp(3.0 * self._f32(wp + "/ZeroConv1d/scale").reshape(-1))[t["zcol"]].contiguous()
        self.an_logdet = None

    def _batched_tables(self):
        """bskip (sum of the layers' skip biases), bz, ez = exp(3 scale) of every flow and the parameter-only part of
        logdet, sum over flows of mean_C(3 logs) (model.py:86-94): ONE ``fwn_gather_tables`` launch over the flat
        masters (fp32 arithmetic, the order of the framework expressions it replaced) + ``fwn_sum_f32``, into buffers
        allocated once (the descriptors of the C step keep pointing at them)."""
        import torch
        hp, P = self.hp, self.params
        flat = self.plan._flat
        lib = self.lib
        if getattr(self, "_bt", None) is None:          # gather tables into the flat master vector, built once
            off = lambda name: int(P[name].storage_offset())
            L = hp.n_layer
            skip = [[] for _ in range(L)]
            bz, ez, an3, anw, spans = [], [], [], [], {}
            zpos = 0
            for (i, j), t in self.flows.items():
                wp = weights.flow_prefix(i, j) + "/WaveNet"
                for l in range(L):
                    skip[l].append(off("%s/ResBlock_%d/skip_conv/bias" % (wp, l)) + np.arange(256))
                zc = t["zcol"].cpu().numpy()
                bz.append(off(wp + "/ZeroConv1d/bias") + zc)
                ez.append(off(wp + "/ZeroConv1d/scale") + zc)
                an3.append(off(weights.flow_prefix(i, j) + "/ActNorm/logs") + np.arange(2 << i))
                anw.append(np.full(2 << i, 3.0 / (2 << i)))          # sum over both planes of mean_C(3 logs)
                spans[(i, j)] = (zpos, zpos + len(zc))
                zpos += len(zc)
            skip = [np.concatenate(a) for a in skip]
            bz, ez, an3, anw = np.concatenate(bz), np.concatenate(ez), np.concatenate(an3), np.concatenate(anw)
            n_s, n_z, n_a = len(skip[0]), len(bz), len(an3)
            total = n_s + 2 * n_z + n_a
            idx = np.full((L, total), -1, dtype=np.int64)
            for l in range(L):
                idx[l, :n_s] = skip[l]
            idx[0, n_s:n_s + n_z] = bz
            idx[0, n_s + n_z:n_s + 2 * n_z] = ez
            idx[0, n_s + 2 * n_z:] = an3
            post = np.ones(total)
            post[n_s + n_z:n_s + 2 * n_z] = 3.0
            post[n_s + 2 * n_z:] = anw
            mode = np.full(total, 2, dtype=np.uint8)
            mode[n_s + n_z:n_s + 2 * n_z] = 3
            out = torch.empty(total, dtype=torch.float32, device=self.dev)
            self._bt = dict(idx=torch.from_numpy(idx).to(self.dev), post=torch.from_numpy(post).to(self.dev),
                            mode=torch.from_numpy(mode).to(self.dev), out=out, L=L, total=total, spans=spans,
                            bskip=out[:n_s], bz=out[n_s:n_s + n_z], ez=out[n_s + n_z:n_s + 2 * n_z], anp=out[n_s + 2 * n_z:],
                            an_logdet=torch.empty(1, dtype=torch.float32, device=self.dev))
            for f, ((i, j), t) in enumerate(self.flows.items()):
                wp = weights.flow_prefix(i, j) + "/WaveNet"
                lo, hi = spans[(i, j)]
                t["bskip"] = self._bt["bskip"][f * 256:(f + 1) * 256]
                t["bfin"] = self._f32(wp + "/Conv_final/bias")
                t["bz"], t["ez"] = self._bt["bz"][lo:hi], self._bt["ez"][lo:hi]
        bt = self._bt
        st = _stream(flat)
        _lib.check(lib.fwn_gather_tables(flat.data_ptr(), bt["idx"].data_ptr(), bt["L"], bt["total"], bt["post"].data_ptr(),
                                         bt["mode"].data_ptr(), bt["out"].data_ptr(), st), "fwn_gather_tables")
        self.an_logdet = None       # (unused since the forward half runs the inference tail: its log-det partials carry the ActNorm terms)

    def _pack_flow_plan(self, i, j):
        """The backward's transposed / natural-order copies as jobs of the plan (transposed packing:
        no separate transposes)."""
        import torch
        hp, dev, plan = self.hp, self.dev, self.plan
        ch, half, L = 1 << i, hp.num_mels // 2, hp.n_layer
        cin = half * (2 << i)
        wp = weights.flow_prefix(i, j) + "/WaveNet"
        bz = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=dev)
        id256 = self._i32("id256", np.arange(256))
        taps = [self._i32(("tap256", tap), tap * 256 + np.arange(256)) for tap in range(3)]
        csrc = self._i32(("csrc", i), packing.cond_src_k(i, half))
        br = packing.bitrev_table(i).astype(np.int64)
        fk = packing.front_src_k(i)
        P = self.params

        def job(name, src_k, k_dst, src_n, n_dst, out, col, transposed, weight_norm=True):
            plan.add(P[name + "/kernel"], P[name + "/g"] if weight_norm else None, src_k, src_n, k_dst, n_dst,
                     out.data_ptr() + 2 * col, int(out.stride(0)), 1.0, transposed)

        t = {}
        t["WfT"] = bz(ch, 768)
        for tap in range(3):
            job(wp + "/Conv_front", self._i32(("fk", i, tap), fk[tap * ch:(tap + 1) * ch]), ch, id256, 256, t["WfT"], tap * 256, True)
        t["WdT"], t["WcT"], t["WresT"], t["WskipT"] = [], [], [], []
        for l in range(L):
            rp = "%s/ResBlock_%d" % (wp, l)
            wdt = bz(256, 1536)
            for tap in range(3):
                job(rp + "/Conv_filter", taps[tap], 256, id256, 256, wdt, tap * 512, True)
                job(rp + "/Conv_gate", taps[tap], 256, id256, 256, wdt, tap * 512 + 256, True)
            t["WdT"].append(wdt)
            if l == 0:
                t["WcT_all"] = bz(cin, L * 512)          # the layers side by side: one K = L * 512 GEMM for the conditioning gradient
            wct = t["WcT_all"][:, l * 512:(l + 1) * 512]
            job(rp + "/filter_conv_c", csrc, cin, id256, 256, wct, 0, True)
            job(rp + "/gate_conv_c", csrc, cin, id256, 256, wct, 256, True)
            t["WcT"].append(wct)
            if l + 1 < L:
                wrt = bz(256, 256)
                job(rp + "/res_conv", id256, 256, id256, 256, wrt, 0, True)
                t["WresT"].append(wrt)
            if l == 0:
                t["WskipT_all"] = bz(L * 256, 256)              # the L transposed skip convs stacked along N: one GEMM for all do_l
            wst = t["WskipT_all"][l * 256:(l + 1) * 256]
            job(rp + "/skip_conv", id256, 256, id256, 256, wst, 0, True)
            t["WskipT"].append(wst)
        # (the forward half runs the inference tail on the inference packing: no natural-order Wskip / Wfin / Wz copies)
        t["WfinT"] = bz(256, 256)
        job(wp + "/Conv_final", id256, 256, id256, 256, t["WfinT"], 0, True)
        zcol = np.concatenate([br, ch + br])
        t["zcol"] = self._i64(("zcol", i), zcol)
        t["zinv32"] = self._i32(("zinv", i), np.argsort(zcol))
        zc32 = self._i32(("zcol", i), zcol)
        n2 = 2 * ch
        t["ldz"] = max(8, n2)
        t["WzT"] = bz(256, t["ldz"])
        job(wp + "/ZeroConv1d", id256, 256, zc32, n2, t["WzT"], 0, True, weight_norm=False)
        return t

    def _i32(self, key, arr):
        import torch
        gk = (str(self.dev), self.hp.num_mels, "train", key)
        if gk not in packing._IDX_CACHE:
            packing._IDX_CACHE[gk] = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int32)).to(self.dev)
        return packing._IDX_CACHE[gk]

    def _i64(self, key, arr):
        import torch
        gk = (str(self.dev), self.hp.num_mels, "train64", key)
        if gk not in packing._IDX_CACHE:
            packing._IDX_CACHE[gk] = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int64)).to(self.dev)
        return packing._IDX_CACHE[gk]

    def _f32(self, name):
        import torch
        v = self.params[name]
        if isinstance(v, torch.Tensor):
            if v.device == self.dev and v.dtype == torch.float32 and v.is_contiguous():
                return v
            return v.to(device=self.dev, dtype=torch.float32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(self.dev)

    def _pack(self, name, src_k, src_n, k_dst, n_dst, out, weight_norm=True):
        import torch
        v = self._f32(name + "/kernel")
        k_src, n_src = v.shape[0] * v.shape[1], v.shape[2]
        st = _stream(v)
        sc = None
        if weight_norm:
            g = self._f32(name + "/g")
            _lib.check(self.lib.fwn_wn_scale(v.data_ptr(), g.data_ptr(), k_src, n_src, self._scale.data_ptr(), st), "fwn_wn_scale")
            sc = self._scale.data_ptr()
        _lib.check(self.lib.fwn_pack_bf16(v.data_ptr(), sc, src_k.data_ptr(), src_n.data_ptr(), n_src, k_dst, n_dst,
                                          int(out.stride(0)), out.data_ptr(), st), "fwn_pack_bf16")

    def _pack_flow(self, i, j):
        import torch
        hp, dev = self.hp, self.dev
        ch, half, L = 1 << i, hp.num_mels // 2, hp.n_layer
        cin = half * (2 << i)
        kcpad = packing.roundup(cin, 64)
        wp = weights.flow_prefix(i, j) + "/WaveNet"
        bz = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=dev)
        id256 = self._i32("id256", np.arange(256))
        id768 = self._i32("id768", np.arange(768))
        rows_f = self._i32("rows_f", np.concatenate([np.arange(256), np.full(256, -1)]))
        rows_g = self._i32("rows_g", np.concatenate([np.full(256, -1), np.arange(256)]))
        csrc = self._i32(("csrc", i), packing.cond_src_k(i, half))
        br = packing.bitrev_table(i).astype(np.int64)
        t = {}
        # front: K = tap*Ch + tau (plane order)
        fk = packing.front_src_k(i)[:3 * ch]
        wf = bz(256, packing.roundup(3 * ch, 8))
        self._pack(wp + "/Conv_front", self._i32(("fk", i), np.concatenate([fk, np.full(wf.shape[1] - 3 * ch, -1)])), id256,
                   wf.shape[1], 256, wf)
        wft = transpose_shift(wf, 256, 3 * ch, ld_dst=256)                      # [3Ch][256]
        t["WfT"] = wft.view(3, ch, 256).permute(1, 0, 2).reshape(ch, 768).contiguous()       # [Ch][tap*256 + n]
        t["Wd"], t["WdT"], t["Wc"], t["WcT"], t["WresT"], t["WskipT"] = [], [], [], [], [], []
        for l in range(L):
            rp = "%s/ResBlock_%d" % (wp, l)
            wd = bz(512, 768)
            self._pack(rp + "/Conv_filter", id768, rows_f, 768, 512, wd)
            self._pack(rp + "/Conv_gate", id768, rows_g, 768, 512, wd)
            wdt = transpose_shift(wd, 512, 768, ld_dst=512)                     # [768][512]
            t["WdT"].append(wdt.view(3, 256, 512).permute(1, 0, 2).reshape(256, 1536).contiguous())
            wc = bz(512, kcpad)
            self._pack(rp + "/filter_conv_c", csrc, rows_f, kcpad, 512, wc)
            self._pack(rp + "/gate_conv_c", csrc, rows_g, kcpad, 512, wc)
            if l == 0:
                t["WcT_all"] = bz(cin, L * 512)
            t["WcT_all"][:, l * 512:(l + 1) * 512] = transpose_shift(wc, 512, cin, ld_dst=512)            # [cin][512] of layer l
            t["WcT"].append(t["WcT_all"][:, l * 512:(l + 1) * 512])
            if l + 1 < L:
                wr = bz(256, 256)
                self._pack(rp + "/res_conv", id256, id256, 256, 256, wr)
                t["WresT"].append(transpose_shift(wr, 256, 256, ld_dst=256))
            ws = bz(256, 256)
            self._pack(rp + "/skip_conv", id256, id256, 256, 256, ws)
            t["WskipT"].append(transpose_shift(ws, 256, 256, ld_dst=256))
        t["WskipT_all"] = torch.cat(t["WskipT"], 0).contiguous()
        wfin = bz(256, 256)
        self._pack(wp + "/Conv_final", id256, id256, 256, 256, wfin)
        t["WfinT"] = transpose_shift(wfin, 256, 256, ld_dst=256)
        # ZeroConv rows in plane order: row fg*Ch + tau serves logical channel fg*Ch + bitrev(tau)
        zcol = np.concatenate([br, ch + br])
        t["zcol"] = self._i64(("zcol", i), zcol)
        t["zinv32"] = self._i32(("zinv", i), np.argsort(zcol))
        n2 = 2 * ch
        ldz = max(8, n2)
        wz = bz(n2, 256)
        self._pack(wp + "/ZeroConv1d", id256, self._i32(("zcol", i), zcol), 256, n2, wz, weight_norm=False)
        t["WzT"] = transpose_shift(wz, n2, 256, ld_dst=packing.roundup(n2, 64))[:, :ldz].contiguous()   # [256][ldz], zero padded
        t["ldz"] = ldz
        return t


class GradEngine:
    """``loss_and_grads(params, x, c)``: one training forward + backward on the current device.

    params: dict name -> fp32 array / tensor in the reference's layouts (``weights.param_shapes``).
    Returns ``(loss, log_p, logdet, grads)`` with grads a dict name -> fp32 device tensor of the
    parameter's shape (``d loss / d param``, loss = -(log_p + logdet), train.py:60)."""

    def __init__(self, hparams, device="cuda"):
        if not hparams.affine or hparams.causality:
            raise NotImplementedError("affine=True, causality=False only")
        self.hp, self.device = hparams, device
        self.lib = _lib.load()
        self._gout = None
        self._on_block = None
        # True: the caller refreshes the host-computed tables itself (``refresh_host_tables``) - a recorded
        # step keeps that device -> host -> device round trip outside its hipGraph
        self.external_host_tables = False
        # True: no consumer of a block's gradients before the end of the call (fwn_train_desc.defer_block_done; slower, see _run)
        self.defer_block_done = False

    def refresh_host_tables(self):
        """Host-computed small tables (only when the parameters are not views of one flat device vector: then
        ``PackPlan.refresh_tables_device`` does it inside the step)."""
        tp = getattr(self, "_tp", None)
        if tp is None or tp.plan is None:
            raise RuntimeError("no recorded packing plan: run one eager step on device-resident parameters first")
        if tp.plan._dev_ready:
            return
        tp.plan.hostview.reset()
        tp.plan.upload_tables()

    # ------------------------------------------------------------------ helpers
    def _side_stream(self, dev):
        """The second stream of a step (FWN_TRAIN_SIDE=0: none): each block's weight gradients run on it under the next
        block's backward chain."""
        import torch
        if os.environ.get("FWN_TRAIN_SIDE", "1") == "0":
            return None
        if getattr(self, "_side", None) is None:            # an engine serves one device
            prio = os.environ.get("FWN_TRAIN_SIDE_PRIO")       # developer switch (same-box A/B): HIP stream priority
            self._side = torch.cuda.Stream(torch.device(dev), priority=int(prio)) if prio else torch.cuda.Stream(torch.device(dev))
        return self._side

    def _call(self, name, *args):
        _lib.check(getattr(self.lib, name)(*args), name)

    def loss_and_grads(self, params, x, c, grad_out=None, on_block_done=None):
        """grad_out: optional dict name -> fp32 tensor (e.g. views of a flat gradient buffer): gradients are
        written there (the large ones in place) and returned as those very tensors.  on_block_done(i) is
        called when every gradient of block i is in grad_out (blocks finish last to first; -1 = the
        up-sampling convs, at the end) - the hook a data-parallel step uses to start that block's
        all-reduce under the rest of the backward pass."""
        self._gout = grad_out
        self._on_block = on_block_done
        try:
            return self._loss_and_grads(params, x, c)
        finally:
            _STREAMS.clear()
            self._gout = None
            self._on_block = None

    def _loss_and_grads(self, params, x, c):
        import torch
        hp, lib = self.hp, self.lib
        dev = torch.device(self.device)
        st = torch.cuda.current_stream(dev).cuda_stream
        _STREAMS.clear()
        _STREAMS[torch.zeros(0, device=dev).device] = st
        shp = self._shapes = weights.param_shapes(hp)
        key = tuple(v.data_ptr() for v in list(params.values())[:4]) if all(hasattr(v, "data_ptr") for v in list(params.values())[:4]) else None
        tp = getattr(self, "_tp", None)
        if tp is not None and tp.plan is not None and key is not None and getattr(self, "_tp_key", None) == key:
            tp.params = params
            if tp.plan._dev_ready:
                tp.plan.refresh_tables_device()
            elif not self.external_host_tables:
                self.refresh_host_tables()
            tp.plan.run_kernels()
            tp._small_tables()
        else:
            tp = self._tp = _TrainPack(params, hp, self.device)
            self._tp_key = key if tp.plan is not None else None
        x = torch.as_tensor(x).to(device=dev, dtype=torch.float32).contiguous()
        c = torch.as_tensor(c).to(device=dev, dtype=torch.float32).contiguous()
        B, T = int(x.shape[0]), int(x.shape[1])
        if T % (1 << hp.n_block) or int(c.shape[1]) * hp.hop_size != T:
            raise ValueError("bad shapes")
        return self._run(tp, params, x.reshape(B, T), c, B, T, st)

    def _run(self, tp, params, x, c, B, T, st):
        """One call of ``fwn_train_loss_and_grads`` (csrc/train_api.hip sequences forward and backward): this method only
        fills the descriptors - packed copies, masters, gradient destinations, per-step tables - and lends a workspace."""
        import torch
        hp, lib, dev = self.hp, self.lib, x.device
        L = hp.n_layer
        shp = self._shapes
        go = self._gout
        if go is None:          # no caller-owned gradient buffers: one fresh contiguous tensor per parameter
            go = {k: torch.empty(v, dtype=torch.float32, device=dev) for k, v in shp.items()}
        elif not all(go[k].is_contiguous() for k in shp):
            raise ValueError("grad_out tensors must be contiguous")
        masters = {k: tp._f32(k) for k in shp}                   # device fp32 copies (the masters themselves when they qualify)
        ptr = lambda t_: t_.data_ptr() if t_ is not None else None
        key = (tuple(masters[k].data_ptr() for k in shp), tuple(go[k].data_ptr() for k in shp))
        if getattr(self, "_desc_key", None) != key:
            nfl = hp.n_block * hp.n_flow
            flows = (_lib.FlowTrainDesc * nfl)()
            td = _lib.TrainDesc()
            keep = []

            def conv(cg, name, weight_norm=True):
                cg.V, cg.dV, cg.db = ptr(masters[name + "/kernel"]), ptr(go[name + "/kernel"]), ptr(go[name + "/bias"])
                if weight_norm:
                    cg.g, cg.dg = ptr(masters[name + "/g"]), ptr(go[name + "/g"])

            for i in range(hp.n_block):
                for j in range(hp.n_flow):
                    f, t = flows[i * hp.n_flow + j], tp.flows[(i, j)]
                    fp = weights.flow_prefix(i, j)
                    wp = fp + "/WaveNet"
                    f.WfT, f.WskipT_all = ptr(t["WfT"]), ptr(t["WskipT_all"])
                    f.WfinT, f.WzT, f.ldz = ptr(t["WfinT"]), ptr(t["WzT"]), int(t["ldz"])
                    f.wct_ld = int(t["WcT_all"].stride(0))
                    for l in range(L):
                        rp = "%s/ResBlock_%d" % (wp, l)
                        f.WdT[l], f.WcT[l] = ptr(t["WdT"][l]), ptr(t["WcT"][l])
                        if l + 1 < L:
                            f.WresT[l] = ptr(t["WresT"][l])
                        conv(f.filt[l], rp + "/Conv_filter"); conv(f.gate[l], rp + "/Conv_gate")
                        conv(f.res[l], rp + "/res_conv"); conv(f.skip[l], rp + "/skip_conv")
                        conv(f.filt_c[l], rp + "/filter_conv_c"); conv(f.gate_c[l], rp + "/gate_conv_c")
                    conv(f.front, wp + "/Conv_front"); conv(f.final_, wp + "/Conv_final"); conv(f.zero, wp + "/ZeroConv1d", weight_norm=False)
                    f.d_an_b, f.d_an_logs = ptr(go[fp + "/ActNorm/b"]), ptr(go[fp + "/ActNorm/logs"])
                    f.d_zscale = ptr(go[wp + "/ZeroConv1d/scale"])
                td.cond_rows[i], td.front_rows[i] = ptr(tp.cond_rows[i]), ptr(tp.front_rows[i])
                td.zinv32[i], td.br[i] = ptr(tp.flows[(i, 0)]["zinv32"]), ptr(tp.br[i])
                td.zcol[i] = ptr(tp.flows[(i, 0)]["zcol"])
            for n in range(len(hp.upsample_scales)):
                td.up_bias_dev[n] = ptr(masters["upsample_%d/bias" % n])
                u = td.up[n]
                u.V, u.g = ptr(masters["upsample_%d/kernel" % n]), ptr(masters["upsample_%d/g" % n])
                u.dV, u.dg, u.db = ptr(go["upsample_%d/kernel" % n]), ptr(go["upsample_%d/g" % n]), ptr(go["upsample_%d/bias" % n])
            td.model = C.pointer(tp.pm.model_desc)
            td.flows = C.cast(flows, C.POINTER(_lib.FlowTrainDesc))
            td.zero_dead_res = 1            # first call on these gradient buffers: the dead res_conv gradients are zeroed once
            self._desc, self._desc_flows, self._desc_key = td, flows, key
        else:
            td, flows = self._desc, self._desc_flows
        # per-step tables (recomputed from the masters before every step: their addresses may move)
        # fwn_train_desc.an_logdet: unused by the library since round 3 (any device pointer)
        an_ld = getattr(self, "_an_ld0", None)
        if an_ld is None or an_ld.device != dev:
            an_ld = self._an_ld0 = torch.zeros(1, dtype=torch.float32, device=dev)
        for i in range(hp.n_block):
            for j in range(hp.n_flow):
                f, t = flows[i * hp.n_flow + j], tp.flows[(i, j)]
                tabs = [t[k] if t[k].is_contiguous() else t[k].contiguous() for k in ("bskip", "bfin", "bz", "ez")]
                f.bskip, f.bfin, f.bz, f.ez = (x_.data_ptr() for x_ in tabs)
                t["_tabs_alive"] = tabs
        td.an_logdet = an_ld.data_ptr()
        # the weight gradients of a block run on a second stream under the next block's chain (fwn.h fwn_train_desc.side_stream)
        side = self._side_stream(dev)
        td.side_stream = side.cuda_stream if side is not None else None
        # fwn_train_desc.defer_block_done: with nothing to exchange the data-gradient chain need not wait for the side
        # stream block by block.  Measured (round 4, same box, recorded step at 8 x 6400): 14.30 ms against 13.48 with
        # the per-block joins - without them the chain's small launches queue behind ever more chip-filling weight-gradient
        # GEMMs - so it stays off; FWN_TRAIN_DEFER=1 turns it on (same-box A/B)
        defer = os.environ.get("FWN_TRAIN_DEFER")
        td.defer_block_done = int(defer) if defer in ("0", "1") else int(bool(self.defer_block_done))
        wkey = (B, T, str(dev), bool(td.side_stream))
        if getattr(self, "_ws_key", None) != wkey:
            need = int(lib.fwn_train_workspace_bytes(C.byref(td), B, T))
            if need == 0:
                _lib.check(-1, "fwn_train_workspace_bytes")
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
            self._ws_key = wkey
        ws = self._ws
        off = (-ws.data_ptr()) % 256
        out3 = torch.empty(3, dtype=torch.float32, device=dev)
        hook, failure = self._on_block, []

        def block_done(user, blk):
            # ctypes prints and DROPS an exception raised inside a callback: a failed all-reduce or graph cut would go
            # unnoticed and the optimiser would step on un-reduced gradients.  Catch it, make the C sequencer stop
            # (non-zero return) and re-raise once the call is back.
            try:
                if hook is not None:
                    hook(blk)
                return 0
            except BaseException as e:
                failure.append(e)
                return 1

        cb = _lib.BLOCK_DONE_FN(block_done)
        rc = lib.fwn_train_loss_and_grads(C.byref(td), B, T, x.data_ptr(), c.data_ptr(), ws.data_ptr() + off, ws.numel() - off,
                                          out3.data_ptr(), cb, None, st)
        if failure:
            raise failure[0]
        _lib.check(rc, "fwn_train_loss_and_grads")
        td.zero_dead_res = 0
        self._alive = (masters, an_ld, x, c)           # until the stream has consumed them
        return out3[0], out3[1], out3[2], {k: go[k] for k in shp}


class Trainer:
    """One rank of the data-parallel training step (train.py:35-83 ``build_model`` + the session loop
    body): gradients of -(log_p + logdet) on this rank's batch (``GradEngine``), then
    ``optim.DataParallelAdam`` = RCCL all-reduce of the flat gradient, global-norm clip, Adam on the
    fp32 masters.  ``ddi`` performs the ActNorm data-dependent init of the first step
    (train.py:221,229 with init=True)."""

    def __init__(self, hparams, params, device="cuda", group=None, graph=None):
        """graph: record the step into hipGraphs after one eager step per input shape and replay it from then on
        (the step is ~5700 launches; replaying removes their host cost).  Default: on for GPU devices, off if
        FWN_TRAIN_GRAPH=0."""
        import os
        from .optim import DataParallelAdam
        self.hp, self.device = hparams, device
        self.opt = DataParallelAdam(hparams, params, device, group=group, grad_reduce_dtype=getattr(hparams, "grad_reduce_dtype", "fp32"),
                                    exchange=getattr(hparams, "grad_exchange", "allreduce"))
        self.engine = GradEngine(hparams, device)
        if graph is None:
            graph = os.environ.get("FWN_TRAIN_GRAPH", "1") != "0"
        self.graph = bool(graph)
        self._recorded = {}
        # False: skip the gradient exchange (bench.py's compute-only step time, to price the overlap)
        self.exchange = True

    def ddi(self, x, c):
        """ActNorm data-dependent init from the GLOBAL batch (train.py:221,229 with init=True): each rank pushes its
        shard through the flows and the per-channel moments are all-reduced flow by flow (``FloWaveNet`` with
        ``group``), so every rank derives bit-identical b / logs - the single-process result on the concatenated
        batch up to summation order.  (The reference lets its towers race on this assign: SURVEY section 2.1 C2.)"""
        from .model import FloWaveNet
        import torch
        views = self.opt.master_views()
        m = FloWaveNet(self.hp, init=True, device=self.device, cond_mode=1, group=self.opt.group).load_params(views)
        xx = torch.as_tensor(x).to(self.device)
        m.forward(xx.reshape(xx.shape[0], -1, 1), torch.as_tensor(c).to(self.device))
        for k, v in m.export_actnorm().items():
            views[k].copy_(torch.as_tensor(v).to(self.device).reshape(views[k].shape))
        # Only the ActNorm tables were made identical above (moment all-reduce).  Any other per-rank difference in the
        # initial masters - caller-supplied params, another seed, a restore that went differently - would survive every
        # all-reduced step: check once, here, that the ranks start in lock-step.
        if not self.opt.weights_identical():
            raise RuntimeError("Trainer.ddi: the ranks hold different master weights after the data-dependent init "
                               "(different initial parameters or checkpoints per rank?)")

    def step(self, x, c):
        """-> (loss, log_p, logdet, grad_norm) device scalars; the masters are updated in place."""
        if self.graph and self.opt.exchange == "allreduce":     # (the sharded exchange is eager only: optim.record_update)
            return self._step_recorded(x, c)
        return self._step_eager(x, c)

    def _ranges(self):
        return {("upsample" if key == "upsample" else int(key.split("_")[1])): (lo, hi) for key, lo, hi in self.opt.block_ranges()}

    def _step_recorded(self, x, c):
        """First call at a shape: eager (creates the packing plan, fills every cache).  Second: record.  The
        recording is a chain of hipGraphs cut where a block's gradients are final, so that with more than one
        rank each block's all-reduce still starts between two replays, under the rest of the backward pass;
        the host-computed tables and the Adam rate are refreshed before the replay."""
        import torch
        import torch.distributed as dist
        dev = torch.device(self.device)
        x = torch.as_tensor(x)
        c = torch.as_tensor(c)
        key = (tuple(x.shape), tuple(c.shape))
        rec = self._recorded.get(key)
        if rec is None:
            self._recorded[key] = "warm"
            return self._step_eager(x, c)
        world = dist.get_world_size(self.opt.group) if dist.is_available() and dist.is_initialized() else 1
        if rec == "warm":
            xs = torch.empty(tuple(x.shape), dtype=torch.float32, device=dev)
            cs = torch.empty(tuple(c.shape), dtype=torch.float32, device=dev)
            params, gv = self.opt.master_views(), self.opt.grad_views()
            segs, pool = [], torch.cuda.graph_pool_handle()
            mprio = os.environ.get("FWN_TRAIN_MAIN_PRIO")     # developer switch (same-box A/B)
            side = torch.cuda.Stream(dev, priority=int(mprio)) if mprio else torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            self.engine.external_host_tables = True
            cur = [None]
            try:
                with torch.cuda.stream(side):
                    cur[0] = torch.cuda.CUDAGraph()
                    # thread_local: other threads (the RCCL watchdog of a data-parallel run polls events) may call
                    # HIP while this thread records
                    cur[0].capture_begin(pool=pool, capture_error_mode="thread_local")

                    def cut(i):
                        if world == 1 and i >= 0 and not self.opt.force_collectives:
                            return
                        if i == -1 and segs and segs[-1][1] == 0:
                            # fwn_train_loss_and_grads reports block 0 and then the up-sampling convs (-1) back to back - block 0 is
                            # joined behind the up-sampling backward since round 5, nothing is enqueued between the two hooks
                            # (csrc/train_api.hip: join_pending, then the -1 hook) - so there is nothing to cut: no empty
                            # hipGraph segment, only the exchange of the up-sampling range to start
                            segs.append((None, -1))
                            return
                        cur[0].capture_end()
                        segs.append((cur[0], i))
                        cur[0] = torch.cuda.CUDAGraph()
                        cur[0].capture_begin(pool=pool, capture_error_mode="thread_local")

                    loss, log_p, logdet, _ = self.engine.loss_and_grads(params, xs, cs, grad_out=gv, on_block_done=cut)
                    gnorm = self.opt.record_update()
                    cur[0].capture_end()
                    segs.append((cur[0], None))
            except Exception as e:      # recording is an optimisation: a runtime that refuses it must not stop training
                import warnings
                try:
                    if cur[0] is not None:
                        cur[0].capture_end()
                except Exception:
                    pass
                torch.cuda.synchronize(dev)
                warnings.warn("recording the training step failed (%s: %s); continuing with eager steps" % (type(e).__name__, e))
                self.graph = False
                self.engine.external_host_tables = False
                return self._step_eager(x, c)
            finally:
                self.engine.external_host_tables = False
            torch.cuda.current_stream(dev).wait_stream(side)
            rec = self._recorded[key] = dict(xs=xs, cs=cs, segs=segs, out=(loss, log_p, logdet, gnorm))
        rec["xs"].copy_(x.reshape(rec["xs"].shape), non_blocking=True)
        rec["cs"].copy_(c, non_blocking=True)
        self.engine.refresh_host_tables()
        self.opt.advance()
        ranges, works = self._ranges(), []
        for g, i in rec["segs"]:
            if i is None:                       # the optimiser: every exchange must have landed
                for w in works:
                    if w is not None:
                        w.wait()
            if g is not None:
                g.replay()
            if i is not None and self.exchange:
                lo, hi = ranges["upsample" if i < 0 else i]
                works.append(self.opt.allreduce_range(lo, hi))
        return rec["out"]

    def _step_eager(self, x, c):
        params = self.opt.master_views()
        gv = self.opt.grad_views()
        ranges = {("upsample" if key == "upsample" else int(key.split("_")[1])): (lo, hi) for key, lo, hi in self.opt.block_ranges()}
        works = []

        def block_done(i):      # block i's gradients are final: its all-reduce runs under the remaining backward
            if self.exchange:
                lo, hi = ranges["upsample" if i < 0 else i]
                works.append(self.opt.allreduce_range(lo, hi))

        loss, log_p, logdet, grads = self.engine.loss_and_grads(params, x, c, grad_out=gv, on_block_done=block_done)
        gnorm = self.opt.step(works=works)
        return loss, log_p, logdet, gnorm
