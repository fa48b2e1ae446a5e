"""Host logic for K10: index tables that map the reference's parameter layouts
(TF ``(k, C_in, C_out)`` kernels, logical squeezed channel order) onto the device
layouts the HIP kernels read, and the driver that runs the packing kernels.

Plane algebra (DESIGN.md "Data layout in HBM"):
  * n squeezes turn time offset ``tau`` (0..2^n-1) inside a row into canonical channel
    ``bitrev_n(tau)``; the even/odd planes are the two channel halves, and inside a plane
    natural position ``tau'`` is logical channel ``bitrev_{n-1}(tau')``;
  * ``change_order`` (model.py:166-174) only flips which plane plays ``in_a``;
  * the conditioning half ``c_a`` is a mel-bin half at all phases: natural K index
    ``tau*half + m'`` is logical channel ``m'*2^n + bitrev_n(tau)``.

Everything in this file except ``pack_model`` is pure NumPy and runs without a GPU.
"""
from __future__ import annotations

import ctypes as C
import os
import math

import numpy as np

from . import _lib
from .weights import FILTER, flow_prefix

GATE_N = 2 * FILTER  # filter + gate output channels
# tanh(f) * sigmoid(g) = (1 - a) / ((1 + a)(1 + b)), a = 2^(-2 log2(e) f), b = 2^(-log2(e) g):
# the packed filter / gate rows and biases are pre-multiplied by these (csrc/common.h gated_unit2).
GATE_MUL = (-2.0 * 1.4426950408889634, -1.4426950408889634)


def bitrev(v: int, nbits: int) -> int:
    out = 0
    for _ in range(nbits):
        out = (out << 1) | (v & 1)
        v >>= 1
    return out


def bitrev_table(nbits: int) -> np.ndarray:
    return np.array([bitrev(v, nbits) for v in range(1 << nbits)], dtype=np.int32)


def roundup(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def gate_row_channel() -> "tuple[np.ndarray, np.ndarray]":
    """Packed gate row n' = nb*128 + wn*64 + fg*32 + j -> (fg, channel nb*64 + wn*32 + j)."""
    n = np.arange(GATE_N)
    nb, wn, fg, j = n // 128, (n // 64) % 2, (n // 32) % 2, n % 32
    return fg.astype(np.int32), (nb * 64 + wn * 32 + j).astype(np.int32)


def front_src_k(block: int) -> np.ndarray:
    """[kfpad] -> source row of the (3*Ch, 256) front kernel, -1 = K padding."""
    ch = 1 << block
    br = bitrev_table(block)
    out = np.full(roundup(3 * ch, 64), -1, dtype=np.int32)
    for tap in range(3):
        out[tap * ch: (tap + 1) * ch] = tap * ch + br
    return out


def front2_src_k(block: int) -> np.ndarray:
    """[6*Ch] -> source row of the (3*Ch, 256) front kernel for the hi|lo layout used when Ch >= 32:
    K = tap*2Ch + half*Ch + tau (the same weight serves the hi and the lo bf16 half of x)."""
    ch = 1 << block
    br = bitrev_table(block)
    out = np.empty(6 * ch, dtype=np.int32)
    for tap in range(3):
        for half in range(2):
            out[tap * 2 * ch + half * ch: tap * 2 * ch + (half + 1) * ch] = tap * ch + br
    return out


def front2_pad32_src_k(block: int) -> np.ndarray:
    """Ch = 16 in the Ch = 32 form of ``front2_src_k`` (front_mfma_kernel<32, 16>: K = 96 does not divide into 64-wide chunks
    per tap): [192], K = tap*64 + half*32 + tau with tau >= 16 zero padding (-1)."""
    ch = 1 << block
    assert ch == 16
    br = bitrev_table(block)
    out = np.full(6 * 32, -1, dtype=np.int32)
    for tap in range(3):
        for half in range(2):
            out[tap * 64 + half * 32: tap * 64 + half * 32 + ch] = tap * ch + br
    return out


def front3_src_k(block: int) -> np.ndarray:
    """[kf3] -> source row of the (3*Ch, 256) front kernel for the chained front conv of tail_chain.h (Ch <= 8):
    K = (tap*Ch + tau)*2 + half - the same weight serves the hi and the lo bf16 half of the fp32 state, and the pair sits
    in adjacent K positions so a lane builds its 8-element MFMA operand from 4 consecutive (tap, tau) inputs.
    kf3 = 6*Ch rounded up to the MFMA k-step of 16 (16, 16, 32, 48 for Ch = 1, 2, 4, 8); -1 = K padding."""
    ch = 1 << block
    br = bitrev_table(block)
    out = np.full(roundup(6 * ch, 16), -1, dtype=np.int32)
    for tap in range(3):
        for tau in range(ch):
            for half in range(2):
                out[(tap * ch + tau) * 2 + half] = tap * ch + br[tau]
    return out


def cond_src_k(block: int, half: int) -> np.ndarray:
    """[kcpad] -> source row of the (cin, 256) conditioning kernel for natural K = tau*half + m'."""
    n = block + 1
    p = 1 << n
    cin = half * p
    br = bitrev_table(n)
    out = np.full(roundup(cin, 64), -1, dtype=np.int32)
    tau, m = np.divmod(np.arange(cin), half)
    out[:cin] = m * p + br[tau]
    return out


def zero_src_n(block: int) -> np.ndarray:
    """[npt*64] -> source column of the (256, C) ZeroConv kernel; -1 = unused pad row.

    Row pt*64 + fg*32 + j serves natural position tau' = pt*32 + j; fg 0 = log_s, 1 = t
    (model.py:133 split order)."""
    ch = 1 << block
    npt = max(1, (ch + 31) // 32)
    br = bitrev_table(block)
    out = np.full(npt * 64, -1, dtype=np.int32)
    for pt in range(npt):
        for j in range(32):
            tau = pt * 32 + j
            if tau < ch:
                out[pt * 64 + j] = br[tau]
                out[pt * 64 + 32 + j] = ch + br[tau]
    return out


def acc_k_perm(k: int = FILTER) -> np.ndarray:
    """Row order of the weights whose OUTPUT feeds the next MFMA chain straight from the accumulator
    registers (tail kernel: skip sum -> final conv -> ZeroConv).  A 32x32 fp32 accumulator tile converted to
    bf16 serves as the B operand of the next v_mfma_f32_32x32x16_bf16 with element j of lane
    half h of k-step s holding tile row 16s + 8(j>>2) + 4h + (j&3); the next A-operand
    fragment is 8 contiguous packed columns k' = tile*32 + s*16 + h*8 + j.  Packing output channel
    acc_k_perm(n') (= n' with bits 2 and 3 swapped, an involution) into weight row n' makes those 8 register
    elements 8 CONSECUTIVE channels: S = ReLU(skip sum) and U = ReLU(final conv) exist in natural channel order
    (K axes of Wfinal / Wzero natural; the copies of S and U the training step keeps are natural too)."""
    kp = np.arange(k)
    s, h, j = (kp >> 4) & 1, (kp >> 3) & 1, kp & 7
    return ((kp & ~31) + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)).astype(np.int32)


def actnorm_table(b: np.ndarray, logs: np.ndarray, block: int) -> np.ndarray:
    """ActNorm (b, logs) in logical channel order -> an[2][4][Ch] (shift, scale, 1/scale, 3*logs)."""
    ch = 1 << block
    br = bitrev_table(block).astype(np.int64)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    l3 = 3.0 * np.asarray(logs, dtype=np.float64).reshape(-1)
    out = np.empty((2, 4, ch), dtype=np.float64)
    for role in range(2):
        idx = role * ch + br
        out[role, 0] = b[idx]
        out[role, 1] = np.exp(l3[idx])
        out[role, 2] = np.exp(-l3[idx])
        out[role, 3] = l3[idx]
    return out.astype(np.float32)


def actnorm_from_table(an: np.ndarray, block: int):
    """Inverse of ``actnorm_table``: an[2][4][Ch] -> (b, logs) in the reference's (1,1,C) layout."""
    ch = 1 << block
    br = bitrev_table(block).astype(np.int64)
    b = np.empty(2 * ch, dtype=np.float32)
    logs = np.empty(2 * ch, dtype=np.float32)
    for role in range(2):
        b[role * ch + br] = an[role, 0]
        logs[role * ch + br] = an[role, 3] / 3.0
    return b.reshape(1, 1, -1), logs.reshape(1, 1, -1)


def upsample_kernel(params, n: int) -> "tuple[np.ndarray, float]":
    """Weight-normed (2s,3) kernel of upsampling stage n: l2_normalize over axis [0,2]
    (convolutional.py:186) times g; returns (kernel fp32 [2s][3], bias)."""
    v = np.asarray(params["upsample_%d/kernel" % n], dtype=np.float64)
    nrm = np.sqrt(np.maximum(np.sum(v * v, axis=(0, 2), keepdims=True), 1e-12))
    w = v / nrm * float(np.asarray(params["upsample_%d/g" % n]).reshape(-1)[0])
    return np.ascontiguousarray(w[:, :, 0, 0], dtype=np.float32), float(
        np.asarray(params["upsample_%d/bias" % n]).reshape(-1)[0])


# ---------------------------------------------------------------------------------------------
# Device side
# ---------------------------------------------------------------------------------------------
_IDX_CACHE = {}


class PackPlan:
    """Recorded packing work for parameters that live in device memory at stable addresses (the fp32
    masters of a training run): ``refresh()`` re-packs every weight with two grouped launches
    (``fwn_pack_jobs``) and re-uploads the small host-computed tables in one copy, into the same
    device buffers - the descriptors built at record time stay valid."""

    def __init__(self, dev):
        self.dev = dev
        self.sjobs, self.slot, self.jobs = [], {}, []
        self.tables, self.recipes, self.host_cbs, self.keep = [], [], [], []
        self._dev_ready = False
        self.hostview = None
        self._built = False
        self.tail_jobs, self._tail_L = [], 0       # (Wskip, Wfinal, out) of the flows whose tail stream follows the weights

    def add(self, v, g, src_k, src_n, k_dst, n_dst, out_ptr, ld_dst, mul=1.0, transposed=False):
        slot = -1
        if g is not None:
            key = (v.data_ptr(), g.data_ptr())
            if key not in self.slot:
                self.slot[key] = len(self.sjobs)
                self.sjobs.append((v.data_ptr(), g.data_ptr(), int(v.shape[0] * v.shape[1]), int(v.shape[2])))
            slot = self.slot[key]
        self.jobs.append((v.data_ptr(), src_k.data_ptr(), src_n.data_ptr(), int(out_ptr), int(ld_dst), int(v.shape[2]),
                          int(k_dst), int(n_dst), slot, int(bool(transposed)), float(mul)))
        self.keep += [v, g, src_k, src_n]
        self._built = False

    def add_tail_stream(self, wskip_ptr, wfinal_ptr, out_ptr, L):
        """The fragment-order copy of a flow's Wskip | Wfinal (csrc/tail_rs.h) is derived from two packed matrices: it is
        re-packed behind the grouped weight packing of every refresh (``fwn_pack_tail_stream_jobs``: one launch for all)."""
        self.tail_jobs.append((int(wskip_ptr), int(wfinal_ptr), int(out_ptr)))
        self._tail_L = int(L)
        self._built = False

    def table(self, setter, fn, recipe=None):
        """recipe (optional): how the table follows from the parameters, for the on-device refresh -
        ``dict(terms=[(param name, index array or -1 per element), ..], post=scalar or array, exp=bool or array)``:
        table = F(post * sum_t param_t[index_t]) with F = exp where flagged; or ``dict(upsample=n)``."""
        self.tables.append((setter, fn))
        self.recipes.append(recipe)

    def enable_device_tables(self, params):
        """Build the gather tables of the on-device refresh (``refresh_tables_device``).  Needs every table to carry
        a recipe and every parameter to be a view of ONE flat fp32 device vector (the optimiser's masters)."""
        import torch
        self._dev_ready = False
        if getattr(self, "_tbuf", None) is None or any(r is None for r in self.recipes):
            return False
        vals = list(params.values())
        base = vals[0]._base if isinstance(vals[0], torch.Tensor) else None
        if base is None or base.dim() != 1 or base.dtype != torch.float32 or any(
                not isinstance(v, torch.Tensor) or v._base is not base or not v.is_contiguous() for v in vals):
            return False
        total = int(self._tbuf.numel())
        nterm = max(len(r.get("terms", ())) for r in self.recipes)
        idx = np.full((nterm, total), -1, dtype=np.int64)
        post = np.ones(total, dtype=np.float64)
        mode = np.zeros(total, dtype=np.uint8)
        self._up_jobs = []
        for (off, size), r in zip(self._toffs, self.recipes):
            if "upsample" in r:
                self._up_jobs.append((r["upsample"], off, size))
                continue
            for t, (name, ix) in enumerate(r["terms"]):
                ix = np.broadcast_to(np.asarray(ix, dtype=np.int64).reshape(-1), (size,))
                idx[t, off:off + size] = np.where(ix >= 0, int(params[name].storage_offset()) + ix, -1)
            post[off:off + size] = np.broadcast_to(np.asarray(r.get("post", 1.0), dtype=np.float64).reshape(-1), (size,))
            mode[off:off + size] = np.broadcast_to(np.asarray(r.get("exp", False), dtype=bool).reshape(-1), (size,))
        dev = self.dev
        self._flat = base
        self._didx = torch.from_numpy(np.ascontiguousarray(idx)).to(dev)
        self._dpost = torch.from_numpy(post).to(dev)
        self._dmode = torch.from_numpy(mode).to(dev)
        self._nterm, self._ttotal = nterm, total
        self._up_params = params
        self._dev_ready = True
        return True

    def refresh_tables_device(self):
        """The small tables recomputed from the masters ON the device: one ``fwn_gather_tables`` launch (fp64 like the
        host path) + one ``fwn_upsample_wn`` per up-sampling stage, straight into the table buffer the descriptors
        point at - no device -> host -> device round trip, and the whole refresh can sit in a hipGraph."""
        import torch
        lib = _lib.load()
        st = torch.cuda.current_stream(self.dev).cuda_stream
        _lib.check(lib.fwn_gather_tables(self._flat.data_ptr(), self._didx.data_ptr(), self._nterm, self._ttotal,
                                         self._dpost.data_ptr(), self._dmode.data_ptr(), self._tbuf.data_ptr(), st), "fwn_gather_tables")
        P = self._up_params
        for n, off, size in self._up_jobs:          # weight-normed up-sampling kernels: v / ||v||_(k) * g per kw column
            v, g = P["upsample_%d/kernel" % n], P["upsample_%d/g" % n]
            _lib.check(lib.fwn_upsample_wn(v.data_ptr(), g.data_ptr(), size // 6, self._tbuf[off:].data_ptr(), st), "fwn_upsample_wn")

    def _build(self):
        import torch
        sj = (_lib.ScaleJob * max(1, len(self.sjobs)))()
        for i, (v, g, k, n) in enumerate(self.sjobs):
            sj[i].v, sj[i].g, sj[i].k_src, sj[i].n_src = v, g, k, n
        pj = (_lib.PackJob * max(1, len(self.jobs)))()
        # jobs that read the same master back to back (stable sort: the jobs are independent of one another), so that the
        # second reader finds it in the Infinity Cache (pack_jobs_kernel dispatches the jobs in table order)
        order = sorted(range(len(self.jobs)), key=lambda i: self.jobs[i][0]) if os.environ.get("FWN_PACK_SORT", "1") != "0" else range(len(self.jobs))
        for i, (v, sk, sn, out, ld, n_src, kd, nd, slot, tr, mul) in enumerate(self.jobs[k] for k in order):
            j = pj[i]
            j.v, j.src_k, j.src_n, j.out, j.ld_dst = v, sk, sn, out, ld
            j.n_src, j.k_dst, j.n_dst, j.scale_slot, j.transposed, j.mul = n_src, kd, nd, slot, tr, mul
        self._sj = torch.frombuffer(bytearray(bytes(sj)), dtype=torch.uint8).to(self.dev)
        self._pj = torch.frombuffer(bytearray(bytes(pj)), dtype=torch.uint8).to(self.dev)
        self._scales = torch.empty(max(1, len(self.sjobs)), 512, dtype=torch.float32, device=self.dev)
        if self.tail_jobs:
            tj = np.asarray(self.tail_jobs, dtype=np.uint64).reshape(-1)
            self._tj = torch.from_numpy(tj.view(np.int64).copy()).to(self.dev)
        self._built = True

    def run_kernels(self):
        import torch
        if not self._built:
            self._build()
        st = torch.cuda.current_stream(self.dev).cuda_stream
        _lib.check(_lib.load().fwn_pack_jobs(self._sj.data_ptr(), len(self.sjobs), self._pj.data_ptr(), len(self.jobs),
                                             self._scales.data_ptr(), 512, st), "fwn_pack_jobs")
        if self.tail_jobs:
            _lib.check(_lib.load().fwn_pack_tail_stream_jobs(self._tj.data_ptr(), len(self.tail_jobs), self._tail_L, st), "fwn_pack_tail_stream_jobs")

    def upload_tables(self):
        import torch
        arrs = [np.ascontiguousarray(fn(), dtype=np.float32) for _, fn in self.tables]
        offs, total = [], 0
        for a in arrs:
            offs.append(total)
            total += (a.size + 3) // 4 * 4
        host = np.zeros(max(total, 4), dtype=np.float32)
        for a, off in zip(arrs, offs):
            host[off:off + a.size] = a.reshape(-1)
        self._toffs = [(off, a.size) for a, off in zip(arrs, offs)]
        if getattr(self, "_tbuf", None) is None:
            self._tbuf = torch.from_numpy(host).to(self.dev)
            for (setter, _), a, off in zip(self.tables, arrs, offs):
                setter(self._tbuf[off:off + a.size].view(a.shape))
        else:
            self._tbuf.copy_(torch.from_numpy(host))
        for cb in self.host_cbs:
            cb()

    def refresh(self):
        """The masters changed: recompute everything that was recorded."""
        if self._dev_ready:
            self.refresh_tables_device()
        else:
            self.hostview.reset()
            self.upload_tables()
        self.run_kernels()


class PackedModel:
    """Owns the packed device tensors and the ctypes descriptors that point into them."""

    def __init__(self, hp, device):
        self.hp = hp
        self.device = device
        self.tensors = []        # keeps every device buffer alive
        self.an = {}             # (i, j) -> fp32 [2][4][Ch] tensor
        self.wd8 = {}            # (flow index * L + layer) -> uint8 [512][768] e4m3 gate weights (gate_fp8 only)
        self.flow_descs = (_lib.FlowDesc * (hp.n_block * hp.n_flow))()
        self.model_desc = _lib.ModelDesc()
        self.weight_bytes = 0

    def keep(self, t):
        self.tensors.append(t)
        return t


def pack_model(params, hp, device="cuda", cond_mode: int = 0, plan: "PackPlan | None" = None,
               gate_fp8: bool = False, persist_mode: int = 0, chain_mode: int = 0, tail_stream: bool = True,
               cond_stream: bool = True) -> PackedModel:
    """Upload ``params`` (reference layouts, fp32) and run the packing kernels (K10).
    With ``plan`` (params must then be device tensors at stable addresses) the work is recorded into it
    and executed once; ``plan.refresh()`` repeats it after the parameters changed."""
    import torch

    lib = _lib.load()
    if hp.n_layer > _lib.FWN_MAX_LAYERS:
        raise ValueError("n_layer=%d exceeds FWN_MAX_LAYERS" % hp.n_layer)
    if hp.num_mels % 8:
        raise ValueError("num_mels must be a multiple of 8 (16-byte rows of the mel half planes)")
    half = hp.num_mels // 2
    pm = PackedModel(hp, device)
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None:       # "cuda" != "cuda:0" for torch: make it concrete
        dev = torch.device("cuda", torch.cuda.current_device())
    stream = torch.cuda.current_stream(dev).cuda_stream
    L = hp.n_layer

    def dev_i32(key, fn):
        # index tables depend only on (block, num_mels): kept across calls (a training step re-packs
        # the weights every step)
        gk = (str(dev), hp.num_mels, key)
        if gk not in _IDX_CACHE:
            _IDX_CACHE[gk] = torch.from_numpy(np.ascontiguousarray(fn(), dtype=np.int32)).to(dev)
        return _IDX_CACHE[gk]

    # Small fp32 tables (biases, ActNorm / ZeroConv tables, up-sampling kernels) are gathered on the
    # host and uploaded in ONE copy at the end; `put(setter, array)` calls setter(view) once the
    # device copy exists.  Every table starts on a 16-byte boundary (float4 loads in the kernels).
    pending = []

    def put(setter, fn, recipe=None):
        """fn() -> the table's current value (re-evaluated by plan.refresh()); recipe: see PackPlan.table."""
        if plan is not None:
            plan.table(setter, fn, recipe)
        else:
            a = fn()
            pending.append((setter, np.ascontiguousarray(a, dtype=np.float32).reshape(-1), np.shape(a)))

    def flush_tables():
        if plan is not None:
            plan.upload_tables()
            return
        offs, total = [], 0
        for _, a, _ in pending:
            offs.append(total)
            total += (a.size + 3) // 4 * 4
        host = np.zeros(max(total, 4), dtype=np.float32)
        for (_, a, _), off in zip(pending, offs):
            host[off:off + a.size] = a
        buf = pm.keep(torch.from_numpy(host).to(dev))
        for (setter, a, shape), off in zip(pending, offs):
            setter(buf[off:off + a.size].view(shape))

    class _HostView(dict):
        """params may hold device tensors (fp32 masters of optim.DataParallelAdam): the few
        host-side reductions (bias sums, ActNorm / ZeroConv scale tables) read them through here.
        All small device tensors come over in ONE copy (a per-tensor .cpu() is a sync each)."""
        _cache = None

        def reset(self):
            self._cache = None

        def _fill(self):
            small = [(k, v) for k, v in params.items() if isinstance(v, torch.Tensor) and v.numel() <= 4096]
            self._cache = {}
            if small:
                flat = torch.cat([v.detach().reshape(-1).to(torch.float32) for _, v in small]).cpu().numpy()
                off = 0
                for k, v in small:
                    n = v.numel()
                    self._cache[k] = flat[off:off + n].reshape(tuple(v.shape))
                    off += n

        def __getitem__(self, k):
            v = params[k]
            if not isinstance(v, torch.Tensor):
                return v
            if self._cache is None:
                self._fill()
            return self._cache[k] if k in self._cache else v.detach().cpu().numpy()

    hostp = _HostView()
    if plan is not None:
        plan.hostview = hostp

    ident256 = dev_i32("id256", lambda: np.arange(FILTER))
    ident768 = dev_i32("id768", lambda: np.arange(3 * FILTER))
    accperm = dev_i32("accperm", lambda: acc_k_perm(FILTER))
    accperm_host = acc_k_perm(FILTER).astype(np.int64)
    fg, gch = gate_row_channel()
    gate_rows = [dev_i32("gate_rows%d" % s, lambda s=s: np.where(fg == s, gch, -1)) for s in (0, 1)]
    scale_buf = torch.empty(FILTER, dtype=torch.float32, device=dev)
    if gate_fp8 and plan is not None:
        raise ValueError("the fp8 gate path is inference-only (no PackPlan)")
    # fp8 gate path: per (flow, layer) one e4m3 copy of the gate-packed dilated conv weights + its power-of-two exponent
    fp8_exp = torch.zeros(max(1, hp.n_block * hp.n_flow * L), dtype=torch.int32, device=dev) if gate_fp8 else None
    fp8_amax = torch.zeros_like(fp8_exp, dtype=torch.float32) if gate_fp8 else None
    fp8_scales = torch.empty(2, FILTER, dtype=torch.float32, device=dev) if gate_fp8 else None
    fp8_slots = []

    def pack_gate_fp8(rp, d, l, slot):
        """Wd8 = e4m3(GATE_MUL * weight-normed (Conv_filter | Conv_gate) * 2^e), gate-packed rows like Wd."""
        wd8 = pm.keep(torch.zeros(GATE_N, 3 * FILTER, dtype=torch.uint8, device=dev))
        pm.weight_bytes += wd8.numel()
        srcs = []
        for s_, nm in enumerate(("/Conv_filter", "/Conv_gate")):
            v = torch.as_tensor(params[rp + nm + "/kernel"]).to(device=dev, dtype=torch.float32).contiguous()
            g = torch.as_tensor(params[rp + nm + "/g"]).to(device=dev, dtype=torch.float32).contiguous()
            _lib.check(lib.fwn_wn_scale(v.data_ptr(), g.data_ptr(), 3 * FILTER, FILTER, fp8_scales[s_].data_ptr(), stream), "fwn_wn_scale")
            _lib.check(lib.fwn_wn_absmax(v.data_ptr(), fp8_scales[s_].data_ptr(), 3 * FILTER, FILTER, GATE_MUL[s_],
                                         fp8_amax[slot:].data_ptr(), stream), "fwn_wn_absmax")
            srcs.append(v)
        for s_, v in enumerate(srcs):
            _lib.check(lib.fwn_pack_e4m3(v.data_ptr(), fp8_scales[s_].data_ptr(), ident768.data_ptr(), gate_rows[s_].data_ptr(), FILTER,
                                         3 * FILTER, GATE_N, 3 * FILTER, GATE_MUL[s_], fp8_amax[slot:].data_ptr(), wd8.data_ptr(),
                                         fp8_exp[slot:].data_ptr(), stream), "fwn_pack_e4m3")
            v.record_stream(torch.cuda.current_stream(dev))
        d.Wd8[l] = wd8.data_ptr()
        pm.wd8[slot] = wd8
        fp8_slots.append((d, l, slot))

    def pack(name, src_k, src_n, k_dst, n_dst, out, ld_dst, col_off=0, weight_norm=True, mul=None):
        """Pack params[name + '/kernel'] into out[:, col_off: col_off + k_dst]; ``mul`` scales every
        output channel (folded into the weight-norm scale before the bf16 rounding)."""
        def up(x):
            if isinstance(x, torch.Tensor):
                if x.device == dev and x.dtype == torch.float32 and x.is_contiguous():
                    return x
                return x.to(device=dev, dtype=torch.float32).contiguous()
            return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)

        v = up(params[name + "/kernel"])
        if plan is not None:
            if v is not params[name + "/kernel"]:
                raise ValueError("a PackPlan needs contiguous fp32 device parameters (%s)" % name)
            if mul is not None and not weight_norm:
                raise ValueError("mul needs weight_norm")
            plan.add(v, up(params[name + "/g"]) if weight_norm else None, src_k, src_n, k_dst, n_dst,
                     out.data_ptr() + 2 * col_off, ld_dst, 1.0 if mul is None else mul)
            return
        k_src = v.shape[0] * v.shape[1]
        n_src = v.shape[2]
        sc = None
        if weight_norm:
            g = up(params[name + "/g"])
            _lib.check(lib.fwn_wn_scale(v.data_ptr(), g.data_ptr(), k_src, n_src, scale_buf.data_ptr(), stream),
                       "fwn_wn_scale")
            if mul is not None:
                scale_buf.mul_(mul)
            sc = scale_buf.data_ptr()
        elif mul is not None:
            raise ValueError("mul needs weight_norm")
        _lib.check(lib.fwn_pack_bf16(v.data_ptr(), sc, src_k.data_ptr(), src_n.data_ptr(), n_src, k_dst, n_dst,
                                     ld_dst, out.data_ptr() + 2 * col_off, stream), "fwn_pack_bf16")
        # v / g are freed by the caching allocator only after the stream passes this point
        v.record_stream(torch.cuda.current_stream(dev))

    def bf16_zeros(*shape):
        t = pm.keep(torch.zeros(*shape, dtype=torch.bfloat16, device=dev))
        pm.weight_bytes += t.numel() * 2
        return t

    cond_blocks = []
    for i in range(hp.n_block):
        ch = 1 << i
        cin = half * (2 << i)
        kcpad = roundup(cin, 64)
        kfpad = roundup(3 * ch, 64)
        npt = max(1, (ch + 31) // 32)
        f_src_k = dev_i32(("front", i), lambda: front_src_k(i))
        c_src_k = dev_i32(("cond", i), lambda: cond_src_k(i, half))
        z_src_n = dev_i32(("zero", i), lambda: zero_src_n(i))
        zsn_host = zero_src_n(i)
        wc_blk = bf16_zeros(hp.n_flow, L, GATE_N, kcpad)     # contiguous: hoisted conditioning
        cond_blocks.append((i, wc_blk, cin, kcpad))
        for j in range(hp.n_flow):
            fp = flow_prefix(i, j)
            wp = fp + "/WaveNet"
            d = pm.flow_descs[i * hp.n_flow + j]
            d.Ch, d.cin, d.kcpad, d.kfpad, d.npt, d.L = ch, cin, kcpad, kfpad, npt, L

            wfront = bf16_zeros(FILTER, kfpad)
            pack(wp + "/Conv_front", f_src_k, ident256, kfpad, FILTER, wfront, kfpad)
            d.Wfront = wfront.data_ptr()
            if ch >= 32:
                f2 = dev_i32(("front2", i), lambda: front2_src_k(i))
                wfront2 = bf16_zeros(FILTER, 6 * ch)
                pack(wp + "/Conv_front", f2, ident256, 6 * ch, FILTER, wfront2, 6 * ch)
                d.Wfront2 = wfront2.data_ptr()
            if ch == 16:     # the MFMA front kernel on a zero-padded 32-channel image (front_mfma_kernel<32, 16>)
                f2 = dev_i32(("front2p", i), lambda: front2_pad32_src_k(i))
                wfront2 = bf16_zeros(FILTER, 192)
                pack(wp + "/Conv_front", f2, ident256, 192, FILTER, wfront2, 192)
                d.Wfront2 = wfront2.data_ptr()
            if ch <= 8:      # chained front conv (the previous flow's tail computes this flow's h0, tail_chain.h)
                f3 = dev_i32(("front3", i), lambda: front3_src_k(i))
                kf3 = roundup(6 * ch, 16)
                wfront3 = bf16_zeros(FILTER, kf3)
                pack(wp + "/Conv_front", f3, ident256, kf3, FILTER, wfront3, kf3)
                d.Wfront3, d.kf3 = wfront3.data_ptr(), kf3
            put(lambda v, d=d: setattr(d, "bfront", v.data_ptr()), lambda wp=wp: hostp[wp + "/Conv_front/bias"],
                dict(terms=[(wp + "/Conv_front/bias", np.arange(FILTER))]))

            wskip = bf16_zeros(FILTER, L * FILTER)
            for l in range(L):
                rp = "%s/ResBlock_%d" % (wp, l)
                wd = bf16_zeros(GATE_N, 3 * FILTER)
                # filter / gate rows carry the exponent scales of tanh / sigmoid (GATE_MUL) so the
                # gate epilogue feeds its accumulators straight into exp2
                pack(rp + "/Conv_filter", ident768, gate_rows[0], 3 * FILTER, GATE_N, wd, 3 * FILTER, mul=GATE_MUL[0])
                pack(rp + "/Conv_gate", ident768, gate_rows[1], 3 * FILTER, GATE_N, wd, 3 * FILTER, mul=GATE_MUL[1])
                wc = wc_blk[j, l]
                pack(rp + "/filter_conv_c", c_src_k, gate_rows[0], kcpad, GATE_N, wc, kcpad, mul=GATE_MUL[0])
                pack(rp + "/gate_conv_c", c_src_k, gate_rows[1], kcpad, GATE_N, wc, kcpad, mul=GATE_MUL[1])
                def gate_bias(rp=rp):
                    bsum = [GATE_MUL[0] * (np.asarray(hostp[rp + "/Conv_filter/bias"], np.float64)
                                           + np.asarray(hostp[rp + "/filter_conv_c/bias"], np.float64)),
                            GATE_MUL[1] * (np.asarray(hostp[rp + "/Conv_gate/bias"], np.float64)
                                           + np.asarray(hostp[rp + "/gate_conv_c/bias"], np.float64))]
                    return np.where(fg == 0, bsum[0][gch], bsum[1][gch])
                d.Wd[l] = wd.data_ptr()
                d.Wc[l] = wc.data_ptr()
                # the same operands in MFMA-fragment order for the register-streamed gate kernel (csrc/gate_rs.h): packed
                # once, so only without a PackPlan (a plan re-packs Wd / Wc every training step, whose forward pass keeps
                # the training gate kernel anyway)
                gs_bytes = int(lib.fwn_gate_stream_bytes(cin)) if plan is None else 0
                if gs_bytes:
                    wgs = pm.keep(torch.empty(gs_bytes, dtype=torch.uint8, device=dev))
                    pm.weight_bytes += gs_bytes
                    _lib.check(lib.fwn_pack_gate_stream(wd.data_ptr(), wc.data_ptr(), cin, kcpad, wgs.data_ptr(), stream),
                               "fwn_pack_gate_stream")
                    d.Wgs[l] = wgs.data_ptr()
                if gate_fp8:
                    pack_gate_fp8(rp, d, l, (i * hp.n_flow + j) * L + l)
                put(lambda v, d=d, l=l: d.bgate.__setitem__(l, v.data_ptr()), gate_bias,
                    dict(terms=[(rp + "/Conv_filter/bias", np.where(fg == 0, gch, -1)), (rp + "/filter_conv_c/bias", np.where(fg == 0, gch, -1)),
                                (rp + "/Conv_gate/bias", np.where(fg == 0, -1, gch)), (rp + "/gate_conv_c/bias", np.where(fg == 0, -1, gch))],
                         post=np.where(fg == 0, GATE_MUL[0], GATE_MUL[1])))
                if l + 1 < L:   # the last layer's res_conv is dead (modules.py:126-128,175-176)
                    wr = bf16_zeros(FILTER, FILTER)
                    pack(rp + "/res_conv", ident256, ident256, FILTER, FILTER, wr, FILTER)
                    d.Wres[l] = wr.data_ptr()
                    put(lambda v, d=d, l=l: d.bres.__setitem__(l, v.data_ptr()), lambda rp=rp: hostp[rp + "/res_conv/bias"],
                        dict(terms=[(rp + "/res_conv/bias", np.arange(FILTER))]))
                # rows (output channels) in accumulator order: S leaves the tail's MFMA chain in natural channel order
                pack(rp + "/skip_conv", ident256, accperm, FILTER, FILTER, wskip, L * FILTER, col_off=l * FILTER)
            d.Wskip = wskip.data_ptr()
            put(lambda v, d=d: setattr(d, "bskip", v.data_ptr()),
                lambda wp=wp: sum(np.asarray(hostp["%s/ResBlock_%d/skip_conv/bias" % (wp, l)], np.float64).reshape(-1) for l in range(L))[accperm_host],
                dict(terms=[("%s/ResBlock_%d/skip_conv/bias" % (wp, l), accperm_host) for l in range(L)]))

            wfin = bf16_zeros(FILTER, FILTER)
            pack(wp + "/Conv_final", ident256, accperm, FILTER, FILTER, wfin, FILTER)
            d.Wfinal = wfin.data_ptr()
            # Wskip | Wfinal once more in MFMA-fragment order for the register-streamed tail (csrc/tail_rs.h): one ZeroConv pair
            # tile only.  Under a PackPlan (training: the weights are re-packed every step) the streams are re-packed too, all in one launch
            ts_bytes = int(lib.fwn_tail_stream_bytes(L)) if (npt == 1 and tail_stream) else 0
            if ts_bytes:
                wts = pm.keep(torch.empty(ts_bytes, dtype=torch.uint8, device=dev))
                pm.weight_bytes += ts_bytes
                if plan is None:
                    _lib.check(lib.fwn_pack_tail_stream(wskip.data_ptr(), wfin.data_ptr(), L, wts.data_ptr(), stream), "fwn_pack_tail_stream")
                else:       # re-packed behind every grouped weight packing, all flows in one launch (PackPlan.run_kernels)
                    plan.add_tail_stream(wskip.data_ptr(), wfin.data_ptr(), wts.data_ptr(), L)
                d.Wts = wts.data_ptr()
            put(lambda v, d=d: setattr(d, "bfinal", v.data_ptr()), lambda wp=wp: np.asarray(hostp[wp + "/Conv_final/bias"]).reshape(-1)[accperm_host],
                dict(terms=[(wp + "/Conv_final/bias", accperm_host)]))

            wz = bf16_zeros(npt * 64, FILTER)
            pack(wp + "/ZeroConv1d", ident256, z_src_n, FILTER, npt * 64, wz, FILTER, weight_norm=False)
            def zero_tables(which, wp=wp, zsn_host=zsn_host, npt=npt):
                valid = zsn_host >= 0
                if which == 0:
                    zb = np.asarray(hostp[wp + "/ZeroConv1d/bias"], np.float64).reshape(-1)
                    out = np.zeros(npt * 64)
                    out[valid] = zb[zsn_host[valid]]
                else:
                    zs = np.asarray(hostp[wp + "/ZeroConv1d/scale"], np.float64).reshape(-1)
                    out = np.ones(npt * 64)
                    out[valid] = np.exp(3.0 * zs[zsn_host[valid]])
                return out
            d.Wzero = wz.data_ptr()
            zidx = np.where(zsn_host >= 0, zsn_host, -1)
            put(lambda v, d=d: setattr(d, "bzero", v.data_ptr()), lambda f=zero_tables: f(0),
                dict(terms=[(wp + "/ZeroConv1d/bias", zidx)]))
            put(lambda v, d=d: setattr(d, "ezero", v.data_ptr()), lambda f=zero_tables: f(1),
                dict(terms=[(wp + "/ZeroConv1d/scale", zidx)], post=3.0, exp=True))

            def set_an(v, d=d, key=(i, j)):
                pm.an[key] = v
                d.an = v.data_ptr()
            an_idx = np.concatenate([r * ch + bitrev_table(i).astype(np.int64) for r in range(2)]).reshape(2, 1, ch)
            rows = np.arange(4).reshape(1, 4, 1)
            put(set_an, lambda fp=fp, i=i: actnorm_table(hostp[fp + "/ActNorm/b"], hostp[fp + "/ActNorm/logs"], i),
                dict(terms=[(fp + "/ActNorm/b", np.where(rows == 0, an_idx, -1)), (fp + "/ActNorm/logs", np.where(rows == 0, -1, an_idx))],
                     post=np.broadcast_to(np.array([1.0, 3.0, -3.0, 3.0]).reshape(1, 4, 1), (2, 4, ch)),
                     exp=np.broadcast_to((rows == 1) | (rows == 2), (2, 4, ch))))

    md = pm.model_desc
    # the hoisted conditioning's weights once more in MFMA-fragment order (csrc/cond_rs.h): blocks whose conditioning can be
    # hoisted and the streamed kernel is the faster form (K >= 640), packed once - so only without a PackPlan, like the gate's and the tail's streams
    for i, wc_blk, cin, kcpad in cond_blocks:
        cs_bytes = int(lib.fwn_cond_stream_bytes(kcpad)) if (plan is None and cond_stream and kcpad >= 640 and i < 16) else 0
        if cs_bytes:
            nz = hp.n_flow * L
            wcs = pm.keep(torch.empty(nz * cs_bytes, dtype=torch.uint8, device=dev))
            pm.weight_bytes += nz * cs_bytes
            _lib.check(lib.fwn_pack_cond_stream(wc_blk.data_ptr(), GATE_N * kcpad, kcpad, nz, wcs.data_ptr(), stream), "fwn_pack_cond_stream")
            md.cond_stream[i] = wcs.data_ptr()
    md.n_block, md.n_flow, md.n_layer, md.num_mels = hp.n_block, hp.n_flow, L, hp.num_mels
    md.n_up = len(hp.upsample_scales)
    if md.n_up > _lib.FWN_MAX_UPSAMPLE:
        raise ValueError("too many upsample stages")
    for n, s in enumerate(hp.upsample_scales):
        md.up_scale[n] = int(s)
        put(lambda v, n=n: md.up_w.__setitem__(n, v.data_ptr()), lambda n=n: upsample_kernel(hostp, n)[0], dict(upsample=n))
        md.up_bias[n] = upsample_kernel(hostp, n)[1]
        if plan is not None:
            plan.host_cbs.append(lambda n=n: md.up_bias.__setitem__(n, upsample_kernel(hostp, n)[1]))
    flush_tables()
    if plan is not None:
        plan.run_kernels()
        pm.plan = plan
    md.flows = C.cast(pm.flow_descs, C.POINTER(_lib.FlowDesc))
    md.cond_mode = int(cond_mode)
    # fwn.h chain_mode: 1 runs every flow on its own like round 2 (same-box A/B, tests); an argument, not a process variable
    md.chain_mode = int(chain_mode)
    # fwn.h persist_mode: 2 runs every small-M flow as ONE launch, 1 none; default 0 = those of <= 512 rows, where the form
    # measured faster (FloWaveNet(..., persist_mode=) / hparams.persist_mode)
    md.persist_mode = int(persist_mode)
    md.gate_fp8 = 1 if gate_fp8 else 0
    torch.cuda.current_stream(dev).synchronize()
    if gate_fp8:                                   # the exponents the pack kernels chose, in one copy
        exps = fp8_exp.cpu().numpy()
        for d, l, slot in fp8_slots:
            d.wd8_exp[l] = int(exps[slot])
        pm.fp8_exponents = exps
    return pm
