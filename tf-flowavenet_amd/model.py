"""``FloWaveNet`` - the reference's model surface (model.py:282-404) on MI355X.

Same class name, constructor arguments, method names, argument order, shapes and
return arity as the reference; torch tensors on a HIP device replace TF tensors
and calls execute eagerly on the current HIP stream.  All arithmetic runs in
``csrc/libfwn.so`` (C ABI in ``include/fwn.h``); there is no fallback path.

Differences that are deliberate (DESIGN.md "Deviations"):
  * ``hparams.dtype`` float16 -> bfloat16 hidden activations / weights with fp32
    accumulation; the flow state, ActNorm, coupling and all reductions stay fp32
    (the reference takes its means in fp16, model.py:135,343);
  * ``reverse`` returns fp32 by default (the flow state is fp32 here); ``reverse(..., dtype="hparams")`` returns
    ``hparams.dtype`` like the reference (model.py:356-357,396);
  * global (speaker) conditioning is inert in the reference (``WaveNet.__call__``
    drops ``g``, modules.py:188-189): ``g`` is validated like the reference does
    (model.py:320-321,353-354) and otherwise ignored;
  * ``affine=False`` / ``causality=True`` are not BASELINE configurations: rejected.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib, packing, weights


class FloWaveNet:
    def __init__(self, hparams, init=False, scope="FloWaveNet", device="cuda", cond_mode=0, group=None, gate_fp8=None,
                 persist_mode=None, chain_mode=None, tail_stream=True, cond_stream=True):
        """persist_mode (default: ``hparams.persist_mode`` if present, else 0): which small-M flows run as ONE launch
        (csrc/flow_persist.h, ``fwn_model_desc.persist_mode``) - 0: those of up to 512 rows, 1: none, 2: wherever the form
        exists.  chain_mode (``fwn_model_desc.chain_mode``): 0 chains the flows of a block, 1 runs every flow on its own.
        Both select among kernels that compute the same arithmetic; they are arguments of the model, not process variables.
        tail_stream=False leaves the fragment-order copy of Wskip | Wfinal unpacked, so that the tail runs the kernels of
        rounds 2 - 5 everywhere (csrc/tail_chain.h and the N-split ring GEMMs instead of csrc/tail_rs.h): the one-launch flows
        reproduce THOSE bit for bit below 4 097 rows (tests, bench.py's identity check).
        cond_stream=False leaves the fragment-order copy of the hoisted conditioning's weights unpacked (csrc/cond_rs.h): the
        ring-tile kernel then runs at every row count (same sums in the same order, other split counts).
        group: the ``torch.distributed`` process group a data-parallel job shards its batch over (None = the
        default group when one is initialised; False: none - a model only one rank builds).  It only matters for ``init=True``: the ActNorm data-dependent
        init then uses the statistics of the GLOBAL batch (moments all-reduced flow by flow) so every rank ends
        with the same parameters - the reference's towers race on that assign (model.py:39, train.py:43-57)."""
        if not hparams.affine:
            raise NotImplementedError("affine=False (additive coupling, model.py:136-139) is out of scope")
        if hparams.causality:
            raise NotImplementedError("causality=True (modules.py:12-13,30-31) is out of scope")
        if hparams.n_block < 1 or hparams.n_flow < 1 or hparams.n_layer < 1:
            raise ValueError("n_block, n_flow and n_layer must be >= 1")
        self._hparams = hparams
        self._scope = scope
        self._init = bool(init)
        self._device = device
        self._cond_mode = cond_mode
        self._group = group
        # fp8 (e4m3) dilated taps where the shape has such a kernel (BASELINE configs[4]); default: hparams.gate_fp8
        self._gate_fp8 = bool(getattr(hparams, "gate_fp8", False) if gate_fp8 is None else gate_fp8)
        self._persist_mode = int(getattr(hparams, "persist_mode", 0) if persist_mode is None else persist_mode)
        self._chain_mode = int(getattr(hparams, "chain_mode", 0) if chain_mode is None else chain_mode)
        if self._persist_mode not in (0, 1, 2) or self._chain_mode not in (0, 1):
            raise ValueError("persist_mode must be 0, 1 or 2 and chain_mode 0 or 1")
        self._tail_stream = bool(tail_stream)
        self._cond_stream = bool(cond_stream)
        self._packed = None
        self._ws = {}
        self._lib = _lib.load()     # fails loudly when libfwn.so is missing
        self.hop = int(np.prod(hparams.upsample_scales))
        if self.hop != hparams.hop_size:
            raise ValueError("prod(upsample_scales)=%d must equal hop_size=%d (model.py:231)"
                             % (self.hop, hparams.hop_size))

    # ------------------------------------------------------------------ parameters
    def load_params(self, params):
        """params: dict name -> array in the reference's layouts (weights.param_shapes)."""
        shapes = weights.param_shapes(self._hparams)
        for name, shape in shapes.items():
            if name not in params:
                raise KeyError("missing parameter %r" % name)
            got = tuple(getattr(params[name], "shape", None) or np.shape(params[name]))
            if got != tuple(shape):
                raise ValueError("parameter %r has shape %r, expected %r" % (name, got, tuple(shape)))
        self._packed = packing.pack_model(params, self._hparams, self._device, self._cond_mode, gate_fp8=self._gate_fp8,
                                          persist_mode=self._persist_mode, chain_mode=self._chain_mode, tail_stream=self._tail_stream,
                                          cond_stream=self._cond_stream)
        return self

    def init_synthetic(self, seed=1234, **kw):
        return self.load_params(weights.synthetic_params(self._hparams, seed, **kw))

    def export_actnorm(self):
        """ActNorm (b, logs) per flow in the reference's layout (after a DDI forward)."""
        out = {}
        hp = self._hparams
        for i in range(hp.n_block):
            for j in range(hp.n_flow):
                an = self._packed.an[(i, j)].cpu().numpy()
                b, logs = packing.actnorm_from_table(an, i)
                out[weights.flow_prefix(i, j) + "/ActNorm/b"] = b
                out[weights.flow_prefix(i, j) + "/ActNorm/logs"] = logs
        return out

    @property
    def weight_bytes(self):
        return self._packed.weight_bytes

    # ------------------------------------------------------------------ helpers
    def _check_g(self, g):
        if g is None and self._hparams.gin_channels > 0:
            raise ValueError("g is None")   # model.py:320-321,353-354

    def _prep(self, x, c, what):
        import torch
        if self._packed is None:
            raise RuntimeError("no parameters loaded: call load_params() / init_synthetic() first")
        hp = self._hparams
        if x.dim() != 3 or x.shape[2] != 1:
            raise ValueError("%s must have shape [B, T, 1], got %r" % (what, tuple(x.shape)))
        if c.dim() != 3 or c.shape[2] != hp.num_mels:
            raise ValueError("c must have shape [B, T/hop, %d], got %r" % (hp.num_mels, tuple(c.shape)))
        b, t = int(x.shape[0]), int(x.shape[1])
        if c.shape[0] != b or int(c.shape[1]) * self.hop != t:
            raise ValueError("c has %d frames for T=%d samples (hop_size=%d)" % (c.shape[1], t, self.hop))
        if t % (1 << hp.n_block):
            raise ValueError("T=%d must be a multiple of 2^n_block=%d (model.py:226)" % (t, 1 << hp.n_block))
        dev = torch.device(self._device)
        x32 = x.to(device=dev, dtype=torch.float32).contiguous()
        c32 = c.to(device=dev, dtype=torch.float32).contiguous()
        return b, t, x32, c32

    def _workspace(self, b, t):
        """Scratch for one pass.  One workspace per (B, T, HIP stream): passes issued on different
        streams (e.g. a forward and an inverse overlapping on the chip) never share scratch."""
        import torch
        key = (b, t, self._stream())
        ws = self._ws.get(key)
        if ws is None:
            n = self._lib.fwn_workspace_bytes(C.byref(self._packed.model_desc), b, t)
            if n == 0:
                _lib.check(-1, "fwn_workspace_bytes")
            for k in [k for k in self._ws if k[:2] != (b, t)]:
                del self._ws[k]   # keep only the current shape's workspaces
            ws = torch.empty(n + 256, dtype=torch.uint8, device=self._device)
            self._ws[key] = ws
        off = (-ws.data_ptr()) % 256
        return ws.data_ptr() + off, ws.numel() - off

    def _stream(self):
        import torch
        return torch.cuda.current_stream(torch.device(self._device)).cuda_stream

    def persist_status(self, b, t):
        """0 unless a one-launch flow (csrc/flow_persist.h) of the last ``forward`` / ``reverse`` with batch ``b`` and length ``t`` on
        the current stream gave up a bounded dependency wait (a pass that was only delayed - a preempted queue, a debugger -
        can: its log_p / logdet / waveform are NaN then).  > 0: the give-up code; retry, or build the model with
        ``persist_mode=1``.  Synchronises the stream (``fwn_model_persist_status``)."""
        ws, _ = self._workspace(b, t)
        return int(self._lib.fwn_model_persist_status(C.byref(self._packed.model_desc), b, t, ws, self._stream()))

    # ------------------------------------------------------------------ reference surface
    def forward(self, x, c, g=None, return_z=False):
        """x [B,T,1], c [B,T/hop,num_mels] -> (log_p, logdet) fp32 0-dim tensors (model.py:317-347)."""
        import torch
        self._check_g(g)
        b, t, x32, c32 = self._prep(x, c, "x")
        wsp, wsn = self._workspace(b, t)
        out2 = torch.empty(2, dtype=torch.float32, device=self._device)
        zp = torch.empty(2, b, t // 2, dtype=torch.float32, device=self._device) if return_z else None
        if self._init and self._dp_world() > 1:
            self._forward_init_dp(b, t, x32, c32, wsp, wsn, out2, zp)
        else:
            rc = self._lib.fwn_model_forward(C.byref(self._packed.model_desc), b, t, x32.data_ptr(), c32.data_ptr(),
                                             wsp, wsn, out2.data_ptr(), zp.data_ptr() if return_z else None,
                                             1 if self._init else 0, self._stream())
            _lib.check(rc, "fwn_model_forward")
        self._init = False       # the reference feeds init=True for one step only (train.py:221,229)
        if return_z:
            return out2[0], out2[1], zp
        return out2[0], out2[1]

    def _dp_world(self):
        import torch.distributed as dist
        if self._group is False or not (dist.is_available() and dist.is_initialized()):
            return 1                    # group=False: this model lives on one rank only (its init uses the local batch)
        return dist.get_world_size(self._group)

    def _forward_init_dp(self, b, t, x32, c32, wsp, wsn, out2, zp):
        """init=True on one rank of a data-parallel job: ``fwn_model_forward_init`` calls back before each flow's
        ActNorm tables are derived; the callback all-reduces that flow's 4 Ch + 1 moment doubles (RCCL; they live
        inside this pass's workspace tensor) in stream order."""
        import torch
        from . import distributed
        ws = self._ws[(b, t, self._stream())]
        base, failure = ws.data_ptr(), []

        def reduce(user, buf, n, stream):
            try:
                off = int(buf) - base
                distributed.allreduce_sum_(ws[off:off + 8 * n].view(torch.float64), self._group)
                return 0
            except BaseException as e:      # must not propagate through the C frame
                failure.append(e)
                return 1

        cb = _lib.REDUCE_FN(reduce)
        rc = self._lib.fwn_model_forward_init(C.byref(self._packed.model_desc), b, t, x32.data_ptr(), c32.data_ptr(),
                                              wsp, wsn, out2.data_ptr(), zp.data_ptr() if zp is not None else None,
                                              cb, None, self._stream())
        if failure:
            raise failure[0]
        _lib.check(rc, "fwn_model_forward_init")

    def reverse(self, z, c, g=None, dtype=None):
        """z [B,T,1], c [B,T/hop,num_mels] -> x [B,T,1] (model.py:350-396).  fp32 unless ``dtype`` is given: a torch dtype, or
        "hparams" for the reference's return type, ``hparams.dtype`` (float16 / bfloat16 / float32; model.py:356-357,396)."""
        import torch
        self._check_g(g)
        if dtype == "hparams":
            dtype = {"float16": torch.float16, "bfloat16": torch.bfloat16, "float32": torch.float32}[str(self._hparams.dtype).replace("tf.", "")]
        b, t, z32, c32 = self._prep(z, c, "z")
        wsp, wsn = self._workspace(b, t)
        x = torch.empty(b, t, 1, dtype=torch.float32, device=self._device)
        rc = self._lib.fwn_model_reverse(C.byref(self._packed.model_desc), b, t, z32.data_ptr(), c32.data_ptr(),
                                         wsp, wsn, x.data_ptr(), self._stream())
        _lib.check(rc, "fwn_model_reverse")
        return x if dtype is None or dtype == torch.float32 else x.to(dtype)

    def upsample(self, c):
        """c [B,F,num_mels] -> [B,F*hop,num_mels] fp32 (model.py:398-404)."""
        import torch
        if self._packed is None:
            raise RuntimeError("no parameters loaded")
        hp = self._hparams
        cur = c.to(device=self._device, dtype=torch.float32).contiguous()
        md = self._packed.model_desc
        for n, s in enumerate(hp.upsample_scales):
            b, h, w = cur.shape
            out = torch.empty(b, h * s, w, dtype=torch.float32, device=self._device)
            rc = self._lib.fwn_upsample_stage(cur.data_ptr(), b, h, w, md.up_w[n], md.up_bias[n], int(s),
                                              out.data_ptr(), None, self._stream())
            _lib.check(rc, "fwn_upsample_stage")
            cur = out
        return cur

    __call__ = forward


def z_planes_to_squeezed(zp, n_block, n_flow=2):
    """planes[2][B][T/2] (device layout) -> the reference's final ``out`` [B, T/2^n, 2^n]
    (logical squeezed channel order after n_block*n_flow change_orders), for comparisons
    against the oracle."""
    import torch
    two, b, ht = zp.shape
    n = n_block
    ch = 1 << (n - 1)
    rows = ht // ch
    v = zp.reshape(2, b, rows, ch)                       # [q][b][t][tau']
    br = torch.as_tensor(packing.bitrev_table(n - 1).astype(np.int64), device=zp.device)
    out = torch.empty(b, rows, 2 * ch, dtype=zp.dtype, device=zp.device)
    swapped = (n_block * n_flow) & 1                     # odd number of swaps: halves exchanged
    for q in range(2):
        out[:, :, (q ^ swapped) * ch + br] = v[q]
    return out
