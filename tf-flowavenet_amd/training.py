"""Training side of the path - WORK IN PROGRESS.

What exists: thin wrappers over the generic GEMM / transpose / split-reduce primitives of
``csrc/train_kernels.hip`` (``fwn_gemm``, ``fwn_transpose_shift``, ``fwn_reduce_splits``) from which the
backward pass is being assembled (data gradients = GEMMs on transposed packed weights; weight
gradients = GEMMs over transposed activation copies with K = rows, split over workgroups and summed
in a fixed order).  The optimiser side (gradient all-reduce, clip, Adam) is ``optim.py``.
All arithmetic runs in ``libfwn.so``; there is no fallback path.
"""
from __future__ import annotations

import ctypes as C

from . import _lib


def _stream(t):
    import torch
    return torch.cuda.current_stream(t.device).cuda_stream


def gemm(segs, w, n, m, *, ti=0, bias=None, res=None, rscale=1.0, mask=None, relu=False, oscale=1.0,
         out=None, out_f32=False, accumulate=False, nsplit=1, split_stride=0):
    """``out[m, n] = oscale * relu?(mask?(sum_s shift(x_s)[:, :k_s] @ w[:, koff_s:koff_s+k_s].T + bias + rscale*res))``.

    segs: list of ``(x, k, shift, koff)`` with ``x`` a bf16 2-D tensor (rows, ld); ``w`` bf16 (n, ldw)."""
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
    _lib.check(lib.fwn_gemm(C.byref(d), _stream(w)), "fwn_gemm")
    return out


def transpose_shift(x, m, c, *, shift=0, ti=0, ones_row=False, ld_dst=None):
    """``dst[c', m'] = x[m' + shift, c']`` (zero where the tap leaves its clip), bf16 ``[c (+1), ld_dst]``."""
    import torch
    lib = _lib.load()
    ld_dst = ld_dst or (m + 63) // 64 * 64
    dst = torch.empty(c + (1 if ones_row else 0), ld_dst, dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.fwn_transpose_shift(x.data_ptr(), int(m), int(c), int(x.stride(0)), int(shift), int(ti),
                                       dst.data_ptr(), int(ld_dst), int(bool(ones_row)), _stream(x)), "fwn_transpose_shift")
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


def weight_grad(x, dy, m, kx, n, *, shifts=(0,), ti=0, nsplit=None):
    """``dW[tap*kx + i, j] = sum_r x[r + shift_tap, i] * dy[r, j]`` and ``db[j] = sum_r dy[r, j]`` (fp32).

    x: bf16 [m, >=kx], dy: bf16 [m, >=n].  Returns (dW [len(shifts)*kx, n], db [n])."""
    import torch
    mp = (m + 63) // 64 * 64
    rows = len(shifts) * kx
    xt = torch.empty(rows + 1, mp, dtype=torch.bfloat16, device=x.device)
    lib = _lib.load()
    for i, sh in enumerate(shifts):
        last = i == len(shifts) - 1
        _lib.check(lib.fwn_transpose_shift(x.data_ptr(), m, kx, int(x.stride(0)), int(sh), int(ti),
                                           xt[i * kx:].data_ptr(), mp, int(last), _stream(x)), "fwn_transpose_shift")
    dyt = transpose_shift(dy, m, n, ld_dst=mp)
    if nsplit is None:
        tiles = ((rows + 1 + 63) // 64) * ((n + 127) // 128)
        nsplit = max(1, min(mp // 64, -(-256 // tiles)))
    part = gemm([(xt, mp, 0, 0)], dyt, n, rows + 1, out_f32=True, nsplit=nsplit)
    full = reduce_splits(part) if nsplit > 1 else part
    return full[:rows], full[rows]
