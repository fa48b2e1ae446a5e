"""Data-parallel optimiser step for the flow model.

Replaces, for one process per GPU:
  * ``average_gradients`` (utils.py:34-60: gather every tower's gradient on one device and
    ``reduce_mean``)  ->  ONE bucketed RCCL all-reduce (sum) of a flat fp32 gradient buffer over
    xGMI; the 1/world factor is folded into the optimiser kernel;
  * ``tf.scalar_mul(1/scale)`` + ``clip_by_global_norm(., 1)`` + ``AdamOptimizer`` (train.py:15-32,
    75-81)  ->  ``fwn_grad_norm`` + ``fwn_clip_adam`` (two HBM-bound passes, deterministic norm);
  * ``fp16_dtype_getter`` (utils.py:3-31: fp32 master variable + cached low-precision cast)  ->
    fp32 master weights in one flat buffer, re-packed to bf16 MFMA layouts after the update
    (``packing.pack_model`` reading device views; bf16 needs no loss scale, ``scale`` is kept only
    to consume gradients of a scaled loss).

``training.GradEngine`` fills ``grad``; ``training.Trainer`` ties the two together.
"""
from __future__ import annotations

import numpy as np

from . import _lib, weights


def learning_rate(step: int) -> float:
    """Step-wise schedule of train.py:17-20 (global_step before the update)."""
    if step >= 600000:
        return 0.001 / 6
    if step >= 400000:
        return 0.001 / 4
    if step >= 200000:
        return 0.001 / 2
    return 0.001


class FlatLayout:
    """name -> (offset, shape) of every trainable tensor inside one flat fp32 vector, in
    ``weights.param_shapes`` order, each tensor start aligned to 4 elements (16 bytes)."""

    def __init__(self, hp):
        self.slots = {}
        off = 0
        for name, shape in weights.param_shapes(hp).items():
            n = int(np.prod(shape))
            self.slots[name] = (off, tuple(shape), n)
            off += (n + 3) // 4 * 4
        self.size = off

    def flatten(self, params):
        out = np.zeros(self.size, dtype=np.float32)
        for name, (off, shape, n) in self.slots.items():
            out[off:off + n] = np.asarray(params[name], dtype=np.float32).reshape(-1)
        return out

    def views(self, flat):
        """dict name -> view (torch or numpy) into ``flat`` with the reference's shapes."""
        return {name: flat[off:off + n].reshape(shape) for name, (off, shape, n) in self.slots.items()}


def bucket_bounds(n: int, bucket_elems: int):
    """[lo, hi) element ranges of the all-reduce buckets (last one may be short)."""
    if bucket_elems <= 0:
        raise ValueError("bucket size must be positive")
    return [(lo, min(n, lo + bucket_elems)) for lo in range(0, n, bucket_elems)]


REDUCE_DTYPES = ("fp32", "bf16")


class _Bf16Work:
    """Handle of one bf16 all-reduce of a slice of the fp32 gradient: ``wait()`` waits for the collective, then widens the
    summed bf16 values back into the fp32 slice (stream-ordered on the GPU like the collective's own wait)."""

    def __init__(self, work, dst, src16):
        self.work, self.dst, self.src16 = work, dst, src16

    def wait(self):
        self.work.wait()
        self.dst.copy_(self.src16)


def allreduce_slice(g, lo, hi, group=None, reduce_dtype="fp32", stage=None):
    """Start the all-reduce (sum) of ``g[lo:hi]`` and return its handle.  reduce_dtype "bf16" (362 MB
    instead of 724.5 MB per step over xGMI): the slice is rounded to bf16 into ``stage[lo:hi]`` (a bf16 buffer as long as
    g), summed there by the collective - every rank receives the same bf16 sums, so the lock-step of the masters is
    untouched - and widened back on ``wait()``.  Each rank's contribution carries one bf16 rounding (2^-9 relative) and so
    does every partial sum of the ring: against the fp32 exchange the reduced gradient moves by <= ~2^-8 per element
    relative to the largest partial sum (tests/test_dist.py bounds the step built from it)."""
    import torch.distributed as dist
    if reduce_dtype not in REDUCE_DTYPES:
        raise ValueError("grad_reduce_dtype must be one of %s" % (REDUCE_DTYPES,))
    if reduce_dtype == "fp32":
        return dist.all_reduce(g[lo:hi], group=group, async_op=True)
    s16 = stage[lo:hi]
    s16.copy_(g[lo:hi])
    return _Bf16Work(dist.all_reduce(s16, group=group, async_op=True), g[lo:hi], s16)


def allreduce_flat(grad, group=None, bucket_elems=64 << 20, async_op=True, reduce_dtype="fp32", stage=None):
    """Sum ``grad`` (flat tensor, any device) over ranks bucket by bucket.  Buckets are issued
    back to back (asynchronously) so a caller can interleave them with backward compute; xGMI is
    point-to-point, so fewer, larger buckets (256 MB of fp32 each by default) keep every link busy.
    Returns the list of work handles (already waited when async_op=False).  reduce_dtype / stage: ``allreduce_slice``."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return []
    if reduce_dtype == "bf16" and stage is None:
        stage = torch.empty(grad.numel(), dtype=torch.bfloat16, device=grad.device)
    works = [allreduce_slice(grad, lo, hi, group, reduce_dtype, stage) for lo, hi in bucket_bounds(grad.numel(), bucket_elems)]
    if not async_op:
        for w in works:
            w.wait()
    return works


# ---- the sharded-optimiser exchange (SURVEY section 8e's alternative: "clip/Adam on the reduce-scattered shard, all-gather
# weights"; ZeRO-1 shape).  Rank r owns elements [lo_r, hi_r) of the flat buffers: the gradient is REDUCED ONTO ITS OWNER
# instead of all-reduced, the global norm is the all-reduced sum of the owners' shard norms, every rank runs clip + Adam on
# its shard only (1 / world of the optimiser's HBM traffic: 28 B per parameter read + 12 B written per step), and the
# updated fp32 masters are gathered back so that every rank packs the same weights.  Exchange volume = the all-reduce's
# (reduce-scatter + all-gather ARE its two halves).  Written with per-shard reduce / broadcast collectives, which every
# backend has (gloo has no reduce_scatter): on RCCL a padded reduce_scatter_tensor / all_gather_into_tensor pair moves the
# same bytes in two calls.
EXCHANGES = ("allreduce", "zero1")


def shard_bounds(n: int, world: int, align: int = 4):
    """[lo, hi) of every rank's shard of an n-element flat buffer: equal chunks rounded up to `align` elements (16-byte
    kernel accesses), the last ones short or empty."""
    chunk = (-(-n // world) + align - 1) // align * align
    return [(min(n, r * chunk), min(n, (r + 1) * chunk)) for r in range(world)]


def _group_ranks(group):
    import torch.distributed as dist
    world = dist.get_world_size(group)
    return [dist.get_global_rank(group, r) if group is not None else r for r in range(world)]


def reduce_to_owners(g, group=None, async_op=False):
    """Sum every shard of the flat gradient ``g`` onto its owner: afterwards ``g[lo_r:hi_r]`` on rank r is the sum over the
    ranks (the rest of g is stale).  Returns (this rank's (lo, hi), work handles)."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    bounds, ranks = shard_bounds(g.numel(), world), _group_ranks(group)
    works = [dist.reduce(g[lo:hi], dst=ranks[r], group=group, async_op=True) for r, (lo, hi) in enumerate(bounds) if hi > lo]
    if not async_op:
        for w in works:
            w.wait()
        works = []
    return bounds[rank], works


def gather_from_owners(w, group=None):
    """Every owner's shard of the flat tensor ``w`` to everyone (the all-gather half)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    bounds, ranks = shard_bounds(w.numel(), world), _group_ranks(group)
    works = [dist.broadcast(w[lo:hi], src=ranks[r], group=group, async_op=True) for r, (lo, hi) in enumerate(bounds) if hi > lo]
    for x in works:
        x.wait()


class DataParallelAdam:
    """fp32 master weights + Adam slots in flat device buffers; ``step()`` = all-reduce -> global
    norm -> clip -> Adam, returning the (device) global gradient norm.  exchange = "zero1": reduce onto the shard owners ->
    norm of the shards (all-reduced) -> clip + Adam on the own shard -> gather the masters (see above)."""

    def __init__(self, hp, params, device="cuda", clip=1.0, beta1=0.9, beta2=0.999, eps=1e-8, group=None,
                 bucket_elems=64 << 20, grad_reduce_dtype="fp32", exchange="allreduce"):
        import torch
        if grad_reduce_dtype not in REDUCE_DTYPES:
            raise ValueError("grad_reduce_dtype must be one of %s" % (REDUCE_DTYPES,))
        if exchange not in EXCHANGES:
            raise ValueError("exchange must be one of %s" % (EXCHANGES,))
        if exchange == "zero1" and grad_reduce_dtype != "fp32":
            raise ValueError("exchange='zero1' reduces in fp32")
        self.exchange = exchange
        self.layout = FlatLayout(hp)
        self.device = device
        self.group = group
        self.bucket_elems = bucket_elems
        # "bf16": the gradient exchange moves bf16 (half the xGMI bytes); masters, Adam slots, the global
        # norm and the update stay fp32.  The staging buffer is allocated on first use.
        self.grad_reduce_dtype = grad_reduce_dtype
        self._g16 = None
        # True: issue the collectives even in a one-rank group (tests / bench exercise the RCCL path on one GPU)
        self.force_collectives = False
        self.clip, self.b1, self.b2, self.eps = clip, beta1, beta2, eps
        self.w = torch.from_numpy(self.layout.flatten(params)).to(device)
        self.g = torch.zeros_like(self.w)
        self.m = torch.zeros_like(self.w)
        self.v = torch.zeros_like(self.w)
        self.global_step = 0
        self._lib = _lib.load()
        self._partial = torch.empty(self._lib.fwn_grad_norm_partials(self.w.numel()), dtype=torch.float64, device=device)
        self._gnorm = torch.empty(1, dtype=torch.float32, device=device)
        self._rate = torch.zeros(1, dtype=torch.float32, device=device)     # lr_t of a recorded (hipGraph) update

    def master_views(self):
        return self.layout.views(self.w)

    def grad_views(self):
        return self.layout.views(self.g)

    def block_ranges(self):
        """[lo, hi) element ranges of the flat buffers: the up-sampling convs, then one per block, in
        layout order (the backward pass finishes them last block first)."""
        out, cur, lo = [], None, 0
        for name, (off, _, n) in self.layout.slots.items():
            head = name.split("/")[0]
            key = head if head.startswith("Block_") else "upsample"
            if key != cur:
                if cur is not None:
                    out.append((cur, lo, off))
                cur, lo = key, off
        out.append((cur, lo, self.layout.size))
        return out

    def weights_checksum(self):
        """64-bit position-weighted hash of the master weights' bit patterns (device tensor, int64).  Two ranks hold the
        same masters iff (up to hash collisions) the sums agree; wrap-around arithmetic, order-free, chunked so the
        int64 temporaries stay small."""
        import torch
        bits = self.w.view(torch.int32)
        acc = torch.zeros((), dtype=torch.int64, device=bits.device)
        chunk = 1 << 24
        for lo in range(0, bits.numel(), chunk):
            b = bits[lo:lo + chunk].to(torch.int64)
            pos = torch.arange(lo, lo + b.numel(), dtype=torch.int64, device=bits.device)
            acc = acc + (b * (2 * (pos % 1000003) + 1)).sum()
        return acc

    def weights_identical(self):
        """True iff every rank of the group holds bit-identical master weights (all-reduce MIN / MAX of the checksum):
        the lock-step invariant of the data-parallel step (utils.py:34-60: one averaged gradient, one update)."""
        import torch
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return True
        h = self.weights_checksum().reshape(1)
        lo, hi = h.clone(), h.clone()
        if lo.is_cuda and dist.get_backend(self.group) == "gloo":      # shared-GPU plumbing tests reduce on the host
            lo, hi = lo.cpu(), hi.cpu()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        return int(lo.item()) == int(hi.item())

    def allreduce_range(self, lo, hi):
        """Start the all-reduce (sum) of ``g[lo:hi]`` now - called by the training step as soon as a
        block's gradients are complete, so the exchange overlaps the rest of the backward pass."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return None
        if dist.get_world_size(self.group) == 1 and not self.force_collectives:
            return None
        if self.exchange == "zero1":         # the sharded exchange runs in step(): a block's range is not a shard
            return None
        return allreduce_slice(self.g, lo, hi, self.group, self.grad_reduce_dtype, self._stage())

    def _stage(self):
        import torch
        if self.grad_reduce_dtype == "bf16" and self._g16 is None:
            self._g16 = torch.empty(self.g.numel(), dtype=torch.bfloat16, device=self.g.device)
        return self._g16

    def step(self, loss_scale=1.0, works=None):
        """``self.g`` holds this rank's gradient of (loss_scale * loss).  works: handles of all-reduces
        already started over the whole buffer (``allreduce_range``); None: reduce everything here."""
        import torch
        import torch.distributed as dist
        world = dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1
        if self.exchange == "zero1" and (world > 1 or (self.force_collectives and dist.is_available() and dist.is_initialized())):
            return self._step_zero1(world, loss_scale)
        for w in (allreduce_flat(self.g, self.group, self.bucket_elems, True, self.grad_reduce_dtype, self._stage()) if works is None else works):
            if w is not None:
                w.wait()
        gscale = 1.0 / (world * float(loss_scale))
        st = torch.cuda.current_stream(torch.device(self.device)).cuda_stream
        n = self.w.numel()
        _lib.check(self._lib.fwn_grad_norm(self.g.data_ptr(), n, gscale, self._partial.data_ptr(),
                                           self._gnorm.data_ptr(), st), "fwn_grad_norm")
        lr = learning_rate(self.global_step)
        self.global_step += 1
        _lib.check(self._lib.fwn_clip_adam(self.w.data_ptr(), self.g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                           n, self._gnorm.data_ptr(), gscale, self.clip, lr, self.global_step,
                                           self.b1, self.b2, self.eps, st), "fwn_clip_adam")
        return self._gnorm

    def _step_zero1(self, world, loss_scale):
        """The sharded update: this rank's clip + Adam touch its shard only."""
        import torch
        import torch.distributed as dist
        (lo, hi), _ = reduce_to_owners(self.g, self.group)
        gscale = 1.0 / (world * float(loss_scale))
        st = torch.cuda.current_stream(torch.device(self.device)).cuda_stream
        n = hi - lo
        # ||g||^2 = the sum of the owners' shard norms squared (fp32 all-reduce of one scalar per rank; the shard norm itself
        # is fwn_grad_norm's deterministic fp64 tree): the same value on every rank, the summation order of the all-reduce
        # path's single tree is not reproduced - the two paths agree to fp32 rounding of the clip factor, not bit for bit
        if n > 0:
            _lib.check(self._lib.fwn_grad_norm(self.g.data_ptr() + 4 * lo, n, gscale, self._partial.data_ptr(),
                                               self._gnorm.data_ptr(), st), "fwn_grad_norm")
        else:
            self._gnorm.zero_()
        sq = (self._gnorm.double() ** 2)
        if sq.is_cuda and dist.get_backend(self.group) == "gloo":
            sq = sq.cpu()
        dist.all_reduce(sq, group=self.group)
        self._gnorm.copy_(sq.sqrt().to(self._gnorm.dtype))
        lr = learning_rate(self.global_step)
        self.global_step += 1
        if n > 0:
            _lib.check(self._lib.fwn_clip_adam(self.w.data_ptr() + 4 * lo, self.g.data_ptr() + 4 * lo, self.m.data_ptr() + 4 * lo,
                                               self.v.data_ptr() + 4 * lo, n, self._gnorm.data_ptr(), gscale, self.clip, lr,
                                               self.global_step, self.b1, self.b2, self.eps, st), "fwn_clip_adam")
        gather_from_owners(self.w, self.group)
        return self._gnorm

    # -- the same update split for a recorded step: ``record_update`` puts the two launches on the current
    # (capturing) stream with the rate read from device memory; ``advance`` is called before every replay.
    def record_update(self, loss_scale=1.0):
        import torch
        import torch.distributed as dist
        if self.exchange == "zero1":
            raise NotImplementedError("the recorded (hipGraph) step runs the all-reduce exchange; exchange='zero1' is eager only")
        world = dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1
        gscale = 1.0 / (world * float(loss_scale))
        st = torch.cuda.current_stream(torch.device(self.device)).cuda_stream
        n = self.w.numel()
        _lib.check(self._lib.fwn_grad_norm(self.g.data_ptr(), n, gscale, self._partial.data_ptr(),
                                           self._gnorm.data_ptr(), st), "fwn_grad_norm")
        _lib.check(self._lib.fwn_clip_adam_dev(self.w.data_ptr(), self.g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                               n, self._gnorm.data_ptr(), gscale, self.clip, self._rate.data_ptr(),
                                               self.b1, self.b2, self.eps, st), "fwn_clip_adam_dev")
        return self._gnorm

    def advance(self):
        """global_step += 1 and the device-resident rate of that step (same arithmetic as fwn_clip_adam)."""
        lr = learning_rate(self.global_step)
        self.global_step += 1
        self._rate.fill_(float(self._lib.fwn_adam_rate(lr, self.global_step, self.b1, self.b2)))
