"""Per-launch time of the tail at the bench shapes (blocks 0 - 3 of the 8-clip pass and of one clip): the register-streamed
kernel (csrc/tail_rs.h, the flow's Wts stream) against the kernels it replaces (the same call with Wts = NULL), interleaved
in one process, rotating over the six flows of the block so that the weights are not L2-warm.

    python tools/probe/tail_rs_bench.py [B] [rounds]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tf_flowavenet_amd import _lib, weights as W          # noqa: E402
from tf_flowavenet_amd.hparams import default_hparams     # noqa: E402
from tf_flowavenet_amd.model import FloWaveNet            # noqa: E402


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    hp = default_hparams()
    model = FloWaveNet(hp, init=True).load_params(W.synthetic_params(hp, 1234))
    inp = W.synthetic_inputs(hp, 2, 16128)
    model.forward(torch.from_numpy(inp["x"]).cuda(), torch.from_numpy(inp["c"]).cuda())
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    T = 16128
    print("B = %d, T = %d; us per launch (median of %d rounds x 6 flows), chained with the next flow's front conv where Ch <= 8" % (nb, T, rounds))
    for blk in range(0, 6):
        ch, L = 1 << blk, hp.n_layer
        ti = T // (2 << blk)
        m = nb * ti
        descs = [model._packed.flow_descs[blk * hp.n_flow + j] for j in range(hp.n_flow)]
        if not descs[0].Wts or m < lib.fwn_tail_stream_rows():
            continue
        plains = []
        for d in descs:
            dp = _lib.FlowDesc.from_buffer_copy(d)
            dp.Wts = None
            plains.append(dp)
        rng = np.random.default_rng(blk)
        o = torch.from_numpy((rng.random((L, m, 256)) * 0.8).astype(np.float32)).cuda().to(torch.bfloat16)
        xa = torch.from_numpy(rng.standard_normal((m, ch)).astype(np.float32)).cuda()
        xb = torch.from_numpy(rng.standard_normal((m, ch)).astype(np.float32)).cuda()
        xo = torch.empty_like(xb)
        h0 = torch.empty(m, 256, device="cuda", dtype=torch.bfloat16)
        scratch = torch.empty(2, m, 256, device="cuda", dtype=torch.bfloat16)
        part = torch.zeros(lib.fwn_tail_partials_chained(m, ch, 1) + 8, device="cuda")
        front = ch <= 8

        def run(ds, j):
            nx = ds[(j + 1) % len(ds)]
            _lib.check(lib.fwn_tail_chained(C.byref(ds[j]), C.byref(nx) if front else None, o.data_ptr(), xa.data_ptr(), xb.data_ptr(), xo.data_ptr(),
                                            h0.data_ptr() if front else None, part.data_ptr(), m, ti, 0, scratch.data_ptr(), st), "fwn_tail_chained")

        times = {"new": [], "old": []}
        for r in range(rounds + 2):
            for name, ds in (("new", descs), ("old", plains)):
                for j in range(len(ds)):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    run(ds, j)
                    e1.record()
                    e1.synchronize()
                    if r >= 2:
                        times[name].append(e0.elapsed_time(e1) * 1e3)
        gf = 2.0 * m * (256 * L + 256 + 2 * ch) * 256 / 1e9
        print("block %d  rows %6d  Ch %2d  tail %5.1f GFLOP   new %6.1f us (min %6.1f)   old %6.1f us (min %6.1f)   new / old %.2f" % (
            blk, m, ch, gf, np.median(times["new"]), np.min(times["new"]), np.median(times["old"]), np.min(times["old"]),
            np.median(times["new"]) / np.median(times["old"])))


if __name__ == "__main__":
    main()
