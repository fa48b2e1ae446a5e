"""Per-launch time of the hoisted conditioning projection (csrc/flow_kernels.hip cond_batch_kernel) at the shapes of the blocks that hoist
it (B clips x 16 128 samples: blocks 4 - 7 at B = 8, 2 - 7 at B = 1), through the C ABI's fwn_cond_split + fwn_cond_reduce: all 12 (flow, layer) matrices of a
block in one launch, nsplit K ranges per tile.  Tile shapes are picked inside the library; with the tuning build
(`make -C tf-flowavenet_amd/csrc tune`, FWN_LIB=.../libfwn_tune.so) FWN_COND_TILE / FWN_COND_SPLIT_TILE override them
(0 = 256 x 256, 1 = 256 x 128, 2 = 128 x 128, 3 = 64 x 128).

    python tools/probe/cond_bench.py [B] [nsplit per block 4..7, comma separated, 0 = the library's choice] [rounds] [nsplit of the streamed form]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tf_flowavenet_amd import _lib          # noqa: E402


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    splits = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 0, 0, 0]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    splits2 = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else None      # split counts of the streamed form
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    T, nflow, L = 16128, 6, 2
    nz = nflow * L
    print("B = %d; tiles: FWN_COND_TILE=%s FWN_COND_SPLIT_TILE=%s" % (nb, os.environ.get("FWN_COND_TILE"), os.environ.get("FWN_COND_SPLIT_TILE")))
    blocks = [b for b in range(8) if nb * (T // (2 << b)) < 4096 and 40 * (2 << b) >= 256]      # the blocks whose conditioning is hoisted
    splits = (splits + [0] * 8)[:len(blocks)]
    if splits2:
        splits2 = (splits2 + [0] * 8)[:len(blocks)]
    for i, blk in enumerate(blocks):
        m = nb * (T // (2 << blk))
        cin = 40 * (2 << blk)
        kcpad = (cin + 63) // 64 * 64
        ns = splits[i] or lib.fwn_cond_splits(m, nz, kcpad)
        g = torch.Generator(device="cuda").manual_seed(blk)
        ca = (torch.rand(2, m, cin, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
        # three sets of weights: rotating over them keeps a launch from finding its matrices in L2 / MALL
        wc = [(torch.rand(nz, 512, kcpad, device="cuda", generator=g) - 0.5).to(torch.bfloat16) for _ in range(3)]
        p = torch.empty(nz, m, 512, device="cuda")
        part = torch.empty(max(ns - 1, 1), nz, m, 512, device="cuda")

        def run(w):
            _lib.check(lib.fwn_cond_split(ca[0].data_ptr(), w.data_ptr(), p.data_ptr(), 512 * kcpad, m * 512, 0, 1, nflow, L, m, cin, kcpad,
                                          part.data_ptr(), nz * m * 512, ns, st), "fwn_cond_split")
            _lib.check(lib.fwn_cond_reduce(p.data_ptr(), part.data_ptr(), nz * m * 512, ns, nz * m * 512, st), "fwn_cond_reduce")

        run(wc[0])
        ref = ca[0, :64].float() @ wc[0][3, :, :cin].float().t()
        err = (p[3, :64] - ref).abs().max().item() / ref.abs().max().item()
        ring_p = p.clone()
        ts = []
        for r in range(rounds + 2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(wc[r % 3])
            e1.record()
            e1.synchronize()
            if r >= 2:
                ts.append(e0.elapsed_time(e1) * 1e3)
        gf = 2.0 * m * cin * 512 * nz / 1e9
        wmb = nz * 512 * kcpad * 2 / 1e6
        med = float(np.median(ts))
        # the register-streamed form (csrc/cond_rs.h) on the same operands: its own split count, streams packed once
        rs_txt = ""
        if hasattr(lib, "fwn_cond_stream") and m >= lib.fwn_cond_stream_rows():
            ns2 = (splits2[i] if splits2 else 0) or lib.fwn_cond_stream_splits(m, nz, kcpad)
            ws = [torch.empty_like(w) for w in wc]
            for w, o in zip(wc, ws):
                _lib.check(lib.fwn_pack_cond_stream(w.data_ptr(), 512 * kcpad, kcpad, nz, o.data_ptr(), st), "fwn_pack_cond_stream")
            part2 = torch.empty(max(ns2 - 1, 1), nz, m, 512, device="cuda")
            p2 = torch.full_like(p, float("nan"))

            def run2(w):
                _lib.check(lib.fwn_cond_stream(ca[0].data_ptr(), None, w.data_ptr(), p2.data_ptr(), nflow, L, m, cin, kcpad, part2.data_ptr(),
                                               nz * m * 512, ns2, st), "fwn_cond_stream")
                _lib.check(lib.fwn_cond_reduce(p2.data_ptr(), part2.data_ptr(), nz * m * 512, ns2, nz * m * 512, st), "fwn_cond_reduce")

            run2(ws[0])
            torch.cuda.synchronize()
            same = bool(torch.equal(p2, ring_p)) if ns2 == ns else None
            dmax = float((p2 - ring_p).abs().max())
            t2 = []
            for r in range(rounds + 2):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run2(ws[r % 3])
                e1.record()
                e1.synchronize()
                if r >= 2:
                    t2.append(e0.elapsed_time(e1) * 1e3)
            rs_txt = "   | streamed: nsplit %d  %6.1f us (min %6.1f)  %6.1f TFLOP/s  max diff to ring %.2e%s" % (
                ns2, float(np.median(t2)), min(t2), gf / float(np.median(t2)) * 1e3, dmax, "" if same is None else (" bit-identical" if same else " NOT bit-identical"))
        print("block %d  rows %5d  cin %5d  nsplit %d  %5.1f GFLOP  weights %6.1f MB   %6.1f us (min %6.1f)  %6.1f TFLOP/s  weights at %4.2f TB/s   rel err %.1e" % (
            blk, m, cin, ns, gf, wmb, med, min(ts), gf / med * 1e3, wmb / med, err) + rs_txt)


if __name__ == "__main__":
    main()
