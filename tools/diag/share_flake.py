"""Diagnostic: N processes on one GPU, each repeating the same forward / gradient; count bit mismatches vs the first."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(rank, what, iters, full):
    import torch
    from conftest import small_hparams
    from tf_flowavenet_amd import weights as W
    from tf_flowavenet_amd.hparams import default_hparams
    from tf_flowavenet_amd.model import FloWaveNet
    from tf_flowavenet_amd.training import GradEngine
    if full:
        hp, b, t = default_hparams(), 2, 6400
    else:
        hp, b, t = small_hparams(n_block=3, n_flow=2, n_layer=2, hop_size=16, upsample_scales=[4, 4], num_mels=16), 2, 256
    inp = W.synthetic_inputs(hp, b, t)
    x, c = torch.from_numpy(inp["x"]).reshape(b, t).cuda(), torch.from_numpy(inp["c"]).cuda()
    params = W.synthetic_params(hp, 11)
    bad, ref = 0, None
    if what == "fwd":
        m = FloWaveNet(hp, device="cuda").load_params(params)
        for it in range(iters):
            lp, ld, z = m.forward(x.reshape(b, t, 1), c, return_z=True) if "return_z" in m.forward.__code__.co_varnames else m.forward(x.reshape(b, t, 1), c) + (None,)
            cur = (float(lp), float(ld))
            if ref is None: ref = cur
            elif cur != ref: bad += 1
    else:
        eng = GradEngine(hp, "cuda")
        dp = {k: torch.from_numpy(v).cuda() for k, v in params.items()}
        for it in range(iters):
            loss, _, _, grads = eng.loss_and_grads(dp, x, c)
            cur = torch.cat([g.reshape(-1) for g in grads.values()]).clone()
            if ref is None: ref = cur
            elif not torch.equal(cur, ref):
                bad += 1
                if bad <= 3:
                    names = [k for k, g in grads.items()]
                    off, first = 0, None
                    for k, g in grads.items():
                        n = g.numel()
                        if not torch.equal(cur[off:off + n], ref[off:off + n]):
                            first = k if first is None else first
                        off += n
                    print("rank", rank, "iter", it, "ndiff", int((cur != ref).sum()), "first differing tensor", first, flush=True)
    print("rank", rank, what, "full" if full else "small", "iters", iters, "mismatching iterations:", bad, flush=True)


if __name__ == "__main__":
    import torch.multiprocessing as mp
    what, nproc, iters, full = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, what, iters, full)) for r in range(nproc)]
    [p.start() for p in ps]; [p.join() for p in ps]
