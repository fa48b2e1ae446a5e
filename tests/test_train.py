"""Training-side primitives (work in progress towards the backward pass): the generic multi-segment
GEMM, the shifted transpose and the deterministic split-K reduction against NumPy fp64."""
import os

import numpy as np
import pytest
import torch

from tf_flowavenet_amd import training as TR

pytestmark = pytest.mark.gpu


def bf(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda().to(torch.bfloat16)


def f64(t):
    return t.float().cpu().numpy().astype(np.float64)


def shifted(x, shift, ti):
    """rows r -> x[r + shift] inside clips of ti rows (zero outside), fp64."""
    m = x.shape[0]
    out = np.zeros_like(x)
    r = np.arange(m)
    t = r % ti if ti else r
    ok = (t + shift >= 0) & (t + shift < (ti if ti else m))
    out[ok] = x[r[ok] + shift]
    return out


@pytest.mark.parametrize("m,ti", [(300, 100), (5000, 1000), (40000, 8000)])
def test_gemm_conv_like_with_epilogue(m, ti):
    rng = np.random.default_rng(m)
    x = bf(rng.standard_normal((m, 256)) * 0.5)
    c = bf(rng.random((m, 72)))                       # k = 72: not a multiple of the 64-wide chunk
    w = bf(rng.standard_normal((200, 3 * 256 + 128)) * 0.05)
    w[:, 768 + 72:] = 0
    bias = torch.from_numpy(rng.standard_normal(200).astype(np.float32)).cuda()
    res = bf(rng.standard_normal((m, 200)))
    mask = bf(rng.standard_normal((m, 200)))
    segs = [(x, 256, -3, 0), (x, 256, 0, 256), (x, 256, 3, 512), (c, 72, 0, 768)]
    got = TR.gemm(segs, w, 200, m, ti=ti, bias=bias, res=res, rscale=0.5, mask=mask, relu=True, oscale=2.0)
    xw, ww = f64(x), f64(w)
    want = sum(shifted(xw, sh, ti) @ ww[:, k0:k0 + 256].T for sh, k0 in ((-3, 0), (0, 256), (3, 512)))
    want = want + f64(c) @ ww[:, 768:768 + 72].T + f64(bias) + 0.5 * f64(res)
    want = 2.0 * np.maximum(np.where(f64(mask) > 0, want, 0.0), 0.0)
    err = np.abs(f64(got) - want)
    assert err.max() < 0.05 * max(1.0, np.abs(want).max()) and err.mean() < 4e-3 * max(1.0, np.abs(want).mean())


@pytest.mark.parametrize("m", [300, 5000, 30000])      # the 32 x 64, 64 x 128 and 256 x 128 tiles
def test_gemm_gate_derivative_epilogue_equals_store_then_gate_bwd(m):
    """fwn_gemm_desc.gate_aux: the 256 output columns from gate_col0 leave through the gate's derivative instead of being
    stored - bit for bit what storing them and running fwn_gate_bwd gives; the other columns are stored as always."""
    rng = np.random.default_rng(m)
    x = bf(rng.standard_normal((m, 256)) * 0.5)
    w = bf(rng.standard_normal((512, 256)) * 0.05)
    res = bf(rng.standard_normal((m, 512)))
    aux = bf(np.concatenate([np.tanh(rng.standard_normal((m, 256))), 1 / (1 + np.exp(-rng.standard_normal((m, 256))))], 1))
    plain = TR.gemm([(x, 256, 0, 0)], w, 512, m, res=res, rscale=0.5, oscale=0.7)
    want = torch.empty(m, 512, dtype=torch.bfloat16, device="cuda")
    TR._lib.check(TR._lib.load().fwn_gate_bwd(plain[:, 256:].data_ptr(), 512, aux.data_ptr(), m, want.data_ptr(), None), "fwn_gate_bwd")
    out = torch.full((m, 512), 3.0, dtype=torch.bfloat16, device="cuda")
    dpre = torch.empty(m, 512, dtype=torch.bfloat16, device="cuda")
    TR.gemm([(x, 256, 0, 0)], w, 512, m, res=res, rscale=0.5, oscale=0.7, out=out, gate=(aux, dpre, 256))
    torch.cuda.synchronize()
    assert torch.equal(dpre, want)
    assert torch.equal(out[:, :256], plain[:, :256]) and bool((out[:, 256:] == 3.0).all())
    with pytest.raises(RuntimeError):
        TR.gemm([(x, 256, 0, 0)], w, 512, m, gate=(aux, dpre, 384))


def test_gemm_fp32_accumulate_and_padding_columns_untouched():
    rng = np.random.default_rng(1)
    m, n = 700, 40
    x, w = bf(rng.standard_normal((m, 64))), bf(rng.standard_normal((n, 64)) * 0.1)
    out = torch.full((m, 48), 7.0, device="cuda")
    TR.gemm([(x, 64, 0, 0)], w, n, m, out=out, accumulate=True)
    want = f64(x) @ f64(w).T + 7.0
    assert np.abs(out[:, :n].cpu().numpy() - want).max() < 1e-3
    assert bool((out[:, n:] == 7.0).all())


@pytest.mark.parametrize("m,ti,shifts", [(333, 111, (0,)), (6000, 1000, (-1, 0, 1)), (25600, 3200, (-3, 0, 3))])
def test_weight_and_bias_gradient_via_transposes_and_split_k(m, ti, shifts):
    rng = np.random.default_rng(m)
    kx, n = 256, 512
    x, dy = bf(rng.standard_normal((m, kx)) * 0.5), bf(rng.standard_normal((m, n)) * 0.1)
    dw, db = TR.weight_grad(x, dy, m, kx, n, shifts=shifts, ti=ti)
    dw2, db2 = TR.weight_grad(x, dy, m, kx, n, shifts=shifts, ti=ti)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)          # fixed summation order
    xw, dyw = f64(x), f64(dy)
    want = np.concatenate([shifted(xw, sh, ti).T @ dyw for sh in shifts])
    scale = np.abs(want).max()
    assert np.abs(dw.cpu().numpy() - want).max() < 2e-3 * scale
    np.testing.assert_allclose(db.cpu().numpy(), dyw.sum(0), atol=2e-3 * np.abs(dyw.sum(0)).max())


# ------------------------------------------------------------------ whole-model gradients
# Per-tensor agreement with fp64 autograd, ||got - want|| / ||want||: activations and their gradients are bf16, so a
# tensor agrees to a few % of its own norm plus a noise floor that scales with the whole gradient (sums that cancel:
# a tensor whose norm is tiny next to the rest of the gradient may deviate more, never beyond 30 %).
GRAD_REL, GRAD_FLOOR, GRAD_REL_MAX = 0.1, 2e-3, 0.3


GRAD_BIG, GRAD_REL_BIG = 0.01, 0.1       # tensors carrying >= 1 % of the whole gradient's norm: no absolute floor
_SEEN_BIG = []


def grad_rel_allowed(norm_ref, norm_total, rel=None):
    """Allowed relative error of one gradient tensor.  Small tensors (cancelling sums of bf16 products) get an absolute floor
    of 2e-3 of the whole gradient's norm; a tensor that carries 1 % of the whole gradient or more must be within GRAD_REL_BIG
    on its own (the floor let such a tensor be 30 % off)."""
    if norm_ref >= GRAD_BIG * norm_total:
        if rel is not None:
            _SEEN_BIG.append(rel)
        return GRAD_REL_BIG
    return min(GRAD_REL_MAX, GRAD_REL + GRAD_FLOOR * norm_total / norm_ref)


def _grad_case(cfg, b, t, seed):
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from conftest import small_hparams
    from oracle import grad_torch as G
    from tf_flowavenet_amd import weights as W
    from tf_flowavenet_amd.training import GradEngine
    hp = small_hparams(**cfg)
    p = W.synthetic_params(hp, seed)
    inp = W.synthetic_inputs(hp, b, t)
    ref = G.loss_and_grads(p, inp["x"], inp["c"], hp)
    got = GradEngine(hp).loss_and_grads(p, torch.from_numpy(inp["x"]).reshape(b, t), torch.from_numpy(inp["c"]))
    return hp, p, ref, got


@pytest.mark.parametrize("cfg,b,t", [
    (dict(n_block=2, n_flow=2, n_layer=2, hop_size=16, upsample_scales=[4, 4], num_mels=8), 2, 128),
    (dict(n_block=3, n_flow=3, n_layer=2, hop_size=16, upsample_scales=[4, 4], num_mels=16), 3, 256),
    (dict(n_block=4, n_flow=2, n_layer=3, hop_size=32, upsample_scales=[4, 8], num_mels=16), 2, 512),
    # 4 layers: 18 weight-gradient GEMMs and 26 weight-norm jobs per flow - more than one group (FWN_MAX_GROUP = 16)
    (dict(n_block=4, n_flow=2, n_layer=4, hop_size=16, upsample_scales=[4, 4], num_mels=16), 2, 512),
])
def test_loss_and_all_parameter_gradients_match_autograd_oracle(cfg, b, t):
    """GradEngine (training forward + backward through the stage kernels) against autograd of the
    fp64 oracle for loss = -(log_p + logdet) (train.py:56-66): every one of the reference's
    trainable tensors, including weight-norm V / g, ZeroConv scale, ActNorm and the up-sampling
    kernels.  Activations and activation gradients are bf16: per-tensor agreement is to bf16 noise
    (a few % of the tensor's norm; small reductions with cancellation up to ~20 %)."""
    hp, p, (loss0, lp0, ld0, g0), (loss, lp, ld, g) = _grad_case(cfg, b, t, 5)
    assert abs(float(lp) - lp0) < 1e-3 * abs(lp0) and abs(float(ld) - ld0) < 1e-3 * max(1.0, abs(ld0))
    assert sorted(g) == sorted(g0)
    rels, dot, na, nb_ = [], 0.0, 0.0, 0.0
    total = float(np.sqrt(sum(float((v * v).sum()) for v in g0.values())))
    for k in g0:
        a, r = g[k].cpu().numpy().astype(np.float64), g0[k]
        assert a.shape == r.shape, k
        if "res_conv" in k and ("ResBlock_%d/" % (hp.n_layer - 1)) in k:
            assert not a.any() and not r.any(), k          # dead conv (modules.py:126-128): zero gradient
            continue
        rel = np.linalg.norm(a - r) / np.linalg.norm(r)
        rels.append(rel)
        assert rel < grad_rel_allowed(np.linalg.norm(r), total, rel), (k, rel, np.linalg.norm(r), total)
        dot += float((a * r).sum()); na += float((a * a).sum()); nb_ += float((r * r).sum())
    assert np.median(rels) < 3e-2, np.median(rels)
    assert dot / np.sqrt(na * nb_) > 0.999            # direction of the whole gradient


@pytest.mark.parametrize("n_block,b,t", [(6, 2, 1024), (8, 2, 1024), ("hp8000", 2, 960)])
def test_full_width_model_gradients_match_autograd_oracle(n_block, b, t):
    """The real architecture (hop 256, 80 mels, n_flow=6, n_layer=2; BASELINE configs[2]'s model at n_block=8, and
    n_block=6 as the cheap case that already reaches the Ch = 32 ring front conv and the hoisted conditioning
    backward): all trainable tensors (2 262 at n_block=8, 181 M elements) against fp64 autograd of the oracle.
    "hp8000": the reference's second configuration (hparams8000.py: n_block=5, hop 96 = 8 x 12)."""
    from oracle import grad_torch as G
    from tf_flowavenet_amd import weights as W
    from tf_flowavenet_amd.hparams import default_hparams, hparams8000
    from tf_flowavenet_amd.training import GradEngine
    hp = hparams8000() if n_block == "hp8000" else default_hparams().replace(n_block=n_block)
    p = W.synthetic_params(hp, 1234, actnorm="random")
    inp = W.synthetic_inputs(hp, b, t)
    loss0, lp0, ld0, g0 = G.loss_and_grads(p, inp["x"], inp["c"], hp)
    loss, lp, ld, g = GradEngine(hp).loss_and_grads(p, torch.from_numpy(inp["x"]).reshape(b, t), torch.from_numpy(inp["c"]))
    torch.cuda.synchronize()
    assert abs(float(lp) - lp0) < 1e-3 * abs(lp0) and abs(float(ld) - ld0) < 1e-3 * max(1.0, abs(ld0))
    assert sorted(g) == sorted(g0)
    rels, dot, na, nb_ = [], 0.0, 0.0, 0.0
    total = float(np.sqrt(sum(float((v * v).sum()) for v in g0.values())))
    for k in sorted(g0):
        a, r = g[k].detach().cpu().numpy().astype(np.float64).reshape(-1), g0[k].reshape(-1)
        nr = np.linalg.norm(r)
        if nr == 0:
            assert not a.any(), k                      # dead res_conv of the last layer
            continue
        rel = np.linalg.norm(a - r) / nr
        rels.append(rel)
        assert rel < grad_rel_allowed(nr, total, rel), (k, rel, nr, total)
        dot += float(a @ r); na += float(a @ a); nb_ += float(r @ r)
    assert len(rels) == len(g0) - 3 * hp.n_block * hp.n_flow
    print("tensors with >= 1 %% of the gradient norm: %d, worst relative error %.3f" % (len(_SEEN_BIG), max(_SEEN_BIG) if _SEEN_BIG else 0.0))
    _SEEN_BIG.clear()
    assert np.median(rels) < 4e-2, np.median(rels)
    assert dot / np.sqrt(na * nb_) > 0.9999


@pytest.mark.parametrize("cfg,b,t", [
    (dict(n_block=2, n_flow=2, n_layer=2, hop_size=16, upsample_scales=[4, 4], num_mels=8), 2, 128),
    (dict(n_block=4, n_flow=2, n_layer=3, hop_size=32, upsample_scales=[4, 8], num_mels=16), 2, 512),
    (dict(n_block=4, n_flow=2, n_layer=4, hop_size=16, upsample_scales=[4, 4], num_mels=16), 2, 512),     # two job groups per flow
    ("full6", 2, 1024),                                                                                   # real widths, Ch = 32 front, hoisted cond
])
def test_the_c_sequenced_step_reproduces_the_python_sequenced_reference_bit_for_bit(cfg, b, t):
    """fwn_train_loss_and_grads (csrc/train_api.hip: one C call per batch) against the same stage kernels sequenced from
    Python (tests/dev/py_sequencing.py, the previous product path): loss, log_p, logdet and every gradient tensor are
    identical in every bit - the C entry point changes who sequences the launches, not the arithmetic."""
    import importlib.util, os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from conftest import small_hparams
    from tf_flowavenet_amd import weights as W
    from tf_flowavenet_amd.hparams import default_hparams
    from tf_flowavenet_amd.training import GradEngine
    spec = importlib.util.spec_from_file_location("py_sequencing", os.path.join(os.path.dirname(os.path.abspath(__file__)), "dev", "py_sequencing.py"))
    ref_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_mod)
    hp = default_hparams().replace(n_block=6) if cfg == "full6" else small_hparams(**cfg)
    p = W.synthetic_params(hp, 21, actnorm="random")
    inp = W.synthetic_inputs(hp, b, t)
    x, c = torch.from_numpy(inp["x"]).reshape(b, t).cuda(), torch.from_numpy(inp["c"]).cuda()
    eng = GradEngine(hp)
    loss, lp, ld, g = eng.loss_and_grads(p, x, c)
    loss, lp, ld = float(loss), float(lp), float(ld)
    g = {k: v.clone() for k, v in g.items()}
    l0, lp0, ld0, g0 = ref_mod.loss_and_grads(eng, eng._tp, p, x, c)
    assert (loss, lp, ld) == (float(l0), float(lp0), float(ld0))
    assert sorted(g) == sorted(g0)
    for k in g0:
        a, r = g[k].reshape(-1), g0[k].reshape(-1)
        if k.startswith("upsample_") and k.endswith("/g"):      # a 3-term sum: torch's reduction order is its own
            assert torch.allclose(a, r, rtol=1e-6, atol=0), k
        else:
            assert torch.equal(a, r), k


@pytest.mark.parametrize("cfg,b,t", [
    (dict(n_block=4, n_flow=2, n_layer=3, hop_size=32, upsample_scales=[4, 8], num_mels=16), 2, 512),
    ("full6", 2, 1024),
])
def test_side_stream_weight_gradients_equal_the_one_stream_call_bit_for_bit(cfg, b, t, monkeypatch):
    """fwn_train_desc.side_stream moves each block's weight-gradient GEMMs and weight-norm backward to a second stream
    under the next block's data-gradient chain: same kernels on per-flow copies of the temporaries - same bits; and the
    block callbacks still arrive last block first, each after its gradients are enqueued on the caller's stream."""
    from conftest import small_hparams
    from tf_flowavenet_amd import weights as W
    from tf_flowavenet_amd.hparams import default_hparams
    from tf_flowavenet_amd.training import GradEngine
    hp = default_hparams().replace(n_block=6) if cfg == "full6" else small_hparams(**cfg)
    p = W.synthetic_params(hp, 23, actnorm="random")
    inp = W.synthetic_inputs(hp, b, t)
    x, c = torch.from_numpy(inp["x"]).reshape(b, t).cuda(), torch.from_numpy(inp["c"]).cuda()
    res = {}
    for side in ("0", "1"):
        monkeypatch.setenv("FWN_TRAIN_SIDE", side)
        eng = GradEngine(hp)
        order = []
        for rep in range(2):        # the second call reuses the workspace and the event pool
            order.clear()
            loss, lp, ld, g = eng.loss_and_grads(p, x, c, on_block_done=order.append)
            torch.cuda.synchronize()
        assert order == list(range(hp.n_block - 1, -1, -1)) + [-1]
        res[side] = (float(loss), float(lp), float(ld), {k: v.clone() for k, v in g.items()})
    assert res["0"][:3] == res["1"][:3]
    for k in res["0"][3]:
        assert torch.equal(res["0"][3][k], res["1"][3][k]), k


def test_directional_derivatives_at_the_training_shape_match_the_oracle_forward():
    """the gradient at the BENCH's training shape (n_block = 8, B = 8 crops of 6400 samples,
    hparams.py:28,36 - 25 600 rows at block 0: the 256 x 256 halo gate, the 256 x 256 TN tiles, the fused tails), which fp64
    autograd cannot reach.  tests/golden/make_grad_dir.py took central differences of the fp64 oracle's FORWARD loss along
    one direction per parameter family (the normalised toy-batch autograd gradient of that family, recomputed here on the
    CPU); <g_HIP, d> must match them.  Tolerance: 2 % of the derivative (bf16 hidden activations; measured below)."""
    import importlib.util
    from tf_flowavenet_amd import weights as W
    from tf_flowavenet_amd.hparams import default_hparams
    from tf_flowavenet_amd.training import GradEngine
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_grad_dir", os.path.join(here, "make_grad_dir.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    fx = np.load(os.path.join(here, mg.NAME + ".npz"))
    hp = default_hparams()
    b, t = int(fx["b"]), int(fx["t"])
    assert (b, t) == (hp.batch_size, hp.max_time_steps)
    p = W.synthetic_params(hp, int(fx["seed"]), actnorm="random")
    dirs = mg.directions(p, hp)                       # fp64 autograd on the toy batch (CPU, ~15 s)
    inp = W.synthetic_inputs(hp, b, t)
    loss, lp, ld, g = GradEngine(hp).loss_and_grads(p, torch.from_numpy(inp["x"]).reshape(b, t), torch.from_numpy(inp["c"]))
    torch.cuda.synchronize()
    assert abs(float(loss) - float(fx["loss"])) <= 1e-3 * abs(float(fx["loss"]))
    for fam in mg.FAMILIES:
        got = sum(float((g[k].detach().cpu().numpy().astype(np.float64).reshape(-1) * d.reshape(-1)).sum()) for k, d in dirs[fam].items())
        want = float(fx["fd_" + fam])
        print("%-9s <g, d> = %.6e   oracle central difference %.6e   rel %.2e" % (fam, got, want, abs(got - want) / abs(want)))
        assert abs(got - want) <= 2e-2 * abs(want), (fam, got, want)


def test_side_stream_training_steps_soak_against_the_one_stream_call():
    """50 gradient calls at the bench's training shape (n_block = 8, 8 crops of 6400 samples) with the
    side stream on - each block's weight-gradient GEMMs under the next block's backward chain, kernels of both queues
    sharing CUs - every one bit-identical, loss and all 181 M gradient elements, to the same call on ONE stream."""
    from tf_flowavenet_amd import weights as W
    from tf_flowavenet_amd.hparams import default_hparams
    from tf_flowavenet_amd.training import GradEngine
    hp = default_hparams()
    b, t = hp.batch_size, hp.max_time_steps
    p = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).cuda() for k, v in W.synthetic_params(hp, 1234, actnorm="random").items()}
    inp = W.synthetic_inputs(hp, b, t)
    x, c = torch.from_numpy(inp["x"]).reshape(b, t).cuda(), torch.from_numpy(inp["c"]).cuda()
    shp = W.param_shapes(hp)
    total = sum(int(np.prod(v)) for v in shp.values())

    def views(flat):
        out, off = {}, 0
        for k, v in shp.items():
            n = int(np.prod(v))
            out[k] = flat[off:off + n].view(*v)
            off += n
        return out

    ref = {}
    for side, steps in (("0", 2), ("1", 50)):
        os.environ["FWN_TRAIN_SIDE"] = side
        try:
            eng = GradEngine(hp)
            flat = torch.zeros(total, dtype=torch.float32, device="cuda")
            go = views(flat)
            bad = 0
            for _ in range(steps):
                flat.zero_()                         # (the dead res_conv gradients are written once per engine: zeros either way)
                loss, lp, ld, _ = eng.loss_and_grads(p, x, c, grad_out=go)
                torch.cuda.synchronize()
                if side == "0":
                    ref = dict(loss=float(loss), lp=float(lp), ld=float(ld), flat=flat.clone())
                else:
                    bad += int(not ((float(loss), float(lp), float(ld)) == (ref["loss"], ref["lp"], ref["ld"])
                                    and torch.equal(flat, ref["flat"])))
            assert bad == 0, "%d of %d side-stream calls differ from the one-stream call" % (bad, steps)
        finally:
            os.environ.pop("FWN_TRAIN_SIDE", None)
    assert bool(torch.isfinite(ref["flat"]).all())


def test_an_exception_in_the_block_callback_stops_the_call_and_reaches_the_caller():
    """ctypes prints and drops an exception raised inside a callback.  A failed all-reduce / graph cut in
    on_block_done must stop the C sequencer (FWN_ERR_CALLBACK) and be re-raised by GradEngine, with one stream and two."""
    from conftest import small_hparams
    from tf_flowavenet_amd import weights as W
    from tf_flowavenet_amd.training import GradEngine
    hp = small_hparams(n_block=3, n_flow=2, n_layer=2, hop_size=16, upsample_scales=[4, 4], num_mels=8)
    p = W.synthetic_params(hp, 5, actnorm="random")
    inp = W.synthetic_inputs(hp, 2, 128)
    x, c = torch.from_numpy(inp["x"]).reshape(2, 128).cuda(), torch.from_numpy(inp["c"]).cuda()
    for side in ("0", "1"):
        os.environ["FWN_TRAIN_SIDE"] = side
        try:
            eng = GradEngine(hp)
            seen = []

            def hook(blk):
                seen.append(blk)
                if blk == 1:
                    raise RuntimeError("all-reduce of block 1 failed")

            with pytest.raises(RuntimeError, match="all-reduce of block 1 failed"):
                eng.loss_and_grads(p, x, c, on_block_done=hook)
            torch.cuda.synchronize()
            assert seen == [2, 1]                      # nothing is reported (or enqueued) after the failure
            # the raw C-ABI: a non-zero return from the callback -> FWN_ERR_CALLBACK with a message
            good = eng.loss_and_grads(p, x, c)
            torch.cuda.synchronize()
            again = eng.loss_and_grads(p, x, c, on_block_done=lambda blk: None)
            assert float(good[0]) == float(again[0])   # and the engine is usable afterwards
        finally:
            os.environ.pop("FWN_TRAIN_SIDE", None)


@pytest.mark.parametrize("m,ch", [(700, 1), (333, 8), (65, 128)])
def test_elementwise_stage_entry_points_match_numpy(m, ch):
    """fwn_actnorm_apply2 / fwn_coupling_fwd / fwn_coupling_bwd (model.py:86-94,124-141; stage entry points of the C-ABI:
    the C-sequenced step folds the first two into the inference tail since round 3) against fp64 numpy."""
    from tf_flowavenet_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(m + ch)
    st = torch.cuda.current_stream().cuda_stream
    an = rng.standard_normal((2, 4, ch)).astype(np.float32) * 0.3
    an[:, 1] = np.exp(an[:, 1]); an[:, 2] = 1.0 / an[:, 1]
    xa, xb = rng.standard_normal((m, ch)).astype(np.float32), rng.standard_normal((m, ch)).astype(np.float32)
    dxa, dxb, dan = torch.from_numpy(xa).cuda(), torch.from_numpy(xb).cuda(), torch.from_numpy(an).cuda()
    _lib.check(lib.fwn_actnorm_apply2(dxa.data_ptr(), dxb.data_ptr(), dan.data_ptr(), m * ch, ch, st), "fwn_actnorm_apply2")
    ya = (xa.astype(np.float64) + an[0, 0]) * an[0, 1]
    yb = (xb.astype(np.float64) + an[1, 0]) * an[1, 1]
    assert np.abs(dxa.cpu().numpy() - ya).max() < 1e-5 and np.abs(dxb.cpu().numpy() - yb).max() < 1e-5
    Z = (rng.standard_normal((m, 2 * ch)) * 0.3).astype(np.float32)
    ez = np.exp(rng.standard_normal(2 * ch) * 0.1).astype(np.float32)
    dZ, dez = torch.from_numpy(Z).cuda(), torch.from_numpy(ez).cuda()
    nb = 3
    part = torch.zeros(nb, device="cuda")
    yb32 = dxb.clone()
    _lib.check(lib.fwn_coupling_fwd(yb32.data_ptr(), dZ.data_ptr(), dez.data_ptr(), m, ch, part.data_ptr(), nb, st), "fwn_coupling_fwd")
    ls, t = Z[:, :ch].astype(np.float64) * ez[:ch], Z[:, ch:].astype(np.float64) * ez[ch:]
    ob = (dxb.cpu().numpy().astype(np.float64) - t) * np.exp(-ls)
    assert np.abs(yb32.cpu().numpy() - ob).max() < 1e-5 * max(1.0, np.abs(ob).max())
    assert abs(float(part.sum()) - float(-ls.sum())) < 1e-4 * max(1.0, np.abs(ls).sum())
    # backward: from out_b back to y_b, dZ with the log-det term
    g = rng.standard_normal((m, ch)).astype(np.float32)
    dg = torch.from_numpy(g).cuda()
    ldz = max(8, 2 * ch)
    dz = torch.zeros(m, ldz, device="cuda", dtype=torch.bfloat16)
    dzz = torch.zeros(m, 2 * ch, device="cuda")
    cls = 1.0 / (2.0 * m * ch)
    _lib.check(lib.fwn_coupling_bwd(dg.data_ptr(), yb32.data_ptr(), dZ.data_ptr(), dez.data_ptr(), m, ch, cls, dz.data_ptr(), ldz,
                                    dzz.data_ptr(), st), "fwn_coupling_bwd")
    torch.cuda.synchronize()
    assert np.abs(yb32.cpu().numpy() - dxb.cpu().numpy()).max() < 2e-5 * max(1.0, np.abs(xb).max())       # out_b -> y_b again
    e = np.exp(-ls)
    dls, dt = -g * ob + cls, -g * e
    want = np.concatenate([dls * ez[:ch], dt * ez[ch:]], 1)
    got = dz.float().cpu().numpy()[:, :2 * ch]
    assert np.abs(got - want).max() < 1e-2 * max(1.0, np.abs(want).max())                                 # bf16 output
    assert np.abs(dg.cpu().numpy() - g * e).max() < 1e-5 * max(1.0, np.abs(g * e).max())
    assert np.abs(dzz.cpu().numpy() - np.concatenate([dls * ls, dt * t], 1)).max() < 1e-4 * max(1.0, np.abs(dls * ls).max())


def test_train_entry_point_rejects_bad_arguments_with_a_message():
    """fwn_train_loss_and_grads / fwn_train_workspace_bytes: argument errors come back as codes with fwn_last_error()
    set (workspace too small: FWN_ERR_WORKSPACE), nothing is launched, and the same descriptors still work afterwards."""
    import ctypes as C
    from conftest import small_hparams
    from tf_flowavenet_amd import weights as W, _lib
    from tf_flowavenet_amd.training import GradEngine
    hp = small_hparams(n_block=2, n_flow=2, n_layer=2, hop_size=16, upsample_scales=[4, 4], num_mels=8)
    p = W.synthetic_params(hp, 3, actnorm="random")
    inp = W.synthetic_inputs(hp, 2, 128)
    x, c = torch.from_numpy(inp["x"]).reshape(2, 128).cuda(), torch.from_numpy(inp["c"]).cuda()
    eng = GradEngine(hp)
    first = eng.loss_and_grads(p, x, c)         # kept: the descriptors point at these gradient tensors
    l0 = float(first[0])
    lib, td = eng.lib, eng._desc
    need = int(lib.fwn_train_workspace_bytes(C.byref(td), 2, 128))
    assert need > 0 and int(lib.fwn_train_workspace_bytes(C.byref(td), 2, 120)) == 0          # T not a multiple of the hop size
    assert int(lib.fwn_train_workspace_bytes(C.byref(td), 0, 128)) == 0
    ws = torch.empty(need + 512, dtype=torch.uint8, device="cuda")
    base = ws.data_ptr() + (-ws.data_ptr()) % 256
    out3 = torch.zeros(3, device="cuda")
    cb = _lib.BLOCK_DONE_FN(lambda user, blk: 0)
    call = lambda B, T, xp, wsp, wsn: lib.fwn_train_loss_and_grads(C.byref(td), B, T, xp, c.data_ptr(), wsp, wsn, out3.data_ptr(), cb, None, None)
    assert call(2, 128, x.data_ptr(), base, need - 1) == -3 and b"workspace" in lib.fwn_last_error()
    assert call(2, 128, None, base, need) == -1 and b"null" in lib.fwn_last_error()
    assert call(2, 128, x.data_ptr(), base + 16, need) == -1 and b"aligned" in lib.fwn_last_error()
    assert call(2, 120, x.data_ptr(), base, need) == -1 and b"multiple" in lib.fwn_last_error()
    assert call(0, 128, x.data_ptr(), base, need) == -1
    stop = _lib.BLOCK_DONE_FN(lambda user, blk: 7)
    assert lib.fwn_train_loss_and_grads(C.byref(td), 2, 128, x.data_ptr(), c.data_ptr(), base, need, out3.data_ptr(), stop, None, None) == -4
    assert b"callback" in lib.fwn_last_error()
    torch.cuda.synchronize()
    out3.zero_()
    torch.cuda.synchronize()
    assert float(out3.abs().sum()) == 0.0                       # nothing ran
    assert call(2, 128, x.data_ptr(), base, need) == 0
    torch.cuda.synchronize()
    assert float(out3[0]) == l0


def test_gradient_is_reproducible_bit_for_bit():
    cfg = dict(n_block=2, n_flow=2, n_layer=2, hop_size=16, upsample_scales=[4, 4], num_mels=8)
    _, _, _, (l1, _, _, g1) = _grad_case(cfg, 2, 128, 7)
    _, _, _, (l2, _, _, g2) = _grad_case(cfg, 2, 128, 7)
    assert float(l1) == float(l2) and all(torch.equal(g1[k], g2[k]) for k in g1)


def test_trainer_steps_reduce_the_loss_on_a_fixed_batch():
    """DDI + a few clip/Adam steps (train.py:15-32,221-229) driven by the HIP gradients."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from conftest import small_hparams
    from tf_flowavenet_amd import weights as W
    from tf_flowavenet_amd.training import Trainer
    hp = small_hparams(n_block=3, n_flow=2, n_layer=2, hop_size=16, upsample_scales=[4, 4], num_mels=16)
    inp = W.synthetic_inputs(hp, 4, 256)
    x, c = torch.from_numpy(inp["x"]).reshape(4, 256).cuda(), torch.from_numpy(inp["c"]).cuda()
    tr = Trainer(hp, W.synthetic_params(hp, 11))
    tr.ddi(x, c)
    losses = [float(tr.step(x, c)[0]) for _ in range(25)]
    assert np.isfinite(losses).all()
    assert min(losses[-5:]) < losses[0] - 0.05, losses


def test_recorded_step_replays_the_eager_step_bit_for_bit():
    """Trainer(graph=True): eager first step, hipGraph recording on the second, replays after - weights, loss and
    gradient norm equal the eager trainer's on changing batches, and the Adam rate follows the step count."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from conftest import small_hparams
    from tf_flowavenet_amd import weights as W
    from tf_flowavenet_amd.training import Trainer
    hp = small_hparams(n_block=3, n_flow=2, n_layer=2, hop_size=16, upsample_scales=[4, 4], num_mels=16)
    inp = W.synthetic_inputs(hp, 4, 256)
    x0, c0 = torch.from_numpy(inp["x"]).reshape(4, 256).cuda(), torch.from_numpy(inp["c"]).cuda()
    batches = [(x0.roll(k, 0) * (1.0 - 0.05 * k), c0.roll(k, 0)) for k in range(5)]
    # a second input shape in between: its own eager step + recording, the first shape's recording stays valid
    half = (x0[:2, :128].contiguous(), c0[:2, :128 // hp.hop_size].contiguous())
    batches = batches[:3] + [half, half, half] + batches[3:]
    runs = {}
    for graph in (False, True):
        tr = Trainer(hp, W.synthetic_params(hp, 11), graph=graph)
        tr.ddi(*batches[0])
        outs = [tuple(float(v) for v in tr.step(x, c)) for x, c in batches]
        runs[graph] = (outs, tr.opt.w.clone(), tr.opt.global_step)
    assert runs[True][2] == runs[False][2] == 8
    assert runs[True][0] == runs[False][0]
    assert torch.equal(runs[True][1], runs[False][1])


def test_train_cli_loop_checkpoint_resume_and_synthesis(tmp_path):
    """preprocess -> train (DDI, steps, summaries, checkpoint) -> resume -> synthesize from the checkpoint."""
    import json, os, sys, wave
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from conftest import small_hparams
    from types import SimpleNamespace
    from tf_flowavenet_amd import preprocessing as P, train as TL, synthesize as S
    hp = small_hparams(n_block=3, n_flow=2, n_layer=2, hop_size=16, upsample_scales=[4, 4], num_mels=16,
                       n_fft=64, sample_rate=8000, fmin=50, fmax=3800, max_time_steps=256, batch_size=4, test_size=2,
                       eval_max_time_steps=512)
    book = tmp_path / "raw" / "book"
    os.makedirs(book / "wavs")
    rng = np.random.default_rng(0)
    lines = []
    for i in range(6):
        n = 900 + 37 * i
        pcm = (np.clip(0.3 * rng.standard_normal(n), -1, 1) * 32767).astype("<i2")
        with wave.open(str(book / "wavs" / ("u%d.wav" % i)), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(hp.sample_rate); w.writeframes(pcm.tobytes())
        lines.append("u%d|t|t" % i)
    (book / "metadata.csv").write_text("\n".join(lines), encoding="utf-8")
    data = tmp_path / "base" / "training_data"
    P.preprocess(str(tmp_path / "raw"), str(data), hp)
    args = SimpleNamespace(base_dir=str(tmp_path / "base"), restore=True, summary_interval=2, checkpoint_interval=3,
                           eval_interval=4, train_steps=6, seed=3)
    log_dir = str(tmp_path / "base" / "logs")
    save_dir = TL.train(log_dir, args, hp, "training_data/train.txt")
    ckpts = sorted(os.listdir(save_dir))
    assert "flowavenet_model.ckpt-3.npz" in ckpts and "flowavenet_model.ckpt-6.npz" in ckpts
    recs = [json.loads(l) for l in open(os.path.join(log_dir, "train", "summary.jsonl"))]
    assert [r["step"] for r in recs] == [2, 4, 6] and all(np.isfinite(r["losses/total_loss"]) for r in recs)
    assert os.path.exists(os.path.join(log_dir, "test", "summary.jsonl"))
    assert os.path.exists(os.path.join(log_dir, "train", "predictions-4.wav"))
    # resume: picks up at step 6 and runs to 8
    args.train_steps = 8
    TL.train(log_dir, args, hp, "training_data/train.txt")
    assert "flowavenet_model.ckpt-8.npz" in os.listdir(save_dir)
    # the checkpoint feeds synthesize.py's loader
    ck = S.load_checkpoint(save_dir)
    from tf_flowavenet_amd.model import FloWaveNet
    from tf_flowavenet_amd import weights as W
    m = FloWaveNet(hp).load_params({k: ck[k] for k in W.param_shapes(hp)})
    mel = np.load(data / "mels" / "dataset-mel-00001.npy")[:16]
    x = m.reverse(torch.randn(1, 16 * hp.hop_size, 1) * 0.7, torch.from_numpy(mel[None]))
    assert bool(torch.isfinite(x).all())


# ------------------------------------------------------------------ data-parallel step, 2 ranks on one GPU
def _dp_worker(rank, world, port, out_dir, backend="gloo"):
    import os, sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    from conftest import small_hparams
    from tf_flowavenet_amd import weights as W
    from tf_flowavenet_amd.training import Trainer
    if backend == "nccl":             # one GPU per rank, RCCL over xGMI: the measured configuration
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:                             # gloo moves CUDA tensors through the host; both ranks share cuda:0
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
    hp = small_hparams(n_block=3, n_flow=2, n_layer=2, hop_size=16, upsample_scales=[4, 4], num_mels=16)
    inp = W.synthetic_inputs(hp, 4, 256)
    x = torch.from_numpy(inp["x"]).reshape(4, 256)[2 * rank:2 * rank + 2].cuda()
    c = torch.from_numpy(inp["c"])[2 * rank:2 * rank + 2].cuda()
    tr = Trainer(hp, W.synthetic_params(hp, 11))
    tr.ddi(x, c)
    w0 = tr.opt.w.clone()
    tr.step(x, c)
    g1, w1 = tr.opt.g.cpu().numpy(), tr.opt.w.cpu().numpy()
    tr.step(x, c)               # with FWN_TRAIN_GRAPH=1: recorded (a chain of hipGraphs cut at the all-reduces) ...
    tr.step(x, c)               # ... and replayed
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), w0=w0.cpu().numpy(), g=g1, w1=w1, w3=tr.opt.w.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_two_rank_data_parallel_step_matches_one_process_on_the_whole_batch(tmp_path, backend):
    """Two processes (gloo, both on cuda:0), two clips each: after DDI (per-flow moment all-reduce) both hold the
    ActNorm init of the whole 4-clip batch - bit-identical on the two ranks and equal to the one-process init on the
    concatenated batch -, the all-reduced gradient equals world x the gradient of the 4-clip batch in one process,
    and both ranks end the step with bit-identical weights (model.py:30-83, utils.py:34-60, train.py:75-81)."""
    import os, socket, sys
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from conftest import small_hparams
    from tf_flowavenet_amd import weights as W
    from tf_flowavenet_amd.training import Trainer
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL with one GPU per rank needs two GPUs (the one-rank RCCL path is covered by tests/test_rccl.py)")
    ctx = mp.get_context("spawn")

    def run_pair(out):
        """Both ranks to completion -> exit codes (None = still running after the time limit, killed)."""
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()      # a fresh port per pair
        procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, str(out), backend), daemon=True) for r in range(2)]
        for p in procs:
            p.start()
        try:
            for p in procs:
                p.join(timeout=150)
            return [p.exitcode for p in procs]
        finally:
            for p in procs:         # a rank that died leaves its peer waiting in a collective: never leave it behind
                if p.is_alive():
                    p.kill()
                    p.join(timeout=10)

    w3 = {}
    for mode in ("1", "0"):     # recorded step, then the eager step: same bits after three steps
        os.environ["FWN_TRAIN_GRAPH"] = mode
        out = tmp_path / ("graph" + mode)
        out.mkdir()
        try:
            codes = run_pair(out)
            if None in codes:       # rendezvous of two fresh processes on one GPU stalled: one more try, then fail
                codes = run_pair(out)
            assert codes == [0, 0], codes
        finally:
            os.environ.pop("FWN_TRAIN_GRAPH")
        r0, r1 = (np.load(out / ("rank%d.npz" % r)) for r in range(2))
        assert np.array_equal(r0["w0"], r1["w0"]) and np.array_equal(r0["g"], r1["g"]) and np.array_equal(r0["w1"], r1["w1"])
        assert np.array_equal(r0["w3"], r1["w3"])
        w3[mode] = r0["w3"]
    assert np.array_equal(w3["1"], w3["0"])
    # one process, the whole batch, same initial weights
    hp = small_hparams(n_block=3, n_flow=2, n_layer=2, hop_size=16, upsample_scales=[4, 4], num_mels=16)
    inp = W.synthetic_inputs(hp, 4, 256)
    tr = Trainer(hp, W.synthetic_params(hp, 11))
    tr.ddi(torch.from_numpy(inp["x"]).reshape(4, 256), torch.from_numpy(inp["c"]))
    w0_single = tr.opt.w.cpu().numpy()
    assert not np.array_equal(w0_single, W.synthetic_params(hp, 11) and tr.opt.layout.flatten(W.synthetic_params(hp, 11)))   # DDI wrote something
    np.testing.assert_allclose(r0["w0"], w0_single, rtol=0, atol=2e-4)     # ActNorm b / logs of the global batch
    tr.opt.w.copy_(torch.from_numpy(r0["w0"]))
    _, _, _, grads = tr.engine.loss_and_grads(tr.opt.master_views(), torch.from_numpy(inp["x"]).reshape(4, 256), torch.from_numpy(inp["c"]))
    gv = tr.opt.grad_views()
    for k, g in grads.items():
        gv[k].copy_(g.reshape(gv[k].shape))
    want, got = 2.0 * tr.opt.g.cpu().numpy().astype(np.float64), r0["g"].astype(np.float64)
    cos = float((want * got).sum() / np.sqrt((want * want).sum() * (got * got).sum()))
    assert cos > 0.999 and abs(np.linalg.norm(got) / np.linalg.norm(want) - 1.0) < 2e-2, (cos, np.linalg.norm(got), np.linalg.norm(want))


@pytest.mark.parametrize("t", [256, 4096])
def test_recorded_packing_matches_immediate_packing(t):
    """Device-resident parameters take the PackPlan path (grouped, transposed packing, refreshed in
    place every step); NumPy parameters the immediate one: same gradients, also after the
    parameters changed in place.  t = 4096: 4 096 / 2 048 / 1 024 rows per block - the forward half runs the register-streamed
    tail (csrc/tail_rs.h, SAVE form), whose fragment-order weights the plan re-packs behind the grouped packing of every
    refresh (fwn_pack_tail_stream_jobs: all flows in one launch) - a stale or mis-ordered stream would show after the change."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from conftest import small_hparams
    from tf_flowavenet_amd import weights as W
    from tf_flowavenet_amd.training import GradEngine
    hp = small_hparams(n_block=3, n_flow=2, n_layer=2, hop_size=16, upsample_scales=[4, 4], num_mels=16)
    p = W.synthetic_params(hp, 9, actnorm="random")
    inp = W.synthetic_inputs(hp, 2, t)
    x, c = torch.from_numpy(inp["x"]).reshape(2, t), torch.from_numpy(inp["c"])
    pd = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).cuda() for k, v in p.items()}
    eng = GradEngine(hp)
    for step in range(2):
        l_d, _, _, g_d = eng.loss_and_grads(pd, x, c)
        assert eng._tp.plan is not None and len(eng._tp.plan.tail_jobs) == hp.n_block * hp.n_flow
        assert all(d.Wts for d in eng._tp.pm.flow_descs)
        l_n, _, _, g_n = GradEngine(hp).loss_and_grads({k: v.cpu().numpy() for k, v in pd.items()}, x, c)
        assert float(l_d) == float(l_n)
        for k in g_n:
            assert torch.equal(g_d[k].reshape(-1), g_n[k].reshape(-1)), k
        for k, v in pd.items():          # "an optimiser step": every parameter moves, in place
            v.mul_(1.01)


@pytest.mark.parametrize("m,ti,kx,n,shifts", [(333, 111, 256, 512, (0,)), (6000, 1000, 256, 512, (-1, 0, 1)),
                                              (25600, 3200, 256, 256, (-3, 0, 3)), (200, 25, 80, 512, (0,)), (4000, 500, 640, 512, (0,))])
def test_tn_gemm_reads_operands_transposed_from_lds(m, ti, kx, n, shifts):
    """fwn_tn_gemm (ds_read_b64_tr_b16 operands, no transposed copies) against fp64, ragged M / Kx, taps with
    clip edges, and the bias column sums."""
    rng = np.random.default_rng(m + kx)
    x, dy = bf(rng.standard_normal((m, kx)) * 0.5), bf(rng.standard_normal((m, n)) * 0.1)
    part = TR.tn_weight_grad_partials(x, dy, m, kx, n, shifts=shifts, ti=ti)
    part2 = TR.tn_weight_grad_partials(x, dy, m, kx, n, shifts=shifts, ti=ti)
    assert torch.equal(part, part2)
    full = TR.reduce_splits(part).cpu().numpy() if part.shape[0] > 1 else part[0].cpu().numpy()
    got, got_b = full[:-1], full[-1]
    xw, dyw = f64(x), f64(dy)
    want = np.concatenate([shifted(xw, sh, ti).T @ dyw for sh in shifts])
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 2e-3 * np.abs(want).max()
    np.testing.assert_allclose(got_b, dyw.sum(0), atol=2e-3 * np.abs(dyw.sum(0)).max())     # the bias row
    np.testing.assert_allclose(TR.colsum_bf16(dy, m, n, scale=0.5).cpu().numpy(), 0.5 * dyw.sum(0), atol=2e-3 * np.abs(dyw.sum(0)).max())


def test_grouped_weight_gradients_and_weight_norm_backward():
    """One grouped TN GEMM over jobs of different shapes, then the grouped split reduction + weight-norm backward
    (convolutional.py:73-80: W = V g / ||V||_col) against float64."""
    torch.manual_seed(3)
    m, ti = 5000, 1000
    f64 = lambda t: t.float().cpu().numpy().astype(np.float64)
    xs = [torch.randn(m, k, device="cuda").to(torch.bfloat16) for k in (256, 256, 80)]
    dys = [(torch.randn(m, n, device="cuda") * 0.1).to(torch.bfloat16) for n in (256, 512, 512)]
    shifts = [(0,), (-3, 0, 3), (0,)]
    jobs = [(xs[i], dys[i], xs[i].shape[1], dys[i].shape[1], shifts[i]) for i in range(3)]
    parts = TR.tn_weight_grad_group(jobs, m, ti)
    assert parts[0].shape[0] == TR.tn_group_splits([(256, 256, 1), (256, 512, 3), (80, 512, 1)], m)

    def shifted(a, sh):
        out = np.zeros_like(a)
        for r in range(a.shape[0]):
            if 0 <= r % ti + sh < ti:
                out[r] = a[r + sh]
        return out

    want = [np.concatenate([shifted(f64(xs[i]), sh).T @ f64(dys[i]) for sh in shifts[i]] + [f64(dys[i]).sum(0, keepdims=True)])
            for i in range(3)]
    for i in range(3):
        got = parts[i].sum(0).cpu().numpy()
        assert np.abs(got - want[i]).max() < 2e-3 * np.abs(want[i]).max()
    # weight-norm backward of the two column halves of job 1 (K = 768) and of job 2 through a row map
    perm = torch.randperm(80, device="cuda").to(torch.int32)
    specs = [(parts[1], 768, 0, None, 0.7), (parts[1], 768, 256, None, 1.0), (parts[2], 80, 256, perm, 1.0), (parts[0], 256, 0, None, 1.0)]
    wn_jobs, keep = [], []
    for part, k, col0, row_src, scale in specs:
        v, g = torch.randn(k, 256, device="cuda"), torch.rand(256, device="cuda") + 0.5
        out = dict(dv=torch.empty(k, 256, device="cuda"), dg=torch.empty(256, device="cuda"), db=torch.empty(256, device="cuda"))
        wn_jobs.append(dict(part=part, k=k, n=256, col0=col0, bias_row=int(part.shape[1]) - 1, scale=scale, row_src=row_src, v=v, g=g, **out))
        keep.append((v, g, out))
    wn_jobs[3]["g"] = wn_jobs[3]["v"] = None                  # no weight norm: dV = dW
    TR.wn_backward_group(wn_jobs)
    for (part, k, col0, row_src, scale), (v, g, out), wj in zip(specs, keep, wn_jobs):
        full = f64(part.sum(0)) * scale
        rows = np.arange(k) if row_src is None else row_src.cpu().numpy()
        dw, db = full[rows][:, col0:col0 + 256], full[-1, col0:col0 + 256]
        np.testing.assert_allclose(out["db"].cpu().numpy(), db, rtol=1e-4, atol=1e-4 * np.abs(db).max())
        if wj["g"] is None:
            np.testing.assert_allclose(out["dv"].cpu().numpy(), dw, rtol=1e-4, atol=1e-5 * np.abs(dw).max())
            continue
        vv, gg = f64(v), f64(g)
        nrm = np.sqrt((vv * vv).sum(0))
        dg = (dw * vv).sum(0) / nrm
        dv = gg / nrm * (dw - vv * dg / nrm)
        np.testing.assert_allclose(out["dg"].cpu().numpy(), dg, rtol=1e-4, atol=1e-4 * np.abs(dg).max())
        np.testing.assert_allclose(out["dv"].cpu().numpy(), dv, rtol=1e-4, atol=1e-4 * np.abs(dv).max())


def test_device_side_table_refresh_equals_the_host_tables():
    """PackPlan.refresh_tables_device (batched float64 gathers from the flat masters) reproduces the host-computed
    biases / ActNorm / ZeroConv / up-sampling tables after the parameters changed."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from conftest import small_hparams
    from tf_flowavenet_amd import weights as W
    from tf_flowavenet_amd.optim import DataParallelAdam
    hp = small_hparams(n_block=3, n_flow=2, n_layer=2, hop_size=16, upsample_scales=[4, 4], num_mels=16)
    opt = DataParallelAdam(hp, W.synthetic_params(hp, 11))
    tp = TR._TrainPack(opt.master_views(), hp, "cuda")
    assert tp.plan._dev_ready
    opt.w.add_(torch.randn_like(opt.w) * 0.05)            # "an optimiser step"
    tp.plan.refresh_tables_device()
    dev_tables = tp.plan._tbuf.clone()
    tp.plan.hostview.reset()
    tp.plan.upload_tables()
    host_tables = tp.plan._tbuf.clone()
    diff = (dev_tables != host_tables).nonzero().reshape(-1)
    # float64 exp on the device vs numpy may differ in the last float64 bit: at most a handful of fp32 roundings flip
    assert diff.numel() <= 4, diff[:10]
    assert torch.allclose(dev_tables, host_tables, rtol=2e-7, atol=0)


def test_full_size_model_recorded_step_equals_eager_step():
    """The reference configuration (n_block=8, n_flow=6: 181 M parameters, conditioning widths up to 10240) on a short
    batch: three steps recorded / replayed give the bits of three eager steps, losses finite (that the loss
    falls is shown on the small model above and by tools/train_probe.py at full size: the first Adam steps of 181 M
    parameters on two short clips are not monotonic)."""
    from tf_flowavenet_amd.hparams import default_hparams
    from tf_flowavenet_amd import weights as W
    from tf_flowavenet_amd.training import Trainer
    hp = default_hparams()
    inp = W.synthetic_inputs(hp, 2, 1024)
    x, c = torch.from_numpy(inp["x"]).reshape(2, 1024).cuda(), torch.from_numpy(inp["c"]).cuda()
    params = W.synthetic_params(hp, 1234)
    res = {}
    for graph in (True, False):
        tr = Trainer(hp, params, graph=graph)
        tr.ddi(x, c)
        outs = [tuple(float(v) for v in tr.step(x, c)) for _ in range(3)]
        res[graph] = (outs, tr.opt.w.clone())
        del tr
        torch.cuda.empty_cache()
    assert all(np.isfinite(v) for o in res[True][0] for v in o)
    assert res[True][0] == res[False][0]
    assert torch.equal(res[True][1], res[False][1])


def test_recording_failure_falls_back_to_eager_steps(monkeypatch):
    """A runtime that refuses stream capture must not stop training: warn once, continue eagerly, same bits."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from conftest import small_hparams
    from tf_flowavenet_amd import weights as W
    from tf_flowavenet_amd.training import Trainer
    hp = small_hparams(n_block=2, n_flow=2, n_layer=2, hop_size=16, upsample_scales=[4, 4], num_mels=8)
    inp = W.synthetic_inputs(hp, 2, 128)
    x, c = torch.from_numpy(inp["x"]).reshape(2, 128).cuda(), torch.from_numpy(inp["c"]).cuda()
    ref = Trainer(hp, W.synthetic_params(hp, 11), graph=False)
    ref.ddi(x, c)
    want = [tuple(float(v) for v in ref.step(x, c)) for _ in range(3)]

    def refuse(self, *a, **k):
        raise RuntimeError("capture refused (test)")
    monkeypatch.setattr(torch.cuda.CUDAGraph, "capture_begin", refuse)
    tr = Trainer(hp, W.synthetic_params(hp, 11), graph=True)
    tr.ddi(x, c)
    with pytest.warns(UserWarning, match="continuing with eager steps"):
        got = [tuple(float(v) for v in tr.step(x, c)) for _ in range(3)]
    assert got == want and tr.graph is False and torch.equal(tr.opt.w, ref.opt.w)
