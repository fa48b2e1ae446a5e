"""CPU tests that pin the oracle: the reference has no tests or golden
vectors and cannot be run here ("parity unpinned"), so the fp64 restatement is pinned by an
independent formulation and by invariants every correct implementation satisfies."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import flowavenet_np as onp
from oracle import flowavenet_torch as ot
from tf_flowavenet_amd import weights as W

from conftest import small_hparams

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _setup(hp, b, t, **kw):
    params = onp.to_f64(W.synthetic_params(hp, 1234, **kw))
    inp = W.synthetic_inputs(hp, b, t)
    return params, {k: v.astype(np.float64) for k, v in inp.items()}


def _unsqueeze_all(z, n):
    for _ in range(n):
        z = onp.unsqueeze(z)
    return z


@pytest.mark.parametrize("cfg", [
    dict(n_block=1, n_flow=2, n_layer=1), dict(n_block=2, n_flow=2), dict(n_block=2, n_flow=3),
    dict(n_block=3, n_flow=3, num_mels=16), dict(n_block=4, n_flow=2, n_layer=3, upsample_scales=[4, 8], hop_size=32),
])
def test_two_formulations_agree(cfg):
    hp = small_hparams(**cfg)
    t = 64 * (2 ** max(0, hp.n_block - 2)) * (hp.hop_size // 16)
    p, inp = _setup(hp, 2, t, actnorm="random")
    lp, ld, z = onp.forward(p, inp["x"], inp["c"], hp)
    fp = ot.fold(p, hp)
    lp2, ld2, z2 = ot.forward(fp, torch.tensor(inp["x"]), torch.tensor(inp["c"]), hp)
    assert abs(lp - lp2) < 1e-12 and abs(ld - ld2) < 1e-12
    np.testing.assert_allclose(z2.numpy(), z, atol=1e-12)
    if (hp.n_block * hp.n_flow) % 2 == 0:
        x1 = onp.reverse(p, inp["z"], inp["c"], hp)
        x2 = ot.reverse(fp, torch.tensor(inp["z"]), torch.tensor(inp["c"]), hp)
        np.testing.assert_allclose(x2.numpy(), x1, atol=1e-11)


@pytest.mark.parametrize("cfg,exact", [(dict(n_block=2, n_flow=2), True), (dict(n_block=2, n_flow=3), True),
                                       (dict(n_block=1, n_flow=3), False)])
def test_round_trip_iff_even(cfg, exact):
    """reverse(forward(x)) == x exactly iff n_block*n_flow is even."""
    hp = small_hparams(**cfg)
    p, inp = _setup(hp, 2, 64, actnorm="random")
    _, _, z = onp.forward(p, inp["x"], inp["c"], hp)
    xr = onp.reverse(p, _unsqueeze_all(z, hp.n_block), inp["c"], hp)
    err = np.abs(xr - inp["x"]).max()
    assert (err < 1e-12) if exact else (err > 1e-3)


def test_logdet_is_log_abs_det_jacobian():
    """Reference-style logdet == slogdet(dz/dx) / T (per-sample nats), T=16, n_block=2, n_flow=2."""
    hp = small_hparams(n_block=2, n_flow=2, hop_size=4, upsample_scales=[2, 2])
    p, inp = _setup(hp, 1, 16, actnorm="random")
    fp = ot.fold(p, hp)
    c = torch.tensor(inp["c"])

    def f(xflat):
        return ot.forward_z(fp, xflat.reshape(1, 16, 1), c, hp).reshape(-1)

    x0 = torch.tensor(inp["x"]).reshape(-1)
    jac = torch.autograd.functional.jacobian(f, x0)
    sign, logabs = torch.linalg.slogdet(jac)
    _, ld, _ = onp.forward(p, inp["x"], inp["c"], hp)
    assert abs(float(logabs) / 16 - ld) < 1e-10


def test_zero_init_known_answer():
    """With ZeroConv1d at its literal zero init the coupling is the identity: logdet is the sum of
    the ActNorm terms and z is x pushed through ActNorms + permutations only."""
    hp = small_hparams(n_block=2, n_flow=2)
    p, inp = _setup(hp, 2, 64, zero_conv="zeros", actnorm="random")
    _, ld, z = onp.forward(p, inp["x"], inp["c"], hp)
    expect = sum(np.mean(3.0 * p["Block_%d/Flow_%d/ActNorm/logs" % (i, j)]) for i in range(2) for j in range(2))
    assert abs(ld - expect) < 1e-13
    # channel-wise affine map only: apply ActNorms by hand
    cur = inp["x"]
    for i in range(2):
        cur = onp.squeeze(cur)
        for j in range(2):
            cur, _ = onp.actnorm_forward(p, "Block_%d/Flow_%d/ActNorm" % (i, j), cur)
            a, b = onp.split2(cur)
            cur = np.concatenate([b, a], 2)
    np.testing.assert_allclose(z, cur, atol=1e-13)


def test_ddi_known_answer():
    """After init=True each ActNorm output has per-channel mean 0 and mean-square 1 (model.py:30-83)."""
    hp = small_hparams(n_block=2, n_flow=2)
    p, inp = _setup(hp, 4, 128, actnorm="zeros")
    x = onp.squeeze(inp["x"])
    y, _ = onp.actnorm_forward(p, "Block_0/Flow_0/ActNorm", x, init=True)
    np.testing.assert_allclose(y.mean(axis=(0, 1)), 0.0, atol=1e-12)
    np.testing.assert_allclose((y ** 2).mean(axis=(0, 1)), 1.0, atol=1e-5)
    # a second forward without init reproduces the init forward (parameters were stored)
    lp1, ld1, _ = onp.forward(p, inp["x"], inp["c"], hp, init=True)
    lp2, ld2, _ = onp.forward(p, inp["x"], inp["c"], hp, init=False)
    assert lp1 == lp2 and ld1 == ld2


@pytest.mark.parametrize("n", [1, 2, 3, 4])
def test_squeeze_closed_form(n):
    """s_n[t, m*2^n + r] = x[t*2^n + bitrev_n(r), m]."""
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 64, 3))
    s = x
    for _ in range(n):
        s = onp.squeeze(s)
    for m in range(3):
        for r in range(1 << n):
            np.testing.assert_array_equal(s[:, :, m * (1 << n) + r], x[:, ot.bitrev(r, n)::(1 << n), m])
    u = s
    for _ in range(n):
        u = onp.unsqueeze(u)
    np.testing.assert_array_equal(u, x)


def test_upsample_scatter_equals_conv_transpose():
    hp = small_hparams(upsample_scales=[4, 6], hop_size=24)
    p, inp = _setup(hp, 2, 96)
    up = onp.upsample(p, inp["c"], hp)
    up2 = ot.upsample(ot.fold(p, hp), torch.tensor(inp["c"]), hp).transpose(1, 2).numpy()
    assert up.shape == (2, 96, hp.num_mels)
    np.testing.assert_allclose(up2, up, atol=1e-13)


def test_param_count_matches_survey():
    from tf_flowavenet_amd.hparams import default_hparams, hparams8000
    assert W.count_params(default_hparams()) == 181129876
    assert len(W.param_shapes(default_hparams())) == 2262
    assert W.count_params(hparams8000()) > 0


@pytest.mark.parametrize("name", ["tiny_b2f2", "tiny_b3f3", "tiny_b4f2l3", "tiny_ddi"])
def test_oracle_reproduces_golden(name):
    """The committed fixtures are regenerable from the oracle (regression pin for both)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    over, b, t, actnorm, ddi = mg.CASES[name]
    hp = mg.hp_of(over)
    p, inp = _setup(hp, b, t, actnorm=actnorm)
    lp, ld, z = onp.forward(p, inp["x"], inp["c"], hp, init=ddi)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert abs(lp - float(g["log_p"])) < 1e-12 and abs(ld - float(g["logdet"])) < 1e-12
    np.testing.assert_allclose(z, g["z"], atol=1e-6)
    if "x_rev" in g:
        np.testing.assert_allclose(onp.reverse(p, inp["z"], inp["c"], hp), g["x_rev"], atol=1e-5)


def test_config0_fixture_is_consistent_with_torch_formulation():
    """BASELINE configs[0] (n_block=2, n_flow=2, one 16128-sample clip): the golden log_p/logdet
    from the NumPy oracle are reproduced by the independent torch formulation in fp64."""
    from tf_flowavenet_amd.hparams import default_hparams
    hp = default_hparams().replace(n_block=2, n_flow=2)
    g = np.load(os.path.join(GOLDEN, "config0_b2f2_T16128.npz"))
    params = onp.to_f64(W.synthetic_params(hp, 1234, actnorm="zeros"))
    inp = W.synthetic_inputs(hp, 1, 16128)
    x, c = inp["x"].astype(np.float64), inp["c"].astype(np.float64)
    onp.forward(params, x[:, :4096], c[:, :16], hp, init=False)  # shape smoke on a prefix
    # DDI in the torch formulation = DDI parameters taken from the numpy pass
    lp, ld, _ = onp.forward(params, x, c, hp, init=True)
    fp = ot.fold(params, hp)
    lp2, ld2, _ = ot.forward(fp, torch.tensor(x), torch.tensor(c), hp)
    assert abs(lp - float(g["log_p"])) < 1e-12 and abs(ld - float(g["logdet"])) < 1e-12
    assert abs(lp2 - lp) < 1e-11 and abs(ld2 - ld) < 1e-11
