"""Training-side primitives (work in progress towards the backward pass): the generic multi-segment
GEMM, the shifted transpose and the deterministic split-K reduction against NumPy fp64."""
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
