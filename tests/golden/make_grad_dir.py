#!/usr/bin/env python3
"""Directional derivatives of the training loss at the BENCH's training shape, from the fp64 oracle FORWARD pass.

    python tests/golden/make_grad_dir.py        ->  tests/golden/grad_dir_b8f6_B8_T6400.npz   (about ten minutes of CPU)

Why: fp64 autograd of the full n_block = 8 model fits in memory only at toy shapes (B = 2, T = 1024: block 0 has 1024
rows), so the per-tensor gradient comparison of tests/test_train.py never reaches the tile variants the training step
runs at the reference's shape (hparams.py:28,36: crops of 6400 samples x batch 8 -> 25 600 rows at block 0).  A central
difference of the ORACLE'S FORWARD pass needs no autograd:

    fd_f = (L(w + eps d_f) - L(w - eps d_f)) / (2 eps),     L = -(log_p + logdet)   (train.py:56-66)

for one direction d_f per parameter family f, compared by the GPU test with <g_HIP, d_f>.  A random direction would carry no
signal (<g, d> ~ |g| / sqrt(n) against an error of the same size), so d_f is the normalised fp64 autograd gradient of the
SAME parameters on the toy batch (B = 2, T = 1024: `oracle/grad_torch.py`), restricted to the family - it correlates with
the gradient at the training shape and both the generator and the test can compute it on the CPU.  Families: dilated
kernels (Conv_filter / Conv_gate), conditioning kernels (filter_conv_c / gate_conv_c), per-channel scales (ActNorm b /
logs, ZeroConv scale), up-sampling kernels, the 1 x 1 / front / ZeroConv kernels, every weight-norm g and bias, and all
trainable tensors at once.  Each derivative is taken at eps and 2 eps, which must agree to 2e-3, and the
fixture holds their Richardson extrapolation (4 fd(eps) - fd(2 eps)) / 3 (error O(eps^4)).
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import flowavenet_np as onp          # noqa: E402
from oracle import grad_torch as G               # noqa: E402
from tf_flowavenet_amd import weights as W       # noqa: E402
from tf_flowavenet_amd.hparams import default_hparams   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
NAME = "grad_dir_b8f6_B8_T6400"
SEED, B, T = 1234, 8, 6400                        # hparams.py:28,36 (max_time_steps, batch_size)
TOY_B, TOY_T = 2, 1024

# family -> (name suffixes ("" = every trainable tensor), eps).  Round 4: "pointwise" - the 1 x 1 /
# front / ZeroConv kernels, i.e. most jobs of the grouped TN weight-gradient GEMM (modules.py:74-95,144,158-159) -,
# "gbias" - every weight-norm g and every bias (convolutional.py:73-80: the weight-norm backward) - and "all".
FAMILIES = {
    "dilated": (("/Conv_filter/kernel", "/Conv_gate/kernel"), 5e-3),
    "cond": (("/filter_conv_c/kernel", "/gate_conv_c/kernel"), 5e-3),
    "scale": (("/ActNorm/b", "/ActNorm/logs", "/ZeroConv1d/scale"), 5e-4),
    "upsample": (("upsample_0/kernel", "upsample_1/kernel"), 5e-4),
    "pointwise": (("/res_conv/kernel", "/skip_conv/kernel", "/Conv_final/kernel", "/ZeroConv1d/kernel", "/Conv_front/kernel"), 2e-3),
    "gbias": (("/g", "/bias"), 5e-4),
    "all": (("",), 2e-4),
}


def in_family(fam, name):
    return any(name.endswith(p) for p in FAMILIES[fam][0])


def directions(params, hp):
    """family -> {tensor name: direction}, unit norm over the family: the toy batch's fp64 autograd gradient."""
    inp = W.synthetic_inputs(hp, TOY_B, TOY_T)
    _, _, _, g = G.loss_and_grads(params, inp["x"], inp["c"], hp)
    out = {}
    for fam in FAMILIES:
        d = {k: np.asarray(v, dtype=np.float64) for k, v in g.items() if in_family(fam, k)}
        nrm = np.sqrt(sum(float((v * v).sum()) for v in d.values()))
        assert d and nrm > 0, fam
        out[fam] = {k: v / nrm for k, v in d.items()}
    return out


def loss(p64, x, c, hp):
    log_p, logdet, _ = onp.forward(p64, x, c, hp)
    return -(log_p + logdet)


def main():
    hp = default_hparams()
    params = W.synthetic_params(hp, SEED, actnorm="random")
    t0 = time.time()
    dirs = directions(params, hp)
    print("directions: %.0f s" % (time.time() - t0), flush=True)
    inp = W.synthetic_inputs(hp, B, T)
    x, c = inp["x"].astype(np.float64), inp["c"].astype(np.float64)
    base = onp.to_f64(params)
    path = os.path.join(HERE, NAME + ".npz")
    old = dict(np.load(path)) if os.path.exists(path) else {}
    if old and (int(old["b"]), int(old["t"]), int(old["seed"]), int(old["toy_b"]), int(old["toy_t"])) == (B, T, SEED, TOY_B, TOY_T):
        out = {k: (v.item() if v.shape == () else v) for k, v in old.items()}      # families already taken at this eps are kept
    else:
        out = dict(b=B, t=T, seed=SEED, toy_b=TOY_B, toy_t=TOY_T, loss=loss(base, x, c, hp))
    print("loss %.9f (%.0f s)" % (out["loss"], time.time() - t0), flush=True)
    for fam, (_, eps) in FAMILIES.items():
        if "fd_" + fam in out and float(out["eps_" + fam]) == eps:
            print("%-9s kept" % fam, flush=True)
            continue
        fds = []
        for e in (eps, 2 * eps):
            vals = []
            for sign in (1.0, -1.0):
                p = dict(base)
                for k, d in dirs[fam].items():
                    p[k] = base[k] + sign * e * d.reshape(base[k].shape)
                vals.append(loss(p, x, c, hp))
            fds.append((vals[0] - vals[1]) / (2 * e))
        out["fd_" + fam], out["fd1_" + fam], out["fd2_" + fam], out["eps_" + fam] = (4.0 * fds[0] - fds[1]) / 3.0, fds[0], fds[1], eps
        print("%-9s fd %.9e  (eps: %.9e, 2 eps: %.9e, rel diff %.2e)  %.0f s" % (fam, out["fd_" + fam], fds[0], fds[1],
                                                                                abs(fds[0] - fds[1]) / abs(fds[0]), time.time() - t0), flush=True)
        assert abs(fds[0] - fds[1]) <= 2e-3 * abs(fds[0]), "curvature: shrink eps"
        np.savez(path, **out)


if __name__ == "__main__":
    main()
