"""Parameter naming, shapes and synthetic initialisation for the flow path.

No checkpoint ships with the reference, so benchmarks and parity tests use a
seeded synthetic weight set that follows the reference's initialisers:
He-uniform ``V`` with ``g = 1`` and He-uniform biases (modules.py:17-22,77-108,
convolutional.py:77), zero bias for the upsampling convs (model.py:303-309).
``ZeroConv1d`` is zero-initialised in the reference (modules.py:41-49); the
synthetic set draws its kernel from N(0, 0.02^2) instead so the coupling is
exercised; ``zero_conv="zeros"`` gives the literal init.

Names follow the reference's variable scopes (model.py:284,297,218,181-183); the
same dict feeds the fp64 oracle and ``FloWaveNet.load_params``.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

FILTER = 256  # model.py:217


def flow_prefix(i: int, j: int) -> str:
    return "Block_%d/Flow_%d" % (i, j)


def from_reference_names(variables: dict) -> dict:
    """Variables dumped from one of the reference's TF checkpoints (``vocoder/FloWaveNet/<scope>/<var>[:0]``,
    train.py:53 + model.py:283) -> this package's parameter names.  Optimiser slots (``.../Adam``,
    ``.../Adam_1``, ``beta*_power``), ``global_step`` and the cached low-precision casts of
    ``fp16_dtype_getter`` (utils.py:19-29) are dropped; names without the prefix pass through."""
    out = {}
    for name, value in variables.items():
        n = name[:-2] if name.endswith(":0") else name
        leaf = n.rsplit("/", 1)[-1]
        if leaf in ("Adam", "Adam_1", "global_step") or leaf.startswith("beta") or "fp16_cast" in n or n.startswith("__opt/"):
            continue
        for prefix in ("vocoder/FloWaveNet/", "FloWaveNet/"):
            if n.startswith(prefix):
                n = n[len(prefix):]
                break
        out[n] = value
    return out


def param_shapes(hp) -> "OrderedDict[str, tuple]":
    """name -> shape in the reference's TF layouts (kernel = (k, C_in, C_out))."""
    s = OrderedDict()
    for n, sc in enumerate(hp.upsample_scales):
        s["upsample_%d/kernel" % n] = (2 * sc, 3, 1, 1)
        s["upsample_%d/g" % n] = (1,)
        s["upsample_%d/bias" % n] = (1,)
    for i in range(hp.n_block):
        c = 2 ** (i + 1)                 # squeezed audio channels (model.py:214,298)
        cin = hp.num_mels * 2 ** i       # conditioning channels seen by the coupling net
        for j in range(hp.n_flow):
            fp = flow_prefix(i, j)
            s[fp + "/ActNorm/b"] = (1, 1, c)
            s[fp + "/ActNorm/logs"] = (1, 1, c)
            wp = fp + "/WaveNet"

            def conv(name, k, ci, co, wn=True):
                s["%s/%s/kernel" % (wp, name)] = (k, ci, co)
                if wn:
                    s["%s/%s/g" % (wp, name)] = (co,)
                s["%s/%s/bias" % (wp, name)] = (co,)

            conv("Conv_front", 3, c // 2, FILTER)
            for n in range(hp.n_layer):
                rp = "ResBlock_%d" % n
                conv(rp + "/Conv_filter", 3, FILTER, FILTER)
                conv(rp + "/Conv_gate", 3, FILTER, FILTER)
                conv(rp + "/res_conv", 1, FILTER, FILTER)
                conv(rp + "/skip_conv", 1, FILTER, FILTER)
                conv(rp + "/filter_conv_c", 1, cin, FILTER)
                conv(rp + "/gate_conv_c", 1, cin, FILTER)
            conv("Conv_final", 1, FILTER, FILTER)
            conv("ZeroConv1d", 1, FILTER, c if hp.affine else c // 2, wn=False)
            s[wp + "/ZeroConv1d/scale"] = (1, 1, c if hp.affine else c // 2)
    return s


def count_params(hp) -> int:
    return int(sum(int(np.prod(v)) for v in param_shapes(hp).values()))


def synthetic_params(hp, seed: int = 1234, zero_conv: str = "normal",
                     actnorm: str = "zeros") -> "OrderedDict[str, np.ndarray]":
    """Seeded fp32 parameter set (numpy PCG64), generated in ``param_shapes`` order.

    actnorm: "zeros" (b = logs = 0; run DDI afterwards) or "random" (small random
    values so non-DDI parity tests exercise the ActNorm arithmetic).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    out = OrderedDict()
    for name, shape in param_shapes(hp).items():
        leaf = name.rsplit("/", 1)[1]
        if name.startswith("upsample_"):
            if leaf == "kernel":
                lim = math.sqrt(6.0 / (shape[0] * shape[1] * shape[3]))
                a = rng.uniform(-lim, lim, size=shape)
            elif leaf == "g":
                a = np.ones(shape)
            else:
                a = np.zeros(shape)
        elif "/ActNorm/" in name:
            if actnorm == "zeros":
                a = np.zeros(shape)
            elif actnorm == "random":
                a = rng.uniform(-0.2, 0.2, size=shape) if leaf == "b" else rng.uniform(-0.1, 0.1, size=shape)
            else:
                raise ValueError(actnorm)
        elif "/ZeroConv1d/" in name:
            if leaf == "kernel" and zero_conv == "normal":
                a = rng.normal(0.0, 0.02, size=shape)
            elif zero_conv in ("normal", "zeros"):
                a = np.zeros(shape)
            else:
                raise ValueError(zero_conv)
        elif leaf == "kernel":
            lim = math.sqrt(6.0 / (shape[0] * shape[1]))
            a = rng.uniform(-lim, lim, size=shape)
        elif leaf == "g":
            a = np.ones(shape)
        elif leaf == "bias":
            lim = math.sqrt(6.0 / shape[0])
            a = rng.uniform(-lim, lim, size=shape)
        else:
            raise KeyError(name)
        out[name] = np.ascontiguousarray(a, dtype=np.float32)
    return out


def synthetic_inputs(hp, batch: int, t: int, want=("x", "c", "z")):
    """Seeded synthetic clip(s) per SURVEY section 8(d): mel~U[0,1) seed 75, audio
    clip(0.3 N(0,1), +-0.999) seed 76, latent temp*N(0,1) seed 77.  fp32."""
    if t % hp.hop_size or t % (2 ** hp.n_block):
        raise ValueError("T=%d must divide by hop_size=%d and 2^n_block=%d"
                         % (t, hp.hop_size, 2 ** hp.n_block))
    out = {}
    if "c" in want:
        out["c"] = np.random.Generator(np.random.PCG64(75)).random(
            (batch, t // hp.hop_size, hp.num_mels), dtype=np.float32)
    if "x" in want:
        g = np.random.Generator(np.random.PCG64(76))
        out["x"] = np.clip(0.3 * g.standard_normal((batch, t, 1)), -0.999, 0.999).astype(np.float32)
    if "z" in want:
        g = np.random.Generator(np.random.PCG64(77))
        out["z"] = (hp.temp * g.standard_normal((batch, t, 1))).astype(np.float32)
    return out
