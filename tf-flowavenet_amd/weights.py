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


_LEAF = {"kernel": "kernel", "wn/g": "g", "bias": "bias"}
_BARE = ("filter_conv_c", "gate_conv_c", "res_conv", "skip_conv")      # first-call order, modules.py:113-127


def _split_layer_var(rest):
    """``<keras layer scope>/<kernel | wn/g | bias>`` (convolutional.py:64-87) -> (layer scope, our leaf) or None."""
    for tf_leaf, leaf in _LEAF.items():
        if rest.endswith("/" + tf_leaf):
            return rest[:-len(tf_leaf) - 1], leaf
    return None


def _layer_index(scope, base):
    """``conv1d`` -> 0, ``conv1d_3`` -> 3 (variable_scope(default_name=...) numbering); None if it is not `base`."""
    if scope == base:
        return 0
    if scope.startswith(base + "_") and scope[len(base) + 1:].isdigit():
        return int(scope[len(base) + 1:])
    return None


def from_reference_names(variables: dict) -> dict:
    """Variables dumped from one of the reference's TF checkpoints -> this package's parameter names.

    The reference's variable names are its nested ``tf.variable_scope``s plus the scope every un-named
    ``tf.layers`` layer opens at its first call (``variable_scope(default_name=<snake-cased class name>)``):

    ======================================================================  ==========================================
    TF variable (train.py:53 + model.py:283 prefix ``vocoder/FloWaveNet/``)   parameter here
    ======================================================================  ==========================================
    ``conv2d_transpose[_n]/{kernel,wn/g,bias}`` (model.py:301-311,398-404)   ``upsample_n/{kernel,g,bias}``
    ``Block_i/Flow_j/ActNorm/{b,logs}`` (model.py:13,57,71,181,218,297)       unchanged
    ``Block_i/Flow_j/AffineCoupling/WaveNet/...`` (model.py:110,114)          ``Block_i/Flow_j/WaveNet/...``
    ``.../Conv_front|Conv_final/conv1d/{kernel,wn/g,bias}`` (modules.py:8,17) ``.../Conv_front|Conv_final/{kernel,g,bias}``
    ``.../ResBlock_0_n/Conv_filter|Conv_gate/conv1d/...`` (modules.py:73,152)  ``.../ResBlock_n/Conv_filter|Conv_gate/...``
    ``.../ResBlock_0_n/conv1d[_k]/...``: the four un-scoped 1x1 convs, k in    ``.../ResBlock_n/{filter_conv_c,gate_conv_c,
    first-call order filter_c, gate_c, res, skip (modules.py:113-127)          res_conv,skip_conv}/...``
    ``.../ZeroConv1d/conv1d/{kernel,bias}``, ``.../ZeroConv1d/scale`` (:41-49)  ``.../ZeroConv1d/{kernel,bias,scale}``
    ======================================================================  ==========================================

    The four bare convs of a ResBlock are told apart by kernel shape (the conditioning pair has C_in = cin != 256
    = the res / skip pair) and, inside a pair, by scope index - which is the same under first-call and under
    construction order, so either numbering convention maps correctly.  Optimiser slots (``.../Adam``,
    ``.../Adam_1``, ``beta*_power``), ``global_step``, ``speaker_embeddings`` (inert, modules.py:188-189) and the
    cached low-precision casts of ``fp16_dtype_getter`` (utils.py:19-29) are dropped; names already in this
    package's form pass through."""
    out, bare = {}, {}
    for name, value in variables.items():
        n = name[:-2] if name.endswith(":0") else name
        leaf = n.rsplit("/", 1)[-1]
        if leaf in ("Adam", "Adam_1", "global_step", "speaker_embeddings") or leaf.startswith("beta") or \
                "fp16_cast" in n or n.startswith("__opt/"):
            continue
        if "FloWaveNet/" in n:
            n = n[n.index("FloWaveNet/") + len("FloWaveNet/"):]
        top = n.split("/", 1)
        if len(top) == 2 and _layer_index(top[0], "conv2d_transpose") is not None and _split_layer_var(n):
            scope, lf = _split_layer_var(n)
            out["upsample_%d/%s" % (_layer_index(scope, "conv2d_transpose"), lf)] = value
            continue
        if "/AffineCoupling/WaveNet/" not in n:
            out[n] = value                                   # ActNorm, or a name already in this package's form
            continue
        flow, rest = n.split("/AffineCoupling/WaveNet/")
        wp = flow + "/WaveNet/"
        parts = rest.split("/")
        if parts[0].startswith("ResBlock_"):                 # 'ResBlock_%d_%d' % (b, n), num_blocks = 1 (model.py:116)
            ids = parts[0].split("_")[1:]
            if len(ids) != 2 or ids[0] != "0":
                raise KeyError("unexpected ResBlock scope in %r" % name)
            rp = wp + "ResBlock_%d/" % int(ids[1])
            sv = _split_layer_var("/".join(parts[1:]))
            if sv is None:
                raise KeyError("unexpected variable %r" % name)
            scope, lf = sv
            if scope.startswith("Conv_filter/") or scope.startswith("Conv_gate/"):
                out[rp + scope.split("/")[0] + "/" + lf] = value
            else:
                k = _layer_index(scope, "conv1d")
                if k is None:
                    raise KeyError("unexpected layer scope in %r" % name)
                bare.setdefault(rp, {}).setdefault(k, {})[lf] = value
            continue
        if parts[0] == "ZeroConv1d" and parts[1:] == ["scale"]:
            out[wp + "ZeroConv1d/scale"] = value
            continue
        sv = _split_layer_var("/".join(parts[1:]))
        if parts[0] not in ("Conv_front", "Conv_final", "ZeroConv1d") or sv is None or _layer_index(sv[0], "conv1d") is None:
            raise KeyError("unexpected variable %r" % name)
        out[wp + parts[0] + "/" + sv[1]] = value
    for rp, layers in bare.items():
        cond = sorted(k for k, v in layers.items() if np.shape(v["kernel"])[1] != FILTER)
        plain = sorted(k for k, v in layers.items() if np.shape(v["kernel"])[1] == FILTER)
        if len(cond) != 2 or len(plain) != 2:
            raise KeyError("%s: expected two conditioning and two res/skip 1x1 convs, found %d + %d (global conditioning "
                           "convs are never built: modules.py:188-189)" % (rp, len(cond), len(plain)))
        for k, nm in zip(cond + plain, _BARE):
            for lf, value in layers[k].items():
                out[rp + nm + "/" + lf] = value
    return out


def to_reference_names(params: dict, prefix: str = "vocoder/FloWaveNet/") -> dict:
    """The inverse map: this package's names -> the reference's TF variable names (see ``from_reference_names``),
    e.g. to hand a checkpoint trained here to the reference's ``tf.train.Saver`` tooling."""
    tf_leaf = {v: k for k, v in _LEAF.items()}
    out = {}
    for name, value in params.items():
        head, leaf = name.rsplit("/", 1)
        if head.startswith("upsample_"):
            n = int(head.split("_")[1])
            out["%sconv2d_transpose%s/%s" % (prefix, "_%d" % n if n else "", tf_leaf[leaf])] = value
        elif "/WaveNet/" not in name:
            out[prefix + name] = value
        else:
            flow, rest = name.split("/WaveNet/")
            parts = rest.split("/")
            base = "%s%s/AffineCoupling/WaveNet/" % (prefix, flow)
            if parts[0].startswith("ResBlock_"):
                base += "ResBlock_0_%d/" % int(parts[0].split("_")[1])
                if parts[1] in _BARE:
                    k = _BARE.index(parts[1])
                    out["%sconv1d%s/%s" % (base, "_%d" % k if k else "", tf_leaf[leaf])] = value
                else:
                    out["%s%s/conv1d/%s" % (base, parts[1], tf_leaf[leaf])] = value
            elif leaf == "scale":
                out[base + "ZeroConv1d/scale"] = value
            else:
                out["%s%s/conv1d/%s" % (base, parts[0], tf_leaf[leaf])] = value
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
