#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the fp64 oracle (oracle/flowavenet_np.py).

The reference has no golden vectors and cannot be executed here (TensorFlow 1.12), so
these fixtures are outputs of the build's own oracle on seeded synthetic inputs/weights
(SURVEY section 8c).  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import flowavenet_np as onp          # noqa: E402
from tf_flowavenet_amd import weights as W       # noqa: E402
from tf_flowavenet_amd.hparams import default_hparams, hparams8000   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (hparams overrides, B, T, actnorm mode, ddi)
    "tiny_b2f2": (dict(n_block=2, n_flow=2, n_layer=2, hop_size=16, upsample_scales=[4, 4], num_mels=8), 2, 128, "random", False),
    "tiny_b3f3": (dict(n_block=3, n_flow=3, n_layer=2, hop_size=16, upsample_scales=[4, 4], num_mels=16), 1, 256, "random", False),
    "tiny_b4f2l3": (dict(n_block=4, n_flow=2, n_layer=3, hop_size=32, upsample_scales=[4, 8], num_mels=16), 3, 512, "random", False),
    "tiny_ddi": (dict(n_block=2, n_flow=2, n_layer=2, hop_size=16, upsample_scales=[4, 4], num_mels=8), 2, 128, "zeros", True),
    "config0_b2f2_T16128": (dict(n_block=2, n_flow=2), 1, 16128, "zeros", True),   # BASELINE configs[0]
    "full_b8f6_T2048": (dict(), 1, 2048, "zeros", True),                            # full architecture, short clip
    "full_b8f6_B2_T1024": (dict(), 2, 1024, "zeros", True),                         # T_i = 2 rows at block 7
    "hp8000_b5f6_T1536": ("8k", 2, 1536, "zeros", True),                            # hparams8000 (hop 96)
    # BASELINE configs at their real sizes (fp64 oracle: 6 s / 50 s / 90 s per direction here)
    "full_b8f6_B1_T16128": (dict(), 1, 16128, "zeros", True),                       # configs[1], latency shape
    "full_b8f6_B8_T16128": (dict(), 8, 16128, "zeros", True),                       # configs[1], the bench.py workload
    "full_b8f6_T220672_10s": (dict(), 1, 220672, "zeros", True),                    # configs[3], 10 s @ 22.05 kHz
    "hp8000_b5f6_B2_T16128": ("8k", 2, 16128, "zeros", True),                       # configs[4]: 8 kHz model at an fp8-gate size
}
# the large cases keep z / x_rev as float16 (|values| < 8: half an ulp <= 2e-3, inside the stated tolerances)
FP16_CASES = ("full_b8f6_B8_T16128", "full_b8f6_T220672_10s", "hp8000_b5f6_B2_T16128")


def hp_of(over):
    if over == "8k":
        return hparams8000()
    return default_hparams().replace(**over)


def make(name):
    over, b, t, actnorm, ddi = CASES[name]
    hp = hp_of(over)
    params = W.synthetic_params(hp, 1234, actnorm=actnorm)
    inp = W.synthetic_inputs(hp, b, t)
    p64 = onp.to_f64(params)
    x, c, z = (inp[k].astype(np.float64) for k in ("x", "c", "z"))
    log_p, logdet, zout = onp.forward(p64, x, c, hp, init=ddi)
    store = np.float16 if name in FP16_CASES else np.float32
    out = dict(log_p=log_p, logdet=logdet, z=zout.astype(store), b=b, t=t)
    if (hp.n_block * hp.n_flow) % 2 == 0:
        out["x_rev"] = onp.reverse(p64, z, c, hp).astype(store)
    if ddi:
        out["an_b_last"] = p64["Block_%d/Flow_%d/ActNorm/b" % (hp.n_block - 1, hp.n_flow - 1)]
        out["an_logs_last"] = p64["Block_%d/Flow_%d/ActNorm/logs" % (hp.n_block - 1, hp.n_flow - 1)]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "log_p %.9f logdet %.9f" % (log_p, logdet))


if __name__ == "__main__":
    for n in (sys.argv[1:] or CASES):
        make(n)
