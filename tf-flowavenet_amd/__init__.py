"""MI355X-native FloWaveNet flow forward (log-likelihood) / inverse (synthesis) path.

Drop-in surface for the reference's hot path:
``hparams`` (hparams.py), ``FloWaveNet(hparams, init).forward/reverse/upsample``
(model.py:283-404) and the ``synthesize`` CLI (synthesize.py).  All arithmetic on
the path runs in hand-written HIP kernels for gfx950 behind the C-ABI declared in
``include/fwn.h`` (``csrc/libfwn.so``); there is no CPU or eager fallback.
"""
__version__ = "0.1.0"
