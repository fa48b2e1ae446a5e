import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:
        import torch
        # The GPU box exposes 256 hardware threads; CPU-side oracle maths is tiny.
        torch.set_num_threads(min(8, os.cpu_count() or 1))
    except Exception:
        pass


def small_hparams(**kw):
    from tf_flowavenet_amd.hparams import default_hparams
    base = dict(n_block=2, n_flow=2, n_layer=2, hop_size=16, upsample_scales=[4, 4], num_mels=8)
    base.update(kw)
    return default_hparams().replace(**base)


@pytest.fixture
def make_hp():
    return small_hparams
