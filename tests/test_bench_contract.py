"""CPU checks of bench.py's contract pieces that need no GPU: the algorithmic FLOP figure the roofline is computed
from, and the loud failure without a GPU (the HIP path has no CPU fallback)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_algorithmic_flops_per_sample_match_the_survey():
    import bench
    from tf_flowavenet_amd.hparams import default_hparams, hparams8000
    assert bench.flop_per_sample(default_hparams()) == 16527360              # n_block=8, n_flow=6, n_layer=2
    assert bench.flop_per_sample(hparams8000()) == 14685696                   # 8 kHz, n_block=5
    assert bench.flop_per_sample(default_hparams().replace(n_block=2, n_flow=2)) == 3478528


def test_bench_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "needs a GPU" in (out.stderr + out.stdout)
