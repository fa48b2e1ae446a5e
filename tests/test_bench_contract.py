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


def test_gpus_flag_launches_ranks_as_a_child_job_and_reports_their_failure():
    """``--gpus 2`` without a torchrun environment: the parent (which never touches a GPU) starts the two ranks with
    torch.distributed.run as a child and must pass their failure on (here: no GPU) instead of printing a 1-rank line."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0
    assert "2-rank job failed" in out.stderr and "needs a GPU" in out.stderr
    assert '"metric"' not in out.stdout


def test_rank_count_must_match_the_gpus_flag():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0 and "started 2 ranks" in (out.stderr + out.stdout)


def test_traffic_is_only_reported_for_the_current_kernel_sources(tmp_path, monkeypatch):
    import json
    import bench
    sha = bench.gate_source_hash()
    assert len(sha) == 16
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    src = tmp_path / "tf-flowavenet_amd" / "csrc"
    os.makedirs(src)
    for name in bench.GATE_SOURCES:
        (src / name).write_text("// " + name)
    cur = bench.gate_source_hash()
    (tmp_path / "profiles" / "r01_gate_traffic.json").write_text(json.dumps({"rows": 64512, "traffic_bytes": 1, "source_sha": "stale"}))
    assert bench.gate_traffic(64512)[0] is None
    (tmp_path / "profiles" / "r02_gate_traffic.json").write_text(json.dumps({"rows": 64512, "traffic_bytes": 7, "source_sha": cur}))
    assert bench.gate_traffic(64512)[0] == 7 and bench.gate_traffic(100)[0] is None
    (src / "gate_halo.h").write_text("// edited")
    assert bench.gate_traffic(64512)[0] is None


def test_path_roofline_quotes_committed_profiles_only_for_the_current_kernel_sources(tmp_path, monkeypatch):
    """the bench line carries the PATH's fractions.  The whole-pass fractions and (round 4)
    the per-block table are measured live (the table is handed in: bench.block_table); the launch counts attached to it
    and the whole-pass HBM bytes come from profiles/ and must be dropped once a kernel source changed."""
    import json
    import bench
    from tf_flowavenet_amd.hparams import default_hparams
    hp = default_hparams()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    src = tmp_path / "tf-flowavenet_amd" / "csrc"
    os.makedirs(src)
    os.makedirs(tmp_path / "profiles")
    (src / "a.hip").write_text("// a")
    (src / "b.h").write_text("// b")
    sha = bench.kernel_source_hash()
    blocks = [{"block": i, "rows": 64512 >> i, "us": 100.0 + i, "gflop": 10.0, "frac": 0.1, "launches": 30} for i in range(8)]
    (tmp_path / "profiles" / "r03_pass_table.json").write_text(json.dumps({"source_sha": sha, "fwd": {"blocks": blocks}, "inv": {"blocks": blocks}}))
    (tmp_path / "profiles" / "r03_pass_traffic.json").write_text(json.dumps(
        {"source_sha": sha, "algorithmic_bytes": 882.5e6, "fwd": {"traffic_bytes": 4000000000, "ratio_to_algorithmic": 4.53},
         "inv": {"traffic_bytes": 3900000000, "ratio_to_algorithmic": 4.42}}))
    live = lambda: {d: [{"block": i, "rows": 64512 >> i, "us": 90.0 + i, "gflop": 10.0, "mfma_frac": 0.04} for i in range(8)] for d in ("fwd", "inv")}
    r = bench.path_roofline(hp, 8, 16128, 4.0e-3, 4.1e-3, live())
    flop = 16527360 * 8 * 16128
    assert abs(r["fwd"]["mfma_frac"] - flop / 4.0e-3 / 2.5e15) < 1e-12 and abs(r["fwd"]["frac_of_survey_bound"] - 853e-6 / 4.0e-3) < 1e-12
    assert abs(r["serial_pair_ms"] - 8.1) < 1e-9
    assert len(r["blocks"]["fwd"]) == 8 and r["blocks"]["inv"][3]["us"] == 93.0 and r["blocks"]["inv"][3]["launches"] == 30
    assert r["blocks_source"].startswith("live")
    assert abs(r["hbm"]["fwd"]["gbs_at_this_pass"] - 1000.0) < 1e-6 and abs(r["hbm"]["fwd"]["hbm_frac"] - 0.125) < 1e-9
    (src / "b.h").write_text("// edited")
    r = bench.path_roofline(hp, 8, 16128, 4.0e-3, 4.1e-3, live())
    assert "launches" not in r["blocks"]["fwd"][0] and r["hbm"] is None
    assert "no profile of the current kernel sources" in r["blocks_launch_counts_source"] and "no profile" in r["hbm_source"]
    assert bench.path_roofline(hp, 8, 16128, 4.0e-3, 4.1e-3)["blocks"] is None
    assert bench.path_roofline(hp, 3, 999 * 256, 1e-3, 1e-3)["survey_bound_us"] is None      # no SURVEY bound for other shapes
