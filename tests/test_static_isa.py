"""Static checks over the gfx950 ISA of the kernel sources (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tf-flowavenet_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_no_ring_barrier_is_crossed_with_lds_reads_in_flight(tmp_path):
    """tools/check_barrier_lgkm.py over the inference kernels: no s_barrier is reached with ds_reads in flight (the
    round-3 front_mfma_kernel race, DESIGN.md section 3.5: hipcc sinks the lgkmcnt wait of a chunk's fragment reads below the
    raw barrier that licenses refilling the slot they read; FWN_RING_BARRIER retires them first).  The checker must also
    still SEE the pattern: a ring barrier without the wait is flagged."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_barrier_lgkm as chk
    flags = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off -fno-slp-vectorize -S --cuda-device-only".split()
    out = tmp_path / "flow_kernels.s"
    subprocess.run([HIPCC] + flags + ["-c", os.path.join(CSRC, "flow_kernels.hip"), "-o", str(out)], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    flagged = [(name, rep) for name, body in chk.kernels(str(out)) for rep in [chk.check(name, body)] if rep]
    assert not flagged, flagged[:3]
    # the checker on a hand-made body: reads in flight at the barrier, a refill behind it
    body = ["\tds_read_b128 v[0:3], v4\n", "\ts_waitcnt vmcnt(4)\n", "\ts_barrier\n",
            "\tbuffer_load_dwordx4 v5, s[0:3], 0 offen lds\n", "\ts_waitcnt lgkmcnt(0)\n"]
    rep = chk.check("k", body)
    assert len(rep) == 1 and rep[0][1] == 1 and rep[0][3] is not None
    body.insert(1, "\ts_waitcnt lgkmcnt(0)\n")
    assert chk.check("k", body) == []
