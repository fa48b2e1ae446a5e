"""GPU box: bench.py's overlapped headline with a process-wide developer option set first (fwn_set_option), interleaved with
the default, e.g.   python tools/diag/ab_option.py rs_persist 1 [rounds]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import sys
sys.path.insert(0, %r)
from tf_flowavenet_amd import _lib
name, val = sys.argv[1], int(sys.argv[2])
if name != "none":
    _lib.load().fwn_set_option(name.encode(), val)
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-train", "--no-rtf", "--no-fp8", "--no-latency"]
import bench
bench.main()
""" % ROOT

name, val = sys.argv[1], sys.argv[2]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
for r in range(rounds):
    for n, v in (("none", "0"), (name, val)):
        out = subprocess.run([sys.executable, "-c", CHILD, n, v], capture_output=True, text=True, cwd=ROOT)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        if not line:
            print(n, v, "FAILED", out.stderr[-400:])
            continue
        d = json.loads(line[-1])
        print("%-12s %s: %.3f M samples/s, %.3f ms per step (regions %s)" % (n, v, d["value"] / 1e6, d["ms_per_step"],
              ["%.3f" % t for t in d["config"]["timed_regions"]["ms_per_step_each"]]), flush=True)
