"""Diagnostic: one forward pass at (B, T) under rocprofv3 --kernel-trace: kernel time vs gaps between dependent launches.
usage (under rocprofv3): pass_gaps.py B T   ;  then: pass_gaps.py --parse <kernel_trace.csv>"""
import os, sys
if sys.argv[1] == "--parse":
    import csv
    rows = list(csv.DictReader(open(sys.argv[2])))
    for r in rows: r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    # last pass = kernels after the last split_kernel
    starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("split_kernel")]
    ps = rows[starts[-1]:]
    busy = sum(r["e"] - r["s"] for r in ps) / 1e3
    wall = (ps[-1]["e"] - ps[0]["s"]) / 1e3
    gaps = [(ps[i + 1]["s"] - ps[i]["e"]) / 1e3 for i in range(len(ps) - 1)]
    print("kernels %d  wall %.1f us  busy %.1f us  gaps %.1f us (mean %.2f, max %.2f)" % (len(ps), wall, busy, sum(gaps), sum(gaps) / len(gaps), max(gaps)))
    import collections
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in ps:
        k = r["Kernel_Name"].split("(")[0][:60]
        d[k][0] += 1; d[k][1] += (r["e"] - r["s"]) / 1e3
    for k, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:16]:
        print("%8.1f us %4d x %6.2f us  %s" % (v[1], v[0], v[1] / v[0], k))
    sys.exit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tf_flowavenet_amd import weights as W
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd.model import FloWaveNet
b, t = int(sys.argv[1]), int(sys.argv[2])
hp = default_hparams()
m = FloWaveNet(hp, init=True).load_params(W.synthetic_params(hp, 1234))
inp = W.synthetic_inputs(hp, b, t)
x, c = (torch.from_numpy(inp[k]).cuda() for k in ("x", "c"))
for _ in range(6):
    m.forward(x, c)
torch.cuda.synchronize()
