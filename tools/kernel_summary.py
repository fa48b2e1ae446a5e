"""Per-kernel totals of the training step from a rocprofv3 kernel trace (csv).

    python tools/kernel_summary.py <kernel_trace.csv> [steps]

The steps are cut at `adam_clip_kernel` (one launch per step, the step's last big kernel): only launches between the first
and the last of them are counted and divided by the number of whole steps in between - the model packing of the
initialisation and the eager first step in front of them are NOT part of any step (round 4's summary divided the whole
trace by the step count: its "48 wn_scale_kernel + 51 pack_kernel launches per step" were the init).  [steps]: fallback
divisor when the trace holds fewer than two adam_clip_kernel launches."""
import collections
import csv
import sys

f = sys.argv[1]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70]))
rows.sort()
marks = [i for i, r in enumerate(rows) if r[2].startswith("adam_clip_kernel")]
if len(marks) >= 2:
    sel, steps, how = rows[marks[0] + 1: marks[-1] + 1], float(len(marks) - 1), "between the first and the last adam_clip_kernel"
else:
    sel, steps, how = rows, float(sys.argv[2]) if len(sys.argv) > 2 else 1.0, "whole trace / %s" % (sys.argv[2] if len(sys.argv) > 2 else "1")
d = collections.defaultdict(lambda: [0, 0.0])
for s, e, k in sel:
    d[k][0] += 1
    d[k][1] += (e - s) / 1e6
tot = sum(v[1] for v in d.values())
print("%d whole steps (%s): total kernel ms/step %.2f, launches/step %.0f" % (steps, how, tot / steps, sum(v[0] for v in d.values()) / steps))
for k, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:28]:
    print("%8.2f ms/step %7.1f calls/step %7.1f us  %s" % (v[1] / steps, v[0] / steps, v[1] / v[0] * 1e3, k))
