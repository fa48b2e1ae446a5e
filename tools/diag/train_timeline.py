"""Diagnostic: per-queue timeline of the LAST training step in a rocprofv3 kernel trace (tools/bench_train.py under
rocprofv3 --kernel-trace): busy time per queue, overlap, and the gaps of the main queue.
usage: train_timeline.py <kernel_trace.csv> [out.csv]"""
import csv, sys, collections, os
rows = list(csv.DictReader(open(sys.argv[1])))
print("columns:", list(rows[0].keys()))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# a step starts at each wn_scale_jobs_kernel (first kernel of the packing refresh)
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("wn_scale_jobs_kernel")]
a, b = starts[-2], starts[-1]
step = rows[a:b]
t0 = step[0]["s"]
print("step: %d kernels, %.3f ms" % (len(step), (step[-1]["e"] - t0) / 1e6))
qs = collections.defaultdict(list)
for r in step:
    qs[r["Queue_Id"]].append(r)
for q, rs in qs.items():
    busy = sum(r["e"] - r["s"] for r in rs) / 1e6
    print("queue %s: %d kernels, busy %.3f ms, first at %.3f, last end %.3f" % (q, len(rs), busy, (rs[0]["s"] - t0) / 1e6, (rs[-1]["e"] - t0) / 1e6))
main = max(qs.values(), key=len)
others = [r for q, rs in qs.items() if rs is not main for r in rs]
# per 1 ms window: main busy, side busy
T = (step[-1]["e"] - t0) / 1e6
nb = int(T) + 1
mb, sb = [0.0] * nb, [0.0] * nb
def add(arr, r):
    s, e = (r["s"] - t0) / 1e6, (r["e"] - t0) / 1e6
    k = int(s)
    while s < e and k < nb:
        seg = min(e, k + 1) - s
        arr[k] += seg
        s += seg; k += 1
for r in main: add(mb, r)
for r in others: add(sb, r)
print("per-ms window: main busy | side busy")
for k in range(nb):
    print("%3d  %.2f  %.2f" % (k, mb[k], sb[k]))
# largest gaps on the main queue
gaps = sorted(((main[i + 1]["s"] - main[i]["e"]) / 1e3, (main[i]["e"] - t0) / 1e6, main[i]["Kernel_Name"][:40], main[i + 1]["Kernel_Name"][:40]) for i in range(len(main) - 1))[-12:]
print("largest main-queue gaps (us, at ms, after, before):")
for g in gaps:
    print("  %.1f us at %.3f ms  %s -> %s" % g)
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as f:
        for r in step:
            f.write("%s,%s,%.2f,%.2f,%s\n" % (r["Queue_Id"], r["Kernel_Name"].split("(")[0][:50].replace(",", ";"), (r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, r["Grid_Size_X"]))
# phases of the step (main markers)
def first(name, after=0):
    for r in step:
        if r["Kernel_Name"].startswith(name) and (r["s"] - t0) / 1e6 >= after:
            return (r["s"] - t0) / 1e6
    return float("nan")
tp, tf = first("pack_jobs_kernel"), first("upsample_kernel")
tl, tb, to = first("prior_kernel"), first("(anonymous namespace)::planes_to_rows_kernel"), first("sqnorm_partial_kernel")
print("phases (ms): refresh+pack 0-%.2f | forward %.2f-%.2f | backward %.2f-%.2f | up-sampling bwd + optimiser %.2f-%.2f" % (tf, tf, tl, tl, tb, tb, T))
print("   optimiser starts at %.2f" % to)
# kernel time of THIS step by name (the whole-trace summaries include the one-off setup launches)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    k = r["Kernel_Name"].split("(")[0][:64]
    agg[k][0] += 1; agg[k][1] += (r["e"] - r["s"]) / 1e3
print("last step: %d launches, %.2f ms of kernel time" % (len(step), sum(v[1] for v in agg.values()) / 1e3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('TOPN', '32'))]:
    print("%8.1f us %4d x %7.2f us  %s" % (v[1], v[0], v[1] / v[0], k))
