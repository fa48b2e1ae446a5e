"""Summarise a rocprofv3 kernel trace CSV per (kernel, grid): python tools/prof_summary.py <trace.csv> [passes]"""
import csv, collections, sys
f = sys.argv[1]
passes = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if name in ("wn_scale_kernel", "pack_kernel") or name.startswith("at::") or name.startswith("__amd"):
        continue
    key = (name, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["Scratch_Size"])
    d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += d
tot = sum(t for _, t in agg.values())
print("total kernel ms per pass: %.3f" % (tot / 1e6 / passes))
by = collections.defaultdict(float)
for k, (n, t) in agg.items(): by[k[0]] += t
for k, t in sorted(by.items(), key=lambda kv: -kv[1]): print("  %-40s %8.3f ms/pass" % (k[:40], t / 1e6 / passes))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%-34s wgs %6d y%3s lds %6s vgpr %4s agpr %4s scr %4s | n %4d avg_us %8.2f ms/pass %7.3f" % (k[0][:34], k[1], k[2], k[3], k[4], k[5], k[6], n, t / n / 1e3, t / 1e6 / passes))
