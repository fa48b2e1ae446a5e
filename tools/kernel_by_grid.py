import csv, sys, collections
d = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] not in r["Kernel_Name"]: continue
    k = (r["Kernel_Name"].split("(")[0][:40], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", "?"))
    d[k][0] += 1; d[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
steps = float(sys.argv[3])
for k, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:24]:
    print("%-42s grid %8s wg %5s  %5.0f calls/step  %7.1f us avg  %6.2f ms/step" % (k[0], k[1], k[2], v[0] / steps, v[1] / v[0], v[1] / steps / 1e3))
