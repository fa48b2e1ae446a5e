import csv, sys, collections
f = sys.argv[1]; steps = float(sys.argv[2])
d = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:70]
    d[k][0] += 1; d[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
tot = sum(v[1] for v in d.values())
print("total kernel ms/step %.2f, launches/step %.0f" % (tot / steps, sum(v[0] for v in d.values()) / steps))
for k, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:28]:
    print("%8.2f ms/step %7.0f calls/step %7.1f us  %s" % (v[1] / steps, v[0] / steps, v[1] / v[0] * 1e3, k))
