#!/bin/bash
# usage (GPU box): tools/pmc_gemm.sh <abl> "<counters>"  -- PMC counters for the gemm micro-benchmark
cd "$(dirname "$0")/.."
R=$PWD
abl=${1:-0}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DFWN_ABL=$abl tools/bench_gemm.hip -o /tmp/bench_gemm_$abl 2>/dev/null
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmc_out
rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/pmc_out -- /tmp/bench_gemm_$abl 8 > /tmp/pmc.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmc_out/*/*counter_collection.csv")
if not f:
    print(open("/tmp/pmc.log").read()[-2000:]); raise SystemExit
rows = list(csv.DictReader(open(f[0])))
agg = collections.OrderedDict()
for r in rows:
    k = (r["Kernel_Name"].split("(")[0].replace("void ", "")[:60], r["Grid_Size"])
    d = agg.setdefault(k, collections.defaultdict(float)); d[r["Counter_Name"]] += float(r["Counter_Value"]); d["_n_" + r["Counter_Name"]] += 1
for k, d in agg.items():
    import os
    if os.environ.get("PMC_FILTER", "Gate") not in k[0]: continue
    print(k, {c: round(v / d["_n_" + c]) for c, v in d.items() if not c.startswith("_n_")})
PY
