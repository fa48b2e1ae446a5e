#!/bin/bash
# GPU box: per-(block, stage) table of the one-stream pass + the overlapped headline, quickly (no training / fp8 / latency legs).
#   usage: tools/diag/quick_pass.sh [tag]      ->  gpurun_out/quick/<tag>_pass_table.txt, <tag>_bench.json
cd "$(dirname "$0")/../.."
R=$PWD
TAG=${1:-q}
O=$R/gpurun_out/quick
mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/rocprof_$TAG -o $TAG -- python3 $R/bench.py --no-cpu-baseline --serial --no-train --no-rtf --no-fp8 --no-latency --steps 12 --warmup 3 > $O/${TAG}_rocprof.log 2>&1)
T=$(ls $O/rocprof_$TAG/*/*kernel_trace.csv $O/rocprof_$TAG/*kernel_trace.csv 2>/dev/null | head -1)
python3 tools/pass_table.py $T > $O/${TAG}_pass_table.txt
python3 tools/prof_summary.py $T 40 > $O/${TAG}_kernel_summary.txt
rm -rf $O/rocprof_$TAG
python3 bench.py --no-cpu-baseline --no-train --no-rtf --no-fp8 --no-latency > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
cat $O/${TAG}_pass_table.txt
python3 - <<PY
import json
d = json.loads(open("$O/${TAG}_bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["unit"], "ms_per_step", d["ms_per_step"], "roofline", d.get("roofline", {}).get("frac"), d.get("roofline", {}).get("launch_us"))
PY
