# GPU box, -DFWN_TUNABLE build (make -C tf-flowavenet_amd/csrc tune): bench.py's overlapped 8-clip step with the one-launch flow's
# default row limit (FWN_PERSIST_AUTO_ROWS) and workgroup count (FWN_PERSIST_GRID; unset = the launcher's rule) varied.
#   bash tools/diag/ab_persist_grid.sh "FWN_PERSIST_AUTO_ROWS=512" "FWN_PERSIST_AUTO_ROWS=1024" ...
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/grid
L=tf-flowavenet_amd/csrc/libfwn_tune.so
for rep in 1 2; do
n=0
for setting in "$@"; do
  n=$((n+1))
  env FWN_LIB=$L $setting python3 bench.py --no-train --no-fp8 --no-rtf --no-cpu-baseline --no-latency > gpurun_out/grid/g_$n.json 2>gpurun_out/grid/g_$n.err
  python3 - "$setting" gpurun_out/grid/g_$n.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("%-60s %.2f M samples/s  %.3f ms" % (sys.argv[1], d["value"] / 1e6, d["ms_per_step"]), flush=True)
except Exception as e:
    print(sys.argv[1], "ERR", e, flush=True)
PY
done; done
