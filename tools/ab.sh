#!/bin/bash
# Same-box A/B (GPU box): ab_base/ holds a built copy of the baseline commit
# (git archive <rev> | tar -x -C ab_base && make -C ab_base/tf-flowavenet_amd/csrc).
# usage: tools/ab.sh [rounds] [extra bench args]
cd "$(dirname "$0")/.."
R=${1:-3}; shift
pick() { python3 -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print("%-5s step %.3f ms  fwd %.3f  inv %.3f  gate0 %.2f us" % (sys.argv[1], d["ms_per_step"], d["fwd_ms"], d["inv_ms"], d["roofline"]["launch_us"]))' "$1"; }
for i in $(seq $R); do
  python3 ab_base/bench.py --no-cpu-baseline "$@" 2>/dev/null | pick base
  python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | pick new
done
