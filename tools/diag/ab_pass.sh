#!/bin/bash
# same-box A/B of two builds of the library: single-pass times (tools/tune.py's child, FWN_LIB selects the build)
# usage: tools/diag/ab_pass.sh libfwn_old.so libfwn.so [repeats]
cd "$(dirname "$0")/../.."
for i in $(seq 1 ${3:-2}); do
for lib in $1 $2; do
python - "$lib" <<'PY'
import os, sys, subprocess
lib = os.path.join("tf-flowavenet_amd", "csrc", sys.argv[1])
src = open("tools/tune.py").read()
child = src.split('CHILD = r"""')[1].split('""" % ROOT')[0] % os.getcwd()
r = subprocess.run([sys.executable, "-c", child], env=dict(os.environ, FWN_LIB=lib), capture_output=True, text=True)
print("%-16s %s" % (sys.argv[1], r.stdout.strip() or r.stderr[-400:]), flush=True)
PY
done; done
