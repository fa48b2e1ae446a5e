# GPU box: single-pass times (tools/tune.py's child) under different environment settings, interleaved
#   bash tools/diag/ab_env.sh "FWN_SIDE_STREAM=0" "FWN_SIDE_STREAM=1"
cd "$(dirname "$0")/../.."
for i in 1 2; do
for setting in "$@"; do
python3 - "$setting" <<'PY'
import os, sys, subprocess
src = open("tools/tune.py").read()
child = src.split('CHILD = r"""')[1].split('""" % ROOT')[0] % os.getcwd()
env = dict(os.environ)
for kv in sys.argv[1].split():
    k, v = kv.split("="); env[k] = v
r = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True)
print("%-34s %s" % (sys.argv[1], r.stdout.strip() or r.stderr[-800:]), flush=True)
PY
done; done
