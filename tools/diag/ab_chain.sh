# GPU box: parity subset, then single-pass times with the flows chained (default) and every flow on its own (FWN_CHAIN_MODE=1)
cd "$(dirname "$0")/../.."
O=gpurun_out/ab_chain
mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_fp8.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
run() {
python3 - "$@" <<'PY'
import os, sys, subprocess
src = open("tools/tune.py").read()
child = src.split('CHILD = r"""')[1].split('""" % ROOT')[0] % os.getcwd()
env = dict(os.environ)
for kv in sys.argv[2:]:
    k, v = kv.split("="); env[k] = v
r = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True)
print("%-34s %s" % (sys.argv[1], r.stdout.strip() or r.stderr[-800:]), flush=True)
PY
}
for i in 1 2; do
run "new, unchained" FWN_CHAIN_MODE=1
run "new, chained"
done 2>&1 | tee $O/times.txt
