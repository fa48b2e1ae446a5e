# GPU box: bench.py's step / pass / gate-launch times under different environment settings, interleaved.
#   bash tools/diag/ab_bench_env.sh [rounds] "FWN_OPT_RS_PERSIST=0" "FWN_OPT_RS_PERSIST=1"
cd "$(dirname "$0")/../.."
rounds=$1; shift
for i in $(seq 1 $rounds); do
for setting in "$@"; do
  r=$(env $setting python3 bench.py --no-cpu-baseline --no-train --no-rtf --no-fp8 --no-latency 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('step %.3f ms  fwd %.3f inv %.3f  serial %.3f  gate0 %.2f us (frac %.3f)' % (d['ms_per_step'], d['fwd_ms'], d['inv_ms'], d['path']['serial_pair_ms'], d['roofline']['launch_us'], d['roofline']['frac']))")
  echo "$setting: $r"
done; done
