# GPU box: bench.py's gate launch / pass times with two builds, interleaved.  usage: ab_bench_lib.sh libA.so libB.so [rounds]
cd "$(dirname "$0")/../.."
for i in $(seq 1 ${3:-3}); do
for lib in "$1" "$2"; do
  r=$(FWN_LIB=$PWD/tf-flowavenet_amd/csrc/$lib python3 bench.py --no-cpu-baseline --no-train --no-rtf --no-fp8 --no-latency 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('step %.3f ms  fwd %.3f inv %.3f  gate0 %.2f us (frac %.3f)' % (d['ms_per_step'], d['fwd_ms'], d['inv_ms'], d['roofline']['launch_us'], d['roofline']['frac']))")
  echo "$lib: $r"
done; done
