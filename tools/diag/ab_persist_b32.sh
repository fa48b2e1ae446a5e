cd "$(dirname "$0")/../.."
for i in 1 2; do for v in 0 1; do
  r=$(FWN_OPT_RS_PERSIST=$v python3 bench.py --batch 32 --steps 10 --no-cpu-baseline --no-train --no-rtf --no-fp8 --no-latency 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('step %.3f ms  fwd %.3f inv %.3f  serial %.3f' % (d['ms_per_step'], d['fwd_ms'], d['inv_ms'], d['path']['serial_pair_ms']))")
  echo "B=32 FWN_OPT_RS_PERSIST=$v: $r"
done; done
