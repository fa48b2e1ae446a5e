cd "$(dirname "$0")/../.."
for i in 1 2; do for l in 2 3 4 6; do
python3 bench.py --no-cpu-baseline --no-train --no-rtf --no-fp8 --no-latency --lanes $l 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes $l step', round(d['ms_per_step'],3), 'value', round(d['value']/1e6,2))"
done; done
