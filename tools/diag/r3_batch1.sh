cd "$(dirname "$0")/../.."
O=gpurun_out/r3b1
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_train.py tests/test_gpu_parity.py -x -q -s -m gpu -k "directional or synthesize_cli or real_sizes or callback" > $O/pytest_new.txt 2>&1
grep -E "passed|failed|z error|wav |<g, d>|Error" $O/pytest_new.txt | tail -30
for i in 1 2; do
for st in 2 0; do
FWN_LIB=tf-flowavenet_amd/csrc/libfwn_tune.so FWN_SMALL_TILE=$st python3 bench.py --no-cpu-baseline --no-train --no-rtf --no-fp8 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SMALL_TILE=$st step', round(d['ms_per_step'],3), 'fwd', round(d['fwd_ms'],3), 'inv', round(d['inv_ms'],3), 'b1', round(d['latency_b1']['fwd_ms'],3))"
done; done
