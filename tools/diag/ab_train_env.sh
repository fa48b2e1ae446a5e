# GPU box: training step time (tools/bench_train.py) with the tunable build under different environment settings
cd "$(dirname "$0")/../.."
for i in 1 2; do
for setting in "$@"; do
  r=$(env FWN_LIB=tf-flowavenet_amd/csrc/libfwn_tune.so $setting python3 tools/bench_train.py --steps 20 --warmup 3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms  loss %.6f  gnorm %.6f' % (d['ms_per_step'], d['loss'], d['grad_norm']))")
  echo "$setting: $r"
done; done
