# GPU box: training step time (tools/bench_train.py, product build) under different environment settings, interleaved
#   bash tools/diag/ab_train_env2.sh [rounds] "FWN_TRAIN_GRAPH=0" "FWN_TRAIN_GRAPH=0 FWN_TRAIN_SIDE_CUMASK=4 FWN_TRAIN_DEFER=1"
cd "$(dirname "$0")/../.."
rounds=$1; shift
for i in $(seq 1 $rounds); do
for setting in "$@"; do
  r=$(env $setting python3 tools/bench_train.py --steps 20 --warmup 4 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms  loss %.6f  gnorm %.6f' % (d['ms_per_step'], d['loss'], d['grad_norm']))" 2>&1 | tail -1)
  echo "$setting: $r"
done; done
