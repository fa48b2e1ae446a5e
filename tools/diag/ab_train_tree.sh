# GPU box: training tests, then the training step of the committed tree (ab_old/: `git archive HEAD` + its libfwn.so) against the
# working tree, interleaved on the same box
cd "$(dirname "$0")/../.."
O=gpurun_out/ab_train
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_train.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
for i in 1 2 3; do
for tree in ab_old .; do
  r=$(cd $tree && python3 tools/bench_train.py --steps 20 --warmup 3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms  loss %.6f  gnorm %.6f' % (d['ms_per_step'], d['loss'], d['grad_norm']))")
  echo "$tree: $r"
done; done
