export FWN_LIB=tf-flowavenet_amd/csrc/libfwn_tune.so
for r in 1 2 3; do for v in 0 1; do
FWN_TRS=$v python tools/bench_train.py --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('FWN_TRS=$v round $r: %.3f ms/step' % d['ms_per_step'])"
done; done
