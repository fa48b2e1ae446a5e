#!/bin/bash
# same-box A/B of a tuning-build variable on the headline: tools/diag/ab_tune_env.sh NAME V0 V1 [rounds]
# (make -C tf-flowavenet_amd/csrc tune first; prints M samples/s and ms per overlapped step of bench.py, interleaved)
N=$1; A=$2; B=$3; R=${4:-3}
export FWN_LIB=tf-flowavenet_amd/csrc/libfwn_tune.so
for r in $(seq $R); do for v in $A $B; do
env $N=$v python bench.py --no-cpu-baseline --no-train --no-rtf --no-fp8 --no-latency --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$N=$v round $r: %.2f M samples/s  %.3f ms/step' % (d['value']/1e6, d['ms_per_step']))"
done; done
