for i in 1 2; do
for lib in libfwn_old.so libfwn.so; do
echo $lib; FWN_LIB=tf-flowavenet_amd/csrc/$lib python bench.py --serial --no-cpu-baseline --no-train --no-rtf --no-fp8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['fwd_ms'], d['inv_ms'])"
FWN_LIB=tf-flowavenet_amd/csrc/$lib python bench.py --no-cpu-baseline --no-train --no-rtf --no-fp8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done; done
