mkdir -p gpurun_out/v2
L=tf-flowavenet_amd/csrc/libfwn_tune.so
run() { # name, env...
  n=$1; shift
  env FWN_LIB=$L "$@" python bench.py --no-train --no-fp8 --no-rtf --no-cpu-baseline --no-latency > gpurun_out/v2/g_$n.json 2>gpurun_out/v2/g_$n.err
  python - "$n" <<PY
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open("gpurun_out/v2/g_%s.json"%n).read().strip().splitlines()[-1])
    print(n, "%.2f M/s"%(d["value"]/1e6), "%.3f ms"%d["ms_per_step"])
except Exception as e: print(n,"ERR",e)
PY
}
for rep in 1 2; do
run base_$rep FWN_PERSIST_AUTO_ROWS=256
run r512_$rep FWN_PERSIST_AUTO_ROWS=512
run r512_g64_$rep FWN_PERSIST_AUTO_ROWS=512 FWN_PERSIST_GRID=64
run r512_g96_$rep FWN_PERSIST_AUTO_ROWS=512 FWN_PERSIST_GRID=96
run r512_g128_$rep FWN_PERSIST_AUTO_ROWS=512 FWN_PERSIST_GRID=128
done
