# GPU box: per-(block, stage) tables of the serial pass for the environment settings given as arguments, e.g.
#   bash tools/diag/pt.sh "FWN_CHAIN_MODE=1" "FWN_CHAIN_MODE=0"    (optional PT_BATCH / PT_SAMPLES)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=$PWD/gpurun_out/pt
mkdir -p $O
n=0
for setting in "$@"; do
  n=$((n+1))
  rm -rf $O/rp$n
  (cd /tmp && env $setting rocprofv3 --kernel-trace --output-format csv -d $O/rp$n -o t -- python3 $OLDPWD/bench.py --no-cpu-baseline --serial --no-train --no-rtf --no-fp8 --no-latency --steps 8 --warmup 2 --batch ${PT_BATCH:-8} --samples ${PT_SAMPLES:-16128} > $O/rocprof$n.log 2>&1)
  T=$(ls $O/rp$n/*/*kernel_trace.csv $O/rp$n/*kernel_trace.csv 2>/dev/null | head -1)
  echo "=== $setting"
  python3 tools/pass_table.py $T --batch ${PT_BATCH:-8} --samples ${PT_SAMPLES:-16128} | tee $O/table$n.txt
  rm -rf $O/rp$n
done
