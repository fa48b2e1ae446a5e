# GPU box: per-(block, stage) tables of the serial pass for the environment settings given as arguments, e.g.
#   bash tools/diag/pt.sh "BENCH_ARGS=--chain-mode=1" "BENCH_ARGS="    (optional PT_BATCH / PT_SAMPLES; BENCH_ARGS: extra bench.py flags,
#   e.g. --chain-mode / --persist-mode - the model's kernel selection is an argument, the package reads no environment; tunable
#   builds' FWN_* thresholds still go in as plain VAR=value settings)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=$PWD/gpurun_out/pt
mkdir -p $O
n=0
for setting in "$@"; do
  n=$((n+1))
  rm -rf $O/rp$n
  BA=$(for kv in $setting; do case $kv in BENCH_ARGS=*) echo ${kv#BENCH_ARGS=};; esac; done)
  (cd /tmp && env $setting rocprofv3 --kernel-trace --output-format csv -d $O/rp$n -o t -- python3 $OLDPWD/bench.py --no-cpu-baseline --serial --no-train --no-rtf --no-fp8 --no-latency --steps 8 --warmup 2 --batch ${PT_BATCH:-8} --samples ${PT_SAMPLES:-16128} $BA > $O/rocprof$n.log 2>&1)
  T=$(ls $O/rp$n/*/*kernel_trace.csv $O/rp$n/*kernel_trace.csv 2>/dev/null | head -1)
  echo "=== $setting"
  python3 tools/pass_table.py $T --batch ${PT_BATCH:-8} --samples ${PT_SAMPLES:-16128} | tee $O/table$n.txt
  rm -rf $O/rp$n
done
