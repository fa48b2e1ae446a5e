set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03_base
mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python3 bench.py --no-cpu-baseline --serial --no-train --no-rtf --no-fp8 > $O/bench_serial.json 2>> $O/bench.err
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/rocprof -o base -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --serial --no-train --no-rtf --no-fp8 --no-latency --steps 10 --warmup 2 > $O/rocprof.log 2>&1)
T=$(ls $O/rocprof/*kernel_trace.csv | head -1)
python3 tools/pass_table.py $T --json $O/pass_table.json > $O/pass_table.txt 2>&1
python3 tools/prof_summary.py $T 29 > $O/kernel_summary.txt
rm -rf $O/rocprof
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
cat $O/pass_table.txt
