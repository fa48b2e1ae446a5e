# GPU box: training step time + rocprofv3 kernel summary (tools/kernel_summary.py) -> gpurun_out/train_prof/
cd "$(dirname "$0")/../.."
R=$PWD
O=$R/gpurun_out/train_prof
mkdir -p $O
export TMPDIR=/tmp
python3 tools/bench_train.py --steps 20 --warmup 3 2> $O/bench_train.err | tail -1 > $O/bench_train.json
cat $O/bench_train.json
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp -o t -- python3 $R/tools/bench_train.py --steps 10 --warmup 3 > $O/rocprof.log 2>&1)
T=$(ls $O/rp/*/*kernel_trace.csv $O/rp/*kernel_trace.csv 2>/dev/null | head -1)
python3 tools/kernel_summary.py $T 14 > $O/kernel_summary_train.txt
python3 tools/diag/train_timeline.py $T > $O/timeline.txt 2>&1
python3 tools/diag/train_blocks.py $T > $O/blocks.txt 2>&1
rm -rf $O/rp
head -50 $O/kernel_summary_train.txt
