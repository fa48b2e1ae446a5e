#!/bin/bash
# GPU box: regenerate the measurements kept under profiles/ (writes to gpurun_out/refresh/; copy what is to be judged
# into profiles/ with the round's tag).   usage: tools/refresh_profiles.sh [tag]
cd "$(dirname "$0")/.."
R=$PWD
TAG=${1:-r03}
O=$R/gpurun_out/refresh
mkdir -p $O
export TMPDIR=/tmp
# PMC passes of the dominant kernel first: bench.py quotes roofline.traffic from the profile of the current kernel sources
python3 tools/gate_pmc.py $O $TAG > $O/gate_pmc.log 2>&1
cp $O/${TAG}_gate_traffic.json $R/profiles/ 2>/dev/null
# whole-pass HBM bytes (FETCH_SIZE / WRITE_SIZE over every dispatch of a forward / inverse pass)
python3 tools/pass_pmc.py $O $TAG > $O/pass_pmc.log 2>&1
cp $O/${TAG}_pass_traffic.json $R/profiles/ 2>/dev/null
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -o $TAG -- python3 $R/bench.py --no-cpu-baseline --serial --no-train --no-rtf --no-fp8 --no-latency --steps 20 --warmup 3 > $O/rocprof.log 2>&1)
T=$(ls $O/rocprof/*/*kernel_trace.csv $O/rocprof/*kernel_trace.csv 2>/dev/null | head -1)
python3 tools/prof_summary.py $T 57 > $O/${TAG}_kernel_summary_B8.txt
# per-(block, stage) table of the one-stream pass (bench.py quotes it while the kernel sources are unchanged)
python3 tools/pass_table.py $T --json $O/${TAG}_pass_table.json > $O/${TAG}_pass_table.txt
cp $O/${TAG}_pass_table.json $R/profiles/ 2>/dev/null
cp $(ls $O/rocprof/*/*kernel_stats.csv $O/rocprof/*kernel_stats.csv 2>/dev/null | head -1) $O/${TAG}_rocprofv3_kernel_stats_B8.csv
rm -f $T   # tens of MB; the stats CSV and the summary are what is kept
# the same table for ONE clip (the latency shape: every block at its dependent-launch floor)
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/rocprof_b1 -o ${TAG}b1 -- python3 $R/bench.py --no-cpu-baseline --serial --no-train --no-rtf --no-fp8 --no-latency --steps 12 --warmup 3 --batch 1 > $O/rocprof_b1.log 2>&1)
T1=$(ls $O/rocprof_b1/*/*kernel_trace.csv $O/rocprof_b1/*kernel_trace.csv 2>/dev/null | head -1)
python3 tools/pass_table.py $T1 --batch 1 > $O/${TAG}_pass_table_B1.txt
rm -rf $O/rocprof_b1
# the bench lines last: they quote the per-block table and the traffic figures collected above
python3 bench.py > $O/${TAG}_bench_B8.json 2> $O/bench_B8.err
python3 bench.py --no-cpu-baseline --serial --no-train --no-rtf --no-fp8 > $O/${TAG}_bench_B8_serial.json 2>> $O/bench_B8.err
python3 tools/bench_extra.py > $O/${TAG}_bench_extra.txt 2>&1
# the data-parallel training step
python3 tools/bench_train.py --steps 20 --warmup 3 2> $O/bench_train.err | tail -1 > $O/${TAG}_bench_train.json
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_train -o ${TAG}t -- python3 $R/tools/bench_train.py --steps 10 --warmup 3 > $O/rocprof_train.log 2>&1)
cp $(ls $O/rocprof_train/*/*kernel_stats.csv $O/rocprof_train/*kernel_stats.csv 2>/dev/null | head -1) $O/${TAG}_rocprofv3_kernel_stats_train.csv
TT=$(ls $O/rocprof_train/*/*kernel_trace.csv $O/rocprof_train/*kernel_trace.csv 2>/dev/null | head -1)
python3 tools/kernel_summary.py $TT 14 > $O/${TAG}_kernel_summary_train.txt
python3 tools/diag/train_timeline.py $TT > $O/${TAG}_train_timeline.txt 2>&1
python3 tools/diag/train_blocks.py $TT > $O/${TAG}_train_blocks.txt 2>&1
rm -rf $O/rocprof $O/rocprof_train
ls -la $O
