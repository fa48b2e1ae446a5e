#!/bin/bash
# GPU box: regenerate the measurements kept under profiles/ (writes to gpurun_out/refresh/).
# usage: tools/refresh_profiles.sh
cd "$(dirname "$0")/.."
R=$PWD
O=$R/gpurun_out/refresh
mkdir -p $O
export TMPDIR=/tmp
python3 bench.py > $O/bench_B8.json 2> $O/bench_B8.err
python3 bench.py --no-cpu-baseline --serial > $O/bench_B8_serial.json 2>> $O/bench_B8.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -o r01 -- python3 $R/bench.py --no-cpu-baseline --serial --steps 20 --warmup 3 > $O/rocprof.log 2>&1)
T=$(ls $O/rocprof/*kernel_trace.csv | head -1)
python3 tools/prof_summary.py $T 57 > $O/kernel_summary_B8.txt
cp $(ls $O/rocprof/*kernel_stats.csv | head -1) $O/rocprofv3_kernel_stats_B8.csv
rm -f $T   # tens of MB; the stats CSV and the summary are what is kept
python3 tools/bench_extra.py > $O/bench_extra.txt 2>&1
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  echo "## $c"; PMC_FILTER=gate_halo_kernel tools/pmc_gemm.sh 0 "$c" 2>&1 | grep "258048\|516096" | head -2
done > $O/gate_pmc_raw.txt 2>&1
# the data-parallel training step
python3 tools/bench_train.py --steps 20 --warmup 3 2> $O/bench_train.err | tail -1 > $O/bench_train.json
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_train -o r01t -- python3 $R/tools/bench_train.py --steps 10 --warmup 3 > $O/rocprof_train.log 2>&1)
cp $(ls $O/rocprof_train/*kernel_stats.csv | head -1) $O/rocprofv3_kernel_stats_train.csv
python3 tools/kernel_summary.py $(ls $O/rocprof_train/*kernel_trace.csv | head -1) 14 > $O/kernel_summary_train.txt
rm -f $O/rocprof_train/*kernel_trace.csv
ls -la $O
