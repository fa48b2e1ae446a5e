# GPU box: ablations of the fused tail (tail_chain.h FWN_TABL): builds libfwn_tabl{1,2,3}.so on the box and times the stages
cd "$(dirname "$0")/../.."
C=tf-flowavenet_amd/csrc
for n in 1 2 3; do
  mkdir -p /tmp/tabl$n
  for f in api flow_kernels aux_kernels train_kernels train_api; do
    if [ $f = flow_kernels ]; then /opt/rocm/bin/hipcc -DFWN_TABL=$n -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off -fno-slp-vectorize -c $C/$f.hip -o /tmp/tabl$n/$f.o || exit 1
    else cp $C/$f.o /tmp/tabl$n/$f.o; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libfwn_tabl$n.so /tmp/tabl$n/*.o
done
python3 tools/stage_bench.py --libs $C/libfwn.so,$C/libfwn_tabl1.so,$C/libfwn_tabl2.so,$C/libfwn_tabl3.so --rounds 2 --blocks ${1:-0,1,2}
