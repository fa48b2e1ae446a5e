#!/bin/bash
# usage: tools/diag/build_rs.sh <tag> [hipcc flags...]   ->  tools/rs_<tag>_bin (+ ISA of the kernels under /tmp/rs/<tag>/)
tag=$1; shift
mkdir -p /tmp/rs/$tag && cd /tmp/rs/$tag && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize "$@" -save-temps -Rpass-analysis=kernel-resource-usage \
    /root/repo/tools/bench_gate_rs.hip -o rs_bin > log.txt 2>&1
grep -E "error" -A5 log.txt | head -30
cp rs_bin /root/repo/tools/rs_${tag}_bin
S=bench_gate_rs-hip-amdgcn-amd-amdhsa-gfx950.s
for k in $(grep -o "^_Z14gate_rs_kernel[A-Za-z0-9_]*:" $S | tr -d :); do
  awk "/^$k:/,/s_endpgm/" $S > $k.s
  echo "$tag $k: $(grep -A12 "Function Name: $k" log.txt | grep -E " VGPRs:| AGPRs:|ScratchSize|Occupancy" | sed 's/.*:0: *//' | sed 's/ \[-Rpass.*//' | tr '\n' ' ') lines $(wc -l < $k.s) mfma $(grep -c v_mfma $k.s) scratch-in-loop $(awk '/s_barrier/{b=1} b&&/scratch_/{n++} /vmcnt\(0\)/{if(b)exit} END{print n+0}' $k.s)"
done
