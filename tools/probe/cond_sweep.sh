# sweep of the hoisted conditioning projection at the bench shapes (tuning build: make -C tf-flowavenet_amd/csrc tune)
export FWN_LIB=tf-flowavenet_amd/csrc/libfwn_tune.so
echo "== library defaults (product lib)"; FWN_LIB= python tools/probe/cond_bench.py 8
for mt in 4 3 2; do for ns in 1,1,1,1 1,1,2,4 1,1,2,5 1,2,3,6; do echo "== streamed: tile 32 x $mt rows, nsplit $ns"; FWN_CRS_MT=$mt python tools/probe/cond_bench.py 8 0,0,0,0 12 $ns; done; done
