export FWN_LIB=tf-flowavenet_amd/csrc/libfwn_tune.so
echo "== library defaults (product lib)"; FWN_LIB= python tools/probe/cond_bench.py 8
for t in 0 1 2 3; do echo "== no split, tile $t"; FWN_COND_TILE=$t python tools/probe/cond_bench.py 8 1,1,1,1 12; done
for t in 0 1 2; do for ns in 2,2,2,2 2,2,4,4 2,3,5,5 2,2,4,8; do echo "== split tile $t nsplit $ns"; FWN_COND_SPLIT_TILE=$t python tools/probe/cond_bench.py 8 $ns 12; done; done
