# GPU box: parity tests, then single-pass times of libfwn_base.so (the committed sources) against libfwn.so, interleaved
cd "$(dirname "$0")/../.."
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_fp8.py -x -q -m gpu 2>&1 | tail -2
bash tools/diag/ab_env.sh "FWN_LIB=tf-flowavenet_amd/csrc/libfwn_base.so" "FWN_LIB=tf-flowavenet_amd/csrc/libfwn.so"
