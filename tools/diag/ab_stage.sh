# GPU box: quick parity subset, then per-stage A/B of libfwn_base.so (the committed sources) against libfwn.so
cd "$(dirname "$0")/../.."
O=gpurun_out/ab_stage
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_fp8.py -x -q -m gpu -k "single_flow or gate_stage or golden or fp8 or real_sizes" > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
python3 tools/stage_bench.py --libs tf-flowavenet_amd/csrc/libfwn_base.so,tf-flowavenet_amd/csrc/libfwn.so --rounds 2 "$@" | tee $O/stage.txt
