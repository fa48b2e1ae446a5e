# GPU box: the whole -m gpu suite, then the default bench line (and optionally the serial one)
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-check}
mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python3 -c "
import json,sys
d=json.load(open('$O/bench.json'))
print('step %.3f ms  value %.2f M/s  fwd %.3f inv %.3f  gate0 %.2f us  train %s  rtf10 %s  fp8 fwd %s' % (d['ms_per_step'], d['value']/1e6, d['fwd_ms'], d['inv_ms'], d['roofline']['launch_us'], d['train'] and d['train'].get('ms_per_step'), d['rtf_10s'] and d['rtf_10s'].get('inverse_ms'), d['fp8'] and d['fp8'].get('fwd_ms')))
"
