"""Developer A/B of dispatch thresholds on the GPU box: the -DFWN_TUNABLE build (make -C tf-flowavenet_amd/csrc tune)
reads FWN_* thresholds from the environment; every setting runs in its own process.

    python tools/tune.py "FWN_TAIL_SPLIT_MAX=0" "FWN_TAIL_SPLIT_MAX=12288 FWN_SMALL_D=4" ""
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, os
sys.path.insert(0, %r)
import torch
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd import weights as W
from tf_flowavenet_amd.model import FloWaveNet
hp = default_hparams()
# the model's kernel-selection arguments from the environment of THIS developer tool (the package itself reads none)
m = FloWaveNet(hp, init=True, chain_mode=int(os.environ.get("FWN_CHAIN_MODE", "0")),
               persist_mode=int(os.environ.get("FWN_PERSIST_MODE", "0"))).load_params(W.synthetic_params(hp, 1234))
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
out = []
first = True
for b, t in ((8, 16128), (1, 16128), (1, 220672)):
    inp = W.synthetic_inputs(hp, b, t)
    x, c, z = (torch.from_numpy(inp[k]).cuda() for k in ("x", "c", "z"))
    if first:
        m.forward(x, c); first = False
    out.append("B=%%d T=%%6d fwd %%7.3f inv %%7.3f ms" %% (b, t, timed(lambda: m.forward(x, c)), timed(lambda: m.reverse(z, c))))
print(" | ".join(out))
""" % ROOT
lib = os.path.join(ROOT, "tf-flowavenet_amd", "csrc", "libfwn_tune.so")
for setting in sys.argv[1:] or [""]:
    env = dict(os.environ, FWN_LIB=lib)
    env.update(kv.split("=") for kv in setting.split())
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print("%-50s %s" % (setting or "(defaults)", r.stdout.strip() or r.stderr[-600:]), flush=True)
