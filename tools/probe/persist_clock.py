import sys, os, runpy
import numpy as np
# reuse the probe, then compute the clock from slots (1, 5) = s_memtime and (0, 6) = s_memrealtime
sys.argv = ["persist_stamps.py"] + sys.argv[1:]
g = runpy.run_path("tools/probe/persist_stamps.py")
raw = g["raw"]
dt_real = (raw[:, 6].astype(np.int64) - raw[:, 0].astype(np.int64)) / 100e6
dt_clk = (raw[:, 5].astype(np.int64) - raw[:, 1].astype(np.int64))
print("in-kernel shader clock per ticket: median %.2f GHz (min %.2f, max %.2f)" % (np.median(dt_clk / dt_real) / 1e9, (dt_clk / dt_real).min() / 1e9, (dt_clk / dt_real).max() / 1e9))
