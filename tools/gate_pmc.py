"""GPU box: PMC counters of the dominant kernel (block-0 gated dilated layer, bf16 and fp8) -> profiles-ready files.

    python tools/gate_pmc.py <out_dir> [tag]

Runs `rocprofv3 --pmc <group> --kernel-trace` once per counter group (separate passes, never combined with other trace
domains: MI355X guide, HBM / rocprofv3 section) on a small driver that launches fwn_gate / fwn_gate_fp8 at the bench.py
shape, averages the counters per launch and writes <tag>_gate_pmc_raw.txt and <tag>_gate_traffic.json (HBM bytes =
2 x FETCH_SIZE + WRITE_SIZE KB: FETCH_SIZE counts half the bytes of 16-byte-per-lane streams on gfx950) with the hash of
the kernel sources bench.py checks before quoting the number."""
import collections, csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = r"""
import sys, ctypes as C
sys.path.insert(0, %r)
import torch
from tf_flowavenet_amd import _lib, weights as W
from tf_flowavenet_amd.hparams import default_hparams
from tf_flowavenet_amd.model import FloWaveNet
hp = default_hparams().replace(n_block=1, n_flow=2)
m8 = FloWaveNet(hp, gate_fp8=True).load_params(W.synthetic_params(hp, 1234))
lib = _lib.load()
d = m8._packed.flow_descs[0]
b, t = 8, 16128
ti = t // 2; m = b * ti
h = (torch.randn(m, 256, device="cuda") * 0.5).to(torch.bfloat16)
ca = torch.rand(m, d.cin, device="cuda").to(torch.bfloat16)
h8 = torch.empty(m, 256, dtype=torch.uint8, device="cuda")
o = torch.empty(m, 256, device="cuda", dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
lib.fwn_cast_e4m3(h.data_ptr(), h8.data_ptr(), m * 256, st)
for _ in range(12):
    _lib.check(lib.fwn_gate(C.byref(d), 0, h.data_ptr(), ca.data_ptr(), None, o.data_ptr(), m, ti, st))
    _lib.check(lib.fwn_gate_fp8(C.byref(d), 0, h8.data_ptr(), ca.data_ptr(), o.data_ptr(), m, ti, st))
torch.cuda.synchronize()
""" % ROOT
GROUPS = ["FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE",
          "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES", "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"]


def main():
    out_dir = sys.argv[1]
    tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
    os.makedirs(out_dir, exist_ok=True)
    drv = "/tmp/gate_pmc_driver.py"
    open(drv, "w").write(DRIVER)
    agg = collections.OrderedDict()
    for grp in GROUPS:
        d = "/tmp/gate_pmc_%s" % grp.split()[0]
        subprocess.run(["rm", "-rf", d])
        subprocess.run(["rocprofv3", "--pmc"] + grp.split() + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, drv],
                       cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        if not files:
            print("no counters for", grp)
            continue
        for r in csv.DictReader(open(files[0])):
            name = r["Kernel_Name"]
            if "gate_halo_kernel" not in name and "gate_rs_kernel" not in name:
                continue
            # bf16: the register-streamed kernel (gate_rs_kernel, round 4) at this shape; fp8: the tap-sharing tile's FP8 form
            key = "bf16" if "gate_rs_kernel" in name else ("fp8" if "true" in name.split("gate_halo_kernel")[1].split(">")[0] or "Lb1" in name else "bf16_halo")
            a = agg.setdefault(key, collections.defaultdict(lambda: [0.0, 0]))
            a[r["Counter_Name"]][0] += float(r["Counter_Value"])
            a[r["Counter_Name"]][1] += 1
    sys.path.insert(0, ROOT)
    import bench
    raw = []
    for key, cs in agg.items():
        means = {c: v / n for c, (v, n) in cs.items()}
        raw.append("%s %s (M = 64512, 504 workgroups), per launch: %s" % (key, "gate_rs_kernel<5>" if key == "bf16" else "gate_halo_kernel<256,256>", json.dumps({c: round(v) for c, v in means.items()})))
        if key == "bf16" and "FETCH_SIZE" in means and "WRITE_SIZE" in means:
            rec = {"kernel": "gate_rs_kernel<5>", "rows": 64512, "fetch_size_kb": means["FETCH_SIZE"],
                   "write_size_kb": means["WRITE_SIZE"], "traffic_bytes": int((2 * means["FETCH_SIZE"] + means["WRITE_SIZE"]) * 1024),
                   "source_sha": bench.gate_source_hash(),
                   "source": "tools/gate_pmc.py: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section"}
            json.dump(rec, open(os.path.join(out_dir, tag + "_gate_traffic.json"), "w"))
    open(os.path.join(out_dir, tag + "_gate_pmc_raw.txt"), "w").write("\n".join(raw) + "\n")
    print("\n".join(raw))


if __name__ == "__main__":
    main()
