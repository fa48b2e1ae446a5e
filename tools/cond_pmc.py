"""GPU box: PMC counters of the hoisted conditioning projection at the shapes of blocks 4 - 7 of the 8-clip pass - the
register-streamed kernel (csrc/cond_rs.h) and the ring tiles it replaces, the same operands.

    python tools/cond_pmc.py <out_dir> [tag]

One `rocprofv3 --pmc <group> --kernel-trace` pass per counter group (never combined with other trace domains: MI355X guide, HBM /
rocprofv3 section); counters averaged per launch -> <tag>_cond_pmc_raw.txt.  HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE KB (FETCH_SIZE
counts half the bytes of 16-byte-per-lane streams on gfx950)."""
import collections, csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = r"""
import sys
sys.path.insert(0, %r)
import torch
from tf_flowavenet_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
T, nb, nflow, L = 16128, 8, 6, 2
nz = nflow * L
for blk in (4, 5, 6, 7):
    m = nb * (T // (2 << blk)); cin = 40 * (2 << blk); kc = cin
    g = torch.Generator(device="cuda").manual_seed(blk)
    ca = (torch.rand(m, cin, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
    wc = (torch.rand(nz, 512, kc, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
    ws = torch.empty_like(wc)
    _lib.check(lib.fwn_pack_cond_stream(wc.data_ptr(), 512 * kc, kc, nz, ws.data_ptr(), st))
    p = torch.empty(nz, m, 512, device="cuda")
    ns_ring = lib.fwn_cond_splits(m, nz, kc); ns_rs = lib.fwn_cond_stream_splits(m, nz, kc)
    part = torch.empty(max(ns_ring, ns_rs, 2) - 1, nz, m, 512, device="cuda")
    for _ in range(6):
        _lib.check(lib.fwn_cond_split(ca.data_ptr(), wc.data_ptr(), p.data_ptr(), 512 * kc, m * 512, 0, 1, nflow, L, m, cin, kc, part.data_ptr(), nz * m * 512, ns_ring, st))
        _lib.check(lib.fwn_cond_stream(ca.data_ptr(), None, ws.data_ptr(), p.data_ptr(), nflow, L, m, cin, kc, part.data_ptr(), nz * m * 512, ns_rs, st))
torch.cuda.synchronize()
""" % ROOT
GROUPS = ["FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum TCC_MISS_sum", "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE",
          "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES", "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"]


def main():
    out_dir = sys.argv[1]
    tag = sys.argv[2] if len(sys.argv) > 2 else "r06"
    os.makedirs(out_dir, exist_ok=True)
    drv = "/tmp/cond_pmc_driver.py"
    open(drv, "w").write(DRIVER)
    agg = collections.OrderedDict()
    for grp in GROUPS:
        d = "/tmp/cond_pmc_%s" % grp.split()[0]
        subprocess.run(["rm", "-rf", d])
        subprocess.run(["rocprofv3", "--pmc"] + grp.split() + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, drv],
                       cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        if not files:
            print("no counters for", grp)
            continue
        for r in csv.DictReader(open(files[0])):
            name = r["Kernel_Name"]
            if "cond_rs_kernel" not in name and "cond_batch_kernel" not in name:
                continue
            grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
            key = (name.split("(")[0].replace("void ", "")[:44], grid)
            a = agg.setdefault(key, collections.defaultdict(lambda: [0.0, 0]))
            a[r["Counter_Name"]][0] += float(r["Counter_Value"])
            a[r["Counter_Name"]][1] += 1
    raw = []
    for (name, grid), cs in agg.items():
        means = {c: v / n for c, (v, n) in cs.items()}
        extra = ""
        if "FETCH_SIZE" in means and "WRITE_SIZE" in means:
            extra += "  HBM %.1f MB" % ((2 * means["FETCH_SIZE"] + means["WRITE_SIZE"]) / 1024)
        if means.get("SQ_WAVE_CYCLES"):
            extra += "  MFMA busy / wave cycles %.3f" % (means.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / means["SQ_WAVE_CYCLES"])
        if means.get("SQ_LDS_IDX_ACTIVE"):
            extra += "  LDS bank-conflict cycles / active %.3f" % (means.get("SQ_LDS_BANK_CONFLICT", 0) / means["SQ_LDS_IDX_ACTIVE"])
        raw.append("%-46s grid %-8s per launch: %s%s" % (name, grid, json.dumps({c: round(v) for c, v in means.items()}), extra))
    open(os.path.join(out_dir, tag + "_cond_pmc_raw.txt"), "w").write("\n".join(raw) + "\n")
    print("\n".join(raw))


if __name__ == "__main__":
    main()
