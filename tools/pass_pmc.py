"""GPU box: HBM traffic of a WHOLE forward / inverse pass from the rocprofv3 PMC counters FETCH_SIZE and WRITE_SIZE.

    python tools/pass_pmc.py <out_dir> [tag]          ->  <tag>_pass_traffic.json

Two separate `rocprofv3 --pmc <counter> --kernel-trace` passes (never combined with other trace domains: MI355X guide,
HBM / rocprofv3 section) over `bench.py --serial` at the bench workload; the dispatches of every whole pass (cut like
tools/pass_table.py: `upsample_kernel` .. `prior_kernel` / `merge_kernel`, 48 coupling launches, no ddi kernel) are summed
and averaged per pass.  HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB): FETCH_SIZE counts half the bytes of 16-byte-per-lane
streams on gfx950 (LDS-DMA and 16-byte loads carry nearly all of this path's reads; the guide calls other widths
uncalibrated, so read the figure as an upper estimate of the fetched bytes).  Compared with SURVEY section 8(d)'s
algorithmic bytes per pass (weights once + minimal activation traffic: 882.5 MB at B = 8, T = 16128).
"""
import csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALGO_BYTES = 882.5e6


def passes_of(rows):
    """rows: [(dispatch order key, kernel name, value)] -> {"fwd": [sum per pass], "inv": [...]}"""
    rows.sort()
    out = {"fwd": [], "inv": []}
    cur = None
    for _, name, val in rows:
        base = name.replace("void ", "").split("(")[0]
        if (base.startswith("upsample_kernel") or base.startswith("upsample8_kernel")):
            if cur is None or any(not n.startswith("upsample") for n, _ in cur):
                cur = []
            cur.append((base, val))
            continue
        if cur is None:
            continue
        cur.append((base, val))
        if base.startswith("prior_kernel") or base.startswith("merge_kernel"):
            closes = sum(1 for n, _ in cur if n.startswith("tail_kernel") or n.startswith("tail_rs_kernel") or "TailZeroProb" in n or n.startswith("flow_persist_kernel"))
            if closes == 48 and not any(n.startswith("ddi_") for n, _ in cur):
                out["fwd" if base.startswith("prior_kernel") else "inv"].append(sum(v for _, v in cur))
            cur = None
    return out


def main():
    out_dir = sys.argv[1]
    tag = sys.argv[2] if len(sys.argv) > 2 else "r03"
    os.makedirs(out_dir, exist_ok=True)
    res = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = "/tmp/pass_pmc_%s" % ctr
        subprocess.run(["rm", "-rf", d])
        subprocess.run(["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                        os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--serial", "--no-train", "--no-rtf", "--no-fp8", "--no-latency", "--steps", "3",
                        "--warmup", "1"], cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        if not files:
            print("no counters for", ctr)
            return 1
        rows = []
        for r in csv.DictReader(open(files[0])):
            if r["Counter_Name"] != ctr:
                continue
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
        ps = passes_of(rows)
        for k, v in ps.items():
            if v:
                res.setdefault(k, {})[ctr] = sum(v) / len(v)
                res[k]["passes"] = len(v)
        subprocess.run(["rm", "-rf", d])
    sys.path.insert(0, ROOT)
    import bench
    rec = {"workload": "configs[1]: B=8, T=16128, one-stream pass (bench.py --serial)", "algorithmic_bytes": ALGO_BYTES,
           "source_sha": bench.kernel_source_hash(),
           "source": "tools/pass_pmc.py: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, summed over the dispatches of a pass; "
                     "FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section"}
    for k, v in res.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            b = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
            rec[k] = {"fetch_size_kb": v["FETCH_SIZE"], "write_size_kb": v["WRITE_SIZE"], "traffic_bytes": int(b),
                      "ratio_to_algorithmic": b / ALGO_BYTES, "passes": v["passes"]}
    json.dump(rec, open(os.path.join(out_dir, tag + "_pass_traffic.json"), "w"), indent=1)
    print(json.dumps(rec, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
