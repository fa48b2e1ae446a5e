"""Per-(block, stage) table of ONE-STREAM passes from a rocprofv3 kernel trace of `bench.py --serial`.

    python tools/pass_table.py <kernel_trace.csv> [--batch 8] [--samples 16128] [--json out.json]

A pass is found by its launch order on the one stream (the trace sorted by start time): `upsample[8]_kernel` x n_up opens a
pass, `prior_kernel` (forward) or `merge_kernel` (inverse) closes it.  Inside a pass every flow CLOSES with the launch that
holds its coupling (`tail_kernel`, or the `TailZeroProb` ring GEMM), 6 flows make a block; `cond_batch_kernel` /
`cond_reduce_kernel` launches belong to the block that follows them.  (A chained flow has no front launch of its own: the
previous flow's tail computed its h0.)  Per block the table gives the average kernel time per pass and stage
(front / gate / res / tail / cond), the algorithmic FLOP of the block (SURVEY 8d formula) and the fraction of the dense
bf16 MFMA peak the block ran at, and the same per pass.  Kernel time only: launch gaps are not in it.
"""
import argparse
import collections
import csv
import json
import sys

MFMA_PEAK = 2.5e15
N_FLOW, N_LAYER, N_BLOCK, HALF_MELS = 6, 2, 8, 40


def block_flop(i, samples):
    """FLOP of block i for `samples` audio samples, one direction (dead last res conv excluded)."""
    c = 2 << i
    cin = HALF_MELS * c
    mac = 3 * (c // 2) * 256 + N_LAYER * (2 * 3 * 256 * 256 + 2 * cin * 256) + (2 * N_LAYER - 1) * 256 * 256 + 256 * 256 + 256 * c
    return 2.0 * mac * N_FLOW / c * samples


def stage_of(name):
    if name.startswith("front_valu") or name.startswith("front_mfma") or name.startswith("xprep") or "FrontRingProb" in name or "FrontProb" in name:
        return "front"
    if name.startswith("gate_halo") or name.startswith("gate_rs") or "GateProb" in name:
        return "gate"
    if "ResProb" in name:
        return "res"
    if name.startswith("tail_kernel") or name.startswith("tail_rs_kernel") or "TailLinProb" in name or "TailZeroProb" in name:
        return "tail"
    if name.startswith("cond_"):
        return "cond"
    if name.startswith("flow_persist_kernel"):
        return "flow"                    # round 5: one launch per flow of the small-M chain (csrc/flow_persist.h)
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--samples", type=int, default=16128)
    ap.add_argument("--json")
    a = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(a.trace)):
        name = r["Kernel_Name"].replace("void ", "")
        rows.append((float(r["Start_Timestamp"]), float(r["End_Timestamp"]), name))
    rows.sort()
    # cut into passes
    passes, cur, direction = [], None, None
    for s, e, name in rows:
        base = name.split("(")[0]
        if (base.startswith("upsample_kernel") or base.startswith("upsample8_kernel")):
            if cur is None or any(not n.startswith("upsample") for _, _, n in cur):
                cur = []
            cur.append((s, e, base))
            continue
        if cur is None:
            continue
        cur.append((s, e, name))
        if base.startswith("prior_kernel") or base.startswith("merge_kernel"):
            passes.append(("fwd" if base.startswith("prior_kernel") else "inv", cur))
            cur = None
    stages = ("front", "gate", "res", "tail", "cond", "flow", "other")
    out = {}
    for direction in ("fwd", "inv"):
        sel = [p for d, p in passes if d == direction]
        # only whole-model passes without the data-dependent init: 48 coupling launches, no ddi kernel
        closes = lambda n: n.startswith("tail_kernel") or n.startswith("tail_rs_kernel") or "TailZeroProb" in n or n.startswith("flow_persist_kernel")
        good = []
        for p in sel:
            if sum(1 for _, _, n in p if closes(n)) == N_BLOCK * N_FLOW and not any(n.startswith("ddi_") for _, _, n in p):
                good.append(p)
        if not good:
            continue
        acc = collections.defaultdict(float)      # (block, stage) -> ns
        nl = collections.defaultdict(int)
        span = 0.0
        for p in good:
            span += p[-1][1] - p[0][0]
            flow = 0
            pending = []                          # cond launches ahead of a block's first flow
            for s, e, n in p:
                st = stage_of(n)
                if st == "cond":
                    pending.append(e - s)
                    continue
                if st == "other" or flow >= N_BLOCK * N_FLOW:
                    acc[("pre/post", "other")] += e - s
                    nl[("pre/post", "other")] += 1
                    continue
                blk = flow // N_FLOW if direction == "fwd" else N_BLOCK - 1 - flow // N_FLOW
                if pending:
                    acc[(blk, "cond")] += sum(pending)
                    nl[(blk, "cond")] += len(pending)
                    pending = []
                acc[(blk, st)] += e - s
                nl[(blk, st)] += 1
                if closes(n):
                    flow += 1
        npass = len(good)
        samples = a.batch * a.samples
        table = []
        tot_ns = tot_flop = 0.0
        for blk in range(N_BLOCK):
            per = {st: acc[(blk, st)] / npass / 1e3 for st in stages}           # us per pass
            us = sum(per.values())
            fl = block_flop(blk, samples)
            tot_ns += us * 1e3
            tot_flop += fl
            table.append({"block": blk, "rows": samples >> (blk + 1), "us": us, "gflop": fl / 1e9, "frac": fl / (us * 1e-6) / MFMA_PEAK,
                          "launches": sum(nl[(blk, st)] for st in stages) // npass, **{st + "_us": per[st] for st in stages if per[st] > 0}})
        other_us = acc[("pre/post", "other")] / npass / 1e3
        kernel_ms = (tot_ns / 1e3 + other_us) / 1e3
        out[direction] = {"passes": npass, "kernel_ms_per_pass": kernel_ms, "span_ms_per_pass": span / npass / 1e6,
                          "gflop_per_pass": tot_flop / 1e9, "frac_of_mfma_peak_kernel_time": tot_flop / (kernel_ms * 1e-3) / MFMA_PEAK,
                          "frac_of_mfma_peak_span": tot_flop / (span / npass * 1e-9) / MFMA_PEAK,
                          "pre_post_us": other_us, "blocks": table}
        print("%s: %d passes, kernel time %.3f ms/pass (first launch -> last end: %.3f ms), %.1f GFLOP -> %.3f of the 2.5 PF bf16 peak (span: %.3f)"
              % (direction, npass, kernel_ms, span / npass / 1e6, tot_flop / 1e9, out[direction]["frac_of_mfma_peak_kernel_time"],
                 out[direction]["frac_of_mfma_peak_span"]))
        print("  block   rows  launches      us   GFLOP   frac |  front    gate     res    tail    cond    flow (one launch per flow)")
        for t in table:
            print("  %5d %6d %9d %7.1f %7.1f %6.3f | %6.1f %7.1f %7.1f %7.1f %7.1f %7.1f" % (
                t["block"], t["rows"], t["launches"], t["us"], t["gflop"], t["frac"], t.get("front_us", 0), t.get("gate_us", 0),
                t.get("res_us", 0), t.get("tail_us", 0), t.get("cond_us", 0), t.get("flow_us", 0)))
        print("  upsample / split / merge / prior / copies: %.1f us" % other_us)
    if a.json:
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        out["source_sha"] = bench.kernel_source_hash()          # bench.py quotes the table only for these kernel sources
        out["workload"] = "B=%d, T=%d, one-stream pass (bench.py --serial), rocprofv3 --kernel-trace" % (a.batch, a.samples)
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)
    if not out:
        print("no whole passes found in the trace", file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
