"""Diagnostic: the LAST training step of a rocprofv3 kernel trace (tools/bench_train.py) by block - forward span, backward
data-gradient chain span (coupling_bwd .. flow_small_grads_kernel of the block's flows) and the weight-gradient kernels.
usage: train_blocks.py <kernel_trace.csv> [n_flow=6]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
NF = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for r in rows:
    r["s"], r["e"], r["n"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0]
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if r["n"].startswith("wn_scale_jobs_kernel")]
step = rows[starts[-2]:starts[-1]]
t0 = step[0]["s"]
ms = lambda t: (t - t0) / 1e6
# forward: flows close with the kernel holding the coupling
closes = [i for i, r in enumerate(step) if r["n"].startswith(("tail_kernel", "tail_rs_kernel")) or "TailZeroProb" in r["n"]]
bwd0 = next(i for i, r in enumerate(step) if r["n"].startswith("coupling_bwd_kernel"))
closes = [i for i in closes if i < bwd0]
nflow = len(closes)
nb = nflow // NF
print("step %.3f ms, %d launches, %d flows" % (ms(step[-1]["e"]), len(step), nflow))
first_fwd = next(i for i, r in enumerate(step) if r["n"].startswith("upsample"))
print("refresh + pack: 0 .. %.3f ms" % ms(step[first_fwd]["s"]))
prev = first_fwd
print("forward:  block  start_ms  span_us  kernel_us  launches")
for b in range(nb):
    last = closes[(b + 1) * NF - 1]
    ks = step[prev:last + 1]
    print("          %5d  %8.3f  %7.1f  %9.1f  %8d" % (b, ms(ks[0]["s"]), (ks[-1]["e"] - ks[0]["s"]) / 1e3, sum(k["e"] - k["s"] for k in ks) / 1e3, len(ks)))
    prev = last + 1
# the second flow of every block, launch by launch
for b in range(nb):
    i0, i1 = closes[b * NF] + 1, closes[b * NF + 1]
    print("forward, block %d, flow 1:" % b)
    for r in step[i0:i1 + 1]:
        print("   %8.3f ms  %6.1f us  grid %6s  %s" % (ms(r["s"]), (r["e"] - r["s"]) / 1e3, r["Grid_Size_X"], r["n"][:80]))
cb = [i for i, r in enumerate(step) if r["n"].startswith("coupling_bwd_kernel")]
sg = [i for i, r in enumerate(step) if r["n"] == "flow_small_grads_kernel"]
side_names = ("tn_gemm_kernel", "tn_gemm_multi_kernel", "tn_table_put_kernel", "wn_group_kernel", "wn_group_mid_kernel", "flow_small_grads_final_kernel")
chain = [r for r in step[bwd0:] if not r["n"].startswith(side_names)]
print("backward: block  start_ms  chain_span_us  (per flow)   tn_us  wn_us  (kernel time of the block's weight-gradient launches)")
tn = [r for r in step if r["n"].startswith("tn_gemm_kernel")]
tnm = [r for r in step if r["n"].startswith("tn_gemm_multi_kernel")]      # round 4: ONE weight-gradient launch per block
wn = [r for r in step if r["n"].startswith("wn_group")]
ntn, nwn = len(tn) // nflow, len(wn) // nflow
for b in range(nb - 1, -1, -1):
    k = nb - 1 - b
    s, e = step[cb[k * NF]]["s"], step[sg[(k + 1) * NF - 1]]["e"]
    tnb = sum(r["e"] - r["s"] for r in tn[k * NF * ntn:(k + 1) * NF * ntn]) / 1e3
    if len(tnm) == nb:
        tnb += (tnm[k]["e"] - tnm[k]["s"]) / 1e3
    wnb = sum(r["e"] - r["s"] for r in wn[k * NF * nwn:(k + 1) * NF * nwn]) / 1e3
    print("          %5d  %8.3f  %13.1f  %10.1f  %6.1f %6.1f" % (b, ms(s), (e - s) / 1e3, (e - s) / 1e3 / NF, tnb, wnb))
last_sg = step[sg[-1]]["e"]
print("after the chain (joins, up-sampling backward, norm, Adam): %.3f .. %.3f ms" % (ms(last_sg), ms(step[-1]["e"])))
print("after the last chain (weight-gradient kernels of the last blocks left out):")
for r in step[sg[-1] + 1:]:
    if not r["n"].startswith(side_names):
        print("   %8.3f ms  %6.1f us  grid %8s  %s" % (ms(r["s"]), (r["e"] - r["s"]) / 1e3, r["Grid_Size_X"], r["n"][:70]))
print("before the forward pass:")
for r in step[:first_fwd]:
    print("   %8.3f ms  %6.1f us  grid %8s  %s" % (ms(r["s"]), (r["e"] - r["s"]) / 1e3, r["Grid_Size_X"], r["n"][:70]))
# one flow of block 0 and one of the last block: the chain's launches
for k, name in [(nflow - 1 - b * NF, "backward, block %d, last flow processed" % b) for b in range(nb)]:
    i0 = cb[k]
    i1 = sg[k]
    print(name + ":")
    for r in step[i0:i1 + 1]:
        if not r["n"].startswith(side_names):
            print("   %8.3f ms  %6.1f us  grid %6s  %s" % (ms(r["s"]), (r["e"] - r["s"]) / 1e3, r["Grid_Size_X"], r["n"][:70]))
