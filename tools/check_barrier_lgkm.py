#!/usr/bin/env python3
"""Static check over gfx950 ISA (hipcc -S --cuda-device-only): barriers crossed with LDS reads outstanding.

    python tools/check_barrier_lgkm.py file.s [...]

Why: an s_barrier orders nothing by itself.  hipcc is free to sink the consumers of a ds_read (MFMAs are register-only
instructions) and with them the s_waitcnt lgkmcnt that retires it BELOW a raw __builtin_amdgcn_s_barrier().  If that
barrier is the one that licenses overwriting the LDS bytes being read (a ring slot refilled by LDS-DMA right behind it),
the refill races with the read still in flight: the round-3 front_mfma_kernel finding - wrong
weights in about one overlapped step in 50, only when another kernel shared the CU and slowed the LDS.
The check walks every kernel in program order with a counter of LDS operations in flight (ds_read / ds_write: +1;
s_waitcnt lgkmcnt(n): min(count, n); scalar loads also count in lgkmcnt and are included) and reports every s_barrier
reached with reads in flight, together with the next LDS write (LDS-DMA `... lds` or ds_write) behind it - the candidate
race.  A report is a prompt to read the code, not a verdict: a barrier that only publishes NEW data may be crossed with
reads of OTHER data in flight.
"""
import re
import sys

LGKM = re.compile(r"lgkmcnt\((\d+)\)")


def kernels(path):
    name, body = None, []
    for ln in open(path):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            if name:
                yield name, body
            name, body = m.group(1), []
        elif name is not None:
            body.append(ln)
            if "s_endpgm" in ln and ".Lfunc_end" in "".join(body[-3:]):
                pass
    if name:
        yield name, body


def check(name, body):
    out = []
    reads = 0          # ds_read* in flight (upper bound)
    pending = []       # (line no, text) of those reads
    for i, ln in enumerate(body):
        t = ln.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        op = t.split()[0]
        if op.startswith("ds_read") or op.startswith("ds_load"):
            reads += 1
            pending.append((i, t))
        elif op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("ds_write") or op.startswith("ds_store"):
            reads += 0   # counted by the hardware in lgkmcnt as well, but they are not reads of ring data
        elif op == "s_waitcnt":
            m = LGKM.search(t)
            if m:
                n = int(m.group(1))
                if n < reads:
                    pending = pending[len(pending) - n:] if n else []
                    reads = n
        elif op == "s_barrier":
            if reads > 0:
                nxt = None
                for j in range(i + 1, min(i + 400, len(body))):
                    u = body[j].strip()
                    if (" lds" in u and u.startswith("buffer_load")) or u.startswith("global_load_lds") or u.startswith("ds_write") or u.startswith("ds_store"):
                        nxt = (j, u)
                        break
                    if u.startswith("s_barrier"):
                        break
                out.append((i, reads, pending[-1], nxt))
    return out


def main():
    bad = 0
    for path in sys.argv[1:]:
        for name, body in kernels(path):
            rep = check(name, body)
            if rep:
                print("%s\n  %s: %d barrier(s) crossed with LDS reads in flight" % (path, name, len(rep)))
                for i, reads, last, nxt in rep[:4]:
                    print("    line +%d: %d read(s) in flight (last: %s); next LDS write behind it: %s"
                          % (i, reads, last[1][:60], ("+%d %s" % (nxt[0], nxt[1][:60])) if nxt else "none before the next barrier"))
                bad += len(rep)
    print("%d barrier(s) flagged" % bad)


if __name__ == "__main__":
    main()
