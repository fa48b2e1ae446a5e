#!/usr/bin/env python3
"""Static check over gfx950 ISA: registers touched while an inline-asm load into them is still in flight.

    python tools/check_async_loads.py file.s [kernel-name-substring]

gate_rs.h / gate_co.h stream weight fragments to registers with `buffer_load_dwordx4` issued from inline asm and wait
for them with hand-counted `s_waitcnt vmcnt(N)` (also inline asm).  hipcc believes an asm statement's output register is
written when the statement executes; it is free to copy, spill or reuse that register before the counted wait - code that
then reads the OLD contents (or whose result the late load overwrites).  The check walks a kernel in program order with the
queue of vector-memory operations (VGPR loads, LDS-DMA, stores: they retire in issue order, `s_waitcnt vmcnt(N)` leaves the N
youngest in flight) and reports every instruction that reads or writes a VGPR whose load has not been waited for."""
import re
import sys

VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
VMCNT = re.compile(r"vmcnt\((\d+)\)")


def regs_of(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


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
    if name:
        yield name, body


def is_vm(op):
    return op.startswith(("buffer_load", "buffer_store", "global_load", "global_store", "buffer_atomic", "global_atomic", "flat_", "scratch_load",
                          "scratch_store"))


BRANCHES = ("s_cbranch_execnz", "s_cbranch_execz", "s_cbranch_scc1", "s_cbranch_scc0", "s_cbranch_vccnz", "s_cbranch_vccz")


def decode(body):
    """One record per line: (kind, payload).  kinds: 'skip', 'label', 'wait' n, 'end', 'jump' target, 'cond' target,
    'vm' (touched, dst, text), 'op' (touched, text)."""
    labels = {}
    for i, ln in enumerate(body):
        m = re.match(r"^(\.LBB\w+):", ln)
        if m:
            labels[m.group(1)] = i
    recs = []
    for ln in body:
        t = ln.strip()
        if not t or t.startswith(";") or (t.startswith(".") and not re.match(r"^\.LBB\w+:", t)):
            recs.append(("skip", None))
            continue
        if re.match(r"^\.LBB\w+:", t) or t.endswith(":"):
            recs.append(("label", None))
            continue
        op = t.split()[0]
        rest = t.split(None, 1)[1] if " " in t else ""
        if op == "s_waitcnt":
            m = VMCNT.search(t)
            recs.append(("wait", int(m.group(1))) if m else ("skip", None))
        elif op == "s_endpgm":
            recs.append(("end", None))
        elif op == "s_branch":
            recs.append(("jump", labels.get(rest.split()[0], -1)))
        elif op in BRANCHES:
            recs.append(("cond", labels.get(rest.split()[0], -1)))
        elif is_vm(op):
            dst = frozenset(regs_of(rest.split(",")[0])) if (op.startswith(("buffer_load", "global_load", "scratch_load", "flat_load")) and " lds" not in t) else frozenset()
            recs.append(("vm", (frozenset(regs_of(rest)), dst, t)))
        else:
            touched = frozenset(regs_of(rest))
            recs.append(("op", (touched, t)) if touched else ("skip", None))
    return recs


class WalkLimit(RuntimeError):
    """The walk ran into max_steps: nothing may be concluded about the kernel (never report it clean)."""


def check(name, body, max_steps=20000000):
    """Walk the control-flow graph (not the listing: hipcc lays a loop's latch out in front of its header): both arms of
    every conditional branch; a state = (program counter at a label, the QUEUE of operations in flight - the destination
    registers of every entry in issue order from the oldest register still in flight on, entries without a destination
    (LDS-DMA, stores) behind it included: a later
    `s_waitcnt vmcnt(N)` retires by position, so two visits of a label with the same registers in flight but queues of
    different depth are different states - round 4's key (registers only) pruned a loop's second iteration whose counted
    waits reached less far than the first's); a state seen before is not walked again.  Running into max_steps raises."""
    recs = decode(body)
    reports, seen_rep, seen = [], set(), set()
    work = [(0, ())]
    steps = 0
    while work and steps < max_steps:
        pc, q0 = work.pop()
        queue = list(q0)                         # (line, dst regs) in issue order
        inflight = {}
        for idx, regs in queue:
            for r in regs:
                inflight[r] = idx
        while pc < len(recs) and steps < max_steps:
            kind, pay = recs[pc]
            pc += 1
            steps += 1
            if kind == "skip":
                continue
            if kind == "label":
                # what matters about the queue: which registers are in flight and how many operations are younger than each
                # (a later `s_waitcnt vmcnt(N)` retires by position); the counter saturates at 63, so counts are capped -
                # a loop that issues stores beside a load it never waits for then reaches a fixed point
                while len(queue) > 80 and not any(inflight.get(r) == queue[0][0] for r in queue[0][1]):
                    del queue[0]                     # register-free entries that no vmcnt(N <= 63) distinguishes any more
                n = len(queue)
                key = (pc, tuple((frozenset(r for r in regs if inflight.get(r) == idx), min(n - 1 - i, 64))
                                 for i, (idx, regs) in enumerate(queue) if any(inflight.get(r) == idx for r in regs)))
                if key in seen:
                    break
                seen.add(key)
            elif kind == "wait":
                if pay < len(queue):
                    for idx, regs in queue[:len(queue) - pay]:
                        for r in regs:
                            if inflight.get(r) == idx:
                                del inflight[r]
                    del queue[:len(queue) - pay]
            elif kind == "end":
                break
            elif kind == "jump":
                if pay < 0:
                    break
                pc = pay
            elif kind == "cond":
                if pay >= 0:
                    work.append((pay, tuple(queue)))
            elif kind == "vm":
                touched, dst, t = pay
                hit = sorted(r for r in touched if r in inflight)
                if hit and pc not in seen_rep:
                    seen_rep.add(pc)
                    reports.append((pc, t, hit, (inflight[hit[0]], body[inflight[hit[0]] - 1].strip())))
                queue.append((pc, dst))
                for r in dst:
                    inflight[r] = pc
            else:
                touched, t = pay
                hit = [r for r in touched if r in inflight]
                if hit and pc not in seen_rep:
                    seen_rep.add(pc)
                    hit.sort()
                    reports.append((pc, t, hit, (inflight[hit[0]], body[inflight[hit[0]] - 1].strip())))
    if steps >= max_steps:
        raise WalkLimit("%s: control-flow walk stopped after %d steps" % (name, steps))
    reports.sort()
    return reports


if __name__ == "__main__":
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, body in kernels(sys.argv[1]):
        if want not in name:
            continue
        rep = check(name, body)
        print("%s: %d instruction(s) touch a register with a load in flight" % (name, len(rep)))
        for ln, t, hit, (lidx, ltxt) in rep[:12]:
            print("   line %d: %s   <- v%s of line %d: %s" % (ln, t[:90], hit[:4], lidx, ltxt[:70]))
