#!/usr/bin/env python3
"""Static check: no packed fp32 arithmetic whose LOW half takes src0.lo and src1.HI (`op_sel:[0,1...]`) in a kernel's gfx950 ISA.

Why (measured, tools/probe/slp_coresidency.hip, profiles/r06_slp_coresidency.txt): on MI355X `v_pk_mul_f32`, `v_pk_fma_f32`
and `v_pk_add_f32` with `op_sel:[0,1]` / `op_sel:[0,1,0]` return, in lanes 48-63 only, a low half computed as if src1's high
dword were +0 - but only while waves of ANOTHER kernel issue bf16 MFMAs on the same CU (either MFMA shape; not beside fp32
MFMAs, LDS-DMA alone, plain VALU + LDS work or a second instance of the same kernel; never when the kernel runs alone).  The
other swizzles (`op_sel:[1,0]`, `op_sel_hi:[1,0]` = what hipcc emits for pair * scalar, `op_sel_hi:[0,1]`, both halves swapped,
the addend's `op_sel:[0,0,1]`), the unswizzled forms, `v_pk_mov_b32` and `v_pk_mul_f16` with the same selects are clean.
hipcc's SLP vectoriser emits exactly the failing form for a broadcast operand (`acc0 += y0 * w; acc1 += y1 * w`): that was the
round-2 front_valu_kernel flake and the 92-98 % failure rate of the round-5 `make slp_repro` build.  No hazard inside one wave
is involved (one instruction, operands long written), so no s_nop or wait repairs it; the product is built with
-fno-slp-vectorize, and this check is what keeps hand-written f32x2 code and future compiler versions from reintroducing the form.

usage: check_packed_f32.py file.s [...]      exit code 1 if any kernel carries the form
"""
import re
import sys

PACKED = re.compile(r"^\s*(v_pk_(?:mul|add|fma)_f32)\b(.*)$")
OP_SEL = re.compile(r"\bop_sel:\[([01]),([01])")


def kernels(path):
    """(name, [instruction lines]) for every kernel of an `hipcc -S --cuda-device-only` file."""
    name, body = None, []
    for line in open(path):
        m = re.match(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$", line)
        if m and not m.group(1).startswith(".L") and name is None:
            name, body = m.group(1), []
            continue
        if name is not None:
            body.append(line)
            if line.strip().startswith("s_endpgm"):
                yield name, body
                name = None


def check(name, body):
    """[(line number in the body, instruction text)] for every packed fp32 mul / add / fma whose op_sel starts [0,1."""
    out = []
    for i, line in enumerate(body):
        m = PACKED.match(line)
        if not m:
            continue
        s = OP_SEL.search(m.group(2))
        if s and s.group(1) == "0" and s.group(2) == "1":
            out.append((i, line.strip()))
    return out


def main(argv):
    bad = 0
    for path in argv[1:]:
        n = 0
        for name, body in kernels(path):
            n += 1
            for i, text in check(name, body):
                bad += 1
                print("%s: %s: +%d: %s" % (path, name, i, text))
        print("%s: %d kernels checked" % (path, n))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
