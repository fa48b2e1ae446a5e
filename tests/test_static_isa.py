"""Static checks over the gfx950 ISA of the kernel sources (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tf-flowavenet_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


_ISA = {}
# (unit, extra flags): every kernel translation unit of libfwn.so, and the diagnostic -DFWN_PS_STAMP build of the one-launch flow
# (flow_persist.hip - weights, P tile and plane tiles by LDS-DMA in flight across a barrier, parks overlaying the
# weight regions, ds_reads kept three k-steps ahead - was the one unit the checks skipped)
# round 6: gate_rs.hip (the register-streamed gate left flow_kernels.hip) and tail_rs.hip (the register-streamed tail: LDS-DMA
# AND weight loads from inline asm, every wait hand-counted); cond_rs.hip (the register-streamed conditioning projection: the same, in a
# run-time loop over pairs of items)
UNITS = (("flow_kernels", ()), ("gate_rs", ()), ("tail_rs", ()), ("cond_rs", ()), ("train_kernels", ()), ("aux_kernels", ()), ("flow_persist", ()),
         ("flow_persist", ("-DFWN_PS_STAMP",)))


def _makefile_flags():
    """The flags libfwn.so is built with, read from the Makefile itself (a hand-copied list would drift: the checks below
    depend on exact code generation)."""
    import re
    text = open(os.path.join(CSRC, "Makefile")).read()
    arch = re.search(r"^ARCH\s*\?=\s*(\S+)", text, re.M).group(1)
    flags = re.search(r"^CXXFLAGS\s*=\s*(.*)$", text, re.M).group(1)
    flags = flags.replace("$(EXTRA)", "").replace("$(ARCH)", arch).split()
    assert "--offload-arch=gfx950" in flags and "-O3" in flags, flags
    return flags


def _isa(unit, tmp_path_factory):
    """(<unit>, extra flags) -> gfx950 ISA, once per session.  Every unit is compiled on the first call, four at a time (the two
    big ones - flow_kernels.hip, gate_rs.hip - take two minutes of hipcc each)."""
    if not _ISA:
        from concurrent.futures import ThreadPoolExecutor
        flags = _makefile_flags()
        root = tmp_path_factory.mktemp("isa")

        def build(u):
            name, extra = u
            out = root / ("%s%s.s" % (name, "_" + "".join(c for c in "".join(extra) if c.isalnum()) if extra else ""))
            subprocess.run([HIPCC] + flags + list(extra) + ["-S", "--cuda-device-only", "-c", os.path.join(CSRC, name + ".hip"), "-o", str(out)],
                           check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            assert os.path.getsize(str(out)) > 0, out
            return u, str(out)

        with ThreadPoolExecutor(max_workers=4) as ex:
            for u, path in ex.map(build, UNITS):
                _ISA[u] = path
    return _ISA[unit]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_no_register_is_touched_while_an_asm_load_into_it_is_in_flight(tmp_path_factory):
    """tools/check_async_loads.py over the inference kernels: the register-streamed gates (gate_rs.h, gate_co.h) load weight
    fragments from inline asm and wait for them with hand-counted s_waitcnt vmcnt(N); hipcc believes an asm output written
    when the statement executes and may copy or reuse the register before the wait.  That was the persistent gate's race
    (round 4: `if (first) wait<n0> else wait<n1>` made hipcc fill the merged "+v" operand with a v_mov IN FRONT of the wait -
    a copy of a ring stage whose load was still in flight).  No kernel may touch a register with a load in flight; and the
    checker must still see the pattern."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_async_loads as chk
    for unit in UNITS:       # round 5: the training and auxiliary kernels too (train_kernels.hip runs the same LDS-DMA rings)
        ks = list(chk.kernels(_isa(unit, tmp_path_factory)))
        assert ks, ("no kernel found in the ISA of", unit)
        flagged = [(name, rep[:2]) for name, body in ks for rep in [chk.check(name, body)] if rep]
        assert not flagged, (unit, flagged[:3])
    body = ["\tbuffer_load_dwordx4 v[0:3], v9, s[0:3], s4 offen offset:0\n", "\tbuffer_load_dwordx4 v[4:7], v9, s[0:3], s4 offen offset:1024\n",
            "\ts_cbranch_vccz .LBB0_1\n", "\tv_mov_b64_e32 v[10:11], v[0:1]\n", "\ts_waitcnt vmcnt(1)\n", ".LBB0_1:\n",
            "\ts_waitcnt vmcnt(1)\n", "\tv_add_u32_e32 v8, v1, v2\n", "\ts_endpgm\n"]
    rep = chk.check("k", body)
    assert len(rep) == 1 and rep[0][1].startswith("v_mov_b64") and rep[0][2] == [0, 1]
    del body[3]
    assert chk.check("k", body) == []
    # a loop whose SECOND iteration is the wrong one: the same registers are in flight at the label both times,
    # but behind one store less - the counted wait of the second visit no longer reaches the load
    loop = ["\tbuffer_load_dwordx4 v[0:3], v9, s[0:3], 0 offen\n", "\tbuffer_store_dword v20, v9, s[0:3], 0 offen\n",
            "\tbuffer_store_dword v20, v9, s[0:3], 0 offen offset:4\n", ".LBB0_2:\n", "\ts_waitcnt vmcnt(2)\n",
            "\tv_add_u32_e32 v8, v0, v1\n", "\tbuffer_load_dwordx4 v[0:3], v9, s[0:3], 0 offen\n",
            "\tbuffer_store_dword v20, v9, s[0:3], 0 offen\n", "\ts_cbranch_scc1 .LBB0_2\n", "\ts_endpgm\n"]
    rep = chk.check("k", loop)
    assert len(rep) == 1 and rep[0][1].startswith("v_add_u32") and rep[0][2] == [0, 1]
    with pytest.raises(chk.WalkLimit):           # running out of steps is an error, never "clean"
        chk.check("k", loop, max_steps=5)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_no_ring_barrier_is_crossed_with_lds_reads_in_flight(tmp_path, tmp_path_factory):
    """tools/check_barrier_lgkm.py over the inference kernels: no s_barrier is reached with ds_reads in flight (the
    round-3 front_mfma_kernel race, DESIGN.md section 3.5: hipcc sinks the lgkmcnt wait of a chunk's fragment reads below the
    raw barrier that licenses refilling the slot they read; FWN_RING_BARRIER retires them first).  The checker must also
    still SEE the pattern: a ring barrier without the wait is flagged."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_barrier_lgkm as chk
    for unit in UNITS:
        out = _isa(unit, tmp_path_factory)
        ks = list(chk.kernels(str(out)))
        assert ks, ("no kernel found in the ISA of", unit)
        flagged = [(name, rep) for name, body in ks for rep in [chk.check(name, body)] if rep]
        assert not flagged, (unit, flagged[:3])
    # the checker on a hand-made body: reads in flight at the barrier, a refill behind it
    body = ["\tds_read_b128 v[0:3], v4\n", "\ts_waitcnt vmcnt(4)\n", "\ts_barrier\n",
            "\tbuffer_load_dwordx4 v5, s[0:3], 0 offen lds\n", "\ts_waitcnt lgkmcnt(0)\n"]
    rep = chk.check("k", body)
    assert len(rep) == 1 and rep[0][1] == 1 and rep[0][3] is not None
    body.insert(1, "\ts_waitcnt lgkmcnt(0)\n")
    assert chk.check("k", body) == []


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_no_packed_fp32_arithmetic_takes_its_low_half_from_src1s_high_dword(tmp_path_factory):
    """tools/check_packed_f32.py over every unit: no v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 with `op_sel:[0,1...]`.
    Round 6 root-caused the SLP co-residency failure (tools/probe/slp_coresidency.hip,
    profiles/r06_slp_coresidency.txt): with that swizzle lanes 48-63 compute the low half as if src1's high dword were 0 while
    waves of another kernel issue bf16 MFMAs on the same CU.  hipcc's SLP vectoriser emits the form for a broadcast operand
    (front_valu_kernel: 72 instructions in the vectorised build, none anywhere else), the build has -fno-slp-vectorize, and this
    check covers hand-written f32x2 code and the next compiler.  The forms the product does carry (`op_sel_hi:[1,0]`, pair *
    scalar) measured clean.  The checker must still see the failing form, and must not flag the clean ones."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_packed_f32 as chk
    assert "-fno-slp-vectorize" in _makefile_flags()
    for unit in UNITS:
        ks = list(chk.kernels(_isa(unit, tmp_path_factory)))
        assert ks, ("no kernel found in the ISA of", unit)
        flagged = [(name, rep[:2]) for name, body in ks for rep in [chk.check(name, body)] if rep]
        assert not flagged, (unit, flagged[:3])
    body = ["\tv_pk_mul_f32 v[6:7], v[8:9], v[10:11] op_sel:[0,1] op_sel_hi:[1,1]\n",      # the failing broadcast form
            "\tv_pk_fma_f32 v[0:1], v[2:3], v[10:11], v[0:1] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n",
            "\tv_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]\n",
            "\tv_pk_mul_f32 v[4:5], v[0:1], 0.5 op_sel_hi:[1,0]\n",                       # clean: pair * scalar
            "\tv_pk_mul_f32 v[4:5], v[0:1], v[2:3] op_sel:[1,0]\n",                        # clean: the commuted broadcast
            "\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,0,1]\n",              # clean: the addend swizzled
            "\tv_pk_add_f32 v[0:1], v[2:3], v[4:5]\n", "\tv_pk_mov_b32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]\n",
            "\tv_pk_mul_f16 v0, v1, v2 op_sel:[0,1]\n", "\ts_endpgm\n"]
    assert [i for i, _ in chk.check("k", body)] == [0, 1, 2]
