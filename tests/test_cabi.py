"""CPU tests of the C-ABI boundary: the library loads, exports every symbol include/fwn.h
declares, and validates arguments (error code + message) before any launch.  No compute."""
import ctypes as C
import os
import re

import pytest

from tf_flowavenet_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def header_functions():
    src = open(os.path.join(ROOT, "include", "fwn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fwn_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = header_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), "libfwn.so does not export %s" % n
        assert n in _lib.SIGNATURES, "ctypes binding missing for %s" % n
    assert sorted(_lib.SIGNATURES) == names


def test_version(lib):
    assert lib.fwn_version() == 321


def test_struct_layout_matches_header(tmp_path):
    """The ctypes mirrors against the C compiler's view of include/fwn.h: sizes and the offsets of the last members."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "fwn.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(fwn_flow_desc), '
                   'offsetof(fwn_flow_desc, an), offsetof(fwn_flow_desc, Wd8), offsetof(fwn_flow_desc, wd8_exp), sizeof(fwn_model_desc), '
                   'offsetof(fwn_model_desc, up_w), offsetof(fwn_model_desc, flows), offsetof(fwn_model_desc, gate_fp8), '
                   'sizeof(fwn_conv_grad), sizeof(fwn_flow_train_desc), offsetof(fwn_flow_train_desc, d_zscale), sizeof(fwn_train_desc), '
                   'offsetof(fwn_train_desc, an_logdet), offsetof(fwn_train_desc, side_stream), sizeof(fwn_gemm_desc), '
                   'offsetof(fwn_gemm_desc, gate_out), offsetof(fwn_flow_desc, Wfront3), offsetof(fwn_flow_desc, kf3), '
                   'offsetof(fwn_model_desc, chain_mode)); return 0; }\n')
    exe = str(tmp_path / "layout")
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe], check=True)
    got = [int(v) for v in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    F, M, FT, T = _lib.FlowDesc, _lib.ModelDesc, _lib.FlowTrainDesc, _lib.TrainDesc
    assert got == [C.sizeof(F), F.an.offset, F.Wd8.offset, F.wd8_exp.offset, C.sizeof(M), M.up_w.offset, M.flows.offset,
                   M.gate_fp8.offset, C.sizeof(_lib.ConvGrad), C.sizeof(FT), FT.d_zscale.offset, C.sizeof(T), T.an_logdet.offset,
                   T.side_stream.offset, C.sizeof(_lib.GemmDesc), _lib.GemmDesc.gate_out.offset, F.Wfront3.offset, F.kf3.offset,
                   M.chain_mode.offset]


def test_argument_validation_reports_errors(lib):
    assert lib.fwn_split_planes(None, 1, 8, None, None) == -1
    assert b"fwn_split_planes" in lib.fwn_last_error()
    assert lib.fwn_split_planes(1 << 20, 1, 7, 1 << 20, None) == -1          # odd T
    assert lib.fwn_upsample_stage(1 << 20, 1, 4, 80, 1 << 20, 0.0, 3, 1 << 20, None, None) == -1   # odd s
    assert b"even" in lib.fwn_last_error()
    assert lib.fwn_pack_bf16(None, None, None, None, 1, 1, 1, 1, None, None) == -1
    d = _lib.FlowDesc()
    assert lib.fwn_front(C.byref(d), 1 << 20, 1 << 20, None, 64, 64, 1, None) == -1
    assert b"power of two" in lib.fwn_last_error() or b"flow desc" in lib.fwn_last_error()
    m = _lib.ModelDesc()
    assert lib.fwn_workspace_bytes(C.byref(m), 1, 256) == 0
    assert lib.fwn_model_forward(C.byref(m), 1, 256, None, None, None, 0, None, None, 0, None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.FwnError, match="no CPU fallback"):
        _lib.load()


def _build_c_consumer(tmp_path):
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    exe = str(tmp_path / "cabi_smoke")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "cabi_smoke.c"), "-o", exe, "-ldl", "-lm"])
    return exe


def test_plain_c_program_binds_the_abi(lib, tmp_path):
    """include/fwn.h compiles as C99 and a torch-free C program gets version, error codes and messages."""
    import subprocess
    exe = _build_c_consumer(tmp_path)
    out = subprocess.run([exe, _lib.LIB_PATH], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    assert "C ABI ok: version 321" in out.stdout


@pytest.mark.gpu
def test_plain_c_program_runs_kernels_on_the_gpu(lib, tmp_path):
    import subprocess
    exe = _build_c_consumer(tmp_path)
    out = subprocess.run([exe, _lib.LIB_PATH, "gpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "GPU round trip ok" in out.stdout


def test_training_primitives_validate_their_arguments(lib):
    """No launch happens for bad descriptors: error code + message (CPU-only check)."""
    g = _lib.GemmDesc()
    assert lib.fwn_gemm(C.byref(g), None) == -1 and b"fwn_gemm" in lib.fwn_last_error()
    g.W, g.Y, g.nseg, g.M, g.N, g.ldw, g.ldy, g.nsplit = 1 << 20, 1 << 21, 1, 64, 64, 64, 64, 1
    g.seg[0].x, g.seg[0].rows, g.seg[0].ld, g.seg[0].k, g.seg[0].koff = 1 << 22, 64, 64, 60, 0       # k not a multiple of 8
    assert lib.fwn_gemm(C.byref(g), None) == -1 and b"multiples of 8" in lib.fwn_last_error()
    g.seg[0].k = 64
    g.nsplit = 4                                                                                     # split-K needs an fp32 output
    assert lib.fwn_gemm(C.byref(g), None) == -1 and b"split-K" in lib.fwn_last_error()
    assert lib.fwn_transpose_shift(None, 4, 4, 4, 0, 0, 1, 0, None, 64, 0, None) == -1
    assert lib.fwn_reduce_splits(None, 2, 16, 16, 1.0, None, None) == -1
    assert lib.fwn_coupling_bwd(None, None, None, None, 4, 1, 0.0, None, 8, None, None) == -1
    assert lib.fwn_wn_backward_group(None, 1, None, None) == -1
    wj = (_lib.WnJob * 1)()
    assert lib.fwn_wn_backward_group(wj, 1, None, None) == -1 and b"bad job" in lib.fwn_last_error()
    assert lib.fwn_wn_backward_group(wj, _lib.FWN_MAX_GROUP + 1, None, None) == -1
    tj = (_lib.TnJob * 1)()
    assert lib.fwn_tn_gemm_group(tj, 1, 64, 0, None) == -1 and b"job 0" in lib.fwn_last_error()
    assert C.sizeof(_lib.TnJob) == 72 and C.sizeof(_lib.WnJob) == 104
    assert lib.fwn_upsample_bwd(1 << 20, 1 << 20, 1 << 20, 1, 4, 8, 3, 1 << 20, None, 1 << 20, 1 << 20, None) == -1   # odd s
    assert lib.fwn_mel_spectrogram(1 << 20, 1, 4096, 1 << 20, 1 << 20, 1000, 256, 80, 20.0, -100.0, 1 << 20, None) == -1
    assert b"power of two" in lib.fwn_last_error()
    assert lib.fwn_pack_jobs(None, 1, None, 0, None, 512, None) == -1
    assert lib.fwn_colsum_partials(1000, 8) > 0 and lib.fwn_upsample_bwd_partials(8, 25, 16) > 0


def test_build_keeps_slp_vectorisation_off_and_the_valu_front_conv_free_of_swizzled_packed_math():
    """ADVICE r1 / DESIGN 3.5: hipcc's SLP pass turned the scalar fp32 code of front_valu_kernel into v_pk_mul_f32 /
    v_pk_add_f32 with op_sel swizzles, which returned wrong lanes whenever another kernel shared the CU.  The build
    must carry -fno-slp-vectorize, and the shipped code object must not contain packed fp32 math in that kernel."""
    import re
    import shutil
    import subprocess
    mk = open(os.path.join(ROOT, "tf-flowavenet_amd", "csrc", "Makefile")).read()
    flags = [ln for ln in mk.splitlines() if ln.startswith("CXXFLAGS")]
    assert flags and "-fno-slp-vectorize" in flags[0]
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    import glob
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "libfwn.so")
        shutil.copy(_lib.LIB_PATH, so)
        subprocess.run([objdump, "--offloading", so], check=True, capture_output=True, cwd=tmp)     # extracts the code objects
        dis = ""
        for co in sorted(glob.glob(so + ".*gfx950")):
            dis += subprocess.run([objdump, "-d", co], capture_output=True, text=True, check=True).stdout
    blocks = re.split(r"\n(?=[0-9a-f]+ <)", dis)
    front = [b for b in blocks if "front_valu_kernel" in b.split("\n", 1)[0]]
    assert front, "front_valu_kernel not found in the code object"
    for b in front:
        assert "v_pk_mul_f32" not in b and "v_pk_add_f32" not in b and "v_pk_fma_f32" not in b


def test_graft_entry_build_runs_and_checks_the_header_version():
    """__graft_entry__.build() (the driver's "does it build" check): make is a no-op on an up-to-date tree, every symbol
    binds, and the version assertion follows include/fwn.h (it once pinned a stale literal and failed after a bump)."""
    import importlib
    g = importlib.import_module("__graft_entry__")
    g.build()
