"""ctypes binding of ``csrc/libfwn.so`` (C ABI declared in ``include/fwn.h``).

There is deliberately no fallback: if the shared library is missing or a symbol
is absent, importing/using the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os

FWN_MAX_LAYERS = 8
FWN_MAX_UPSAMPLE = 4

_HERE = os.path.dirname(os.path.abspath(__file__))
# FWN_LIB: developer override (tools/tune.py loads the -DFWN_TUNABLE build); the product is csrc/libfwn.so
LIB_PATH = os.environ.get("FWN_LIB") or os.path.join(_HERE, "csrc", "libfwn.so")

vp = C.c_void_p
i32 = C.c_int32
i64 = C.c_int64


class FlowDesc(C.Structure):
    """Mirror of ``fwn_flow_desc`` (include/fwn.h)."""
    _fields_ = [
        ("Ch", i32), ("cin", i32), ("kcpad", i32), ("kfpad", i32), ("npt", i32), ("L", i32),
        ("Wfront", vp), ("bfront", vp), ("Wfront2", vp),
        ("Wd", vp * FWN_MAX_LAYERS), ("Wc", vp * FWN_MAX_LAYERS), ("bgate", vp * FWN_MAX_LAYERS),
        ("Wres", vp * FWN_MAX_LAYERS), ("bres", vp * FWN_MAX_LAYERS),
        ("Wskip", vp), ("bskip", vp), ("Wfinal", vp), ("bfinal", vp),
        ("Wzero", vp), ("bzero", vp), ("ezero", vp),
        ("an", vp),
        ("Wd8", vp * FWN_MAX_LAYERS), ("wd8_exp", i32 * FWN_MAX_LAYERS),
        ("Wfront3", vp), ("kf3", i32), ("reserved", i32), ("Wgs", vp * FWN_MAX_LAYERS),
        ("Wts", vp),
    ]


class ModelDesc(C.Structure):
    """Mirror of ``fwn_model_desc`` (include/fwn.h)."""
    _fields_ = [
        ("n_block", i32), ("n_flow", i32), ("n_layer", i32), ("num_mels", i32),
        ("n_up", i32),
        ("up_scale", i32 * FWN_MAX_UPSAMPLE),
        ("up_w", vp * FWN_MAX_UPSAMPLE),
        ("up_bias", C.c_float * FWN_MAX_UPSAMPLE),
        ("flows", C.POINTER(FlowDesc)),
        ("cond_mode", i32),
        ("gate_fp8", i32),
        ("chain_mode", i32),
        ("persist_mode", i32),
        ("block_events", C.POINTER(vp)),
        ("cond_stream", vp * 16),
    ]


REDUCE_FN = C.CFUNCTYPE(C.c_int, vp, vp, C.c_int, vp)      # fwn_reduce_fn(user, buf, n, stream)
BLOCK_DONE_FN = C.CFUNCTYPE(C.c_int, vp, C.c_int)         # fwn_block_done_fn(user, block) -> 0 to go on


class ConvGrad(C.Structure):
    """fwn_conv_grad (include/fwn.h)."""
    _fields_ = [("V", vp), ("g", vp), ("dV", vp), ("dg", vp), ("db", vp)]


class FlowTrainDesc(C.Structure):
    """fwn_flow_train_desc (include/fwn.h)."""
    _fields_ = [("WfT", vp), ("WdT", vp * FWN_MAX_LAYERS), ("WcT", vp * FWN_MAX_LAYERS), ("WresT", vp * FWN_MAX_LAYERS),
                ("Wskip", vp), ("WskipT_all", vp), ("Wfin", vp), ("WfinT", vp), ("Wz", vp), ("WzT", vp),
                ("bskip", vp), ("bfin", vp), ("bz", vp), ("ez", vp), ("ldz", i32), ("wct_ld", i32),
                ("front", ConvGrad), ("final_", ConvGrad), ("zero", ConvGrad),
                ("filt", ConvGrad * FWN_MAX_LAYERS), ("gate", ConvGrad * FWN_MAX_LAYERS), ("res", ConvGrad * FWN_MAX_LAYERS),
                ("skip", ConvGrad * FWN_MAX_LAYERS), ("filt_c", ConvGrad * FWN_MAX_LAYERS), ("gate_c", ConvGrad * FWN_MAX_LAYERS),
                ("d_an_b", vp), ("d_an_logs", vp), ("d_zscale", vp)]


class TrainDesc(C.Structure):
    """fwn_train_desc (include/fwn.h)."""
    _fields_ = [("model", C.POINTER(ModelDesc)), ("flows", C.POINTER(FlowTrainDesc)),
                ("cond_rows", vp * 16), ("front_rows", vp * 16), ("zinv32", vp * 16), ("br", vp * 16), ("zcol", vp * 16),
                ("up_bias_dev", vp * FWN_MAX_UPSAMPLE), ("up", ConvGrad * FWN_MAX_UPSAMPLE),
                ("an_logdet", vp), ("zero_dead_res", i32), ("defer_block_done", i32), ("side_stream", vp)]


# name -> (restype, argtypes); every symbol include/fwn.h declares.
class ScaleJob(C.Structure):
    _fields_ = [("v", C.c_void_p), ("g", C.c_void_p), ("k_src", C.c_int32), ("n_src", C.c_int32)]


class PackJob(C.Structure):
    _fields_ = [("v", C.c_void_p), ("src_k", C.c_void_p), ("src_n", C.c_void_p), ("out", C.c_void_p), ("ld_dst", C.c_int64),
                ("n_src", C.c_int32), ("k_dst", C.c_int32), ("n_dst", C.c_int32), ("scale_slot", C.c_int32),
                ("transposed", C.c_int32), ("mul", C.c_float)]


class GemmSeg(C.Structure):
    _fields_ = [("x", C.c_void_p), ("rows", C.c_int32), ("ld", C.c_int32), ("k", C.c_int32), ("shift", C.c_int32),
                ("koff", C.c_int32), ("pad_", C.c_int32)]


FWN_GEMM_MAXSEG = 8


FWN_MAX_GROUP = 16


class TnJob(C.Structure):
    """fwn_tn_job (include/fwn.h)."""
    _fields_ = [("x", C.c_void_p), ("dy", C.c_void_p), ("part", C.c_void_p), ("split_stride", C.c_int64),
                ("ldx", C.c_int32), ("Kx", C.c_int32), ("ntap", C.c_int32), ("shift0", C.c_int32), ("dshift", C.c_int32),
                ("ldy", C.c_int32), ("N", C.c_int32), ("nsplit", C.c_int32), ("bias_row", C.c_int32), ("reserved", C.c_int32)]


class WnJob(C.Structure):
    """fwn_wn_job (include/fwn.h)."""
    _fields_ = [("part", C.c_void_p), ("row_src", C.c_void_p), ("V", C.c_void_p), ("g", C.c_void_p), ("dV", C.c_void_p),
                ("dg", C.c_void_p), ("db", C.c_void_p), ("split_stride", C.c_int64),
                ("nsplit", C.c_int32), ("ldp", C.c_int32), ("col0", C.c_int32), ("bias_row", C.c_int32), ("K", C.c_int32),
                ("N", C.c_int32), ("scale", C.c_float), ("reserved", C.c_int32), ("col_src", C.c_void_p)]


class GemmDesc(C.Structure):
    """include/fwn.h fwn_gemm_desc."""
    _fields_ = [("seg", GemmSeg * FWN_GEMM_MAXSEG),
                ("nseg", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("Ti", C.c_int32),
                ("W", C.c_void_p), ("ldw", C.c_int32), ("pad0_", C.c_int32),
                ("bias", C.c_void_p),
                ("R", C.c_void_p), ("ldr", C.c_int32), ("rscale", C.c_float),
                ("mask", C.c_void_p), ("ldmask", C.c_int32), ("relu", C.c_int32),
                ("Y", C.c_void_p), ("ldy", C.c_int32), ("out_f32", C.c_int32),
                ("accumulate", C.c_int32), ("nsplit", C.c_int32),
                ("split_stride", C.c_int64),
                ("oscale", C.c_float), ("gate_col0", C.c_int32),
                ("gate_aux", C.c_void_p), ("gate_out", C.c_void_p)]


SIGNATURES = {
    "fwn_version": (C.c_int, []),
    "fwn_last_error": (C.c_char_p, []),
    "fwn_wn_scale": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp]),
    "fwn_pack_bf16": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, i64, vp, vp]),
    "fwn_gather_tables": (C.c_int, [vp, vp, C.c_int, i64, vp, vp, vp, vp]),
    "fwn_sum_f32": (C.c_int, [vp, i64, vp, vp]),
    "fwn_upsample_wn": (C.c_int, [vp, vp, C.c_int, vp, vp]),
    "fwn_upsample_stage": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, C.c_float, C.c_int, vp, vp, vp]),
    "fwn_upsample_stage_dev": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, vp]),
    "fwn_split_planes": (C.c_int, [vp, i64, i64, vp, vp]),
    "fwn_merge_planes": (C.c_int, [vp, i64, i64, vp, vp]),
    "fwn_actnorm_ddi": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp]),
    "fwn_actnorm_moments": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp]),
    "fwn_actnorm_from_moments": (C.c_int, [vp, C.c_int, vp, vp]),
    "fwn_front": (C.c_int, [C.POINTER(FlowDesc), vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "fwn_gate": (C.c_int, [C.POINTER(FlowDesc), C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, vp]),
    "fwn_tail_can_chain": (C.c_int, [C.POINTER(FlowDesc), C.c_int, C.c_int]),
    "fwn_tail_partials_chained": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "fwn_tail_chained": (C.c_int, [C.POINTER(FlowDesc), C.POINTER(FlowDesc), vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "fwn_gate_stream_bytes": (i64, [C.c_int]),
    "fwn_pack_gate_stream": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp]),
    "fwn_gate_stream_rows": (C.c_int, []),
    "fwn_tail_stream_bytes": (i64, [C.c_int]),
    "fwn_pack_tail_stream": (C.c_int, [vp, vp, C.c_int, vp, vp]),
    "fwn_tail_stream_rows": (C.c_int, []),
    "fwn_gate_clock": (C.c_int, [C.POINTER(FlowDesc), C.c_int, vp, vp, vp, C.c_int, C.c_int, vp, vp]),
    "fwn_gate_fp8_supported": (C.c_int, [C.c_int, C.c_int]),
    "fwn_gate_fp8": (C.c_int, [C.POINTER(FlowDesc), C.c_int, vp, vp, vp, C.c_int, C.c_int, vp]),
    "fwn_cast_e4m3": (C.c_int, [vp, vp, i64, vp]),
    "fwn_wn_absmax": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_float, vp, vp]),
    "fwn_pack_e4m3": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, i64, C.c_float, vp, vp, vp, vp]),
    "fwn_flow_run_fp8": (C.c_int, [C.POINTER(FlowDesc), i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int,
                                   C.c_int, vp, vp, vp]),
    "fwn_res": (C.c_int, [C.POINTER(FlowDesc), C.c_int, vp, vp, vp, C.c_int, vp]),
    "fwn_cond": (C.c_int, [vp, vp, vp, i64, i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                           C.c_int, vp]),
    "fwn_cond_splits": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "fwn_cond_split": (C.c_int, [vp, vp, vp, i64, i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, i64,
                                 C.c_int, vp]),
    "fwn_cond_reduce": (C.c_int, [vp, vp, i64, C.c_int, i64, vp]),
    "fwn_pack_tail_stream_jobs": (C.c_int, [vp, C.c_int, C.c_int, vp]),
    "fwn_cond_stream_bytes": (i64, [C.c_int]),
    "fwn_cond_stream_rows": (C.c_int, []),
    "fwn_cond_stream_splits": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "fwn_pack_cond_stream": (C.c_int, [vp, i64, C.c_int, C.c_int, vp, vp]),
    "fwn_cond_stream": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, i64, C.c_int, vp]),
    "fwn_tail_partials": (C.c_int, [C.c_int]),
    "fwn_tail_partials_desc": (C.c_int, [C.POINTER(FlowDesc), C.c_int, C.c_int]),
    "fwn_tail": (C.c_int, [C.POINTER(FlowDesc), vp, vp, vp, vp, C.c_int, C.c_int, vp, vp]),
    "fwn_tail_train": (C.c_int, [C.POINTER(FlowDesc), vp, C.c_int64, vp, vp, vp, C.c_int, vp, vp, vp, vp]),
    "fwn_flow_run": (C.c_int, [C.POINTER(FlowDesc), i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int,
                               C.c_int, vp]),
    "fwn_flow_persist_supported": (C.c_int, [C.POINTER(FlowDesc), i64, i64]),
    "fwn_flow_persist_sync_bytes": (i64, [C.c_int, C.c_int]),
    "fwn_flow_run_persist": (C.c_int, [C.POINTER(FlowDesc), i64, i64, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp]),
    "fwn_flow_persist_status": (C.c_int, [vp, vp]),
    "fwn_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "fwn_prior_logp": (C.c_int, [vp, i64, vp, C.c_int, vp, vp]),
    "fwn_pack_jobs": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, vp]),
    "fwn_tn_gemm": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, i64, C.c_int, vp]),
    "fwn_colsum_bf16": (C.c_int, [vp, i64, C.c_int, C.c_int, C.c_float, vp, vp, vp]),
    "fwn_gemm": (C.c_int, [C.POINTER(GemmDesc), vp]),
    "fwn_upsample_bwd_partials": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "fwn_upsample_bwd": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
    "fwn_gate_train": (C.c_int, [C.POINTER(FlowDesc), C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp]),
    "fwn_actnorm_apply": (C.c_int, [vp, vp, i64, C.c_int, vp]),
    "fwn_actnorm_apply2": (C.c_int, [vp, vp, vp, i64, C.c_int, vp]),
    "fwn_coupling_fwd": (C.c_int, [vp, vp, vp, i64, C.c_int, vp, C.c_int, vp]),
    "fwn_coupling_bwd": (C.c_int, [vp, vp, vp, vp, i64, C.c_int, C.c_float, vp, C.c_int, vp, vp]),
    "fwn_gate_bwd": (C.c_int, [vp, C.c_int, vp, i64, vp, vp]),
    "fwn_colsum_partials": (C.c_int, [i64, C.c_int]),
    "fwn_colsum_prod": (C.c_int, [vp, vp, i64, C.c_int, C.c_float, vp, vp, vp]),
    "fwn_actnorm_bwd": (C.c_int, [vp, vp, vp, i64, C.c_int, vp]),
    "fwn_flow_small_grads_partials": (i64, [i64, C.c_int]),
    "fwn_flow_small_grads": (C.c_int, [vp, vp, vp, vp, vp, vp, i64, C.c_int, vp, vp, vp, vp, vp, vp, vp]),
    "fwn_tn_gemm_tile": (C.c_int, [C.c_int]),
    "fwn_tn_gemm_group": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp]),
    "fwn_wn_group_scratch": (i64, [vp, C.c_int]),
    "fwn_wn_backward_group": (C.c_int, [vp, C.c_int, vp, vp]),
    "fwn_transpose_shift": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp]),
    "fwn_reduce_splits": (C.c_int, [vp, C.c_int, i64, i64, C.c_float, vp, vp]),
    "fwn_mel_spectrogram": (C.c_int, [vp, i64, i64, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, vp, vp]),
    "fwn_grad_norm_partials": (C.c_int, [i64]),
    "fwn_grad_norm": (C.c_int, [vp, i64, C.c_float, vp, vp, vp]),
    "fwn_clip_adam": (C.c_int, [vp, vp, vp, vp, i64, vp, C.c_float, C.c_float, C.c_float, i64, C.c_float,
                                C.c_float, C.c_float, vp]),
    "fwn_adam_rate": (C.c_double, [C.c_float, i64, C.c_float, C.c_float]),
    "fwn_clip_adam_dev": (C.c_int, [vp, vp, vp, vp, i64, vp, C.c_float, C.c_float, vp, C.c_float, C.c_float, C.c_float, vp]),
    "fwn_workspace_bytes": (C.c_size_t, [C.POINTER(ModelDesc), i64, i64]),
    "fwn_model_forward": (C.c_int, [C.POINTER(ModelDesc), i64, i64, vp, vp, vp, C.c_size_t, vp, vp,
                                    C.c_int, vp]),
    "fwn_model_forward_init": (C.c_int, [C.POINTER(ModelDesc), i64, i64, vp, vp, vp, C.c_size_t, vp, vp, vp, vp, vp]),
    "fwn_train_workspace_bytes": (C.c_size_t, [C.POINTER(TrainDesc), i64, i64]),
    "fwn_train_loss_and_grads": (C.c_int, [C.POINTER(TrainDesc), i64, i64, vp, vp, vp, C.c_size_t, vp, BLOCK_DONE_FN, vp, vp]),
    "fwn_model_reverse": (C.c_int, [C.POINTER(ModelDesc), i64, i64, vp, vp, vp, C.c_size_t, vp, vp]),
    "fwn_model_persist_status": (C.c_int, [C.POINTER(ModelDesc), i64, i64, vp, vp]),
}

_lib = None


class FwnError(RuntimeError):
    pass


def load():
    """Load libfwn.so and bind every symbol; raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FwnError(
            "HIP extension %s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C tf-flowavenet_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
    # libfwn.so depends on libamdhip64; in a process that also uses torch on the GPU that name must resolve to torch's own
    # copy, i.e. torch has to be loaded first (the other order left torch with the system runtime: "no ROCm-capable device")
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    # developer switches of the same-box A/B scripts (tools/diag): read ONCE here, handed to the library as options - the
    # launch path itself reads no environment (round 4's did, on every gate launch)
    for opt in ("rs_persist",):
        v = os.environ.get("FWN_OPT_" + opt.upper())
        if v is not None:
            lib.fwn_set_option(opt.encode(), int(v))
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().fwn_last_error()
        raise FwnError("%s failed (%d): %s" % (what or "libfwn call", rc, msg.decode() if msg else "?"))
