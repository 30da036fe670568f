"""ctypes binding of libmacx.so -- the C ABI declared in include/macx.h.

The product path has NO CPU fallback: if the library is missing or does not export the ABI this
module raises, and every op built on it fails loudly.
"""
import ctypes as C
import os

from . import build as _build

ABI_VERSION = 5

MACX_OK, MACX_EINVAL, MACX_EUNSUPPORTED, MACX_EREJECTED, MACX_ESMALL = 0, -1, -2, -3, -4
ACT = {"NON": 0, "TANH": 1, "SIGMOID": 2, "ELU": 3, "RELU": 4}
INIT = {"PRM": 0, "ZERO": 1, "Q": 2}
WRITE_INPUTS = {"MEM": 0, "INFO": 1, "SUM": 2, "BOTH": 3}
SEG = {"controls": 0, "memories": 1, "infos": 2, "att_question": 3, "att_kb": 4, "att_self": 5, "att_gate": 6}


class MacxOpts(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "abi_version", "init_ctrl", "init_mem", "control_input_unshared", "control_input_act", "control_feed_prev",
        "control_feed_prev_att", "control_feed_inputs", "control_cont_act", "read_mem_act", "read_ctrl_act",
        "write_inputs", "write_self_att", "write_self_att_cont", "write_mem_act", "write_gate", "write_gate_shared")]
    _fields_ += [("write_gate_bias", C.c_float), ("memory_variational_dropout", C.c_int32), ("gemm_family", C.c_int32),
                 ("tune", C.c_int32 * 16)]


# macx_opts.tune keys (include/macx.h MACX_TUNE_*): the per-call A/B hooks and the profiling tools' phase mask.  A value v travels
# as v + 1; 0 = the shipped default of that key.
TUNE = {"native_waves": 0, "phase_mask": 1, "row_tiles": 2, "pre_fill": 3, "chain": 4, "sb_defer": 5, "chain_kv": 7, "sb_wide": 8,
        "wgrad_pipe": 10, "sb_cont": 13, "dkb_uni": 14, "dkb_fill": 15}


def set_tune(opts, key, value):
    """opts.tune[key] = value (None: back to the default).  key: a name of TUNE or its number."""
    k = TUNE[key] if isinstance(key, str) else int(key)
    if k not in TUNE.values():
        raise KeyError("no tuning key %r (have %s)" % (key, sorted(TUNE)))
    opts.tune[k] = 0 if value is None else int(value) + 1


class MacxShapes(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "S", "N", "d", "p", "b0", "d_logical")]


class MacxDropout(C.Structure):
    _fields_ = [("keep_memory", C.c_float), ("keep_read", C.c_float), ("keep_write", C.c_float), ("seed", C.c_uint32),
                ("mask_word", C.c_void_p)]


PARAM_FIELDS = (
    "initMem", "initCtrl", "qInput_W", "qInput_b", "qInputU_W", "qInputU_b", "ctrlLogits_w", "ctrlLogits_b",
    "contControl_W", "contControl_b", "contControl2_W", "contControl2_b", "projX_W", "projX_b", "projY_W", "projY_b",
    "memKbProj_W", "memKbProj_b", "memKbProj2_W", "memKbProj2_b", "kbLogits_w", "kbLogits_b", "newMemory_W",
    "newMemory_b", "selfCtrl_W", "selfCtrl_b", "selfLogits_w", "selfLogits_b", "gate_W", "gate_b")


class MacxParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in PARAM_FIELDS]


class MacxParamGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in PARAM_FIELDS]


class MacxOutShapes(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "d", "hidden", "answers", "b0")]


OUT_FIELDS = ("outQuestion_W", "outQuestion_b", "fc0_W", "fc0_b", "fc1_W", "fc1_b")


class MacxOutParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in OUT_FIELDS]


class MacxOutGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in OUT_FIELDS]


class MacxStemShapes(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "H", "W", "Cin", "Cmid", "Cout", "b0")]


STEM_FIELDS = ("kernel0", "bias0", "kernel1", "bias1")


class MacxStemParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in STEM_FIELDS]


class MacxStemGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in STEM_FIELDS]


class MacxEncShapes(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "S", "V", "E", "h", "b0")]


ENC_FIELDS = ("emb", "fw_kernel", "fw_bias", "bw_kernel", "bw_bias")


class MacxEncParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ENC_FIELDS]


class MacxEncGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ENC_FIELDS]


class MacxInputs(C.Structure):
    _fields_ = [("vecQuestions", C.c_void_p), ("words", C.c_void_p), ("questionLengths", C.c_void_p),
                ("knowledgeBase", C.c_void_p)]


class MacxInputGrads(C.Structure):
    _fields_ = [("vecQuestions", C.c_void_p), ("words", C.c_void_p), ("knowledgeBase", C.c_void_p)]


EXPORTS = ("macx_abi_version", "macx_strerror", "macx_check", "macx_saved_floats", "macx_ws_floats",
           "macx_saved_segment", "macx_cell_begin", "macx_cell_step", "macx_cell_forward", "macx_cell_backward",
           "macx_cell_backward_phase",
           "macx_linear", "macx_pack_weight", "macx_kb_project", "macx_control_attend", "macx_dropout_mask", "macx_dropout_mask_w", "macx_wgrad_splits",
           "macx_wgrad", "macx_output_saved_floats", "macx_output_ws_floats",
           "macx_output_forward", "macx_output_backward", "macx_adam_ema_step",
           "macx_stem_saved_floats", "macx_stem_ws_floats", "macx_stem_forward", "macx_stem_backward",
           "macx_encoder_saved_floats", "macx_encoder_ws_floats", "macx_encoder_forward", "macx_encoder_backward",
           "macx_images_to_nhwc", "macx_gemm_mode", "macx_h2_floats", "macx_h2_from_f32", "macx_h2_to_f32", "macx_h2_gemm",
           "macx_h2_pack_weight", "macx_h2_gemm_planes", "macx_op_act", "macx_op_act_bwd", "macx_op_binary", "macx_op_reduce",
           "macx_op_softmax", "macx_op_softmax_bwd", "macx_op_dropout", "macx_op_dropout_w", "macx_kb_attend_fwd", "macx_kb_attend_bwd",
           "macx_kb_attend_bwd_ws_floats", "macx_answer_loss", "macx_workspace_bytes", "macx_embed_lookup", "macx_embed_lookup_bwd", "macx_control_attend_bwd",
           "macx_control_attend_bwd_ws_floats", "macx_read_fwd", "macx_read_bwd",
           "macx_write_fwd", "macx_write_bwd", "macx_read_chain_time", "macx_cell_forward_chain_time", "macx_saved_activation", "macx_ctrl_inputs_ws_floats",
           "macx_ctrl_inputs_fwd", "macx_ctrl_inputs_bwd")

_lib = None


class MacxError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib().macx_strerror(code).decode() if _lib is not None else str(code)
        super().__init__("%s failed: %s (code %d)" % (where, msg, code))


def lib():
    """Load libmacx.so, rebuilding it first when the digest of csrc/ + include/ differs from the one it was built from
    (a sha256 comparison; a no-op when nothing changed).  Where hipcc is absent a library whose digest matches is loaded
    as it is; a stale one raises.  Raises if the library cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.build(force=bool(os.environ.get("MACX_REBUILD")))
    L = C.CDLL(path)
    missing = [n for n in EXPORTS if not hasattr(L, n)]
    if missing:
        raise ImportError("libmacx.so lacks symbols: %s" % missing)
    L.macx_abi_version.restype = C.c_int
    if L.macx_abi_version() != ABI_VERSION:
        raise ImportError("libmacx.so ABI %d != binding ABI %d" % (L.macx_abi_version(), ABI_VERSION))
    L.macx_strerror.restype = C.c_char_p
    L.macx_strerror.argtypes = [C.c_int]
    P = C.POINTER
    L.macx_check.argtypes = [P(MacxOpts), P(MacxShapes)]
    L.macx_saved_floats.restype = C.c_size_t
    L.macx_saved_floats.argtypes = [P(MacxOpts), P(MacxShapes), C.c_int]
    L.macx_ws_floats.restype = C.c_size_t
    L.macx_ws_floats.argtypes = [P(MacxOpts), P(MacxShapes), C.c_int]
    L.macx_saved_segment.argtypes = [P(MacxOpts), P(MacxShapes), C.c_int, C.c_int, P(C.c_size_t), P(C.c_size_t)]
    common = [P(MacxOpts), P(MacxShapes), P(MacxDropout), P(MacxParams), P(MacxInputs), C.c_void_p, C.c_size_t,
              C.c_void_p, C.c_size_t]
    L.macx_cell_begin.argtypes = common + [C.c_int, C.c_void_p]
    L.macx_cell_step.argtypes = common + [C.c_int, C.c_int, C.c_void_p]
    L.macx_cell_forward.argtypes = common + [C.c_int, C.c_void_p]
    L.macx_cell_backward.argtypes = common + [C.c_void_p, C.c_void_p, P(MacxParamGrads), P(MacxInputGrads), C.c_void_p]
    L.macx_cell_backward_phase.argtypes = common + [C.c_void_p, C.c_void_p, P(MacxParamGrads), P(MacxInputGrads), C.c_int, C.c_void_p]
    L.macx_linear.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float,
                              C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.macx_pack_weight.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.macx_kb_project.argtypes = [P(MacxShapes), P(MacxDropout), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]
    L.macx_control_attend.argtypes = [P(MacxShapes)] + [C.c_void_p] * 8
    L.macx_dropout_mask.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, C.c_size_t, C.c_void_p,
                                    C.c_void_p]
    L.macx_dropout_mask_w.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, C.c_size_t, C.c_void_p, C.c_void_p,
                                      C.c_void_p]
    L.macx_read_chain_time.argtypes = [P(MacxOpts), P(MacxShapes), P(MacxDropout), P(MacxParams), P(MacxInputs), C.c_void_p, C.c_size_t,
                                       C.c_int, C.c_int, P(C.c_float), C.c_void_p]
    L.macx_cell_forward_chain_time.argtypes = common + [P(C.c_float), C.c_void_p]
    L.macx_saved_activation.argtypes = [P(MacxOpts), P(MacxShapes), C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.macx_ctrl_inputs_ws_floats.restype = C.c_size_t
    L.macx_ctrl_inputs_ws_floats.argtypes = [P(MacxOpts), P(MacxShapes)]
    L.macx_ctrl_inputs_fwd.argtypes = [P(MacxOpts), P(MacxShapes), P(MacxParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                       C.c_void_p]
    L.macx_ctrl_inputs_bwd.argtypes = [P(MacxOpts), P(MacxShapes), P(MacxParams), C.c_void_p, C.c_void_p, C.c_void_p, P(MacxParamGrads),
                                       C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.macx_gemm_mode.argtypes = [C.c_int]
    L.macx_adam_ema_step.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                     C.c_float, C.c_float, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    L.macx_adam_ema_step.restype = C.c_int
    L.macx_stem_saved_floats.restype = C.c_size_t
    L.macx_stem_saved_floats.argtypes = [P(MacxStemShapes)]
    L.macx_stem_ws_floats.restype = C.c_size_t
    L.macx_stem_ws_floats.argtypes = [P(MacxStemShapes)]
    L.macx_stem_forward.restype = C.c_int
    L.macx_stem_forward.argtypes = [P(MacxStemShapes), C.c_int, C.c_float, C.c_uint32, P(MacxStemParams), C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_size_t, C.c_void_p]
    L.macx_stem_backward.restype = C.c_int
    L.macx_stem_backward.argtypes = [P(MacxStemShapes), C.c_int, C.c_float, C.c_uint32, P(MacxStemParams), C.c_void_p, C.c_void_p,
                                     C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, P(MacxStemGrads), C.c_void_p]
    L.macx_output_saved_floats.restype = C.c_size_t
    L.macx_output_saved_floats.argtypes = [P(MacxOutShapes)]
    L.macx_output_ws_floats.restype = C.c_size_t
    L.macx_output_ws_floats.argtypes = [P(MacxOutShapes)]
    L.macx_output_forward.argtypes = [P(MacxOutShapes), C.c_int, C.c_float, C.c_uint32, P(MacxOutParams), C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.macx_output_backward.argtypes = [P(MacxOutShapes), C.c_int, C.c_float, C.c_uint32, P(MacxOutParams), C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, P(MacxOutGrads), C.c_void_p,
                                       C.c_void_p, C.c_void_p]
    L.macx_wgrad_splits.argtypes = [C.c_int, C.c_int, C.c_int]
    L.macx_wgrad.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                             C.c_void_p]
    for n in EXPORTS:
        f = getattr(L, n)
        if f.restype is C.c_int or n in ("macx_check",):
            f.restype = C.c_int
    L.macx_images_to_nhwc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.macx_encoder_saved_floats.argtypes = [P(MacxEncShapes)]
    L.macx_encoder_ws_floats.argtypes = [P(MacxEncShapes)]
    L.macx_encoder_forward.argtypes = [P(MacxEncShapes), C.c_float, C.c_float, C.c_uint32, P(MacxEncParams), C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.macx_encoder_backward.argtypes = [P(MacxEncShapes), C.c_float, C.c_float, C.c_uint32, P(MacxEncParams), C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, P(MacxEncGrads),
                                        C.c_void_p]
    L.macx_h2_floats.argtypes = [C.c_size_t, C.c_size_t]
    L.macx_h2_from_f32.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.macx_h2_to_f32.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.macx_h2_pack_weight.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.macx_h2_gemm_planes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.macx_h2_gemm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                               C.c_void_p, C.c_size_t, C.c_void_p]
    L.macx_answer_loss.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    P_, V_ = C.c_void_p, C.c_void_p
    L.macx_embed_lookup.argtypes = [V_, V_, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint32, C.c_uint32, V_, V_]
    L.macx_embed_lookup_bwd.argtypes = [V_, V_, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint32, C.c_uint32, V_, V_]
    L.macx_control_attend_bwd_ws_floats.restype = C.c_size_t
    L.macx_control_attend_bwd_ws_floats.argtypes = [P_]
    L.macx_control_attend_bwd.argtypes = [P_] + [V_] * 6 + [C.c_size_t] + [V_] * 5
    L.macx_workspace_bytes.restype = C.c_size_t
    L.macx_workspace_bytes.argtypes = [P_, P_, C.c_int]
    L.macx_read_fwd.argtypes = [P_] * 4 + [V_] * 3 + [V_, C.c_size_t, V_, V_, V_]
    L.macx_read_bwd.argtypes = [P_] * 4 + [V_, V_, C.c_size_t, V_, C.c_size_t, V_, P_, V_, V_, V_, V_]
    L.macx_write_fwd.argtypes = [P_] * 4 + [V_] * 3 + [V_, C.c_size_t, V_, V_]
    L.macx_write_bwd.argtypes = [P_] * 4 + [V_, C.c_size_t, V_, C.c_size_t, V_, P_, V_, V_, V_, V_]
    L.macx_kb_attend_fwd.argtypes = [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6
    L.macx_kb_attend_bwd_ws_floats.argtypes = [C.c_int, C.c_int, C.c_int]
    L.macx_kb_attend_bwd.argtypes = [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    L.macx_op_act.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    L.macx_op_act_bwd.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.macx_op_binary.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    L.macx_op_reduce.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.macx_op_softmax.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    L.macx_op_softmax_bwd.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    L.macx_op_dropout.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, C.c_void_p,
                                  C.c_void_p]
    L.macx_op_dropout_w.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, C.c_void_p,
                                    C.c_void_p, C.c_void_p]
    for n in EXPORTS:
        if n.endswith("_floats"):
            getattr(L, n).restype = C.c_size_t
        elif n not in ("macx_strerror",):
            getattr(L, n).restype = C.c_int
    _lib = L
    return L


def check(code, where):
    if code != 0:
        raise MacxError(code, where)


PACK_TRANSPOSE, PACK_F32MFMA, PACK_BF16X3, PACK_KMAJOR = 1, 0 << 1, 1 << 1, 2 << 1


def kb_pack_flags():
    """macx_pack_weight flags for a weight handed to macx_kb_project (an fp32-operand entry point: native f32 MFMA in mode 0,
    the split-bf16 kernel otherwise) under the GEMM mode in force."""
    return PACK_BF16X3 if lib().macx_gemm_mode(-1) else PACK_F32MFMA
