"""ctypes binding of libcmpc_hip.so (C ABI declared in include/cmpc.h).

The library is the product's only compute path for the CMPC head: if it is missing the import
fails loudly -- there is no CPU or PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libcmpc_hip.so")

DT_F32, DT_BF16, DT_F16 = 0, 1, 2
MODEL_CMPC, MODEL_V5_BILSTM, MODEL_VIDEO = 0, 1, 2
ABI_VERSION = 3          # CMPC_ABI_VERSION of include/cmpc.h this binding was written against
ACT_NONE, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3


class CmpcError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.run(["make", "-C", csrc, "clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", csrc, "-j8"], check=True)
    return LIB_PATH


class GemmNtArgs(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("nseg", C.c_int),
        ("A", C.c_void_p * 3), ("Bt", C.c_void_p * 3),
        ("K", C.c_int * 3), ("lda", C.c_int * 3), ("ldb", C.c_int * 3),
        ("sA", C.c_int64 * 3), ("sB", C.c_int64 * 3),
        ("C", C.c_void_p), ("ldc", C.c_int), ("sC", C.c_int64),
        ("c_f32", C.c_int),
        ("M", C.c_int), ("N", C.c_int), ("n_valid", C.c_int), ("batch", C.c_int),
        ("bias", C.c_void_p),
        ("sbias", C.c_void_p), ("ld_sbias", C.c_int),
        ("pbias", C.c_void_p), ("ld_pbias", C.c_int),
        ("rows_per_sample", C.c_int),
        ("act", C.c_int), ("alpha", C.c_float), ("accumulate", C.c_int),
    ]


class GemmTnArgs(C.Structure):
    _fields_ = [
        ("dtype", C.c_int),
        ("A", C.c_void_p), ("lda", C.c_int), ("Ka", C.c_int),
        ("D", C.c_void_p), ("ldd", C.c_int), ("Nd", C.c_int),
        ("out", C.c_void_p), ("ldo", C.c_int),
        ("R", C.c_int), ("Kv", C.c_int), ("Nv", C.c_int),
        ("nb", C.c_int), ("a_off", C.c_int64 * 8), ("d_off", C.c_int64 * 8), ("o_off", C.c_int64 * 8),
        ("nb2", C.c_int), ("a_bs", C.c_int64), ("d_bs", C.c_int64), ("o_bs", C.c_int64),
        ("rsplit", C.c_int), ("alpha", C.c_float), ("zeros", C.c_void_p),
        ("conv_H", C.c_int), ("conv_W", C.c_int), ("conv_dy", C.c_int), ("conv_dx", C.c_int),
    ]


class ConvArgs(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("X", C.c_void_p), ("ldx", C.c_int), ("Wt", C.c_void_p), ("ldw", C.c_int),
        ("bias", C.c_void_p), ("res", C.c_void_p), ("Y", C.c_void_p), ("ldy", C.c_int),
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int),
        ("ksize", C.c_int), ("stride", C.c_int), ("dil", C.c_int), ("relu", C.c_int), ("zeros", C.c_void_p),
    ]


class ConvLstmLn(C.Structure):
    _fields_ = [("beta", C.c_void_p * 5), ("gamma", C.c_void_p * 5)]


class ConvLstmDln(C.Structure):
    _fields_ = [("dbeta", C.c_void_p * 5), ("dgamma", C.c_void_p * 5)]


class PackDesc(C.Structure):
    _fields_ = [
        ("src_off", C.c_int64), ("ld_src", C.c_int),
        ("dst_off", C.c_int64), ("dst_dt", C.c_int), ("transpose", C.c_int),
        ("rows", C.c_int), ("cols", C.c_int), ("ld_dst", C.c_int),
        ("nks", C.c_int), ("ks_src", C.c_int * 4), ("ks_len", C.c_int * 4), ("ks_dst", C.c_int * 4),
        ("nns", C.c_int), ("ns_src", C.c_int * 5), ("ns_len", C.c_int * 5), ("ns_dst", C.c_int * 5),
    ]


class AdamSeg(C.Structure):
    _fields_ = [("off", C.c_int64), ("count", C.c_int), ("wd", C.c_float), ("gmult", C.c_float)]


class EngineCfg(C.Structure):
    """cmpc_cfg (include/cmpc.h)"""
    _fields_ = [
        ("batch_size", C.c_int), ("num_steps", C.c_int), ("vf_h", C.c_int), ("vf_w", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("vf_dim", C.c_int), ("c4_dim", C.c_int), ("c3_dim", C.c_int),
        ("vocab_size", C.c_int), ("v_emb_dim", C.c_int), ("mlp_dim", C.c_int), ("rnn_size", C.c_int), ("glove_dim", C.c_int),
        ("parse_dim", C.c_int),
        ("start_lr", C.c_double), ("end_lr", C.c_double), ("lr_power", C.c_double), ("lr_decay_step", C.c_int),
        ("weight_decay", C.c_float), ("loss_w", C.c_float * 4),
        ("dtype", C.c_int), ("loss_scale", C.c_float), ("n_lanes", C.c_int), ("device", C.c_int),
        ("model", C.c_int), ("hsv", C.c_int), ("bn_train", C.c_int), ("bn_decay", C.c_float),
        ("c2_dim", C.c_int), ("c2_h", C.c_int), ("c2_w", C.c_int), ("aspp_depth", C.c_int), ("low_dim", C.c_int), ("aspp_rates", C.c_int * 3), ("sample_frames", C.c_int), ("freeze_bn", C.c_int), ("conv5", C.c_int),
    ]


class Feeds(C.Structure):
    """cmpc_feeds"""
    _fields_ = [("words", C.c_void_p), ("seq_len", C.c_void_p), ("c3", C.c_void_p), ("c4", C.c_void_p), ("c5", C.c_void_p),
                ("target_fine", C.c_void_p), ("feats_ready", C.c_void_p), ("c2", C.c_void_p), ("im", C.c_void_p), ("feats_ready_lv", C.c_void_p * 3), ("levels_done", C.c_void_p)]


class Fetches(C.Structure):
    """cmpc_fetches"""
    _fields_ = [("pred", C.c_void_p), ("up", C.c_void_p), ("sigm", C.c_void_p)]


_P, _I, _F, _L = C.c_void_p, C.c_int, C.c_float, C.c_int64
_PP = C.POINTER(C.c_void_p)

# name -> argtypes (return type is always int).  Must list every symbol include/cmpc.h declares.
SIGNATURES = {
    "cmpc_gemm_nt_pair": [_P, _P, _P],
    "cmpc_gemm_nt": [C.POINTER(GemmNtArgs), _P],
    "cmpc_gemm_tn": [C.POINTER(GemmTnArgs), _P],
    "cmpc_gemm_tn_grouped": [C.POINTER(GemmTnArgs), C.c_int, _P],
    "cmpc_conv_nhwc": [C.POINTER(ConvArgs), _P],
    "cmpc_cast": [_I, _P, _I, _P, _L, _P],
    "cmpc_act_bwd": [_I, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _I, _P],
    "cmpc_wcolsum": [_I, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    "cmpc_rowdot1": [_I, _P, _P, _I, _P, _I, _I, _I, _I, _F, _P],
    "cmpc_rank1_update": [_I, _P, _P, _P, _P, _P, _I, _F, _F, _I, _I, _I, _I, _P],
    "cmpc_axpy": [_I, _P, _P, _F, _L, _P],
    "cmpc_bias_act_res": [_I, _P, _P, _P, _I, _L, _I, _P],
    "cmpc_l2norm_rows_fwd": [_I, _P, _P, _P, _P, _I, _I, _I, _P],
    "cmpc_l2norm_rows_bwd": [_I, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "cmpc_sample_stats": [_I, _P, _P, _I, _I, _I, _I, _P],
    "cmpc_lowrank_nn": [_I, _P, _I, _L, _P, _I, _L, _P, _I, _L, _I, _I, _I, _I, _I, _F, _I, _P],
    "cmpc_mutan_fwd": [_I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "cmpc_mutan_bwd": [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "cmpc_graph_softmax_fwd": [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "cmpc_graph_softmax_bwd": [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "cmpc_gconv_pre_fwd": [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "cmpc_gconv_pre_bwd": [_I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "cmpc_gconv_post_fwd": [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "cmpc_gconv_post_bwd": [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "cmpc_softmax_n_fwd": [_P, _P, _I, _I, _P],
    "cmpc_softmax_n_bwd": [_P, _P, _P, _I, _I, _P],
    "cmpc_l2norm_all_fwd": [_P, _P, _P, _I, _P],
    "cmpc_l2norm_all_bwd": [_P, _P, _P, _P, _I, _P],
    "cmpc_exchange_combine_fwd": [_I, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _P],
    "cmpc_exchange_combine_bwd": [_I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "cmpc_convlstm_a": [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "cmpc_convlstm_b": [_I, _P, _P, _P, C.POINTER(ConvLstmLn), _P, _P, _I, _I, _I, _I, _P],
    "cmpc_convlstm_c": [_I, _P, _P, C.POINTER(ConvLstmLn), _P, _P, _P, _I, _I, _I, _I, _P],
    "cmpc_convlstm_bwd": [_I, _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(ConvLstmLn), _P, _P, _P,
                          _P, _P, _P, C.POINTER(ConvLstmDln), _P, _P, _I, _I, _I, _I, _P],
    "cmpc_score_conv_fwd": [_I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "cmpc_score_conv_bwd": [_I, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P],
    "cmpc_upsample_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "cmpc_upsample_loss_bwd": [_P, _P, _P, _F, _I, _I, _I, _I, _I, _P],
    "cmpc_embed_gather": [_P, _P, _P, _I, _I, _I, _I, _P],
    "cmpc_embed_scatter": [_P, _I, _P, _P, _I, _I, _I, _P],
    "cmpc_lstm_cell_fwd": [_P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P],
    "cmpc_lstm_cell_bwd": [_P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _P],
    "cmpc_lstm_bwd_step": [_P, _P, _I, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _P],
    "cmpc_lstm_seq_fwd": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "cmpc_lstm_seq_bwd": [_P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "cmpc_parse_softmax_fwd": [_P, _I, _P, _P, _I, _I, _P],
    "cmpc_parse_softmax_bwd": [_P, _P, _P, _P, _I, _I, _I, _P],
    "cmpc_lang_pool_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "cmpc_lang_pool_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "cmpc_bn_stats": [_I, _P, _I, _I, _I, _I, _F, _P, _P, _P],
    "cmpc_bn_from_moving": [_P, _P, _I, _I, _F, _P, _P],
    "cmpc_bn_update_moving": [_P, _I, _F, _P, _P, _I, _I, _P],
    "cmpc_bn_apply_fwd": [_I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "cmpc_bn_bwd": [_I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _P, _P, _P, _I, _I, _I, _P],
    "cmpc_resize_bilinear_fwd": [_I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "cmpc_resize_bilinear_bwd": [_I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "cmpc_hsv_map": [_I, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "cmpc_reverse_sequence": [_P, _P, _P, _I, _I, _I, _P],
    "cmpc_reverse_words_tb": [_P, _P, _P, _I, _I, _P],
    "cmpc_rows_nonzero2": [_P, _P, _P, _I, _I, _I, _P],
    "cmpc_conv_to1_fwd": [_I, _P, _I, _P, _P, _P, _I, _I, _P],
    "cmpc_conv_to1_bwd": [_I, _P, _P, _I, _P, _P, _P, _P, _I, _I, _P],
    "cmpc_pack_weights": [_P, _P, _P, _P, _I, _I, _P],
    "cmpc_pack_weights_range": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "cmpc_adam_step": [_P, _P, _P, _P, _P, _I, _F, _F, _F, _F, _F, _P, _P],
    # whole-path entry points (handle = void*)
    "cmpc_default_cfg": [C.POINTER(EngineCfg)],
    "cmpc_default_cfg_model": [C.POINTER(EngineCfg), _I, _I],
    "cmpc_state_info": [_P, _I, C.POINTER(C.c_char_p), C.POINTER(_L)],
    "cmpc_get_state": [_P, C.c_char_p, _P, _L],
    "cmpc_set_state": [_P, C.c_char_p, _P, _L],
    "cmpc_create": [C.POINTER(EngineCfg), _PP],
    "cmpc_destroy": [_P],
    "cmpc_get_cfg": [_P, C.POINTER(EngineCfg)],
    "cmpc_param_info": [_P, _I, C.POINTER(C.c_char_p), C.POINTER(_L), C.POINTER(_I), C.POINTER(_L * 4)],
    "cmpc_buffers": [_P, _PP, _PP, _PP, _PP, C.POINTER(_L)],
    "cmpc_set_weights": [_P, C.c_char_p, _P, _L],
    "cmpc_get_weights": [_P, C.c_char_p, _P, _L],
    "cmpc_pack": [_P, _P],
    "cmpc_get_step": [_P, C.POINTER(_L)],
    "cmpc_set_step": [_P, _L],
    "cmpc_forward": [_P, C.POINTER(Feeds), C.POINTER(Fetches), _P],
    "cmpc_backward": [_P, _P],
    "cmpc_set_bwd_levels_event": [_P, _P],
    "cmpc_debug_check_guards": [_P],
    "cmpc_optimizer_step": [_P, _F, _P, C.POINTER(C.c_double)],
    "cmpc_optimizer_bucket": [_P, _I, _F, _P, C.POINTER(C.c_double)],
    "cmpc_tap": [_P, C.c_char_p, _PP, C.POINTER(_I), C.POINTER(_I), C.POINTER(_L * 4)],
    "cmpc_tap_name": [_P, _I, C.POINTER(C.c_char_p)],
    "cmpc_launch_count": [_P, C.POINTER(_L)],
    "cmpc_grad_bucket": [_P, _I, C.POINTER(_I), C.POINTER(_L * 4), C.POINTER(_L * 4)],
    "cmpc_grad_bucket_wait": [_P, _I, _P],
    "cmpc_phase_marks": [_P, _I],
    "cmpc_phase_marks_read": [_P, _I, C.POINTER(C.c_char_p), C.POINTER(C.c_float)],
    "cmpc_set_lanes": [_P, _I],
    "cmpc_dense_crf": [_P, _P, _I, _I, _F, _F, _F, _F, _F, _I, _P, _P, _P],
    "cmpc_launch_trace": [_I, _P],
    "cmpc_launch_trace_read": [_I, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(_L)],
    "cmpc_kernel_timing": [_P, _I],
    "cmpc_kernel_timing_read": [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_L)],
    "cmpc_plan_info": [_P, C.POINTER(C.POINTER(PackDesc)), C.POINTER(_I), C.POINTER(_L), C.POINTER(_L), C.POINTER(_I)],
    "cmpc_operand_info": [_P, C.c_char_p, C.POINTER(_L), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)],
}
# entry points that return a count, not a status
COUNTS = {"cmpc_param_count": [_P], "cmpc_tap_count": [_P], "cmpc_grad_bucket_count": [_P], "cmpc_state_count": [_P]}

_lib = None


def load():
    """Load the shared library (once) and attach prototypes; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    global LIB_PATH
    LIB_PATH = os.environ.get("CMPC_LIB_PATH", LIB_PATH)       # debug builds (e.g. the -DCMPC_GEMM_TRACE one)
    if not os.path.exists(LIB_PATH):
        raise CmpcError(
            f"{LIB_PATH} not found: the CMPC head has no fallback path. Build it with "
            f"`python -c 'import __graft_entry__ as g; g.build()'` or `make -C cmpc-refseg_amd/csrc`.")
    lib = C.CDLL(LIB_PATH)
    lib.cmpc_last_error.restype = C.c_char_p
    lib.cmpc_last_error.argtypes = []
    lib.cmpc_abi_version.restype = C.c_int
    lib.cmpc_abi_version.argtypes = []
    if lib.cmpc_abi_version() != ABI_VERSION:
        raise CmpcError(f"{LIB_PATH} exports ABI version {lib.cmpc_abi_version()}, this binding needs {ABI_VERSION}: rebuild it "
                        f"(`make -C cmpc-refseg_amd/csrc`)")
    if hasattr(lib, "cmpc_crc32c"):      # host utility (tf_bundle.py falls back to Python without it; older A/B builds lack it)
        lib.cmpc_crc32c.restype = C.c_uint32
        lib.cmpc_crc32c.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = C.c_int
        fn.argtypes = argtypes
    for name, argtypes in COUNTS.items():
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = argtypes
    _lib = lib
    return lib


_DEBUG_SYNC = bool(os.environ.get("CMPC_DEBUG_SYNC"))
_DEBUG_TRACE = os.environ.get("CMPC_DEBUG_TRACE")


def call(name: str, *args):
    lib = load()
    if _DEBUG_TRACE:                     # CMPC_DEBUG_TRACE=<file>: last line = the launch in flight when a fault hit
        with open(_DEBUG_TRACE, "a") as f:
            f.write(name + "\n")
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise CmpcError(f"{name} failed ({rc}): {lib.cmpc_last_error().decode()}")
    if _DEBUG_SYNC:                      # CMPC_DEBUG_SYNC=1: surface an asynchronous fault at the launch that caused it
        import torch
        torch.cuda.synchronize()
