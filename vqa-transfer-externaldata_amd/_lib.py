"""ctypes binding of libvqahot.so (include/vqa_hot.h).  Fails loudly when the
library is absent: the product path has no fallback."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvqahot.so")


class VqaHotError(RuntimeError):
    pass


def lib_path():
    return _LIB_PATH


class Dims(C.Structure):
    _fields_ = [("B", C.c_int32), ("R", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("T", C.c_int32),
                ("W", C.c_int32), ("A", C.c_int32), ("Vq", C.c_int32), ("N_img", C.c_int64),
                ("model_type", C.c_int32), ("keep_att", C.c_float), ("keep_joint", C.c_float),
                ("inv_global_batch", C.c_float), ("flags", C.c_int32),
                ("num_marginal", C.c_int32), ("ent_cols", C.c_int32), ("extra_weight", C.c_float),
                ("map_dim", C.c_int32), ("La", C.c_int32)]


FLAG_DETERMINISTIC = 1
FLAG_FUSED_GATHER = 2
FLAG_SHARED_LN = 4


class Fc(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("beta", C.c_void_p), ("gamma", C.c_void_p)]


class Params(C.Structure):
    _fields_ = [("embed", C.c_void_p), ("v_linear_v", Fc),
                ("gru_wg", C.c_void_p), ("gru_bg", C.c_void_p), ("gru_wc", C.c_void_p), ("gru_bc", C.c_void_p),
                ("q_linear_v", Fc), ("score", Fc), ("pooled_linear_l", Fc), ("q_linear_l", Fc),
                ("joint_fc", Fc), ("head", Fc), ("answer_glove", C.c_void_p), ("head2", Fc), ("joint2", Fc),
                ("q_L_ft2", Fc), ("q_L_mean", Fc), ("q_L_log_sigma_sq", Fc), ("v_adapt", Fc),
                ("embed2", C.c_void_p), ("gru_bw_wg", C.c_void_p), ("gru_bw_bg", C.c_void_p), ("gru_bw_wc", C.c_void_p),
                ("gru_bw_bc", C.c_void_p), ("q_att_key", Fc), ("q_att_query", Fc), ("word_score", Fc), ("v_word_fc", Fc),
                ("glove_fixed", C.c_void_p), ("glove_learn", C.c_void_p), ("lstm_k", C.c_void_p), ("lstm_b", C.c_void_p),
                ("l2v", Fc * 3), ("v2l", Fc * 3), ("answer_layer1", Fc), ("pooled_layer1", Fc), ("q_layer1", Fc),
                ("classifier", Fc)]


class Batch(C.Structure):
    _fields_ = [("table", C.c_void_p), ("nbox_table", C.c_void_p), ("image_idx", C.c_void_p),
                ("q_intseq", C.c_void_p), ("q_intseq_len", C.c_void_p), ("answer_target", C.c_void_p),
                ("train_mask", C.c_void_p), ("obj_mask", C.c_void_p), ("attr_mask", C.c_void_p),
                ("exist_mask", C.c_void_p), ("keep_att", C.c_void_p), ("keep_joint", C.c_void_p),
                ("keep_joint2", C.c_void_p), ("live_rows", C.c_void_p), ("noise", C.c_void_p), ("keep_tile", C.c_void_p),
                ("keep_word", C.c_void_p), ("answer_intseq", C.c_void_p), ("answer_intseq_len", C.c_void_p)]


class PtDims(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("B", "n", "R", "D", "H", "W", "A", "Vq", "n_ws", "L", "flags")] + \
               [("keep_att", C.c_float), ("keep_joint", C.c_float), ("global_valid", C.c_float * 2)]


class PtFc(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("beta", C.c_void_p * 4), ("gamma", C.c_void_p * 4)]


class PtParams(C.Structure):
    _fields_ = [("wordset_map", C.c_void_p), ("l_glove", C.c_void_p), ("spat_v_linear_v", PtFc),
                ("spat_q_linear_v", PtFc), ("spat_att_score", PtFc), ("gru_wg", C.c_void_p), ("gru_bg", C.c_void_p),
                ("gru_wc", C.c_void_p), ("gru_bc", C.c_void_p), ("pooled_linear_l", PtFc), ("q_linear_l", PtFc),
                ("joint_fc", PtFc), ("wordset_ft", PtFc), ("classifier", PtFc)]


class PtKind(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("normal_boxes", "fills", "blanks", "blanks_len", "wordsets", "num", "keep_att",
                                          "keep_bf_joint", "keep_ws_joint")]


class PtBatch(C.Structure):
    _fields_ = [("image_ft", C.c_void_p), ("spatial_ft", C.c_void_p), ("num_boxes", C.c_void_p), ("kind", PtKind * 2),
                ("perm", C.c_void_p), ("inv", C.c_void_p), ("live_rows", C.c_void_p)]


_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes); every symbol declared in include/vqa_hot.h
SIGNATURES = {
    "vqa_hot_version": (_I, []),
    "vqa_hot_error_string": (C.c_char_p, [_I]),
    "vqa_gather_features": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _L, _P]),
    "vqa_embed_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "vqa_gru_pack_wx": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "vqa_gru_unpack_dwx": (_I, [_P, _P, _P, _I, _I, _P]),
    "vqa_embed_fwd_ld": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "vqa_gru_unpack_dwx_bias": (_I, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "vqa_embed_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "vqa_embed_bwd_len": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "vqa_embed_bwd_len_det": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "vqa_set_deterministic": (_I, [_I]),
    "vqa_gemm_f32": (_I, [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _P, _L, _P]),
    "vqa_gemm_workspace_floats": (_L, [_I, _I, _I, _I, _I, _I]),
    "vqa_gemm_f32_ex": (_I, [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _P, _L, _I, _P]),
    "vqa_gemm_f32_gather": (_I, [_I, _I, _I, _P, _I, _P, _I, _L, _P, _I, _P, _I, _P, _P, _I, _P]),
    "vqa_gemm_set_tall_config": (_I, [_I]),
    "vqa_gemm_set_max_blocks": (_I, [_I]),
    "vqa_gemm_set_order": (_I, [_I]),
    "vqa_gemm_set_config": (_I, [_I]),
    "vqa_conv_set_config": (_I, [_I]),
    "vqa_gemm_set_gru_config": (_I, [_I]),
    "vqa_gru_seq_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "vqa_gru_seq_fwd_persistent": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "vqa_gru_fwd_persistent_supported": (_I, [_I, _I, _I]),
    "vqa_gru_persistent_sync_bytes": (_L, []),
    "vqa_gru_set_persistent": (_I, [_I]),
    "vqa_gru_persistent_set_census": (_I, [_P]),
    "vqa_gru_seq_fwd_ws": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "vqa_gru_ws_supported": (_I, [_I, _I, _I]),
    "vqa_gru_seq_bwd_ws": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "vqa_gru_ws_bwd_supported": (_I, [_I, _I, _I]),
    "vqa_gru_ws_workspace_bytes": (_L, [_I]),
    "vqa_gru_ws_set_mode": (_I, [_I]),
    "vqa_gru_ws_set_form": (_I, [_I]),
    "vqa_gru_ws_set_stamps": (_I, [_P]),
    "vqa_gru_seq_fwd_rows": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "vqa_gru_seq_bwd_rows": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "vqa_gru_seq_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "vqa_gru_seq_fwd_live": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "vqa_gru_seq_bwd_live": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "vqa_gru_fill_finished": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "vqa_gru_zero_finished": (_I, [_P, _P, _I, _I, _I, _P]),
    "vqa_ln_set_fast": (_I, [_I]),
    "vqa_ln_relu_fwd": (_I, [_P, _P, _P, _P, _F, _P, _P, _P, _I, _I, _I, _P]),
    "vqa_ln_relu_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _I, _I, _I, _P]),
    "vqa_ln_act_fwd": (_I, [_P, _P, _P, _P, _F, _P, _P, _P, _I, _I, _I, _I, _P]),
    "vqa_ln_act_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "vqa_ln_pair_mul_supported": (_I, [_I, _P, _I]),
    "vqa_ln_pair_mul_fwd": (_I, [_P] * 13 + [_I, _I, _P]),
    "vqa_ln_pair_mul_bwd": (_I, [_P] * 20 + [_I, _I, _P]),
    "vqa_tanh_fwd": (_I, [_P, _P, _L, _P]),
    "vqa_tanh_bwd": (_I, [_P, _P, _P, _L, _P]),
    "vqa_softmax_ce_fwd": (_I, [_P, _P, _P, _I, _P, _P, _P, _I, _I, _P]),
    "vqa_softmax_set_fast": (_I, [_I]),
    "vqa_colsum": (_I, [_P, _I, _I, _I, _P, _P, _L, _P]),
    "vqa_colsum_workspace_floats": (_L, [_I, _I]),
    "vqa_colsum3": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _L, _P]),
    "vqa_colsum_acc": (_I, [_P, _I, _I, _I, _P, _I, _P, _L, _P]),
    "vqa_colsum3_acc": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _I, _P, _L, _P]),
    "vqa_mul": (_I, [_P, _P, _P, _L, _P]),
    "vqa_mul_bwd": (_I, [_P, _P, _P, _P, _P, _L, _P]),
    "vqa_add_inplace": (_I, [_P, _P, _L, _P]),
    "vqa_gru_gates_fwd": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _P]),
    "vqa_gru_cand_fwd": (_I, [_P, _I, _P, _P, _P, _I, _P, _P, _I, _I, _P]),
    "vqa_gru_bwd_a": (_I, [_P, _P, _P, _P, _P, _I, _P, _I, _P, _I, _P, _I, _I, _P]),
    "vqa_gru_bwd_b": (_I, [_P, _P, _P, _P, _I, _P, _I, _I, _P]),
    "vqa_attn_set_fast": (_I, [_I]),
    "vqa_attn_pool_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _I, _I, _P]),
    "vqa_attn_pool_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "vqa_attn_pool_fwd_rep": (_I, [_P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _I, _I, _I, _P]),
    "vqa_attn_pool_bwd_rep": (_I, [_P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "vqa_loss_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _F, _P, _P, _P, _I, _I, _P]),
    "vqa_loss2_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "vqa_rowmin_mask_fwd": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "vqa_rowmin_mask_bwd": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "vqa_report_reduce": (_I, [_P, _I, _P, _P]),
    "vqa_report_key": (C.c_char_p, [_I]),
    "vqa_sumsq": (_I, [_P, _L, _P, _P, _P, _L, _P]),
    "vqa_sumsq_workspace_floats": (_L, [_L]),
    "vqa_clip_adam": (_I, [_P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _F, _P]),
    "vqa_dropout_mask": (_I, [_P, _L, C.c_uint64, C.c_uint64, _F, _P]),
    "vqa_clip_adam_dev": (_I, [_P, _P, _P, _P, _L, _P, _F, _P, _F, _F, _F, _P]),
    "vqa_embed2_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "vqa_embed2_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "vqa_lstm_step_fwd": (_I, [_P, _P, _P, _P, _I, _P, _P, _I, _I, _P]),
    "vqa_lstm_step_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _P]),
    "vqa_relu_fwd": (_I, [_P, _P, _L, _P]),
    "vqa_relu_bwd": (_I, [_P, _P, _P, _L, _P]),
    "vqa_fill": (_I, [_P, _L, _F, _P]),
    "vqa_score_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "vqa_score_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "vqa_reverse_tokens": (_I, [_P, _P, _P, _I, _I, _P]),
    "vqa_bi_outputs_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "vqa_bi_outputs_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "vqa_bi_dx_combine": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "vqa_gru_seq_bwd_outs": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "vqa_adam_lr_step": (_I, [_P, _P, C.c_double, C.c_double, _P, _P]),
    "vqa_graph_capture_begin": (_I, [_P]),
    "vqa_graph_capture_end": (_I, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "vqa_graph_capture_abort": (_I, [_P]),
    "vqa_graph_launch": (_I, [_P, _P]),
    "vqa_graph_destroy": (_I, [_P]),
    "vqa_stream_is_capturing": (_I, [_P]),
    "vqa_conv2d_nhwc": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P]),
    "vqa_im2col_nhwc": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, C.POINTER(C.c_float), _P, _I, _P]),
    "vqa_pad_c3c4_nhwc": (_I, [_P, _I, _I, _I, C.POINTER(C.c_float), _P, _P]),
    "vqa_maxpool3x3s2_same_nhwc": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "vqa_subsample_nhwc": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "vqa_crop_and_resize_nhwc": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _P, _P]),
    "vqa_probe_enable": (_I, [C.c_char_p, _I]),
    "vqa_probe_read": (_I, [C.POINTER(C.c_float), _I, C.POINTER(C.c_int)]),
    "vqa_probe_read_label": (_I, [C.c_char_p, C.POINTER(C.c_float), _I, C.POINTER(_I)]),
    "vqa_probe_labels": (_I, [C.c_char_p, _I]),
    "vqa_roctx_enable": (_I, [_I]),
    "vqa_stream_delay_us": (_I, [_F, _P]),
    "vqa_reparam_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "vqa_reparam_bwd": (_I, [_P, _P, _P, _P, _F, _P, _P, _L, _P]),
    "vqa_outer_rows": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "vqa_tile_mul_fwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "vqa_tile_mul_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "vqa_marginal_entropy": (_I, [_P, _P, _P, _F, _P, _P, _I, _I, _I, _I, _I, _P]),
    "vqa_extra_report": (_I, [_P, _P, _I, _F, _P, _P]),
    "vqa_normal_noise": (_I, [_P, _L, C.c_uint64, C.c_uint64, _P]),
    "vqa_conv2d_bwd_workspace_floats": (_L, [_I, _I, _I, _I, _I, _I, _I, _I]),
    "vqa_conv2d_nhwc_bwd": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _L, _P]),
    "vqa_clock_sample": (_I, [_F, _I, _I, _P, _P]),
    "vqa_gemm_shortk_supported": (_I, [_I, _I, _I, _I, _I, _I]),
    "vqa_gemm_shortk_nn": (_I, [_I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _P]),
    "vqa_gemm_shortk_set_grid": (_I, [_I]),
    "vqa_gemm_shortk_set_waves": (_I, [_I]),
    "vqa_gemm_shortk_set_mode": (_I, [_I]),
    "vqa_gemm_bf16x3_supported": (_I, [_I, _I, _I]),
    "vqa_gemm_bf16x3_nn": (_I, [_I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P]),
    "vqa_gemm_bf16x3_workspace_floats": (_L, [_I, _I, _I, _I]),
    "vqa_gemm_bf16x3": (_I, [_I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _L, _P]),
    "vqa_gemm_bf16x3_set_mode": (_I, [_I]),
    "vqa_probe_disable": (_I, []),
    "vqa_fusion_workspace_bytes": (_L, [C.POINTER(Dims)]),
    "vqa_fusion_tensor": (_I, [C.POINTER(Dims), C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "vqa_fusion_forward": (_I, [C.POINTER(Dims), C.POINTER(Params), C.POINTER(Batch), _P, _L, _I, _P]),
    "vqa_fusion_backward_phases": (_I, [C.POINTER(Dims), C.POINTER(Params), C.POINTER(Params), C.POINTER(Batch), _P, _L,
                                        _P, _I, _P]),
    "vqa_pretrain_workspace_bytes": (_L, [C.POINTER(PtDims)]),
    "vqa_pretrain_tensor": (_I, [C.POINTER(PtDims), C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "vqa_pretrain_report_key": (C.c_char_p, [_I]),
    "vqa_pretrain_forward": (_I, [C.POINTER(PtDims), C.POINTER(PtParams), C.POINTER(PtBatch), _P, _L, _I, _P]),
    "vqa_pretrain_backward": (_I, [C.POINTER(PtDims), C.POINTER(PtParams), C.POINTER(PtParams), C.POINTER(PtBatch), _P, _L,
                                   _P, _P]),
    "vqa_pretrain_backward_phases": (_I, [C.POINTER(PtDims), C.POINTER(PtParams), C.POINTER(PtParams),
                                          C.POINTER(PtBatch), _P, _L, _P, _I, _P]),
    "vqa_fusion_backward": (_I, [C.POINTER(Dims), C.POINTER(Params), C.POINTER(Params), C.POINTER(Batch), _P, _L,
                                 _P, _P]),
}

ABI_VERSION = 5      # VQA_HOT_ABI_VERSION of include/vqa_hot.h

_lib = None


def load():
    """Returns the loaded CDLL with typed signatures; raises VqaHotError if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise VqaHotError(
            "libvqahot.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the hot path)" % _LIB_PATH)
    lib = C.CDLL(_LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError if the ABI and the header drift apart
        fn.restype = res
        fn.argtypes = args
    if lib.vqa_hot_version() != ABI_VERSION:
        raise VqaHotError("libvqahot.so ABI version %d != %d (rebuild: python -c 'import __graft_entry__ as g; "
                          "g.build()')" % (lib.vqa_hot_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().vqa_hot_error_string(rc)
        raise VqaHotError("%s failed: %s (%d)" % (what, msg.decode() if msg else "?", rc))
