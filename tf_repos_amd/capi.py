"""ctypes binding of libdeepctr_hip.so (the C ABI declared in include/deepctr_hip.h).

This is the only module that touches the shared library.  There is NO CPU fallback: if the
library is missing or a call fails, a Python exception is raised (TF-style classes from
tf_repos_amd.errors).  torch is used by callers purely as a device-memory / stream provider:
tensors cross the boundary as raw pointers (`tensor.data_ptr()`).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from . import errors

_HERE = os.path.dirname(os.path.abspath(__file__))
# DCTR_IEEE_ADAM=1: the library built with -DDCTR_IEEE_ADAM (Adam's update term correctly rounded in the table kernels too,
# csrc/opt_rules.h); DCTR_LIB_VARIANT=<name>: an experimental build made by `python -m tf_repos_amd.build --variant <name> ...`
_VARIANT = os.environ.get("DCTR_LIB_VARIANT") or ("ieee" if os.environ.get("DCTR_IEEE_ADAM") == "1" else "")
LIB_PATH = os.path.join(_HERE, "_lib", "libdeepctr_hip%s.so" % ("_" + _VARIANT if _VARIANT else ""))

DCTR_OK = 0
DCTR_ERR_UNSUPPORTED = -6
MODELS = {"deepfm": 0, "fnn": 1, "ipnn": 2, "opnn": 3, "nfm": 4, "afm": 5, "dcn": 6, "wide": 7, "deep": 8, "wide_n_deep": 9, "mvm": 10,
          "din": 11, "esmm": 12}
OPTIMIZERS = {"Adam": 0, "Adagrad": 1, "Momentum": 2, "ftrl": 3}
TABLE_MODES = {"dense_exact": 0, "touched_rows": 1}
GATHER_RAW, GATHER_FM, GATHER_BI = 0, 1, 2
MAX_LAYERS = 8
# dropout sites (include/deepctr_hip.h DCTR_DROPOUT_SITE_*)
SITE_MLP = lambda i: 0x1000 + i          # noqa: E731
SITE_MLP2 = lambda i: 0x2000 + i         # noqa: E731
SITE_NFM_BI, SITE_AFM_ATT, SITE_AFM_YEMB = 0xB1, 0xA0, 0xA1
INPUT_SLOTS = 8


class Config(C.Structure):
    _fields_ = [
        ("model", C.c_int32), ("field_size", C.c_int32), ("embedding_size", C.c_int32),
        ("feature_size", C.c_int64), ("n_deep_layers", C.c_int32),
        ("deep_layers", C.c_int32 * MAX_LAYERS), ("keep_prob", C.c_float * MAX_LAYERS),
        ("cross_layers", C.c_int32), ("n_attention_layers", C.c_int32),
        ("attention_layers", C.c_int32 * MAX_LAYERS), ("l2_reg", C.c_float),
        ("learning_rate", C.c_float), ("optimizer", C.c_int32), ("table_mode", C.c_int32),
        ("batch_norm", C.c_int32), ("batch_norm_decay", C.c_float), ("max_batch", C.c_int32),
        ("seed", C.c_uint64), ("shard_rank", C.c_int32), ("shard_world", C.c_int32),
        ("use_graph", C.c_int32), ("dense_size", C.c_int32), ("lin_optimizer", C.c_int32),
        ("lin_learning_rate", C.c_float), ("loss_sum", C.c_int32), ("max_entries", C.c_int32), ("ctr_task_wgt", C.c_float),
        ("n_att_pairs", C.c_int32), ("att_user_slot", C.c_int32 * 8), ("att_ad_slot", C.c_int32 * 8),
        ("table_sweep_period", C.c_int32), ("batch_norm_biased_moving_variance", C.c_int32), ("gemm_mode", C.c_int32),
    ]


class SlotSpec(C.Structure):
    _fields_ = [("ids_feature", C.c_char_p), ("vals_feature", C.c_char_p), ("fixed_len", C.c_int32)]


_lib: Optional[C.CDLL] = None

_P = C.c_void_p
_SIGS = {
    "dctr_version": ([], C.c_int),
    "dctr_last_error": ([], C.c_char_p),
    "dctr_device_count": ([C.POINTER(C.c_int)], C.c_int),
    "dctr_set_device": ([C.c_int], C.c_int),
    "dctr_malloc": ([C.POINTER(_P), C.c_size_t], C.c_int),
    "dctr_free": ([_P], C.c_int),
    "dctr_memcpy_h2d": ([_P, _P, C.c_size_t, _P], C.c_int),
    "dctr_memcpy_d2h": ([_P, _P, C.c_size_t, _P], C.c_int),
    "dctr_memset": ([_P, C.c_int, C.c_size_t, _P], C.c_int),
    "dctr_stream_sync": ([_P], C.c_int),
    "dctr_parse_libsvm": ([C.c_char_p, C.c_size_t, C.c_int, C.c_int64, _P, _P, _P,
                           C.POINTER(C.c_int64), C.POINTER(C.c_size_t)], C.c_int),
    # (text as void*: bytes objects and mapped files both pass)
    "dctr_parse_libsvm_mt": ([_P, C.c_size_t, C.c_int, C.c_int, _P, _P, _P, C.c_int64, C.POINTER(C.c_int64)], C.c_int),
    "dctr_parse_csv": ([C.c_char_p, C.c_size_t, C.c_int, _P, _P, _P, C.c_int64, _P, _P, C.POINTER(C.c_int64),
                        C.POINTER(C.c_size_t)], C.c_int),
    "dctr_embed_gather_fwd": ([_P, _P, C.c_int64, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int,
                               _P, C.c_int, _P, _P, _P, _P, _P], C.c_int),
    "dctr_embed_gather_strided": ([_P, C.c_int, _P, C.c_int, C.c_int64, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int,
                                   _P, C.c_int, _P, _P, _P, _P, _P], C.c_int),
    "dctr_group_create": ([C.c_int64, C.c_int64, C.c_int, C.POINTER(_P)], C.c_int),
    "dctr_group_destroy": ([_P], C.c_int),
    "dctr_group_ids": ([_P, _P, C.c_int, C.c_int, _P], C.c_int),
    "dctr_group_num_unique": ([_P, C.POINTER(C.c_int32), _P], C.c_int),
    "dctr_group_buffers": ([_P] + [C.POINTER(_P)] * 8, C.c_int),
    "dctr_embed_scatter_bwd": ([_P, _P, C.c_int, _P, C.c_int, _P, _P, _P, _P,
                                C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P], C.c_int),
    "dctr_embed_scatter_apply": ([_P, C.c_int, _P, _P, _P, _P, _P, _P, _P, C.c_float, _P, _P, C.c_int, _P, C.c_int, _P, _P, _P, _P,
                                  C.c_int, C.c_int, C.c_int, C.c_int, _P], C.c_int),
    "dctr_embed_lookup_sparse_fwd": ([_P, C.c_int64, C.c_int, _P, _P, _P, C.c_int, _P, C.c_int, _P, _P], C.c_int),
    "dctr_embed_lookup_sparse_bwd": ([_P, _P, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P], C.c_int),
    "dctr_opt_dense": ([C.c_int, _P, _P, _P, _P, _P, C.c_int, C.c_int64, C.c_int64, C.c_float, _P], C.c_int),
    "dctr_opt_table": ([C.c_int, _P, C.c_int, C.c_int64, C.c_int, _P, _P, _P, _P, _P, _P, _P,
                        C.c_float, _P, _P], C.c_int),
    "dctr_fc_fwd": ([_P, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                     C.c_uint64, _P], C.c_int),
    "dctr_fc_bwd_data": ([_P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int,
                          C.c_float, _P], C.c_int),
    "dctr_fc_bwd_weights": ([_P, C.c_int, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, _P,
                             C.c_size_t, _P], C.c_int),
    "dctr_pnn_inner_fwd": ([_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P], C.c_int),
    "dctr_pnn_inner_bwd": ([_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P], C.c_int),
    "dctr_pnn_outer_fwd": ([_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int64, _P], C.c_int),
    "dctr_pnn_outer_bwd": ([_P, C.c_int, _P, C.c_int64, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P], C.c_int),
    "dctr_pnn_outer_fc_workspace_bytes": ([C.c_int, C.c_int], C.c_size_t),
    "dctr_pnn_outer_fc_fwd": ([_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64, _P, C.c_size_t, _P], C.c_int),
    "dctr_pnn_outer_fc_bwd_weights": ([_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, _P, _P], C.c_int),
    "dctr_pnn_outer_fc_bwd_data": ([_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, _P, C.c_int, _P], C.c_int),
    "dctr_dcn_cross_fwd": ([_P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P], C.c_int),
    "dctr_dcn_cross_bwd": ([_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P,
                            _P, C.c_size_t, _P], C.c_int),
    "dctr_loss_head": ([_P, _P, _P, _P, _P, C.c_int, C.c_float, _P, _P, _P, _P, _P], C.c_int),
    "dctr_auc_update": ([_P, _P, C.c_int, _P, _P], C.c_int),
    "dctr_auc_result": ([_P, C.POINTER(C.c_float), _P], C.c_int),
    "dctr_create": ([C.POINTER(Config), C.POINTER(_P)], C.c_int),
    "dctr_destroy": ([_P], C.c_int),
    "dctr_param_count": ([_P, C.POINTER(C.c_int)], C.c_int),
    "dctr_param_info": ([_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int64 * 4)], C.c_int),
    "dctr_param_set": ([_P, C.c_char_p, _P, C.c_size_t], C.c_int),
    "dctr_param_get": ([_P, C.c_char_p, _P, C.c_size_t], C.c_int),
    "dctr_param_grad_get": ([_P, C.c_char_p, _P, C.c_size_t], C.c_int),
    "dctr_afm_fwd": ([_P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P], C.c_int),
    "dctr_afm_bwd": ([_P, _P, C.c_int, C.c_int, _P, C.c_int, _P], C.c_int),
    "dctr_main_stream": ([_P, C.POINTER(_P)], C.c_int),
    "dctr_slot_get": ([_P, C.c_char_p, C.c_int, _P, C.c_size_t], C.c_int),
    "dctr_slot_set": ([_P, C.c_char_p, C.c_int, _P, C.c_size_t], C.c_int),
    "dctr_param_device_ptr": ([_P, C.c_char_p, C.POINTER(_P)], C.c_int),
    "dctr_step_timer_layer": ([_P, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)], C.c_int),
    "dctr_param_device_view": ([_P, C.c_char_p, C.POINTER(_P), C.POINTER(C.c_int64)], C.c_int),
    "dctr_set_global_step": ([_P, C.c_int64], C.c_int),
    "dctr_get_global_step": ([_P, C.POINTER(C.c_int64)], C.c_int),
    "dctr_train_step": ([_P, _P, _P, _P, C.c_int, C.POINTER(C.c_float), _P], C.c_int),
    "dctr_predict": ([_P, _P, _P, C.c_int, _P, _P, _P], C.c_int),
    "dctr_parse_csv_mt": ([_P, C.c_size_t, C.c_int, _P, _P, _P, C.c_int, _P, _P, C.c_int64, C.POINTER(C.c_int64)], C.c_int),
    "dctr_prefetch_ids": ([_P, _P, C.c_int], C.c_int),
    "dctr_prefetch_cancel": ([_P], C.c_int),
    "dctr_tables_sync": ([_P, _P], C.c_int),
    "dctr_input_slot_rewrite": ([_P, C.c_int], C.c_int),
    "dctr_input_slot_fill": ([_P, C.c_int, _P, _P, _P, C.c_int], C.c_int),
    "dctr_input_slot_acquire": ([_P, C.c_int, _P], C.c_int),
    "dctr_input_slot_release": ([_P, C.c_int, _P], C.c_int),
    "dctr_input_slot_wait_released": ([_P, C.c_int], C.c_int),
    "dctr_input_slot_ready": ([_P, C.c_int, C.POINTER(C.c_int)], C.c_int),
    "dctr_crc32c": ([C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_uint32)], C.c_int),
    "dctr_tfrecord_scan": ([C.c_char_p, C.c_size_t, C.c_int64, C.c_int, _P, _P, C.POINTER(C.c_int64), C.POINTER(C.c_size_t)], C.c_int),
    "dctr_tfrecord_frame": ([C.c_char_p, C.c_size_t, _P], C.c_int),
    "dctr_examples_to_slot_csr": ([C.c_char_p, _P, _P, C.c_int64, C.POINTER(SlotSpec), C.c_int, C.POINTER(C.c_char_p), C.c_int,
                                   C.c_int64, C.c_int64, _P, _P, _P, _P, C.POINTER(C.c_int64)], C.c_int),
    "dctr_train_step_csr": ([_P, _P, _P, _P, C.c_int, _P, _P, C.c_int, C.POINTER(C.c_float), _P], C.c_int),
    "dctr_predict_csr": ([_P, _P, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P], C.c_int),
    "dctr_eval_batch_csr": ([_P, _P, _P, _P, C.c_int, _P, _P, C.c_int, _P], C.c_int),
    "dctr_eval_auc_extra": ([_P, C.c_int, C.POINTER(C.c_float), _P], C.c_int),
    "dctr_eval_reset": ([_P, _P], C.c_int),
    "dctr_eval_batch": ([_P, _P, _P, _P, C.c_int, _P], C.c_int),
    "dctr_eval_result": ([_P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int64), _P], C.c_int),
    "dctr_input_slot": ([_P, C.c_int, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)], C.c_int),
    "dctr_check_ids": ([_P, _P], C.c_int),
    "dctr_route_unique": ([_P, C.c_int, _P, _P, _P, _P], C.c_int),
    "dctr_entry_index": ([_P, _P, C.c_int, _P, _P, _P], C.c_int),
    "dctr_table_gather_packed": ([_P, _P, C.c_int, _P, _P], C.c_int),
    "dctr_table_group_rows": ([_P, C.c_int, _P, C.c_int, _P], C.c_int),
    "dctr_table_apply_packed": ([_P, C.c_int, C.c_int, _P, _P], C.c_int),
    "dctr_sharded_forward_backward": ([_P, _P, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P], C.c_int),
    "dctr_sharded_pack_row_grads": ([_P, _P, C.c_int, _P, _P, _P], C.c_int),
    "dctr_dense_grads": ([_P, C.POINTER(_P), C.POINTER(C.c_int64), _P], C.c_int),
    "dctr_dense_apply": ([_P, _P], C.c_int),
    "dctr_read_scalars": ([_P, C.POINTER(C.c_float * 4), _P], C.c_int),
    "dctr_last_outputs": ([_P, C.POINTER(_P), C.POINTER(_P)], C.c_int),
    "dctr_time_kernel": ([_P, C.c_char_p, C.c_int, C.POINTER(C.c_float), _P], C.c_int),
    "dctr_step_timer": ([_P, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)], C.c_int),
    "dctr_debug_tensor": ([_P, C.c_char_p, C.POINTER(_P), C.POINTER(C.c_int64), C.POINTER(C.c_int)], C.c_int),
    "dctr_set_dense_input": ([_P, _P], C.c_int),
    "dctr_measure_copy_bw": ([C.c_size_t, C.c_int, C.POINTER(C.c_float), _P], C.c_int),
    "dctr_rccl_unique_id": ([C.c_char_p, C.c_char_p], C.c_int),
    "dctr_dist_create_rccl": ([_P, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.POINTER(_P)], C.c_int),
    "dctr_dist_create": ([_P, C.c_int, C.c_int, _P, C.POINTER(_P)], C.c_int),
    "dctr_dist_destroy": ([_P], C.c_int),
    "dctr_dist_train_step": ([_P, _P, _P, _P, C.c_int, _P, C.c_int, C.POINTER(C.c_float), _P], C.c_int),
    "dctr_dist_predict": ([_P, _P, _P, C.c_int, _P, _P], C.c_int),
}

RCCL_ID_BYTES = 128
# dctr_transport (include/deepctr_hip.h): collective callbacks for the native sharded-step driver
ALL_GATHER_I32_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p)
ALL_TO_ALL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64),
                            C.c_int64, C.c_void_p)
ALL_REDUCE_F32_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p)


class Transport(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("all_gather_i32", ALL_GATHER_I32_FN), ("all_to_all", ALL_TO_ALL_FN),
                ("all_reduce_f32", ALL_REDUCE_F32_FN)]


_SIGS["dctr_dropout_mask"] = ([C.c_uint64, C.c_int64, C.c_uint64, C.c_int64, C.c_float, _P], C.c_int)
_SIGS["dctr_gemm_plan"] = ([C.c_char, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int], C.c_int)
_SIGS["dctr_gemm_split_plane_bytes"] = ([C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)], C.c_int)
_SIGS["dctr_gemm_wsplit"] = ([_P, C.c_int, C.c_int, _P, _P, _P], C.c_int)
_SIGS["dctr_fc_fwd_split"] = ([_P, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64, _P], C.c_int)
_SIGS["dctr_fc_bwd_data_split"] = ([_P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_float, _P], C.c_int)
_SIGS["dctr_fc_bwd_weights_split"] = ([_P, C.c_int, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_size_t, _P], C.c_int)
_SIGS["dctr_gemm_split_launches"] = ([], C.c_int64)
_SIGS["dctr_ts_plane_bytes"] = ([C.c_int, C.c_int, _P], C.c_int)
_SIGS["dctr_fc_fwd_dot_split"] = ([_P, C.c_int, _P, _P, _P, C.c_int, C.c_int64, C.c_int, C.c_int, _P, _P, _P, _P, _P], C.c_int)
_SIGS["dctr_fc_bwd_weights_gate_split"] = ([_P, C.c_int, _P, C.c_int, _P, _P, _P, _P, _P, C.c_int64, C.c_int, C.c_int, _P, C.c_size_t, _P], C.c_int)
_SIGS["dctr_pairs_fc_fwd_dot_split"] = ([_P, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, _P, C.c_int, C.c_int64, C.c_int, C.c_int, _P, _P, _P, _P, _P], C.c_int)
_SIGS["dctr_pairs_fc_bwd_weights_gate_split"] = ([_P, C.c_int, C.c_int, _P, _P, C.c_int, _P, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_int, C.c_int,
                                                  _P, C.c_size_t, _P], C.c_int)
_SIGS["dctr_fc_bwd_data_gate_split"] = ([_P, C.c_int, _P, _P, _P, _P, _P, C.c_int, C.c_int64, C.c_int, C.c_int, _P, _P], C.c_int)
_SIGS["dctr_set_stat_sync"] = ([_P, ALL_REDUCE_F32_FN, _P, C.c_int], C.c_int)

DECLARED_SYMBOLS = tuple(_SIGS)


def lib() -> C.CDLL:
    """Loads the HIP library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise errors.NotFoundError(
                "libdeepctr_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
        # PyTorch-ROCm bundles its own libamdhip64 under the same SONAME this library links against: whichever copy is loaded
        # first serves the whole process.  If ours (/opt/rocm) came first, torch would come up on a runtime it was not built
        # for and report "No HIP GPUs are available" -- so when torch is installed it is imported before the library.
        if os.environ.get("DCTR_NO_TORCH_PRELOAD") is None:
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        l = C.CDLL(LIB_PATH)
        for name, (args, res) in _SIGS.items():
            fn = getattr(l, name)          # AttributeError if the ABI and the header drift apart
            fn.argtypes = args
            fn.restype = res
        _lib = l
    return _lib


def last_error() -> str:
    return (lib().dctr_last_error() or b"").decode()


_EXC = {-1: errors.InvalidArgumentError, -2: errors.InternalError, -3: errors.NotFoundError,
        -4: errors.OutOfRangeError, -5: errors.InvalidArgumentError, -6: errors.UnimplementedError}


def check(rc: int) -> None:
    if rc != DCTR_OK:
        raise _EXC.get(rc, errors.InternalError)("%s (status %d)" % (last_error(), rc))


def ptr(t) -> C.c_void_p:
    """Raw device (or host) pointer of a torch tensor / numpy array / None."""
    if t is None:
        return C.c_void_p(0)
    if hasattr(t, "data_ptr"):
        return C.c_void_p(t.data_ptr())
    if hasattr(t, "ctypes"):
        return C.c_void_p(t.ctypes.data)
    return C.c_void_p(int(t))


def current_stream() -> C.c_void_p:
    """The hipStream_t torch is currently launching on (so torch ops and our kernels order)."""
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
