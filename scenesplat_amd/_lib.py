"""ctypes binding of libscenesplat_hip.so (the C-ABI declared in include/scenesplat_hip.h).

Loading fails loudly: there is no CPU fallback behind these symbols."""
import ctypes
import os

# torch first: PyTorch-ROCm ships its own libamdhip64; it must be the HIP runtime of the process
# before this library (linked against the same soname) is mapped, or kernels and streams would
# belong to two different runtimes and every launch fails.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libscenesplat_hip.so")

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_i64 = ctypes.c_int64
c_f = ctypes.c_float
c_sz = ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol include/scenesplat_hip.h declares
PROTOTYPES = {
    "ss_version": (c_i, []),
    "ss_serialize_encode": (c_i, [c_p, c_p, c_i64, c_i, ctypes.POINTER(c_i), c_i, c_p, c_p]),
    "ss_offsets_to_batch": (c_i, [c_p, c_i, c_i64, c_p, c_p]),
    "ss_count_duplicates": (c_i, [c_p, c_i64, c_p, c_p]),
    "ss_grid_coord_max": (c_i, [c_p, c_i64, c_p, c_p]),
    "ss_argsort_workspace_bytes": (c_sz, [c_i64, c_i]),
    "ss_argsort_i64": (c_i, [c_p, c_i, c_i64, c_i, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "ss_pool_partition_workspace_bytes": (c_sz, [c_i64]),
    "ss_pool_partition": (c_i, [c_p, c_p, c_i64, c_i, c_p, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "ss_pool_level_attrs": (c_i, [c_p, c_i64, c_i64, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p]),
    "ss_batch_offsets": (c_i, [c_p, c_p, c_i64, c_i, c_p, c_p]),
    "ss_window_index": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i64, c_p, c_p, c_p]),
    "ss_window_attn_fwd": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i64, c_i64, c_i, c_i, c_f, c_i, c_i, c_p, c_p, c_p]),
    "ss_window_attn_bwd_workspace_bytes": (c_sz, [c_i64, c_i64, c_i, c_i, c_i]),
    "ss_window_attn_bwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i64, c_i64, c_i, c_i, c_f, c_i, c_i,
                                 c_p, c_p, c_sz, c_p]),
    "ss_gelu": (c_i, [c_p, c_p, c_p, c_i64, c_i, c_p]),
    "ss_row_keep_scales": (c_i, [c_p, c_p, c_p, c_i64, c_p]),
    "ss_stream_capture_status": (c_i, [c_p]),
    "ss_stream_capture_id": (ctypes.c_ulonglong, [c_p]),
    "ss_cast_bf16_group": (c_i, [c_p, c_p, c_i, c_i, c_p]),
    "ss_cast_bf16_group_elems_per_workgroup": (c_i, []),
    "ss_linear_fwd_headmajor": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_i, c_f, c_p]),
    "ss_headmajor_pack": (c_i, [c_p, c_i, c_p, c_p, c_i64, c_i, c_i, c_i, c_f, c_p]),
    "ss_window_attn_hm_fwd": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i64, c_i64, c_i, c_i, c_p, c_p, c_p]),
    "ss_window_attn_hm_bwd_workspace_bytes": (c_sz, [c_i64, c_i64, c_i, c_i]),
    "ss_window_attn_hm_bwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i64, c_i64, c_i, c_i, c_f, c_p, c_p, c_sz, c_p]),
    "ss_subm_rulebook": (c_i, [c_p, c_p, c_i64, c_i, c_p, c_p, c_i, c_i, c_p, c_p]),
    "ss_subm_conv_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_i, c_p]),
    "ss_stream_create_cu_mask": (c_i, [c_i, c_p, c_p]),
    "ss_ln_add_ln_fwd": (c_i, [c_p, c_i, c_p, c_i, c_p, c_p, c_f, c_p, c_p, c_f, c_p, c_p, c_i, c_p, c_i64, c_i, c_p]),
    "ss_ln_add_ln_bwd": (c_i, [c_p, c_p, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_i64, c_i, c_i, c_p]),
    "ss_segment_minmax": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_p]),
    "ss_segment_minmax_bwd": (c_i, [c_p, c_p, c_p, c_i64, c_i, c_i, c_p]),
    "ss_subm_rulebook_table_size": (c_i64, [c_i64]),
    "ss_subm_rulebook_hashed": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_p, c_p, ctypes.c_size_t, c_p]),
    "ss_subm_tap_mask_keys": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_p, c_p]),
    "ss_subm_weight_mirror": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p]),
    "ss_subm_im2col": (c_i, [c_p, c_p, c_p, c_i64, c_i, c_i64, c_p]),
    "ss_gemm8_ok": (c_i, [c_i64, c_i, c_i, c_i]),
    "ss_subm_conv_fwd_pipe": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_i, c_p]),
    "ss_linear_fwd": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_p]),
    "ss_wgrad_xcd_order": (c_i, []),
    "ss_wgrad_set_xcd_order": (c_i, [c_i]),
    "ss_wgrad8_ok": (c_i, [c_i64, c_i, c_i, c_i]),
    "ss_subm_conv_wgrad_pipe": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_p]),
    "ss_linear_wgrad": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_p]),
    "ss_linear_wgrad_group_plan": (c_i, [c_i64, c_i, c_i, c_p]),
    "ss_linear_wgrad_tiles": (c_i, [c_i, c_i]),
    "ss_linear_wgrad_group_plan2": (c_i, [c_i64, c_i, c_i, c_i, c_p]),
    "ss_linear_wgrad_group": (c_i, [c_p, c_p, c_i, c_i, c_p]),
    "ss_subm_conv_splits": (c_i, [c_i64, c_i, c_i]),
    "ss_subm_conv_fwd_splitk": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_i, c_p]),
    "ss_subm_block_lists": (c_i, [c_p, c_p, c_i64, c_i, c_p, c_p, c_p]),
    "ss_subm_conv_wgrad": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_p]),
    "ss_subm_conv_fwd_pipe_walk": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_i, c_p]),
    "ss_subm_conv_fwd_uses_pipe": (c_i, [c_i64, c_i, c_i, c_i]),
    "ss_subm_conv_fwd_walk": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_i, c_p]),
    "ss_subm_conv_wgrad_pipe_walk": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_p]),
    "ss_subm_conv_wgrad_uses_pipe": (c_i, [c_i64, c_i, c_i, c_i]),
    "ss_subm_conv_wgrad_walk": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_p]),
    "ss_subm_f32_ok": (c_i, [c_i, c_i]),
    "ss_subm_f32_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_p]),
    "ss_subm_f32_dgrad_dup": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_p]),
    "ss_dup_fold_rows": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_p]),
    "ss_dup_zero_rows": (c_i, [c_p, c_p, c_p, c_i64, c_i64, c_p]),
    "ss_subm_f32_wgrad": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_i, c_p]),
    "ss_add_layernorm_fwd": (c_i, [c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_f, c_p, c_i, c_p, c_p, c_i, c_p, c_p, c_i64, c_i, c_p]),
    "ss_add_layernorm_bwd_blocks": (c_i, [c_i64]),
    "ss_group_partial_sums_outputs_per_workgroup": (c_i, []),
    "ss_group_partial_sums": (c_i, [c_p, c_p, c_i, c_i, c_p]),
    "ss_transpose16_group": (c_i, [c_p, c_p, c_i, c_i, c_p]),
    "ss_subm_weight_mirror_group_tile": (c_i, []),
    "ss_subm_weight_mirror_group": (c_i, [c_p, c_p, c_i, c_i, c_p]),
    "ss_add_layernorm_bwd": (c_i, [c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_p,
                                   c_i64, c_i, c_i, c_p]),
    "ss_col_stats": (c_i, [c_p, c_i, c_p, c_p, c_p, c_i64, c_i, c_i, c_p]),
    "ss_bn_stats_finish": (c_i, [c_p, c_p, c_i, c_i, c_i64, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p]),
    "ss_bn_bwd_finish": (c_i, [c_p, c_i, c_i, c_i64, c_p, c_p, c_p]),
    "ss_bn_act_fwd": (c_i, [c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_i64, c_i, c_p]),
    "ss_bn_act_bwd_reduce": (c_i, [c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_i64, c_i, c_i, c_p]),
    "ss_bn_act_bwd_apply": (c_i, [c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_i64, c_i, c_p]),
    "ss_feat_text_scan": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    "ss_lang_head_blocks": (c_i, [c_i64]),
    "ss_lang_head_fwd": (c_i, [c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_i64, c_i, c_p]),
    "ss_lang_head_bwd": (c_i, [c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_i, c_i64, c_i, c_p]),
    "ss_gather_rows": (c_i, [c_p, c_p, c_p, c_i64, c_i64, c_p]),
    "ss_scatter_rows": (c_i, [c_p, c_p, c_p, c_i64, c_i64, c_p]),
    "ss_gather_add_rows": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_p]),
    "ss_segment_reduce": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_p]),
    "ss_segment_bcast": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_p]),
    "ss_knn_grid_table_size": (c_i64, [c_i64]),
    "ss_knn_grid_workspace_bytes": (c_sz, [c_i64]),
    "ss_knn_grid_keys": (c_i, [c_p, c_p, c_i, c_i64, c_f, c_f, c_f, c_f, c_p, c_p]),
    "ss_knn_grid_build": (c_i, [c_p, c_p, c_p, c_i64, c_p, c_sz, c_p, c_p]),
    "ss_knn_grid_query": (c_i, [c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i64, c_p, c_p, c_p, c_p]),
    "ss_ball_grid_query": (c_i, [c_i, c_i, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_i, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i64, c_p, c_p, c_p, c_p]),
    "ss_knn_query": (c_i, [c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p]),
    "ss_ball_query_workspace_bytes": (c_sz, [c_i]),
    "ss_ball_query": (c_i, [c_i, c_i, c_f, c_f, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_sz, c_p]),
    "ss_random_ball_query": (c_i, [c_i, c_i, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p]),
    "ss_farthest_point_sampling": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "ss_grouping_fwd": (c_i, [c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    "ss_grouping_bwd": (c_i, [c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    "ss_subtraction_fwd": (c_i, [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    "ss_subtraction_bwd": (c_i, [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    "ss_aggregation_fwd": (c_i, [c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "ss_aggregation_bwd": (c_i, [c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "ss_interpolation_fwd": (c_i, [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    "ss_interpolation_bwd": (c_i, [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    "ss_attention_relation_fwd": (c_i, [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "ss_attention_relation_bwd": (c_i, [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "ss_attention_fusion_fwd": (c_i, [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "ss_attention_fusion_bwd": (c_i, [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "ss_rpe_dot_prod_fwd": (c_i, [c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "ss_rpe_dot_prod_bwd": (c_i, [c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "ss_rpe_attn_step2_fwd": (c_i, [c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "ss_rpe_attn_step2_bwd": (c_i, [c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "ss_majority_vote": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_i, c_p, c_p]),
    "ss_ballquery_batch_p": (c_i, [c_i, c_i, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "ss_bfs_cluster": (c_i, [c_p, c_p, c_p, c_i, c_i, c_p, c_i64, c_p, c_i64, c_p, c_p]),
}

_lib = None


def load():
    """Return the loaded library (cached).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m scenesplat_amd.build` "
            "(hipcc --offload-arch=gfx950). scenesplat_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class NativeError(RuntimeError):
    pass


_STATUS = {1: "bad argument", 2: "kernel launch failed", 3: "workspace too small"}


def check(rc, name):
    if rc != 0:
        raise NativeError(f"{name} failed: status {rc} ({_STATUS.get(rc, 'unknown')})")
