"""ctypes binding of the C ABI declared in include/me_amd.h (libme_amd.so, built by
__graft_entry__.build()).

The product path has no CPU fallback: if the HIP library is missing or a call fails, a
RuntimeError is raised (mirroring the reference's ASSERT -> std::runtime_error -> RuntimeError,
/root/reference/src/utils.hpp:141-150).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (ME_AMD_LIB_TAG: an A/B build made by build.py under the same tag — tuning scripts only)
_TAG = os.environ.get("ME_AMD_LIB_TAG", "")
LIB_PATH = os.path.join(_HERE, f"libme_amd_{_TAG}.so" if _TAG else "libme_amd.so")

ME_MAX_DIM = 7
ME_MAX_TILE_ROWS = 256
ME_GROUP_ROWS = 16
ME_MAX_BATCH_GROUPS = 4

c_i32, c_i64, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p


class MeRegion(ctypes.Structure):
    """struct me_region (include/me_amd.h)."""
    _fields_ = [
        ("ncol", c_i32),
        ("region_type", c_i32),
        ("kernel_size", c_i32 * ME_MAX_DIM),
        ("dilation", c_i32 * ME_MAX_DIM),
        ("tensor_stride", c_i32 * ME_MAX_DIM),
    ]


class MePackJob(ctypes.Structure):
    """me_pack_job of include/me_amd.h"""
    _fields_ = [("w", ctypes.c_void_p), ("wp", ctypes.c_void_p), ("volume", ctypes.c_int64),
                ("threads", ctypes.c_int64), ("c_src", ctypes.c_int32), ("c_dst", ctypes.c_int32),
                ("transposed", ctypes.c_int32), ("w_is_f32", ctypes.c_int32), ("mode", ctypes.c_int32),
                ("kc", ctypes.c_int32), ("nchunks", ctypes.c_int32), ("ncb", ctypes.c_int32)]


ME_PACK_BF16, ME_PACK_F32X3 = 0, 1


class MePlanJob(ctypes.Structure):
    """me_plan_job of include/me_amd.h"""
    _fields_ = [("tbl", ctypes.c_void_p), ("order", ctypes.c_void_p), ("n_tgt", ctypes.c_int64), ("volume", ctypes.c_int64),
                ("tile_rows", ctypes.c_int32), ("batch_groups", ctypes.c_int32), ("plan_src", ctypes.c_void_p),
                ("plan_dst", ctypes.c_void_p), ("batch_desc", ctypes.c_void_p), ("tile_bptr", ctypes.c_void_p),
                ("item_gptr", ctypes.c_void_p), ("n_tiles", ctypes.c_int64), ("n_items", ctypes.c_int64),
                ("item_base", ctypes.c_int64)]


class MeSpatialGrid(ctypes.Structure):
    """struct me_spatial_grid (include/me_amd.h)."""
    _fields_ = [
        ("ncol", c_i32),
        ("shift", c_i32 * ME_MAX_DIM),
        ("sc_min", c_i32 * (ME_MAX_DIM + 1)),
        ("sc_dim", c_i32 * (ME_MAX_DIM + 1)),
        ("tensor_stride", c_i32 * ME_MAX_DIM),
    ]


_P_GRID = ctypes.POINTER(MeSpatialGrid)
_P_REGION = ctypes.POINTER(MeRegion)
_P_I64 = ctypes.POINTER(c_i64)
_P_I32 = ctypes.POINTER(c_i32)

# name -> (restype, argtypes); must list every symbol include/me_amd.h declares
SIGNATURES = {
    "me_version": (ctypes.c_int, []),
    "me_preload": (ctypes.c_int, []),
    "me_last_error": (ctypes.c_char_p, []),
    "me_region_volume": (c_i64, [_P_REGION]),
    "me_hash_capacity": (c_i64, [c_i64]),
    "me_insert_workspace_bytes": (c_i64, [c_i64]),
    "me_coords_insert_and_map": (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_i64, c_vp, c_vp, c_vp, _P_I64,
                                                c_vp, c_i64, c_vp]),
    "me_coords_insert_and_map_bbox": (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_i64, c_vp, c_vp, c_vp, _P_I64, _P_I32,
                                                     c_vp, c_i64, c_vp]),
    "me_spatial_cells": (c_i64, [_P_GRID]),
    "me_spatial_index_workspace_bytes": (c_i64, [c_i64, c_i64]),
    "me_spatial_index_build": (ctypes.c_int, [c_vp, c_i64, _P_GRID, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "me_kernel_map_probe_lds_bytes": (c_i64, [_P_REGION, _P_GRID, _P_GRID]),
    "me_kernel_map_probe_lds": (ctypes.c_int, [_P_GRID, c_vp, c_vp, c_i64, _P_GRID, c_vp, c_vp, c_vp, _P_REGION, c_vp,
                                               c_vp, c_vp, c_i64, c_vp]),
    "me_kernel_map_count": (ctypes.c_int, [c_vp, c_i64, c_i64, _P_I64, c_vp, c_vp, c_i64, c_vp]),
    "me_kernel_map_compact_ordered": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "me_kernel_map_transpose_ordered": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "me_coords_stride": (ctypes.c_int, [c_vp, c_i64, c_i32, _P_I32, c_vp, c_vp]),
    "me_coords_spatial_keys": (ctypes.c_int, [c_vp, c_i64, c_i32, _P_I32, c_vp, c_vp]),
    "me_coords_zorder_workspace_bytes": (c_i64, [c_i64]),
    "me_coords_zorder": (ctypes.c_int, [c_vp, c_i64, c_i32, _P_I32, _P_I32, c_vp, c_vp, c_i64, c_vp]),
    "me_coords_find": (ctypes.c_int, [c_vp, c_i64, c_vp, c_i32, c_vp, c_i64, c_vp, c_vp]),
    "me_kernel_map_workspace_bytes": (c_i64, [c_i64, c_i64]),
    "me_kernel_map_probe": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, _P_REGION, c_vp, _P_I64, c_vp, c_vp,
                                           c_i64, c_vp]),
    "me_kernel_map_compact": (ctypes.c_int, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "me_kernel_map_transpose": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp]),
    "me_plan_num_tiles": (c_i64, [c_i64, c_i32]),
    "me_plan_tile_bptr_elems": (c_i64, [c_i64, c_i32]),
    "me_plan_max_groups": (c_i64, [c_i64, c_i64, c_i64, c_i32]),
    "me_plan_workspace_bytes": (c_i64, [c_i64, c_i64, c_i32]),
    "me_plan_build": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                     c_i64, c_vp]),
    "me_plan_jobs_init": (c_i64, [c_vp, c_i32]),
    "me_plan_multi_workspace_bytes": (c_i64, [c_i64]),
    "me_plan_build_multi": (ctypes.c_int, [c_vp, c_vp, c_i32, c_vp, c_i64, c_vp]),
    "me_conv_packed_weight_elems": (c_i64, [c_i64, c_i32, c_i32]),
    "me_conv_pack_weights_f32": (ctypes.c_int, [c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "me_conv_target_f32": (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp,
                                          c_vp, c_i64, c_i32, c_i32, c_vp]),
    "me_conv_target_f32_fused": (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp,
                                                c_vp, c_i64, c_i32, c_i32, c_vp]),
    "me_conv_plan_config": (ctypes.c_int, [c_i64, c_i64, c_i64, c_i32, c_i32, _P_I32, _P_I32]),
    "me_transpose_kernel_f32": (ctypes.c_int, [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp]),
    "me_conv_wgrad_workspace_bytes": (c_i64, [_P_I64, c_i64, c_i32, c_i32]),
    "me_conv_wgrad_f32": (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_vp, _P_I64, c_vp, c_i64, c_vp,
                                         c_vp, c_i64, c_vp]),
    "me_bn_workspace_bytes": (c_i64, [c_i64, c_i32]),
    "me_bn_stats": (ctypes.c_int, [c_vp, c_i32, c_i64, c_i32, ctypes.c_float, ctypes.c_float, c_vp, c_vp, c_vp, c_vp,
                                   c_vp, c_vp, c_i64, c_vp]),
    "me_bn_stats_from_tiles": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_i32, ctypes.c_float, ctypes.c_float, c_vp, c_vp,
                                              c_vp, c_vp, c_vp, c_vp]),
    "me_bn_apply": (ctypes.c_int, [c_vp, c_i32, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "me_bn_backward": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp,
                                      c_vp, c_i64, c_vp]),
    "me_bn_apply_residual": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "me_bn_backward_residual": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp,
                                               c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "me_coords_expand_region": (ctypes.c_int, [c_vp, c_i64, c_i32, _P_REGION, _P_I32, c_vp, c_vp, c_vp]),
    "me_coords_quantize_labels": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp]),
    "me_segment_sum_f32": (ctypes.c_int, [c_vp, c_i32, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp]),
    "me_conv_f32x3_supported": (c_i32, [c_i32, c_i32]),
    "me_conv_plan_config_f32x3": (ctypes.c_int, [c_i64, c_i64, c_i64, c_i32, c_i32, _P_I32, _P_I32]),
    "me_conv_packed_weight_elems_f32x3": (c_i64, [c_i64, c_i32, c_i32]),
    "me_conv_pack_weights_f32x3": (ctypes.c_int, [c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "me_conv_target_f32x3": (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp,
                                            c_vp, c_i64, c_i32, c_i32, c_vp]),
    "me_conv_stem_use_bf16": (c_i32, [c_i64, c_i64, c_i32, c_i32]),
    "me_conv_stem_tile_rows": (c_i32, []),
    "me_conv_stem_bf16": (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_i32, c_i32, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_i64,
                                         c_vp, c_vp, c_vp]),
    "me_conv_rowwise_supported_bf16": (c_i32, [c_i64, c_i32, c_i32]),
    "me_conv_rowwise_bf16": (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64,
                                            c_vp]),
    "me_conv_halo_min_uses": (c_i32, []),
    "me_conv_halo_use_bf16": (c_i32, [c_i64, c_i64, c_i64, c_i32, c_i32]),
    "me_conv_halo_config_bf16": (c_i32, [c_i64, c_i64, c_i64, c_i32, c_i32, _P_I32, _P_I32]),
    "me_halo_plan_num_tiles": (c_i64, [c_i64, c_i32]),
    "me_halo_plan_build": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "me_conv_halo_bf16": (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                         c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "me_conv_pack_chunk_bf16": (c_i32, [c_i32, c_i32]),
    "me_conv_pack_chunk_f32x3": (c_i32, [c_i32, c_i32]),
    "me_conv_pack_job_init": (ctypes.c_int, [ctypes.POINTER(MePackJob)]),
    "me_conv_pack_weights_multi": (ctypes.c_int, [c_vp, c_i32, c_vp, c_i64, c_vp]),
    "me_conv_plan_config_bf16": (ctypes.c_int, [c_i64, c_i64, c_i64, c_i32, c_i32, _P_I32, _P_I32]),
    "me_conv_packed_weight_elems_bf16": (c_i64, [c_i64, c_i32, c_i32]),
    "me_conv_pack_weights_bf16": (ctypes.c_int, [c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "me_conv_target_bf16": (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp,
                                           c_vp, c_i64, c_i32, c_i32, c_vp]),
    "me_conv_target_bf16_fused": (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp,
                                           c_vp, c_i64, c_i32, c_i32, c_vp]),
    "me_conv_stats_supported_bf16": (c_i32, [c_i32, c_i32]),
    "me_conv_target_bf16_stats": (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp,
                                           c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "me_conv_plan_config_bf16_ex": (ctypes.c_int, [c_i64, c_i64, c_i64, c_i32, c_i32, _P_I32, _P_I32, _P_I32]),
    "me_conv_splitk_workspace_bytes": (c_i64, [c_i64, c_i32, c_i32, c_i32]),
    "me_conv_target_bf16_ex": (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp,
                                              c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "me_conv_gather_supported_bf16": (c_i32, [c_i32, c_i32]),
    "me_conv_gather_weight_elems_bf16": (c_i64, [c_i64, c_i32, c_i32]),
    "me_conv_gather_pack_weights_bf16": (ctypes.c_int, [c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "me_conv_gather_bf16": (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "me_conv_wgrad_workspace_bytes_bf16": (c_i64, [_P_I64, c_i64, c_i32, c_i32]),
    "me_conv_wgrad_bf16": (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_vp, _P_I64, c_vp, c_i64, c_vp,
                                          c_vp, c_i64, c_vp]),
    "me_conv_target_f64": (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_i32, c_i64, c_i32, c_vp, c_vp, c_i64, c_vp]),
    "me_conv_wgrad_f64": (ctypes.c_int, [c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "me_pool_sum_f64": (ctypes.c_int, [c_vp, c_i32, c_vp, c_i64, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "me_pool_max_f64": (ctypes.c_int, [c_vp, c_i32, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "me_pool_max_backward_f64": (ctypes.c_int, [c_vp, c_i32, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "me_global_pool_f64": (ctypes.c_int, [c_vp, c_vp, c_i32, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "me_broadcast_f64": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp]),
    "me_pool_sum_f32": (ctypes.c_int, [c_vp, c_i32, c_vp, c_i64, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "me_pool_max_f32": (ctypes.c_int, [c_vp, c_i32, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "me_pool_max_backward_f32": (ctypes.c_int, [c_vp, c_i32, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "me_global_pool_workspace_bytes": (c_i64, [c_i64, c_i32, c_i32]),
    "me_global_pool_f32": (ctypes.c_int, [c_vp, c_vp, c_i32, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp,
                                          c_i64, c_vp]),
    "me_broadcast_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp]),
    "me_pool_sum_bf16": (ctypes.c_int, [c_vp, c_i32, c_vp, c_i64, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "me_pool_max_bf16": (ctypes.c_int, [c_vp, c_i32, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "me_pool_max_backward_bf16": (ctypes.c_int, [c_vp, c_i32, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "me_global_pool_bf16": (ctypes.c_int, [c_vp, c_vp, c_i32, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp,
                                           c_i64, c_vp]),
    "me_broadcast_bf16": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp]),
    "me_conv_forward_naive_f32": (ctypes.c_int, [c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_i64, c_i64,
                                                 c_vp, c_vp]),
    "me_conv_backward_naive_f32": (ctypes.c_int, [c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i64,
                                                  c_i64, c_vp, c_vp, c_vp]),
}

# test / tuning hooks (csrc/me_amd_debug.h): exported by the library, not part of include/me_amd.h
DEBUG_SIGNATURES = {
    "me_debug_variants_compiled": (c_i32, []),
    "me_debug_set_conv_variant": (ctypes.c_int, [ctypes.c_int]),
    "me_debug_conv_timing": (ctypes.c_int, [c_vp, c_i32]),
    "me_debug_conv_timing_f32x3": (ctypes.c_int, [c_vp, c_i32]),
    "me_debug_set_wgrad_config": (None, [ctypes.c_int, ctypes.c_int]),
    "me_debug_set_wgrad_order": (None, [ctypes.c_int]),
    "me_debug_set_wgrad_ws": (None, [ctypes.c_int]),
    "me_debug_set_tile_dispatch": (None, [ctypes.c_int]),
    "me_debug_set_wgrad_mb": (None, [ctypes.c_int]),
    "me_debug_set_bf16_shape": (None, [ctypes.c_int, ctypes.c_int]),
    "me_debug_set_bf16_deep": (None, [ctypes.c_int]),
    "me_debug_set_bf16_twobuf": (None, [ctypes.c_int]),
    "me_debug_set_f32_fused_split": (None, [ctypes.c_int]),
    "me_debug_set_bf16_ws": (None, [ctypes.c_int]),
    "me_debug_set_halo": (None, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "me_debug_halo_mode": (c_i32, []),
    "me_debug_set_stem": (None, [ctypes.c_int, ctypes.c_int]),
    "me_debug_halo_timing": (ctypes.c_int, [c_vp, c_i32]),
    "me_debug_set_bf16_ws_fuse": (None, [ctypes.c_int]),
    "me_debug_set_bf16_ws_depth": (None, [ctypes.c_int]),
    "me_debug_set_bf16_ws_ncw": (None, [ctypes.c_int]),
    "me_debug_set_rowwise_groups": (None, [ctypes.c_int]),
    "me_debug_set_insert_fused": (None, [ctypes.c_int]),
    "me_debug_ws_timing": (ctypes.c_int, [c_vp, c_i32]),
    "me_debug_set_bf16_offsync": (None, [ctypes.c_int]),
    "me_debug_set_bf16_splitk": (None, [ctypes.c_int]),
    "me_debug_set_bf16_splitk_mode": (None, [ctypes.c_int]),
    "me_debug_bf16_timing": (ctypes.c_int, [c_vp, c_i32]),
}

_lib = None


def load():
    """Load libme_amd.so (once) and attach the prototypes.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"minkowskiengine_amd: HIP library {LIB_PATH} not found. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the MI355X path.")
    lib = ctypes.CDLL(LIB_PATH)
    for table in (SIGNATURES, DEBUG_SIGNATURES):
        for name, (restype, argtypes) in table.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = restype
            fn.argtypes = argtypes
    _lib = lib
    # (no device work here: importing the package must not create a HIP context — ranks import before
    # torch.cuda.set_device(local_rank), and a parent that imports and then forks GPU workers must keep torch's lazy
    # initialisation intact; the code objects are preloaded per device at first use: preload_device)
    # tuning switches of the library by environment (A/B runs of bench.py): ME_AMD_BF16_WS=0 keeps k_conv_tile_bf16
    # everywhere, ME_AMD_BF16_WS_DEPTH=2 its shallower producer pipeline
    if os.environ.get("ME_AMD_BF16_WS", "") != "":
        lib.me_debug_set_bf16_ws(int(os.environ["ME_AMD_BF16_WS"]))
    if os.environ.get("ME_AMD_F32_FUSED_SPLIT", "") != "":
        lib.me_debug_set_f32_fused_split(int(os.environ["ME_AMD_F32_FUSED_SPLIT"]))
    if os.environ.get("ME_AMD_WGRAD_WS", "") != "":
        lib.me_debug_set_wgrad_ws(int(os.environ["ME_AMD_WGRAD_WS"]))
    if os.environ.get("ME_AMD_BF16_WS_NCW", "") != "":
        lib.me_debug_set_bf16_ws_ncw(int(os.environ["ME_AMD_BF16_WS_NCW"]))
    if os.environ.get("ME_INSERT_FUSED", "") != "":
        lib.me_debug_set_insert_fused(int(os.environ["ME_INSERT_FUSED"]))
    if os.environ.get("ME_AMD_RW_G", "") != "":
        lib.me_debug_set_rowwise_groups(int(os.environ["ME_AMD_RW_G"]))
    if os.environ.get("ME_AMD_BF16_WS_DEPTH", "") != "":
        lib.me_debug_set_bf16_ws_depth(int(os.environ["ME_AMD_BF16_WS_DEPTH"]))
    return lib


_preloaded = set()


def preload_device(index):
    """Load the device code of every translation unit on device `index` now rather than at the first launch from each
    (the first backward pass of a process paid 88 ms for that).  Called by the hosts at the first map insert on a device
    (under that device's guard) and by distributed.init_from_env after set_device — never at import.  Code objects are
    per device: every device a process uses is preloaded once.  ME_AMD_PRELOAD=0 keeps HIP's lazy loading."""
    if index is None or index in _preloaded:
        return
    _preloaded.add(index)
    if os.environ.get("ME_AMD_PRELOAD", "1") == "0":
        return
    import torch
    with torch.cuda.device(index):
        load().me_preload()


def check(rc):
    if rc != 0:
        msg = load().me_last_error()
        raise RuntimeError(msg.decode() if msg else f"libme_amd call failed with code {rc}")


def make_region(ncol, region_type, kernel_size, dilation, tensor_stride):
    rg = MeRegion()
    rg.ncol = int(ncol)
    rg.region_type = int(region_type)
    for d in range(ME_MAX_DIM):
        rg.kernel_size[d] = int(kernel_size[d]) if d < ncol - 1 else 1
        rg.dilation[d] = int(dilation[d]) if d < ncol - 1 else 1
        rg.tensor_stride[d] = int(tensor_stride[d]) if d < ncol - 1 else 1
    return rg
