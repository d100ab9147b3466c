"""ctypes binding of libd4gs.so (the C ABI declared in include/d4gs.h).

The product path has NO fallback: if the HIP library is missing this module raises at first use, and every
entry point raises RuntimeError on a non-zero return code with `d4gs_last_error()`.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("D4GS_LIB_PATH") or os.path.join(HERE, "libd4gs.so")  # override: A/B builds (scripts/)

F = C.c_void_p  # device pointers travel as void*


class Dims(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("G", C.c_int32), ("K", C.c_int32), ("T", C.c_int32), ("S", C.c_int32), ("D", C.c_int32),
        ("width", C.c_int32), ("height", C.c_int32), ("depth_mode", C.c_int32), ("flags", C.c_int32),
        ("n_sigmoid", C.c_int32),
        ("near_plane", C.c_float), ("far_plane", C.c_float), ("eps2d", C.c_float), ("radius_clip", C.c_float),
    ]


class ProjIn(C.Structure):
    _fields_ = [(n, F) for n in ("means", "quats", "scales", "opacities", "colors", "motion_coefs", "rots", "transls",
                                 "times", "RTs", "viewmat", "Kmat")]


class ProjOut(C.Structure):
    _fields_ = [(n, F) for n in ("means2d", "depths", "conics", "radii", "opac_act", "ctab", "geom", "tile_rects", "tiles_touched",
                                 "isect_offsets", "lazy_ws", "tile_counts", "tile_offsets", "n_isect", "scan_ws", "blend_bases",
                                 "tile_masks")]


class Isect(C.Structure):
    _fields_ = [("n_isect", C.c_int64), ("max_tile_count", C.c_int64), ("near_target", C.c_int64)] + \
               [(n, F) for n in ("keys", "gid_of_emit", "sorted_gid", "sorted_emit")]


class Raster(C.Structure):
    _fields_ = [(n, F) for n in ("background", "render_colors", "render_alphas", "last_ids", "final_T", "seg_state")]


class RasterGrads(C.Structure):
    _fields_ = [(n, F) for n in ("v_render_colors", "v_render_alphas", "isect_grad", "isect_live", "v_means2d", "v_conics",
                                 "v_depths", "v_opac_act", "v_ctab", "stats_grad_norm_acc", "stats_vis_count",
                                 "stats_max_radii")] + [("stats_batch_size", C.c_int32), ("stats_update_max_radii", C.c_int32), ("row_mode", C.c_int32)]


class Sizes(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("means2d", "depths", "conics", "radii", "opac_act", "ctab", "geom", "tile_rects",
                                         "tiles_touched", "isect_offsets", "tile_counts", "tile_offsets", "n_isect",
                                         "scan_ws", "render_colors", "render_alphas", "last_ids", "final_T",
                                         "isect_grad_row", "bwd_partials", "seg_state", "lazy_ws")] + \
               [(n, C.c_int32) for n in ("tiles_x", "tiles_y", "channels")] + [("blend_bases", C.c_int64), ("tile_masks", C.c_int64)]


class MoveModelParams(C.Structure):
    _fields_ = [("w", F * 9), ("b", F * 9), ("time_params", F), ("n_time_params", C.c_int32)]


class MoveModelOut(C.Structure):
    _fields_ = [(n, F) for n in ("enc", "acts", "delta", "RTs", "jac", "times", "dtimes", "deltaT")]


class MoveModelGrads(C.Structure):
    _fields_ = [("v_w", F * 9), ("v_b", F * 9), ("v_time_params", F), ("v_delta", F), ("v_enc", F)]


class LeafGrads(C.Structure):
    _fields_ = [(n, F) for n in ("v_means", "v_quats", "v_scales", "v_opacities", "v_colors", "v_motion_coefs",
                                 "v_rots", "v_transls", "v_times", "v_RTs", "v_viewmat", "partials")]


class ShardBlend(C.Structure):
    _fields_ = [("S_total", C.c_int32), ("S_local", C.c_int32), ("s_first", C.c_int32), ("s_stride", C.c_int32), ("C", C.c_int32),
                ("n_pixels", C.c_int64), ("policy", C.POINTER(C.c_int32))]


class FrameIO(C.Structure):
    _fields_ = [(n, F) for n in ("blended", "acc", "renders", "alphas", "means2d", "radii", "n_isect", "background")] + \
               [("policy", C.POINTER(C.c_int32)), ("near_target", C.c_int64), ("counts_pinned", F)]


class FrameGrads(C.Structure):
    _fields_ = [(n, F) for n in ("v_blended", "v_acc", "v_renders", "v_alphas", "v_means2d", "stats_grad_norm_acc",
                                 "stats_vis_count", "stats_max_radii")] + \
               [("stats_batch_size", C.c_int32), ("stats_update_max_radii", C.c_int32), ("row_mode", C.c_int32)]


class Poses(C.Structure):
    _fields_ = [("means", F), ("quats", F), ("transforms", F), ("g_major", C.c_int32)]


RAW_PARAMS, RAW_COLORS, EXACT_CULL, LAZY_SORT, EXACT_TILES = 1, 2, 4, 8, 16
DEPTH_NONE, DEPTH_ED, DEPTH_D = 0, 1, 2
ROWS_AUTO, ROWS_DENSE, ROWS_SPARSE = 0, 1, 2
TILE = 16
VERSION = 305  # D4GS_VERSION of include/d4gs.h
GEOM_STRIDE = 8

EXPORTS = (
    "d4gs_version", "d4gs_copy_counts", "d4gs_last_error", "d4gs_scan_ws_elems", "d4gs_bwd_partials_elems", "d4gs_project_fwd",
    "d4gs_bin_sort", "d4gs_raster_fwd", "d4gs_raster_bwd", "d4gs_project_bwd", "d4gs_blend_fwd", "d4gs_blend_bwd",
    "d4gs_points_fwd", "d4gs_points_bwd", "d4gs_poses_fwd", "d4gs_poses_bwd", "d4gs_forward", "d4gs_backward", "d4gs_forward_cpu", "d4gs_backward_cpu", "d4gs_frame_workspace_bytes", "d4gs_frame_workspace_bytes_fwd", "d4gs_blend_shard_partial_fwd", "d4gs_blend_shard_finish_fwd",
    "d4gs_blend_shard_winner", "d4gs_blend_shard_bwd", "d4gs_control_stats", "d4gs_control_plan", "d4gs_gather_rows", "d4gs_camera_path_fwd", "d4gs_camera_path_bwd",
    "d4gs_pose_encode", "d4gs_pose_encode_bwd", "d4gs_move_model_fwd", "d4gs_move_model_bwd",
    "d4gs_photometric_blocks", "d4gs_photometric_fwd", "d4gs_photometric_bwd", "d4gs_query_sizes", "d4gs_profile_enable", "d4gs_profile_collect", "d4gs_measure_peaks",
)

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
                "Build it with `python -m deblur4dgs_amd.build` (hipcc, --offload-arch=gfx950)."
            )
        import torch  # noqa: F401  -- torch's bundled HIP runtime must be the one both sides use: load it first
        L = C.CDLL(LIB_PATH)
        for name in EXPORTS:
            getattr(L, name)  # AttributeError if the symbol is missing
        L.d4gs_last_error.restype = C.c_char_p
        L.d4gs_scan_ws_elems.restype = C.c_size_t
        L.d4gs_scan_ws_elems.argtypes = [C.c_int64]
        L.d4gs_bwd_partials_elems.restype = C.c_size_t
        L.d4gs_bwd_partials_elems.argtypes = [C.POINTER(Dims)]
        P = C.POINTER
        vp = C.c_void_p
        L.d4gs_copy_counts.argtypes = [vp, vp, vp]
        L.d4gs_measure_peaks.argtypes = [vp, C.c_size_t, P(C.c_double), vp]
        L.d4gs_project_fwd.argtypes = [P(Dims), P(ProjIn), P(ProjOut), vp]
        L.d4gs_bin_sort.argtypes = [P(Dims), P(ProjOut), P(Isect), vp]
        L.d4gs_raster_fwd.argtypes = [P(Dims), P(ProjOut), P(Isect), P(Raster), vp]
        L.d4gs_raster_bwd.argtypes = [P(Dims), P(ProjOut), P(Isect), P(Raster), P(RasterGrads), vp]
        L.d4gs_project_bwd.argtypes = [P(Dims), P(ProjIn), P(ProjOut), vp, vp, vp, vp, vp, P(LeafGrads), vp]
        L.d4gs_points_fwd.argtypes = [P(Dims), P(ProjIn), vp, vp]
        L.d4gs_points_bwd.argtypes = [P(Dims), P(ProjIn), vp, P(LeafGrads), vp]
        L.d4gs_poses_fwd.argtypes = [P(Dims), P(ProjIn), P(Poses), vp]
        L.d4gs_poses_bwd.argtypes = [P(Dims), P(ProjIn), P(Poses), P(LeafGrads), vp]
        L.d4gs_blend_shard_partial_fwd.argtypes = [P(ShardBlend), vp, vp, vp, vp, vp]
        L.d4gs_blend_shard_finish_fwd.argtypes = [P(ShardBlend), vp, vp, vp, vp, vp]
        L.d4gs_blend_shard_winner.argtypes = [P(ShardBlend), vp, vp, vp, vp]
        L.d4gs_blend_shard_bwd.argtypes = [P(ShardBlend), vp, vp, vp, vp, vp, vp]
        L.d4gs_frame_workspace_bytes.argtypes = [P(Dims), C.c_int64]
        L.d4gs_frame_workspace_bytes.restype = C.c_size_t
        L.d4gs_frame_workspace_bytes_fwd.argtypes = [P(Dims), C.c_int64]
        L.d4gs_frame_workspace_bytes_fwd.restype = C.c_size_t
        L.d4gs_forward.argtypes = [P(Dims), P(ProjIn), P(FrameIO), vp, C.c_size_t, C.c_int64, C.c_int64, vp]
        L.d4gs_backward.argtypes = [P(Dims), P(ProjIn), P(FrameIO), P(FrameGrads), P(LeafGrads), vp, C.c_size_t, C.c_int64,
                                    C.c_int64, vp]
        L.d4gs_forward_cpu.argtypes = [P(Dims), P(ProjIn), P(FrameIO)]
        L.d4gs_backward_cpu.argtypes = [P(Dims), P(ProjIn), P(FrameIO), P(FrameGrads), P(LeafGrads)]
        L.d4gs_control_stats.argtypes = [C.c_int32, C.c_int32, vp, vp, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, C.c_int32, vp]
        L.d4gs_control_plan.argtypes = [C.c_int32, vp, vp, vp, vp, vp]
        L.d4gs_gather_rows.argtypes = [vp, C.c_int64, C.c_int32, vp, vp, C.c_int64, C.c_int64, C.c_float, vp]
        L.d4gs_camera_path_fwd.argtypes = [vp, vp, C.c_int32, vp, C.c_int32, C.c_int32, C.c_float, vp, vp, vp, vp, vp, vp]
        L.d4gs_camera_path_bwd.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp]
        L.d4gs_pose_encode.argtypes = [vp, C.c_int32, vp, C.c_int32, vp, vp]
        L.d4gs_pose_encode_bwd.argtypes = [vp, C.c_int32, vp, C.c_int32, vp, vp, vp, vp]
        L.d4gs_move_model_fwd.argtypes = [vp, C.c_int32, vp, C.c_int32, P(MoveModelParams), C.c_int32, C.c_int32, C.c_float,
                                          C.c_int32, P(MoveModelOut), vp]
        L.d4gs_move_model_bwd.argtypes = [P(MoveModelParams), P(MoveModelOut), vp, vp, vp, C.c_int32, C.c_int32,
                                          P(MoveModelGrads), vp]
        L.d4gs_query_sizes.argtypes = [P(Dims), P(Sizes)]
        L.d4gs_photometric_blocks.argtypes = [C.c_int32, C.c_int32, C.c_int32]
        L.d4gs_photometric_blocks.restype = C.c_int64
        L.d4gs_photometric_fwd.argtypes = [vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, vp, vp,
                                           vp, vp]
        L.d4gs_photometric_bwd.argtypes = [vp, vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                           C.c_float, vp, vp]
        L.d4gs_blend_fwd.argtypes = [C.c_int32, C.c_int64, C.c_int32, P(C.c_int32), vp, vp, vp, vp, vp]
        L.d4gs_blend_bwd.argtypes = [C.c_int32, C.c_int64, C.c_int32, P(C.c_int32), vp, vp, vp, vp, vp, vp, vp]
        if L.d4gs_version() != VERSION:
            raise RuntimeError(f"libd4gs.so version {L.d4gs_version()} != {VERSION} (stale build?)")
        _lib = L
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {lib().d4gs_last_error().decode()}")


def ptr(t):
    """Device pointer of a contiguous tensor (or NULL for None)."""
    if t is None:
        return None
    assert t.is_contiguous(), "d4gs expects contiguous tensors"
    return t.data_ptr()


def fill(struct, **tensors):
    for k, v in tensors.items():
        setattr(struct, k, ptr(v))
    return struct


_RAW_STREAM = None


def raw_stream(index=None) -> int:
    """The current HIP stream of device `index` (default: the current device) as an integer handle.  One C call:
    torch.cuda.current_stream() builds a Python Stream object every time (~12 us, six times per rendered frame)."""
    global _RAW_STREAM
    import torch

    if _RAW_STREAM is None:
        _RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if index is None:
        index = torch.cuda.current_device()
    if _RAW_STREAM:
        return _RAW_STREAM(index)
    return torch.cuda.current_stream(index).cuda_stream
