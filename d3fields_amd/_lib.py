"""ctypes binding of libd3fields_hip.so (the C ABI declared in include/d3fields_hip.h).

There is deliberately no fallback: if the shared library cannot be loaded, or a call returns
an error status, an exception is raised.  Nothing in this package computes the field query on
the CPU or through torch ops.
"""
import ctypes
import os

from . import build as _build

ABI_VERSION = 6

# status codes (include/d3fields_hip.h)
OK = 0
ERR_INVALID_ARG, ERR_BAD_SHAPE, ERR_BAD_DTYPE, ERR_BAD_LAYOUT, ERR_HIP, ERR_WORKSPACE = -1, -2, -3, -4, -5, -6
FLAG_FINITE_MAPS = 1
FLAG_UNORDERED_POINTS = 2
FLAG_REFERENCE_ROUNDING = 4
FLAG_REUSE_POINT_ORDER = 8
FLAG_LOCAL_POINTS = 128
PROBE_WORDS = 40
TUNE_XCD_REMAP, TUNE_NO_REORDER, TUNE_FORCE_REORDER = 1 << 12, 1 << 13, 1 << 14
TUNE_DIRECT_GATHER = 1 << 4
TUNE_NO_WINDOW_GATE, TUNE_WINDOW_SIDE = 1 << 5, 1 << 6
GATE_SAMPLES, GATE_MIN_FIT = 128, 96
CHECK_WORDS_ARE_ZERO = 1
TRACK_STALL_SENTINEL = 0x57A11ED
MAX_VIEWS = 64
MAX_MAPS = 8
DTYPE_F32 = 0
DTYPE_F16 = 1
DIST_L2, DIST_SQUARE = 0, 1
SIM_DIST, SIM_EXP, SIM_SOFTMAX_DIM0 = 0, 1, 2

_vp = ctypes.c_void_p
_i32 = ctypes.c_int32
_i64 = ctypes.c_int64
_u32 = ctypes.c_uint32
_f32 = ctypes.c_float


class Views(ctypes.Structure):
    """struct d3f_views"""
    _fields_ = [("V", _i32), ("H", _i32), ("W", _i32), ("depth", _vp), ("K", _vp), ("pose", _vp), ("depth_nonfinite", _vp)]


class ChannelMap(ctypes.Structure):
    """struct d3f_channel_map"""
    _fields_ = [("data", _vp), ("fh", _i32), ("fw", _i32), ("C", _i32), ("dtype", _i32),
                ("stride_v", _i64), ("stride_y", _i64), ("stride_x", _i64), ("nonfinite", _vp)]


class Grid(ctypes.Structure):
    """struct d3f_grid"""
    _fields_ = [("x", _vp), ("y", _vp), ("z", _vp), ("nx", _i32), ("ny", _i32), ("nz", _i32), ("reserved", _i32)]


class TrackState(ctypes.Structure):
    """struct d3f_track_state"""
    _fields_ = [("t", _vp), ("w", _vp), ("adam_m", _vp), ("adam_v", _vp), ("step", _vp), ("out_pts", _vp), ("loss", _vp), ("scratch", _vp)]


class EvalPlan(ctypes.Structure):
    """struct d3f_eval_plan"""
    _fields_ = [("tile_points", _i32), ("reorder", _i32), ("lds_bytes", _i32), ("reserved", _i32), ("workgroups", _i64),
                ("vector_floats", _i32 * MAX_MAPS), ("lanes_per_point", _i32 * MAX_MAPS),
                ("vectors_per_lane", _i32 * MAX_MAPS), ("staged", _i32 * MAX_MAPS), ("gated_window", _i32), ("reserved2", _i32), ("family", _i32), ("reserved3", _i32)]


# name -> (restype, argtypes); every symbol include/d3fields_hip.h declares
SIGNATURES = {
    "d3f_abi_version": (ctypes.c_int, []),
    "d3f_version": (ctypes.c_char_p, []),
    "d3f_last_error": (ctypes.c_char_p, []),
    "d3f_build_has_experiments": (ctypes.c_int, []),
    "d3f_eval": (ctypes.c_int, [ctypes.POINTER(Views), _vp, _i64, ctypes.POINTER(ChannelMap), _i32, _f32, _u32,
                                _vp, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp, _i64, _vp]),
    "d3f_map_check": (ctypes.c_int, [ctypes.POINTER(ChannelMap), _i32, _vp, _vp]),
    "d3f_map_check_many": (ctypes.c_int, [ctypes.POINTER(ChannelMap), ctypes.POINTER(_i32), _i32, ctypes.POINTER(_vp), _u32, _vp]),
    "d3f_eval_workspace_bytes": (_i64, [_i64]),
    "d3f_eval_dist_workspace_bytes": (_i64, [ctypes.POINTER(Views), _i64]),
    "d3f_eval_gate_offset": (_i64, [_i64]),
    "d3f_plan_family_name": (ctypes.c_char_p, [_i32]),
    "d3f_plan_family_takes": (ctypes.c_char_p, [_i32]),
    "d3f_profile_next_eval": (None, [_vp, _vp]),
    "d3f_eval_plan_query": (ctypes.c_int, [ctypes.POINTER(Views), _i64, ctypes.POINTER(ChannelMap), _i32, _u32, _i32, _i32,
                                           ctypes.POINTER(EvalPlan)]),
    "d3f_eval_grid": (ctypes.c_int, [ctypes.POINTER(Views), ctypes.POINTER(Grid), ctypes.POINTER(ChannelMap), _i32, _f32, _u32,
                                     _vp, _vp, ctypes.POINTER(_vp), _vp]),
    "d3f_eval_lattice": (ctypes.c_int, [ctypes.POINTER(Views), _vp, _i32, _i32, _i32, ctypes.POINTER(ChannelMap), _i32, _f32, _u32,
                                        _vp, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp]),
    "d3f_eval_plan_query_lattice": (ctypes.c_int, [ctypes.POINTER(Views), _i32, _i32, _i32, ctypes.POINTER(ChannelMap), _i32, _u32, _i32,
                                                   ctypes.POINTER(EvalPlan)]),
    "d3f_lattice_probe": (ctypes.c_int, [_vp, _i64, _vp, _vp]),
    "d3f_grid_shell_workspace_bytes": (_i64, [ctypes.POINTER(Grid)]),
    "d3f_grid_shell": (ctypes.c_int, [ctypes.POINTER(Views), ctypes.POINTER(Grid), _f32, _f32, _i64, _vp, _vp, _vp, _i64, _vp]),
    "d3f_fps_workspace_bytes": (_i64, [_i64]),
    "d3f_fps_pixels_workspace_bytes": (_i64, [_i64]),
    "d3f_farthest_point_sampling": (ctypes.c_int, [_vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp]),
    "d3f_backproject_workspace_bytes": (_i64, [_i32, _i32]),
    "d3f_backproject_view": (ctypes.c_int, [_vp, _vp, _i32, _i32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                            ctypes.POINTER(ctypes.c_double), _i64, _vp, _vp, _vp, _vp, _vp]),
    "d3f_pcd_nearest": (ctypes.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    "d3f_pcd_to_index": (ctypes.c_int, [_vp, _i64, ctypes.POINTER(ctypes.c_double), ctypes.c_double, ctypes.POINTER(_i32), _vp, _vp, _vp]),
    "d3f_vox_iou_workspace_bytes": (_i64, [_i64, _i64]),
    "d3f_vox_idx_iou": (ctypes.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp]),
    "d3f_erode": (ctypes.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "d3f_compose_labels": (ctypes.c_int, [_vp, _i32, _i64, _vp, _vp, _vp]),
    "d3f_voxel_downsample_workspace_bytes": (_i64, [_i64]),
    "d3f_voxel_downsample": (ctypes.c_int, [_vp, _vp, _i64, ctypes.c_double, _vp, _vp, _vp, _vp, _i64, _vp]),
    "d3f_mask_gate": (ctypes.c_int, [_vp, _i64, _i64, _vp, _i32, _i32, _f32, _f32, _vp, _vp]),
    "d3f_nonzero_pixels": (ctypes.c_int, [_vp, _i32, _i32, _i64, _vp, _vp, _vp, _vp]),
    "d3f_fps_pixels": (ctypes.c_int, [_vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp]),
    "d3f_eval_backward": (ctypes.c_int, [ctypes.POINTER(Views), _vp, _i64, ctypes.POINTER(ChannelMap), _i32, _f32,
                                         _vp, ctypes.POINTER(_vp), _vp, _vp]),
    "d3f_eval_dist_backward": (ctypes.c_int, [ctypes.POINTER(Views), _vp, _i64, _vp, _vp, _vp]),
    "d3f_eval_dist": (ctypes.c_int, [ctypes.POINTER(Views), _vp, _i64, _vp, _vp, _vp]),
    "d3f_onehot2instance": (ctypes.c_int, [_vp, _i64, _i32, _vp, _vp]),
    "d3f_instance2onehot": (ctypes.c_int, [_vp, _i64, _i32, _vp, _vp]),
    "d3f_softmax_workspace_bytes": (_i64, [_i64, _i64]),
    "d3f_similarity_to_target": (ctypes.c_int, [_vp, _i64, _i64, _i32, _i64, _i64, _i64, _vp, _f32, _i32, _i32,
                                                _vp, _vp, _i64, _vp]),
    "d3f_pairwise_similarity": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i32, _f32, _i32, _i32, _vp, _vp, _vp, _i64,
                                               _vp]),
    "d3f_pairwise_topk_workspace_bytes": (_i64, [_i64, _i64]),
    "d3f_pairwise_similarity_topk": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i32, _f32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i64,
                                                    _vp]),
    "d3f_topk_smallest": (ctypes.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _vp, _i64, _vp]),
    "d3f_topk_merge": (ctypes.c_int, [_vp, _vp, _i64, _i32, _i64, _vp, _vp, _vp]),
    "d3f_pairwise_softmax_local": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i32, _f32, _i32, _i64, _vp, _vp, _vp, _i64,
                                                  _vp]),
    "d3f_point_order_locality": (ctypes.c_int, [_vp, _i64, _vp, _vp]),
    "d3f_points_probe": (ctypes.c_int, [_vp, _i64, _vp, _vp]),
    "d3f_rigid_transform": (ctypes.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "d3f_track_loss_grad": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp, _vp, _vp, _vp]),
    "d3f_track_step_scratch_bytes": (_i64, [_i32, _i32]),
    "d3f_track_step": (ctypes.c_int, [ctypes.POINTER(Views), ctypes.POINTER(ChannelMap), _vp, _i32, _i32, _vp, _f32, _f32, _f32, _f32, _f32,
                                      _f32, _f32, ctypes.POINTER(TrackState), _vp]),
    "d3f_track_run": (ctypes.c_int, [ctypes.POINTER(Views), ctypes.POINTER(ChannelMap), _vp, _i32, _i32, _vp, _f32, _f32, _f32, _f32, _f32,
                                     _f32, _f32, _i32, ctypes.POINTER(TrackState), _vp]),
    "d3f_track_run_max_keypoints": (_i32, []),
    "d3f_track_stall_word": (_i64, [_i32, _i32]),
    "d3f_rigid_update": (ctypes.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _f32, _vp]),
    "d3f_softmax_merge": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp, _vp]),
    "d3f_softmax_apply": (ctypes.c_int, [_vp, _i64, _i64, _f32, _vp, _vp]),
}

_lib = None


class D3FError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("libd3fields_hip: status %d: %s" % (code, text))
        self.code = code


def library_path():
    return _build.LIB_PATH


def load():
    """Loads (building first if the .so is absent and hipcc exists) and type-annotates the ABI."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if _build.is_stale():
        # missing, or built from other sources than the ones in the tree (content fingerprint, not mtimes):
        # rebuild -- raises if hipcc is unavailable, so a stale library is never loaded silently
        try:
            _build.build_library()
        except (RuntimeError, OSError) as exc:
            raise ImportError("libd3fields_hip.so is %s and hipcc is not available to build it: %s"
                              % ("missing" if not os.path.exists(path) else "out of date with csrc/", exc))
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    got = lib.d3f_abi_version()
    if got != ABI_VERSION:
        raise ImportError("libd3fields_hip ABI %d != binding ABI %d; rebuild with python -m d3fields_amd.build --force"
                          % (got, ABI_VERSION))
    _lib = lib
    return lib


def check(code):
    if code != OK:
        raise D3FError(code, load().d3f_last_error().decode("utf-8", "replace"))


def ptr(t):
    """Device (or host) address of a torch tensor as c_void_p; None -> NULL."""
    return _vp(t.data_ptr()) if t is not None else _vp(None)


def current_stream_handle(device):
    import torch
    return _vp(torch.cuda.current_stream(device).cuda_stream)
