"""CPU: the C-ABI library loads, exports every symbol include/d3fields_hip.h declares, and its
argument validation returns status codes (no compute call is made -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from d3fields_amd import _lib

HEADER = os.path.join(ROOT, "include", "d3fields_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(d3f_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load()
    names = declared_functions()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), "libd3fields_hip.so lacks %s" % n
        assert n in _lib.SIGNATURES, "ctypes binding lacks %s" % n
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_constants_match_header():
    lib = _lib.load()
    hdr = open(HEADER).read()
    assert lib.d3f_abi_version() == int(re.search(r"#define D3F_ABI_VERSION (\d+)", hdr).group(1)) == _lib.ABI_VERSION
    assert b"gfx950" in lib.d3f_version()
    for name, val in [("D3F_ERR_INVALID_ARG", _lib.ERR_INVALID_ARG), ("D3F_ERR_BAD_SHAPE", _lib.ERR_BAD_SHAPE),
                      ("D3F_ERR_BAD_DTYPE", _lib.ERR_BAD_DTYPE), ("D3F_ERR_BAD_LAYOUT", _lib.ERR_BAD_LAYOUT),
                      ("D3F_ERR_HIP", _lib.ERR_HIP), ("D3F_ERR_WORKSPACE", _lib.ERR_WORKSPACE),
                      ("D3F_MAX_VIEWS", _lib.MAX_VIEWS), ("D3F_MAX_MAPS", _lib.MAX_MAPS)]:
        assert int(re.search(r"#define %s \(?(-?\d+)\)?" % name, hdr).group(1)) == val, name


def test_struct_layouts_match_header():
    # d3f_views: 3 x int32 (+4 pad) + 4 pointers ; d3f_channel_map: ptr + 4 x int32 + 3 x int64 + ptr  (ABI 4: the trailing
    # pointers are the device words of d3f_map_check)
    assert ctypes.sizeof(_lib.Views) == 48 and _lib.Views.depth.offset == 16 and _lib.Views.depth_nonfinite.offset == 40
    assert ctypes.sizeof(_lib.ChannelMap) == 56 and _lib.ChannelMap.stride_v.offset == 24 and _lib.ChannelMap.nonfinite.offset == 48
    hdr = open(os.path.join(ROOT, "include", "d3fields_hip.h")).read()
    assert re.search(r"const float \*pose;[^}]*const uint32_t \*depth_nonfinite;[^}]*\} d3f_views;", hdr)
    assert re.search(r"int64_t stride_v, stride_y, stride_x;\s*const uint32_t \*nonfinite;[^}]*\} d3f_channel_map;", hdr)
    # d3f_col_stat travels between ranks as raw bytes: float, float, int64 = 16 B (tests/test_sharding_gloo.py uses it)
    hdr = open(os.path.join(ROOT, "include", "d3fields_hip.h")).read()
    assert re.search(r"typedef struct d3f_col_stat \{\s*float max_logit;[^}]*float sum_exp;[^}]*int64_t argmax;[^}]*\} d3f_col_stat;", hdr)


def _views(V=2, H=8, W=8, depth=1, K=1, pose=1):
    return _lib.Views(V, H, W, depth, K, pose)


def test_validation_returns_status_codes_not_aborts():
    lib = _lib.load()
    one = ctypes.c_void_p(16)
    # n == 0 is a no-op success even with NULL buffers (empty batch, reference returns empty tensors)
    assert lib.d3f_eval(ctypes.byref(_views()), None, 0, None, 0, 0.02, 0, None, None, None, None, None, 0, None) == 0
    assert lib.d3f_eval(None, one, 4, None, 0, 0.02, 0, one, one, None, None, None, 0, None) == _lib.ERR_INVALID_ARG
    assert b"views" in lib.d3f_last_error()
    assert lib.d3f_eval(ctypes.byref(_views(V=0)), one, 4, None, 0, 0.02, 0, one, one, None, None, None, 0, None) == _lib.ERR_BAD_SHAPE
    assert lib.d3f_eval(ctypes.byref(_views(V=65)), one, 4, None, 0, 0.02, 0, one, one, None, None, None, 0, None) == _lib.ERR_BAD_SHAPE
    assert lib.d3f_eval(ctypes.byref(_views(W=1)), one, 4, None, 0, 0.02, 0, one, one, None, None, None, 0, None) == _lib.ERR_BAD_SHAPE
    assert lib.d3f_eval(ctypes.byref(_views()), None, 4, None, 0, 0.02, 0, one, one, None, None, None, 0, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_eval(ctypes.byref(_views()), one, -1, None, 0, 0.02, 0, one, one, None, None, None, 0, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_eval(ctypes.byref(_views()), one, 4, None, 0, 0.0, 0, one, one, None, None, None, 0, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_eval(ctypes.byref(_views()), one, 4, None, 9, 0.02, 0, one, one, None, None, None, 0, None) == _lib.ERR_BAD_SHAPE
    maps = (_lib.ChannelMap * 1)(_lib.ChannelMap(16, 4, 4, 8, 7, 128, 32, 8))
    outs = (ctypes.c_void_p * 1)(16)
    assert lib.d3f_eval(ctypes.byref(_views()), one, 4, maps, 1, 0.02, 0, one, one, outs, None, None, 0, None) == _lib.ERR_BAD_DTYPE
    maps[0].dtype = 0
    maps[0].stride_x = 4            # stride_x < C: not channels-last
    assert lib.d3f_eval(ctypes.byref(_views()), one, 4, maps, 1, 0.02, 0, one, one, outs, None, None, 0, None) == _lib.ERR_BAD_LAYOUT
    assert lib.d3f_eval_backward(ctypes.byref(_views()), one, 4, None, 0, 0.02, None, None, None, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_eval_backward(ctypes.byref(_views()), None, 0, None, 0, 0.02, None, None, None, None) == 0
    assert lib.d3f_onehot2instance(one, 4, 0, one, None) == _lib.ERR_BAD_SHAPE
    assert lib.d3f_instance2onehot(None, 4, 3, one, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_similarity_to_target(one, 2, 2, 4, 8, 4, 1, one, 1.0, 5, 0, one, None, 0, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_similarity_to_target(one, 2, 2, 4, 8, 4, 1, one, 1.0, 0, 2, one, None, 0, None) == _lib.ERR_WORKSPACE
    assert lib.d3f_pairwise_similarity(one, one, 10, 10, 4, 1.0, 0, 2, one, None, None, 0, None) == _lib.ERR_WORKSPACE
    assert lib.d3f_pairwise_similarity(one, one, 10, 0, 4, 1.0, 0, 2, one, None, None, 0, None) == 0
    with pytest.raises(_lib.D3FError) as e:
        _lib.check(lib.d3f_eval_dist(None, one, 1, one, one, None))
    assert e.value.code == _lib.ERR_INVALID_ARG
    # fp16-stored maps
    maps[0].dtype, maps[0].stride_x = 1, 8
    maps[0].data = 18                # 2-byte aligned is enough for scalar fp16 lanes, 1-byte is not
    assert lib.d3f_eval(ctypes.byref(_views()), one, 0, maps, 1, 0.02, 0, one, one, outs, None, None, 0, None) == 0
    maps[0].data = 17
    assert lib.d3f_eval(ctypes.byref(_views()), one, 4, maps, 1, 0.02, 0, one, one, outs, None, None, 0, None) == _lib.ERR_BAD_LAYOUT
    # row-sharded softmax steps and the point-order probe
    assert lib.d3f_pairwise_softmax_local(one, one, 10, 10, 4, 1.0, 0, 0, one, None, one, 1 << 20, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_pairwise_softmax_local(one, one, 10, 10, 4, 1.0, 0, 0, one, one, None, 0, None) == _lib.ERR_WORKSPACE
    assert lib.d3f_pairwise_softmax_local(one, one, 10, 10, 4, 1.0, 0, -1, one, one, one, 1 << 20, None) == _lib.ERR_BAD_SHAPE
    assert lib.d3f_pairwise_softmax_local(one, one, 10, 10, 4, 1.0, 3, 0, one, one, one, 1 << 20, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_pairwise_softmax_local(None, one, 0, 0, 4, 1.0, 0, 0, None, None, None, 0, None) == 0       # no columns
    assert lib.d3f_softmax_merge(None, 2, 5, one, None, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_softmax_merge(one, -1, 5, one, None, None) == _lib.ERR_BAD_SHAPE
    assert lib.d3f_softmax_merge(None, 0, 0, None, None, None) == 0
    assert lib.d3f_softmax_apply(None, 3, 5, 1.0, one, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_softmax_apply(None, 0, 5, 1.0, None, None) == 0
    assert lib.d3f_point_order_locality(None, 5, one, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_point_order_locality(one, 5, None, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_point_order_locality(one, -5, one, None) == _lib.ERR_INVALID_ARG
    # tracking step
    assert lib.d3f_rigid_transform(one, 2, 5, one, one, one, None, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_rigid_transform(one, -1, 5, one, one, one, one, None) == _lib.ERR_BAD_SHAPE
    assert lib.d3f_track_loss_grad(one, one, one, one, 4, 0, 100.0, one, one, one, None) == _lib.ERR_BAD_SHAPE
    assert lib.d3f_track_loss_grad(one, None, one, one, 4, 8, 100.0, one, one, one, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_rigid_update(one, 2, 5, one, one, one, one, one, one, one, 1.0, 0.0, 0.9, 0.999, 1e-8, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_rigid_update(one, 2, 5, one, one, None, one, one, one, one, 1.0, 0.01, 0.9, 0.999, 1e-8, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_rigid_update(one, 0, 5, None, None, None, None, None, None, None, 1.0, 0.01, 0.9, 0.999, 1e-8, None) == 0


def test_validation_of_association_entry_points():
    lib = _lib.load()
    one = ctypes.c_void_p(16)
    lower = (ctypes.c_double * 3)(0.0, 0.0, 0.0)
    num = (ctypes.c_int32 * 3)(4, 4, 4)
    assert lib.d3f_pcd_to_index(None, 0, lower, 0.03, num, None, None, None) == 0
    assert lib.d3f_pcd_to_index(one, 5, None, 0.03, num, one, None, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_pcd_to_index(one, 5, lower, 0.0, num, one, None, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_pcd_to_index(None, 5, lower, 0.03, num, one, None, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_pcd_to_index(one, -1, lower, 0.03, num, one, None, None) == _lib.ERR_INVALID_ARG
    # one hash set of 8-byte slots at load factor <= 1/2, power-of-two capacity
    assert lib.d3f_vox_iou_workspace_bytes(0, 0) == 1024 * 8
    assert lib.d3f_vox_iou_workspace_bytes(3000, 1000) == 8192 * 8
    assert lib.d3f_vox_idx_iou(one, 10, one, 10, one, None, 0, None) == _lib.ERR_WORKSPACE
    assert lib.d3f_vox_idx_iou(one, 10, one, 10, one, one, 100, None) == _lib.ERR_WORKSPACE
    assert lib.d3f_vox_idx_iou(None, 10, one, 10, one, one, 1 << 20, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_vox_idx_iou(one, 10, one, 10, None, one, 1 << 20, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_vox_idx_iou(one, -1, one, 10, one, one, 1 << 20, None) == _lib.ERR_BAD_SHAPE
    assert lib.d3f_vox_idx_iou(one, 10, one, 10, one, ctypes.c_void_p(20), 1 << 20, None) == _lib.ERR_BAD_LAYOUT
    assert lib.d3f_erode(None, 0, 10, 3, 3, None, None) == 0
    assert lib.d3f_erode(one, 4, 4, 0, 3, ctypes.c_void_p(64), None) == _lib.ERR_BAD_SHAPE
    assert lib.d3f_erode(one, 4, 4, 3, 3, one, None) == _lib.ERR_INVALID_ARG              # in place is not supported
    assert lib.d3f_erode(one, 4, 4, 3, 3, None, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_fps_pixels(one, 0, 3, 0, one, None, one, None) == _lib.ERR_BAD_SHAPE   # fps_np asserts a non-empty set
    assert lib.d3f_fps_pixels(one, 5, 0, 0, None, None, None, None) == 0
    assert lib.d3f_fps_pixels(one, 5, 3, 5, one, None, one, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_fps_pixels(one, 5, 3, 0, one, None, None, None) == _lib.ERR_INVALID_ARG


def test_validation_of_topk_and_lattice_entry_points():
    lib = _lib.load()
    one = ctypes.c_void_p(16)
    assert lib.d3f_pairwise_topk_workspace_bytes(0, 10) == 0
    assert lib.d3f_pairwise_topk_workspace_bytes(100000, 300) > lib.d3f_softmax_workspace_bytes(100000, 300)
    assert lib.d3f_pairwise_similarity_topk(one, one, 10, 10, 4, 1.0, 0, 2, 9, one, one, one, one, 1 << 30, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_pairwise_similarity_topk(one, one, 10, 10, 4, 1.0, 0, 2, 0, one, one, one, one, 1 << 30, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_pairwise_similarity_topk(one, one, 10, 10, 4, 1.0, 0, 2, 3, one, None, one, one, 1 << 30, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_pairwise_similarity_topk(one, one, 10, 10, 4, 1.0, 0, 2, 3, one, one, one, None, 0, None) == _lib.ERR_WORKSPACE
    assert lib.d3f_pairwise_similarity_topk(one, one, 10, 0, 4, 1.0, 0, 2, 3, one, one, one, None, 0, None) == 0
    assert lib.d3f_pairwise_similarity_topk(one, one, 10, 10, 4, 1.0, 7, 2, 3, one, one, one, one, 1 << 30, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_lattice_probe(None, 5, one, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_lattice_probe(one, 5, None, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_lattice_probe(one, -1, one, None) == _lib.ERR_INVALID_ARG
    assert lib.d3f_eval_lattice(ctypes.byref(_views()), None, 0, 4, 4, None, 0, 0.02, 0, None, None, None, None, None) == 0
    assert lib.d3f_eval_lattice(ctypes.byref(_views()), one, -1, 4, 4, None, 0, 0.02, 0, one, one, None, None, None) == _lib.ERR_BAD_SHAPE
    assert lib.d3f_eval_lattice(ctypes.byref(_views()), None, 4, 4, 4, None, 0, 0.02, 0, one, one, None, None, None) == _lib.ERR_INVALID_ARG
    # launch plans of lattices (host logic only): brick walk on large maps, cell runs in caller order on patch-res maps
    plan = _lib.EvalPlan()
    big = (_lib.ChannelMap * 1)(_lib.ChannelMap(1 << 20, 480, 640, 384, 0, 480 * 640 * 384, 640 * 384, 384))
    v = _views(V=4, H=480, W=640)
    assert lib.d3f_eval_plan_query_lattice(ctypes.byref(v), 160, 140, 44, big, 1, _lib.FLAG_FINITE_MAPS, 0, ctypes.byref(plan)) == 0
    # a dense wide map alone: channel-sliced launch, 16 points (four 2 x 2 x 1 tiles) x one 512-byte slice per workgroup,
    # units of 256 workgroups spread over the XCDs
    assert (plan.reorder, plan.tile_points, plan.reserved, plan.staged[0]) == (2, 16, 152, 0)
    assert plan.workgroups == ((80 * 70 * 44 // 4 + 255) // 256 * 3 + 7) // 8 * 8 * 256
    assert lib.d3f_eval_plan_query_lattice(ctypes.byref(v), 161, 141, 45, big, 1, _lib.FLAG_FINITE_MAPS, 0, ctypes.byref(plan)) == 0
    assert plan.tile_points == 16 and plan.workgroups == ((81 * 71 * 45 // 4 + 1 + 255) // 256 * 3 + 7) // 8 * 8 * 256
    odd = (_lib.ChannelMap * 1)(_lib.ChannelMap(1 << 20, 480, 640, 200, 0, 480 * 640 * 200, 640 * 200, 200))
    assert lib.d3f_eval_plan_query_lattice(ctypes.byref(v), 161, 141, 45, odd, 1, _lib.FLAG_FINITE_MAPS, 0, ctypes.byref(plan)) == 0
    # no whole 512-byte slices: the whole-texel kernel on 2 x 2 x 4 bricks (16 lanes per point), clipped at the upper
    # faces, no padding
    assert (plan.reorder, plan.tile_points, plan.reserved, plan.workgroups) == (2, 16, 0, 81 * 71 * 12)
    both = (_lib.ChannelMap * 2)(_lib.ChannelMap(1 << 20, 480, 640, 384, 0, 480 * 640 * 384, 640 * 384, 384),
                                 _lib.ChannelMap(1 << 20, 480, 640, 8, 0, 480 * 640 * 8, 640 * 8, 8))
    assert lib.d3f_eval_plan_query_lattice(ctypes.byref(v), 200, 175, 55, both, 2, _lib.FLAG_FINITE_MAPS, 0, ctypes.byref(plan)) == 0
    # a wide map with a thin companion on a lattice: channel-sliced launch, 512-byte slices (32 lanes), two views in flight
    assert plan.reorder == 2 and plan.reserved == 152 and plan.tile_points == 32 and plan.workgroups % 1024 == 0
    patch = (_lib.ChannelMap * 1)(_lib.ChannelMap(1 << 20, 48, 64, 384, 0, 48 * 64 * 384, 64 * 384, 384))
    assert lib.d3f_eval_plan_query_lattice(ctypes.byref(v), 160, 140, 44, patch, 1, _lib.FLAG_FINITE_MAPS, 0, ctypes.byref(plan)) == 0
    # a patch-resolution wide map on a lattice: 4 x 4 x 4 bricks through LDS texel windows, 4 workgroups per CU
    assert (plan.reorder, plan.staged[0], plan.tile_points, plan.reserved, plan.workgroups) == (2, 3, 64, 2113, 40 * 35 * 11)
    assert lib.d3f_eval_plan_query(ctypes.byref(v), 985600, patch, 1, _lib.FLAG_FINITE_MAPS, 1, 0, ctypes.byref(plan)) == 0
    # the same map without lattice dims (a cloud in caller order): the cell-run gather
    assert plan.reorder == 0 and plan.staged[0] == 16 + 4 and plan.tile_points == 64 and plan.lanes_per_point[0] == 32
    assert plan.vectors_per_lane[0] == 1 and plan.reserved == 7          # the (1,4) variant built for 7 waves per SIMD
    wide = (_lib.ChannelMap * 1)(_lib.ChannelMap(1 << 20, 72, 128, 512, 0, 72 * 128 * 512, 128 * 512, 512))
    v8 = _views(V=8, H=720, W=1280)
    assert lib.d3f_eval_plan_query(ctypes.byref(v8), 200000, wide, 1, _lib.FLAG_FINITE_MAPS | _lib.FLAG_UNORDERED_POINTS, 1, 0, ctypes.byref(plan)) == 0
    assert plan.reorder == 1 and plan.staged[0] == 16 + 8 and plan.vectors_per_lane[0] == 2 and plan.lanes_per_point[0] == 64
    assert plan.reserved == 3                                            # (2,8) at 3 waves per SIMD: spill-free
    # config 4's own map (1024 fp32 channels, 8 views): the register-rows kernel, 32 points per workgroup, Hilbert order, no gate
    wide = (_lib.ChannelMap * 1)(_lib.ChannelMap(1 << 20, 72, 128, 1024, 0, 72 * 128 * 1024, 128 * 1024, 1024))
    assert lib.d3f_eval_plan_query(ctypes.byref(v8), 1000000, wide, 1, _lib.FLAG_FINITE_MAPS, 1, 0, ctypes.byref(plan)) == 0
    assert (plan.family, plan.reorder, plan.staged[0], plan.tile_points, plan.workgroups, plan.gated_window) == (5, 1, 5, 32, 31250, 0)
    assert (plan.lanes_per_point[0], plan.vectors_per_lane[0]) == (256, 1) and plan.lds_bytes <= 32 * 1024
    assert lib.d3f_eval_plan_query_lattice(ctypes.byref(v8), 40, 280, 88, wide, 1, _lib.FLAG_FINITE_MAPS, 0, ctypes.byref(plan)) == 0
    assert (plan.family, plan.reorder, plan.tile_points, plan.workgroups) == (5, 2, 32, 10 * 70 * 44)      # 4 x 4 x 2 bricks of the lattice
    assert lib.d3f_eval_plan_query_lattice(ctypes.byref(v), 160, 140, 44, patch, 1, 0, 0, ctypes.byref(plan)) == 0
    assert plan.staged[0] == 0 and plan.tile_points == 128     # maps not known to be finite: the direct gather


def test_workspace_size():
    lib = _lib.load()
    assert lib.d3f_eval_workspace_bytes(0) == 0
    assert lib.d3f_eval_workspace_bytes(1000000) >= 16 * 1000000
    # the distance-only pass: a tiled copy of the depth maps (4 x 8-pixel tiles of 128 bytes) from 2^22 points on, up to eight views
    v4 = _lib.Views(4, 480, 640, None, None, None)
    assert lib.d3f_eval_dist_workspace_bytes(ctypes.byref(v4), 123200000) == 4 * 60 * 160 * 128
    assert lib.d3f_eval_dist_workspace_bytes(ctypes.byref(v4), (1 << 22) - 1) == 0
    assert lib.d3f_eval_dist_workspace_bytes(ctypes.byref(_lib.Views(3, 243, 321, None, None, None)), 1 << 22) == 3 * 31 * 81 * 128
    assert lib.d3f_eval_dist_workspace_bytes(ctypes.byref(_lib.Views(9, 480, 640, None, None, None)), 1 << 23) == 0
    assert lib.d3f_eval_dist_workspace_bytes(None, 1 << 23) == 0
    assert lib.d3f_softmax_workspace_bytes(0, 10) == 0
    assert lib.d3f_softmax_workspace_bytes(1, 1) == 2 * 16
    assert lib.d3f_softmax_workspace_bytes(100000, 300) == (1563 + 1) * 300 * 16      # one 16-B record per 64-row tile and column


def _plan(V, H, W, n, maps, flags=0, ws=1, inter=0, dtype=0):
    lib = _lib.load()
    v = _lib.Views(V, H, W, 16, 16, 16)
    arr = (_lib.ChannelMap * max(len(maps), 1))()
    for i, (fh, fw, C) in enumerate(maps):
        arr[i] = _lib.ChannelMap(16, fh, fw, C, dtype, fh * fw * C, fw * C, C)
    p = _lib.EvalPlan()
    assert lib.d3f_eval_plan_query(ctypes.byref(v), n, arr, len(maps), flags, ws, inter, ctypes.byref(p)) == 0
    return p


def test_launch_plan_host_logic():
    """The launch geometry / lane mapping the host picks (no GPU needed: d3f_eval_plan_query)."""
    # C2 patch-res: cache-resident map -> caller order, 128-point tiles, 32 lanes x 3 float4 per point, batched loads
    p = _plan(4, 480, 640, 985600, [(48, 64, 384)])
    assert (p.tile_points, p.reorder, p.workgroups) == (128, 0, 7700)
    assert (p.vector_floats[0], p.lanes_per_point[0], p.vectors_per_lane[0], p.staged[0]) == (4, 32, 3, 0)
    # the same maps with a cloud the caller declares unordered: Morton walk, but still the big tiles (maps are cache-resident)
    p = _plan(4, 480, 640, 985600, [(48, 64, 384)], flags=_lib.FLAG_UNORDERED_POINTS)
    assert (p.tile_points, p.reorder, p.workgroups, p.vectors_per_lane[0]) == (128, 1, 7700, 3)
    p = _plan(4, 480, 640, 985600, [(48, 64, 384)], flags=_lib.FLAG_UNORDERED_POINTS, ws=0)
    assert p.reorder == 0                                      # no scratch, no walk
    assert _plan(4, 480, 640, 60000, [(48, 64, 384)], flags=_lib.FLAG_UNORDERED_POINTS).reorder == 0
    # fp16-stored dense maps: 0.94 GB; 8 channels per 16-B load -> 16 lanes x 3 vectors per point, 16-point tiles on the walk
    p = _plan(4, 480, 640, 985600, [(480, 640, 384)], dtype=_lib.DTYPE_F16)
    assert (p.tile_points, p.reorder, p.vector_floats[0], p.lanes_per_point[0], p.vectors_per_lane[0]) == (16, 1, 8, 16, 3)
    p = _plan(4, 480, 640, 1000, [(48, 64, 5)], dtype=_lib.DTYPE_F16)             # odd channel count: scalar fp16 lanes
    assert (p.vector_floats[0], p.lanes_per_point[0], p.vectors_per_lane[0]) == (1, 8, 1)      # thin family: one pass over 8 lanes (3 idle)
    p = _plan(4, 480, 640, 985600, [(48, 64, 384)], dtype=_lib.DTYPE_F16)
    assert (p.tile_points, p.reorder) == (128, 0)
    # C2 dense: 1.9 GB of maps -> Morton walk feeding the channel-sliced kernel (16-point tiles, 512-byte slices, two views
    # in flight: reserved 152); the direct gather on the same walk: 8-point tiles, 3 batched float4 per lane; without
    # scratch: 64-point tiles in caller order
    p = _plan(4, 480, 640, 985600, [(480, 640, 384)])
    assert (p.tile_points, p.reorder, p.reserved, p.workgroups) == (16, 1, 152, (241 * 3 + 7) // 8 * 8 * 256)
    p = _plan(4, 480, 640, 985600, [(480, 640, 384)], flags=_lib.TUNE_DIRECT_GATHER)
    assert (p.tile_points, p.reorder, p.vectors_per_lane[0], p.reserved) == (8, 1, 3, 0)
    p = _plan(4, 480, 640, 985600, [(480, 640, 384)], ws=0)
    assert (p.tile_points, p.reorder) == (64, 0)
    # C4: 8 views x 1024 channels -> whole wave per point, 4 float4 per lane
    p = _plan(8, 720, 1280, 1000000, [(72, 128, 1024)])
    assert (p.lanes_per_point[0], abs(p.vectors_per_lane[0]), p.reorder) == (64, 4, 1)
    # small batches are never reordered; mask (C=8) -> 2 lanes per point; colour (C=3) -> 4 scalar lanes (one idle)
    p = _plan(4, 480, 640, 60000, [(48, 64, 384), (480, 640, 8), (480, 640, 3)])
    assert p.reorder == 0 and p.tile_points == 32            # < 1024 workgroups of 128 -> smaller tiles
    assert _plan(4, 480, 640, 300, [(48, 64, 384)]).tile_points == 8
    assert (p.vector_floats[1], p.lanes_per_point[1], p.vectors_per_lane[1]) == (4, 2, 1)
    assert (p.vector_floats[2], p.lanes_per_point[2], p.vectors_per_lane[2]) == (1, 4, 1)       # thin family: one pass, one vector per lane
    # the distance-only pass (return_names=[]): one lane per point, four points per lane on big batches, nothing per point in LDS
    p = _plan(4, 480, 640, 123200000, [])
    assert (p.tile_points, p.reorder, p.workgroups) == (4096, 0, (123200000 + 4095) // 4096) and p.lds_bytes <= 512
    assert _plan(4, 480, 640, 5000000, []).tile_points == 1024
    assert _plan(4, 480, 640, 100000, []).tile_points == 256
    # many views shrink the tile so that the per-(point,view) records fit LDS
    p = _plan(64, 48, 64, 5000, [(6, 8, 16)])
    assert p.tile_points * 64 * 24 <= 64 * 1024 and p.lds_bytes <= 64 * 1024
    # D3F_TUNE_DIRECT_GATHER: no cell runs / windows / slices, the chosen point order stays
    p = _plan(4, 480, 640, 985600, [(48, 64, 384)], flags=_lib.FLAG_FINITE_MAPS)
    assert p.staged[0] == 16 + 4
    p = _plan(4, 480, 640, 985600, [(48, 64, 384)], flags=_lib.FLAG_FINITE_MAPS | _lib.TUNE_DIRECT_GATHER)
    assert (p.staged[0], p.reorder, p.tile_points, p.reserved) == (0, 0, 128, 0)


def test_window_launch_plan(monkeypatch):
    """LDS texel windows replace the cell-run gather for a patch-resolution wide map on a lattice (D3F_TUNE_DIRECT_GATHER
    switches them off; in experiments builds D3F_EXP_WINDOW forces them for clouds too); infeasible shapes fall back
    instead of failing."""
    lib = _lib.load()
    exp = bool(lib.d3f_build_has_experiments())

    def plan_lattice(V, dims, maps, flags=_lib.FLAG_FINITE_MAPS):
        v = _lib.Views(V, 480, 640, 16, 16, 16)
        arr = (_lib.ChannelMap * len(maps))()
        for i, (fh, fw, C) in enumerate(maps):
            arr[i] = _lib.ChannelMap(16, fh, fw, C, 0, fh * fw * C, fw * C, C)
        p = _lib.EvalPlan()
        assert lib.d3f_eval_plan_query_lattice(ctypes.byref(v), dims[0], dims[1], dims[2], arr, len(maps), flags, 0, ctypes.byref(p)) == 0
        return p

    p = plan_lattice(4, (160, 140, 44), [(48, 64, 384)])
    assert p.staged[0] == 3 and p.tile_points == 64                 # default on a lattice
    # the wide map need not come first in the call (return_names=['mask', 'dino_feats']): same launch, reported in caller order
    p = plan_lattice(4, (160, 140, 44), [(480, 640, 8), (48, 64, 384)])
    assert (p.staged[0], p.staged[1], p.lanes_per_point[0], p.lanes_per_point[1], p.reserved) == (0, 3, 2, 16, 2113)
    big_first = plan_lattice(4, (200, 175, 55), [(480, 640, 384), (480, 640, 8)])
    big_last = plan_lattice(4, (200, 175, 55), [(480, 640, 8), (480, 640, 384)])
    assert big_first.reserved == big_last.reserved == 152 and big_first.workgroups == big_last.workgroups
    p = plan_lattice(4, (160, 140, 44), [(48, 64, 384)], flags=_lib.FLAG_FINITE_MAPS | _lib.TUNE_DIRECT_GATHER)
    assert (p.staged[0], p.reorder, p.reserved) == (0, 0, 0)                          # switched off: direct gather, caller order
    p = plan_lattice(4, (160, 140, 44), [(48, 64, 384), (480, 640, 8)])
    assert (p.staged[0], p.tile_points, p.reorder, p.workgroups, p.reserved) == (3, 64, 2, 40 * 35 * 11, 2113)      # ~17 pool slots per view: three workgroups per CU
    assert 160 * 1024 // 4 < p.lds_bytes <= 160 * 1024 // 3
    if exp:
        monkeypatch.setenv("D3F_EXP_WINDOW", "-1")
        assert plan_lattice(4, (160, 140, 44), [(48, 64, 384)]).staged[0] == 16 + 4      # windows off: cell runs, caller order
        monkeypatch.setenv("D3F_EXP_WINDOW", "64")
        monkeypatch.setenv("D3F_EXP_WINDOW_U", "3")
        p = plan_lattice(4, (160, 140, 44), [(48, 64, 384)])
        assert (p.staged[0], p.reserved) == (3, 2312) and 64 * 1024 < p.lds_bytes <= 80 * 1024
        monkeypatch.delenv("D3F_EXP_WINDOW_U")
        monkeypatch.delenv("D3F_EXP_WINDOW")
    assert plan_lattice(4, (160, 140, 44), [(48, 64, 384)], flags=0).staged[0] == 0      # maps not known finite: direct gather
    assert plan_lattice(4, (160, 140, 44), [(48, 64, 200)]).staged[0] != 3               # no whole 512-byte slices
    assert plan_lattice(4, (160, 140, 44), [(480, 640, 384)]).staged[0] == 0             # dense map: not a window case
    assert plan_lattice(16, (160, 140, 44), [(48, 64, 384)]).staged[0] != 3              # more than 8 views
    p = _plan(8, 720, 1280, 1000000, [(72, 128, 512)], flags=_lib.FLAG_FINITE_MAPS | _lib.FLAG_UNORDERED_POINTS)
    assert (p.staged[0], p.reorder, p.gated_window) == (16 + 8, 1, 1)                    # a big cloud: the plan describes the cell-run side of the gated pair ...
    assert 2000 <= p.reserved2 < 3000 and p.reserved2 % 10 == 3                          # ... and names the window side's kernel (3 workgroups per CU and fewer)
    p = _plan(8, 720, 1280, 1000000, [(72, 128, 512)], flags=_lib.FLAG_FINITE_MAPS | _lib.FLAG_UNORDERED_POINTS | _lib.TUNE_NO_WINDOW_GATE)
    assert (p.staged[0], p.reorder, p.gated_window) == (16 + 8, 1, 0)
    # 1024 fp32 channels: the register rows take the cloud with more than four views (no gate), with four views they are the
    # OTHER side of the window kernel's gate, and they take what the windows do not (a cloud below kWindowCloudMin)
    p = _plan(8, 720, 1280, 1000000, [(72, 128, 1024)], flags=_lib.FLAG_FINITE_MAPS | _lib.FLAG_UNORDERED_POINTS)
    assert (p.family, p.staged[0], p.reorder, p.gated_window) == (5, 5, 1, 0)
    p = _plan(4, 480, 640, 1000000, [(48, 64, 1024)], flags=_lib.FLAG_FINITE_MAPS | _lib.FLAG_UNORDERED_POINTS)
    assert (p.family, p.staged[0], p.reorder, p.gated_window, p.reserved2) == (5, 5, 1, 1, 2113)
    p = _plan(4, 480, 640, 100000, [(48, 64, 1024)], flags=_lib.FLAG_FINITE_MAPS)
    assert (p.family, p.staged[0], p.reorder, p.gated_window, p.tile_points) == (5, 5, 0, 0, 32)
    assert plan_lattice(4, (160, 140, 44), [(48, 64, 1024)]).staged[0] == 3               # four views on a lattice: the windows
    assert _plan(4, 480, 640, 100000, [(48, 64, 1024)], flags=0).family == 4             # maps not known to be finite
    p = _plan(4, 480, 640, 262143, [(48, 64, 384)], flags=_lib.FLAG_FINITE_MAPS | _lib.FLAG_UNORDERED_POINTS)
    assert (p.staged[0], p.reorder, p.gated_window) == (16 + 4, 1, 0)                    # kWindowCloudMin = 262 144 points: below it cell runs only
    p = _plan(4, 480, 640, 262144, [(48, 64, 384)], flags=_lib.FLAG_FINITE_MAPS | _lib.FLAG_UNORDERED_POINTS)
    assert (p.staged[0], p.reorder, p.gated_window, p.reserved2) == (16 + 4, 1, 1, 2113)
    assert _plan(4, 480, 640, 985600, [(48, 64, 384)], flags=_lib.FLAG_FINITE_MAPS).gated_window == 0      # caller order (not declared unordered): no gate
    assert _plan(4, 480, 640, 985600, [(48, 64, 384)], flags=_lib.FLAG_UNORDERED_POINTS).gated_window == 0   # maps not known finite
    assert _plan(4, 480, 640, 985600, [(48, 64, 384)], flags=_lib.FLAG_FINITE_MAPS | _lib.FLAG_UNORDERED_POINTS, ws=0).gated_window == 0
    if exp:
        monkeypatch.setenv("D3F_EXP_ROWS", "-1")
        p = _plan(8, 720, 1280, 1000000, [(72, 128, 1024)], flags=_lib.FLAG_FINITE_MAPS | _lib.FLAG_UNORDERED_POINTS)
        assert (p.staged[0], p.reorder, p.gated_window) == (16 + 8, 1, 1)                # without the register rows: rounds 2-5's gated pair
        monkeypatch.delenv("D3F_EXP_ROWS")
        monkeypatch.setenv("D3F_EXP_WINDOW", "64")
        p = _plan(8, 720, 1280, 1000000, [(72, 128, 1024)], flags=_lib.FLAG_FINITE_MAPS | _lib.FLAG_UNORDERED_POINTS)
        assert (p.staged[0], p.reorder, p.workgroups) == (3, 1, 15625)                   # forced: 64 consecutive points of the Morton order
        monkeypatch.setenv("D3F_EXP_WINDOW", "128")
        p = _plan(8, 720, 1280, 1000000, [(72, 128, 1024)], flags=_lib.FLAG_FINITE_MAPS | _lib.FLAG_UNORDERED_POINTS)
        assert p.staged[0] == 3 and p.lds_bytes <= 160 * 1024 // 2    # 32 KB of records for 128 x 8 pairs + ~11 pool slots per view: 2 workgroups per CU


def test_family_table():
    """The planner is a walk over a table of kernel families (csrc/d3f_plan.h): every row is reachable, reports its id, and the
    library names it."""
    lib = _lib.load()
    F = _lib.FLAG_FINITE_MAPS
    names = [lib.d3f_plan_family_name(k) for k in range(7)]
    assert names == [b"dist-only", b"lds-window", b"cell-runs", b"channel-sliced", b"direct", b"register-rows", None]
    assert all(lib.d3f_plan_family_takes(k) for k in range(6)) and lib.d3f_plan_family_takes(7) is None
    assert _plan(4, 480, 640, 100000, []).family == 0                                         # return_names=[]
    v = _lib.Views(4, 480, 640, 16, 16, 16)
    arr = (_lib.ChannelMap * 1)(_lib.ChannelMap(16, 48, 64, 384, 0, 48 * 64 * 384, 64 * 384, 384))
    p = _lib.EvalPlan()
    assert lib.d3f_eval_plan_query_lattice(ctypes.byref(v), 160, 140, 44, arr, 1, F, 0, ctypes.byref(p)) == 0 and p.family == 1
    assert _plan(4, 480, 640, 985600, [(48, 64, 384)], F).family == 2                          # patch-resolution map, caller order: cell runs
    assert _plan(4, 480, 640, 985600, [(480, 640, 384)], F).family == 3                        # dense map on the Hilbert walk: channel slices
    assert _plan(4, 480, 640, 985600, [(480, 640, 384)], F | _lib.TUNE_DIRECT_GATHER).family == 4
    assert _plan(4, 480, 640, 300, [(48, 64, 384)], F).family == 4                             # a small batch
    assert _plan(4, 480, 640, 985600, [(48, 64, 384)], 0).family == 4                          # maps not known to be finite
    assert _plan(8, 720, 1280, 985600, [(72, 128, 1024)], F).family == 5                       # 1024 fp32 channels, eight views: the rows in registers


def test_plan_table():
    """One row per launch-plan threshold of d3f_api.hip (kSmallBatch, kCacheResidentBytes, kInfinityCacheBytes, kBatchedLoadBytes, kBeyondLlcBytes):
    the plan on either side of each boundary, from d3f_eval_plan_query alone."""
    F = _lib.FLAG_FINITE_MAPS

    def one_map(nbytes):                        # one view, 96 channels (384-byte texels: a wide map, not sliceable), 1024 texels per row
        return [(nbytes // 384 // 1024, 1024, 96)]

    # kSmallBatch = 65536 points: below it nothing is reordered, whatever the maps and the scratch
    assert _plan(1, 480, 640, 65535, one_map(200 << 20), F).reorder == 0
    assert _plan(1, 480, 640, 65536, one_map(200 << 20), F).reorder == 1
    # kCacheResidentBytes = 64 MiB of maps: at most that -> caller order with 128-point tiles; more (with scratch) -> Morton walk
    p = _plan(1, 480, 640, 200000, one_map(64 << 20), F)
    assert (p.reorder, p.tile_points) == (0, 128)
    p = _plan(1, 480, 640, 200000, one_map((64 << 20) + (384 << 10)), F)
    assert (p.reorder, p.tile_points) == (1, 16)
    assert _plan(1, 480, 640, 200000, one_map((64 << 20) + (384 << 10)), F, ws=0).reorder == 0      # no scratch, no sort
    # D3F_FLAG_LOCAL_POINTS (ABI 6) + kInfinityCacheBytes = 256 MiB + kWindowCloudMin: a small cloud whose caller order is local keeps
    # that order on maps inside the Infinity Cache -- not on larger maps, not at 262 144 points, and UNORDERED wins over LOCAL
    L = _lib.FLAG_LOCAL_POINTS
    assert _plan(1, 480, 640, 200000, one_map(200 << 20), F | L).reorder == 0
    assert _plan(1, 480, 640, 200000, one_map(256 << 20), F | L).reorder == 0
    assert _plan(1, 480, 640, 200000, one_map((256 << 20) + (384 << 10)), F | L).reorder == 1
    assert _plan(1, 480, 640, 262143, one_map(200 << 20), F | L).reorder == 0
    assert _plan(1, 480, 640, 262144, one_map(200 << 20), F | L).reorder == 1
    assert _plan(1, 480, 640, 200000, one_map(200 << 20), F | L | _lib.FLAG_UNORDERED_POINTS).reorder == 1
    assert _plan(1, 480, 640, 200000, one_map(200 << 20), F | L | _lib.TUNE_FORCE_REORDER).reorder == 1
    # kBatchedLoadBytes = 128 MiB per map: batched corner loads up to it, load-use per vector beyond (caller order)
    assert _plan(1, 480, 640, 200000, one_map(128 << 20), F, ws=0).vectors_per_lane[0] == 3          # 24 float4 = 8 lanes x 3
    assert _plan(1, 480, 640, 200000, one_map((128 << 20) + (384 << 10)), F, ws=0).vectors_per_lane[0] == -3
    # ... but a thin map (<= 256 bytes per texel) has ONE kernel form: one batched vector per lane
    assert _plan(1, 480, 640, 200000, [((200 << 20) // 256 // 1024, 1024, 64)], F, ws=0).vectors_per_lane[0] == 1
    # kBeyondLlcBytes = 512 MiB of maps in caller order without scratch: 64-point tiles at 2 workgroups per CU (64 KiB LDS pad)
    p = _plan(1, 480, 640, 200000, one_map(512 << 20), F, ws=0)
    assert p.tile_points == 128 and p.lds_bytes < 16 * 1024
    p = _plan(1, 480, 640, 200000, one_map((512 << 20) + (384 << 10)), F, ws=0)
    assert p.tile_points == 64 and p.lds_bytes > 64 * 1024
    # D3F_TUNE_DIRECT_GATHER never changes the point order, only the gather
    assert _plan(1, 480, 640, 200000, one_map((64 << 20) + (384 << 10)), F | _lib.TUNE_DIRECT_GATHER).reorder == 1
    # channel-sliced kernel: one wide fp32 map of 128..1024 channels in whole 512-byte slices, beyond the caches, on the
    # Morton walk (or a lattice, test_window_launch_plan); 1152 channels, 192 channels, a second wide map: whole texels
    assert _plan(4, 480, 640, 200000, [(480, 640, 1024)], F).reserved == 152
    assert _plan(4, 480, 640, 200000, [(480, 640, 128)], F).reserved == 152
    assert _plan(4, 480, 640, 200000, [(480, 640, 1152)], F).reserved == 0
    assert _plan(4, 480, 640, 200000, [(480, 640, 192)], F).reserved == 0
    assert _plan(4, 480, 640, 200000, [(480, 640, 384), (480, 640, 128)], F).reserved == 0
    assert _plan(4, 480, 640, 200000, [(480, 640, 384), (480, 640, 8)], F).reserved == 152      # a thin map rides along
    # ... and only while its records fit the 64 KiB of dynamic LDS a launch gets without opting in (32-point tiles x 36+ views do not)
    assert _plan(8, 480, 640, 200000, [(480, 640, 384), (480, 640, 8)], F).reserved == 152
    p = _plan(40, 480, 640, 200000, [(480, 640, 384), (480, 640, 8)], F)
    assert (p.reserved, p.tile_points) == (0, 16) and p.lds_bytes <= 64 * 1024
