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
    # d3f_views: 3 x int32 (+4 pad) + 3 pointers ; d3f_channel_map: ptr + 4 x int32 + 3 x int64
    assert ctypes.sizeof(_lib.Views) == 40 and _lib.Views.depth.offset == 16
    assert ctypes.sizeof(_lib.ChannelMap) == 48 and _lib.ChannelMap.stride_v.offset == 24


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


def test_workspace_size():
    lib = _lib.load()
    assert lib.d3f_eval_workspace_bytes(0) == 0
    assert lib.d3f_eval_workspace_bytes(1000000) >= 16 * 1000000
    assert lib.d3f_softmax_workspace_bytes(0, 10) == 0
    assert lib.d3f_softmax_workspace_bytes(1, 1) == 2 * 16
    assert lib.d3f_softmax_workspace_bytes(100000, 300) == (391 + 1) * 300 * 16
