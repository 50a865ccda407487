"""Descriptor-similarity helpers with the reference's names and semantics (utils/corr_utils.py).

Each function keeps the reference signature (argument names, meaning, accepted dist_type
values, NotImplementedError for unknown ones, assertion on mismatched channel counts) and runs
on the ROCm device through libd3fields_hip.so.  The [B1,B2,C] difference tensor of the
reference and its out-of-memory retry loop do not exist here.  No CPU / torch-op fallback.
"""
import numpy as np
import torch

from . import _lib

__all__ = ["compute_similarity", "compute_similarity_tensor", "compute_dist_tensor",
           "compute_similarity_tensor_multi", "nearest_descriptor"]

_DIST = {"l2": _lib.DIST_L2, "square": _lib.DIST_SQUARE}


def _dist_code(dist_type):
    if dist_type not in _DIST:
        raise NotImplementedError            # same exception as the reference for unknown dist_type
    return _DIST[dist_type]


def _need_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError("%s must be on the ROCm device; there is no CPU path" % what)


def _workspace(rows, cols, dev):
    nbytes = _lib.load().d3f_softmax_workspace_bytes(rows, cols)
    return torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev), nbytes


def _to_target(src, tgt, scale, dist_type, mode, channel_axis):
    """src [B, ...] with channels on `channel_axis` (1 or -1) against one target [C]."""
    code = _dist_code(dist_type)
    _need_cuda(src, "src_feat_map")
    dev = src.device
    src = src.to(torch.float32).contiguous()
    tgt = tgt.to(device=dev, dtype=torch.float32).contiguous()
    B = src.shape[0]
    if channel_axis == 1:
        C = src.shape[1]
        inner = src.numel() // max(B * C, 1)
        sb, si, sc = C * inner, 1, inner
        oshape = (B,) + tuple(src.shape[2:])
    else:
        C = src.shape[-1]
        inner = src.numel() // max(B * C, 1)
        sb, si, sc = inner * C, C, 1
        oshape = tuple(src.shape[:-1])
    out = torch.empty(oshape, dtype=torch.float32, device=dev)
    ws, ws_bytes = (None, 0)
    if mode == _lib.SIM_SOFTMAX_DIM0:
        ws, ws_bytes = _workspace(B, inner, dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().d3f_similarity_to_target(
            _lib.ptr(src), B, inner, C, sb, si, sc, _lib.ptr(tgt), float(scale), code, mode, _lib.ptr(out),
            _lib.ptr(ws), ws_bytes, _lib.current_stream_handle(dev)))
    return out


def compute_similarity(src_feat_map, tgt_feat, scale, dist_type="l2"):
    """[B,H,W,C] numpy, [C] numpy -> [B,H,W] numpy: exp(-dist*scale)  (corr_utils.py:4-19).  Output dtype follows the
    inputs like the reference's; the arithmetic is fp32 on the device."""
    assert src_feat_map.shape[-1] == tgt_feat.shape[0]
    _dist_code(dist_type)
    dev = torch.device("cuda", torch.cuda.current_device())
    src = torch.from_numpy(np.ascontiguousarray(src_feat_map, dtype=np.float32)).to(dev)
    tgt = torch.from_numpy(np.ascontiguousarray(tgt_feat, dtype=np.float32)).to(dev)
    out = _to_target(src, tgt, scale, dist_type, _lib.SIM_EXP, channel_axis=-1)
    res = out.cpu().numpy()
    assert res.shape == src_feat_map.shape[:3]
    # the numpy reference returns its input's dtype (float64 in -> float64 out); the values here are computed in fp32 on
    # the device (relative difference ~1e-7 from a float64 evaluation) and handed back in that dtype
    return res.astype(np.result_type(src_feat_map.dtype, tgt_feat.dtype), copy=False) if np.issubdtype(src_feat_map.dtype, np.floating) else res


def compute_similarity_tensor(src_feat_map, tgt_feat, scale, dist_type="l2"):
    """[B,C,*dim] tensor, [C] tensor -> [B,*dim]: softmax(-dist*scale, dim=0)  (corr_utils.py:21-42)."""
    assert src_feat_map.shape[1] == tgt_feat.shape[0]
    out = _to_target(src_feat_map, tgt_feat, scale, dist_type, _lib.SIM_SOFTMAX_DIM0, channel_axis=1)
    assert out.shape[0] == src_feat_map.shape[0]
    return out


def compute_dist_tensor(src_feat_map, tgt_feat, dist_type="l2"):
    """[B,C,*dim] tensor, [C] tensor -> [B,*dim] distances  (corr_utils.py:44-61)."""
    assert src_feat_map.shape[1] == tgt_feat.shape[0]
    return _to_target(src_feat_map, tgt_feat, 1.0, dist_type, _lib.SIM_DIST, channel_axis=1)


def _pairwise(src, tgt, scale, dist_type, mode, want_argmax):
    code = _dist_code(dist_type)
    _need_cuda(src, "src_feat_map")
    dev = src.device
    src = src.to(torch.float32).contiguous()
    tgt = tgt.to(device=dev, dtype=torch.float32).contiguous()
    B1, C = src.shape
    B2 = tgt.shape[0]
    out = torch.empty((B1, B2), dtype=torch.float32, device=dev)
    am = torch.empty(B2, dtype=torch.int64, device=dev) if want_argmax else None
    ws, ws_bytes = (None, 0)
    if mode == _lib.SIM_SOFTMAX_DIM0 or want_argmax:
        ws, ws_bytes = _workspace(B1, B2, dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().d3f_pairwise_similarity(
            _lib.ptr(src), _lib.ptr(tgt), B1, B2, C, float(scale), code, mode, _lib.ptr(out), _lib.ptr(am),
            _lib.ptr(ws), ws_bytes, _lib.current_stream_handle(dev)))
    return out, am


def compute_similarity_tensor_multi(src_feat_map, tgt_feats, src_pts, last_match_pts, scale, dist_type="l2"):
    """[B1,C], [B2,C] -> [B1,B2] with columns summing to 1  (corr_utils.py:63-106).

    src_pts / last_match_pts are accepted and ignored, exactly as in the reference (its
    temporal regulariser is commented out, corr_utils.py:97-100).
    """
    assert src_feat_map.shape[1] == tgt_feats.shape[1]
    assert len(src_feat_map.shape) == 2
    assert len(tgt_feats.shape) == 2
    out, _ = _pairwise(src_feat_map, tgt_feats, scale, dist_type, _lib.SIM_SOFTMAX_DIM0, False)
    assert out.shape[0] == src_feat_map.shape[0]
    assert out.shape[1] == tgt_feats.shape[0]
    return out


def nearest_descriptor(src_feat_map, tgt_feats, scale=1.0, dist_type="l2"):
    """Build-defined extension: the k=1 nearest-descriptor lookup that
    ``compute_similarity_tensor_multi(...).argmax(0)`` gives in the reference, fused into the
    same launch sequence.  Returns (similarity [B1,B2], index [B2] int64)."""
    assert src_feat_map.shape[1] == tgt_feats.shape[1]
    return _pairwise(src_feat_map, tgt_feats, scale, dist_type, _lib.SIM_SOFTMAX_DIM0, True)


def knn_descriptors(src_feat_map, tgt_feats, k, scale=1.0, dist_type="l2"):
    """Build-defined extension (the north star's "KNN correspondence lookup"): for every target
    descriptor the k (<= 8) nearest source descriptors, i.e. the k-NN generalisation of the reference's best match
    ``compute_similarity_tensor_multi(...).argmax(0)``.  Returns (similarity [B1,B2] -- the reference's softmax matrix,
    index [k,B2] int64 with row j = the j-th nearest source row, similarity at those rows [k,B2]).  Neighbours are
    ranked by distance (ties -> lower row index); one launch sequence, no [B1,B2,C] tensor, no torch.topk pass."""
    assert src_feat_map.shape[1] == tgt_feats.shape[1]
    if not 1 <= int(k) <= 8:
        raise ValueError("k must be in [1, 8]")
    code = _dist_code(dist_type)
    _need_cuda(src_feat_map, "src_feat_map")
    dev = src_feat_map.device
    src = src_feat_map.to(torch.float32).contiguous()
    tgt = tgt_feats.to(device=dev, dtype=torch.float32).contiguous()
    B1, C = src.shape
    B2 = tgt.shape[0]
    lib = _lib.load()
    out = torch.empty((B1, B2), dtype=torch.float32, device=dev)
    idx = torch.empty((int(k), B2), dtype=torch.int64, device=dev)
    val = torch.empty((int(k), B2), dtype=torch.float32, device=dev)
    ws_bytes = lib.d3f_pairwise_topk_workspace_bytes(B1, B2)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.d3f_pairwise_similarity_topk(_lib.ptr(src), _lib.ptr(tgt), B1, B2, C, float(scale), code,
                                                    _lib.SIM_SOFTMAX_DIM0, int(k), _lib.ptr(out), _lib.ptr(idx), _lib.ptr(val),
                                                    _lib.ptr(ws), ws_bytes, _lib.current_stream_handle(dev)))
    return out, idx, val
