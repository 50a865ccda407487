"""TEST INFRASTRUCTURE ONLY -- ctypes/numpy front end of oracle/d3f_oracle.c.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module (as the checker).  The product package d3fields_amd never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libd3f_oracle.so")
_SRC = os.path.join(_HERE, "d3f_oracle.c")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_i32p = ctypes.POINTER(ctypes.c_int)
_i64 = ctypes.c_int64


def build(force=False):
    """Compile the C restatement (gcc, recipe in oracle/Makefile)."""
    stale = (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(_SRC)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libd3f_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _fp(a):
    return a.ctypes.data_as(_f32p)


def _c32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def threads():
    """OpenMP threads d3f_oracle_eval runs on."""
    return int(lib().d3f_oracle_threads())


def eval_field(depth, K, Rt, pts, maps=(), mu=0.02, mode="eval", return_inter=False,
               return_margin=False):
    """Fusion.eval / eval_dist restated (see d3f_oracle_eval).

    depth [V,H,W], K [V,3,3], Rt [V,3,4], pts [n,3]; maps: sequence of channels-last
    [V,fh,fw,C] arrays.  Returns dict(dist, valid_mask, sets=[...], inter=[...], margin).
    """
    depth, K, Rt, pts = _c32(depth), _c32(K), _c32(Rt), _c32(pts)
    maps = [_c32(m) for m in maps]
    V, H, W = depth.shape
    n = pts.shape[0]
    ns = len(maps)
    out_dist = np.empty(n, np.float32)
    out_valid = np.empty(n, np.uint8)
    outs = [np.empty((n, m.shape[3]), np.float32) for m in maps]
    inters = [np.empty((V, n, m.shape[3]), np.float32) if return_inter else None for m in maps]
    margin = np.empty(n, np.float32) if return_margin else None
    MapArr = _f32p * max(ns, 1)
    IntArr = ctypes.c_int * max(ns, 1)
    c_maps = MapArr(*[_fp(m) for m in maps])
    c_outs = MapArr(*[_fp(o) for o in outs])
    c_int = MapArr(*[(_fp(t) if t is not None else None) for t in inters])
    fh = IntArr(*[m.shape[1] for m in maps])
    fw = IntArr(*[m.shape[2] for m in maps])
    C = IntArr(*[m.shape[3] for m in maps])
    rc = lib().d3f_oracle_eval(
        ctypes.c_int(V), ctypes.c_int(H), ctypes.c_int(W), _fp(depth), _fp(K), _fp(Rt), _fp(pts),
        _i64(n), ctypes.c_int(ns), c_maps, fh, fw, C, ctypes.c_float(mu),
        ctypes.c_int(0 if mode == "eval" else 1), _fp(out_dist),
        out_valid.ctypes.data_as(_u8p), c_outs, c_int if return_inter else None,
        _fp(margin) if margin is not None else None)
    if rc != 0:
        raise ValueError("d3f_oracle_eval rc=%d" % rc)
    return {"dist": out_dist, "valid_mask": out_valid.astype(bool), "sets": outs,
            "inter": inters, "margin": margin}


def onehot2instance(onehot):
    onehot = _c32(onehot)
    NI = onehot.shape[-1]
    n = onehot.size // NI
    out = np.empty(onehot.shape[:-1], np.uint8)
    lib().d3f_oracle_onehot2instance(_fp(onehot), _i64(n), ctypes.c_int(NI),
                                     out.ctypes.data_as(_u8p))
    return out


def instance2onehot(inst, NI):
    inst = np.ascontiguousarray(inst, dtype=np.uint8)
    out = np.empty(inst.shape + (NI,), np.uint8)
    lib().d3f_oracle_instance2onehot(inst.ctypes.data_as(_u8p), _i64(inst.size), ctypes.c_int(NI),
                                     out.ctypes.data_as(_u8p))
    return out.astype(bool)


_DT = {"l2": 0, "square": 1}


def _target_args(src, channel_axis):
    """src [B, ...] with channels at `channel_axis` (1 or -1) -> B, inner, C, strides."""
    src = _c32(src)
    B = src.shape[0]
    if channel_axis in (-1, src.ndim - 1):
        C = src.shape[-1]
        inner = src.size // (B * C)
        return src, B, inner, C, inner * C, C, 1, src.shape[:-1]
    assert channel_axis == 1
    C = src.shape[1]
    inner = src.size // (B * C)
    return src, B, inner, C, C * inner, 1, inner, (B,) + src.shape[2:]


def dist_to_target(src, tgt, dist_type="l2", channel_axis=1):
    src, B, inner, C, sb, si, sc, oshape = _target_args(src, channel_axis)
    tgt = _c32(tgt)
    out = np.empty(B * inner, np.float32)
    lib().d3f_oracle_dist_to_target(_fp(src), _i64(B), _i64(inner), ctypes.c_int(C), _i64(sb),
                                    _i64(si), _i64(sc), _fp(tgt), ctypes.c_int(_DT[dist_type]),
                                    _fp(out))
    return out.reshape(oshape)


def similarity_exp(src, tgt, scale, dist_type="l2", channel_axis=-1):
    src, B, inner, C, sb, si, sc, oshape = _target_args(src, channel_axis)
    tgt = _c32(tgt)
    out = np.empty(B * inner, np.float32)
    lib().d3f_oracle_similarity_exp(_fp(src), _i64(B), _i64(inner), ctypes.c_int(C), _i64(sb),
                                    _i64(si), _i64(sc), _fp(tgt), ctypes.c_float(scale),
                                    ctypes.c_int(_DT[dist_type]), _fp(out))
    return out.reshape(oshape)


def similarity_softmax(src, tgt, scale, dist_type="l2", channel_axis=1):
    src, B, inner, C, sb, si, sc, oshape = _target_args(src, channel_axis)
    tgt = _c32(tgt)
    out = np.empty(B * inner, np.float32)
    lib().d3f_oracle_similarity_softmax(_fp(src), _i64(B), _i64(inner), ctypes.c_int(C), _i64(sb),
                                        _i64(si), _i64(sc), _fp(tgt), ctypes.c_float(scale),
                                        ctypes.c_int(_DT[dist_type]), _fp(out))
    return out.reshape(oshape)


def pairwise(src, tgt, scale=1.0, dist_type="l2", mode="softmax", return_argmax=False):
    src, tgt = _c32(src), _c32(tgt)
    B1, C = src.shape
    B2 = tgt.shape[0]
    out = np.empty((B1, B2), np.float32)
    am = np.empty(B2, np.int64) if return_argmax else None
    lib().d3f_oracle_pairwise(_fp(src), _fp(tgt), _i64(B1), _i64(B2), ctypes.c_int(C),
                              ctypes.c_float(scale), ctypes.c_int(_DT[dist_type]),
                              ctypes.c_int(0 if mode == "softmax" else 1), _fp(out),
                              am.ctypes.data_as(ctypes.POINTER(_i64)) if am is not None else None)
    return (out, am) if return_argmax else out


def fps(pts, k, init_idx):
    """fps_np restated (d3f_oracle_fps): returns (index array [k], max remaining distance)."""
    pts = _c32(pts)
    n = pts.shape[0]
    idx = np.empty(k, np.int64)
    md = ctypes.c_float(0.0)
    rc = lib().d3f_oracle_fps(_fp(pts), _i64(n), ctypes.c_int(k), _i64(init_idx),
                              idx.ctypes.data_as(ctypes.POINTER(_i64)), ctypes.byref(md))
    if rc != 0:
        raise ValueError("d3f_oracle_fps rc=%d" % rc)
    return idx, float(md.value)
