"""Point-cloud helpers on the mask side of the path, with the reference's names and semantics:
``depth2fgpcd`` (utils/my_utils.py:522-537), ``aggr_point_cloud_from_data`` (utils/draw_utils.py:325-413,
numpy outputs only) and ``pcd_iou`` (Fusion.pcd_iou, fusion.py:724-741).  numpy in / numpy out like the
reference; the per-pixel and per-pair work runs on the ROCm device in fp64.  open3d (voxel down-sampling,
o3d point clouds) is an upstream dependency and is not reproduced: downsample=True / out_o3d=True raise.
"""
import ctypes

import numpy as np
import torch

from . import _lib

__all__ = ["depth2fgpcd", "aggr_point_cloud_from_data", "pcd_iou"]


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("pcd_utils needs the ROCm device; there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def _dbl(values):
    arr = (ctypes.c_double * len(values))(*[float(v) for v in values])
    return arr


def _backproject(depth, mask, cam_params, cam_to_world, bounds, dev):
    """One view -> (points [n,3] float64 tensor, pixel index [n] int32 tensor), ascending pixel order."""
    lib = _lib.load()
    H, W = depth.shape
    d = torch.from_numpy(np.ascontiguousarray(depth, dtype=np.float64)).to(dev)
    m = torch.from_numpy(np.ascontiguousarray(mask).astype(np.uint8)).to(dev) if mask is not None else None
    cap = H * W
    pts = torch.empty((cap, 3), dtype=torch.float64, device=dev)
    pix = torch.empty(cap, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = torch.empty(lib.d3f_backproject_workspace_bytes(H, W), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.d3f_backproject_view(_lib.ptr(d), _lib.ptr(m), H, W, _dbl(cam_params), _dbl(np.asarray(cam_to_world).reshape(-1)),
                                            _dbl(bounds) if bounds is not None else None, cap, _lib.ptr(pts), _lib.ptr(pix),
                                            _lib.ptr(cnt), _lib.ptr(ws), _lib.current_stream_handle(dev)))
    n = int(cnt.item())
    return pts[:n], pix[:n]


def depth2fgpcd(depth, mask, cam_params):
    """(h,w) depth, (h,w) bool mask, [fx,fy,cx,cy] -> (n,3) float64 camera-frame points of mask & depth>0."""
    dev = _device()
    mask = np.logical_and(mask, depth > 0)
    pts, _ = _backproject(depth, mask, cam_params, np.eye(4), None, dev)
    return pts.cpu().numpy()


def aggr_point_cloud_from_data(colors, depths, Ks, poses, downsample=True, masks=None, boundaries=None, out_o3d=True):
    """Reference utils/draw_utils.py:325-413 for downsample=False, out_o3d=False: returns (pcds [n,3], colors [n,3])."""
    if out_o3d or downsample:
        raise NotImplementedError("open3d outputs / voxel down-sampling are upstream (open3d) functionality; "
                                  "call with downsample=False, out_o3d=False")
    dev = _device()
    N = colors.shape[0]
    colors = colors / 255.
    bounds = None
    if boundaries is not None:
        bounds = [boundaries[k] for k in ("x_lower", "x_upper", "y_lower", "y_upper", "z_lower", "z_upper")]
    pcds, pcd_colors = [], []
    for i in range(N):
        K = Ks[i]
        cam_param = [K[0, 0], K[1, 1], K[0, 2], K[1, 2]]
        pose = np.linalg.inv(poses[i])
        pts, pix = _backproject(depths[i], None if masks is None else masks[i], cam_param, pose, bounds, dev)
        pcds.append(pts.cpu().numpy())
        pcd_colors.append(colors[i].reshape(-1, 3)[pix.cpu().numpy()])
    return np.concatenate(pcds, axis=0), np.concatenate(pcd_colors, axis=0)


def _nearest(a, b, dev):
    lib = _lib.load()
    md = torch.empty(a.shape[0], dtype=torch.float64, device=dev)
    am = torch.empty(a.shape[0], dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.d3f_pcd_nearest(_lib.ptr(a), a.shape[0], _lib.ptr(b), b.shape[0], _lib.ptr(md), _lib.ptr(am),
                                       _lib.current_stream_handle(dev)))
    return md, am


def pcd_iou(pcd_1, pcd_2, threshold):
    """Fusion.pcd_iou (fusion.py:724-741) without the [N,M] distance matrix.  Returns the same 7-tuple:
    iou, iou_1, iou_2, overlap_idx_1, overlap_idx_2, min_idx_from_1_to_2, min_idx_from_2_to_1."""
    dev = _device()
    a = torch.from_numpy(np.ascontiguousarray(pcd_1, dtype=np.float64)).to(dev)
    b = torch.from_numpy(np.ascontiguousarray(pcd_2, dtype=np.float64)).to(dev)
    d12, i12 = _nearest(a, b, dev)
    d21, i21 = _nearest(b, a, dev)
    d12, d21 = d12.cpu().numpy(), d21.cpu().numpy()
    n1, n2 = (d12 < threshold).sum(), (d21 < threshold).sum()
    iou = (n1 + n2) / (pcd_1.shape[0] + pcd_2.shape[0])
    return (iou, n1 / pcd_1.shape[0], n2 / pcd_2.shape[0], np.where(d12 < threshold)[0], np.where(d21 < threshold)[0],
            i12.cpu().numpy(), i21.cpu().numpy())
