"""Point-cloud helpers on the mask side of the path, with the reference's names and semantics:
``depth2fgpcd`` (utils/my_utils.py:522-537), ``aggr_point_cloud_from_data`` and ``voxel_downsample``
(utils/draw_utils.py:318-413) and ``pcd_iou`` (Fusion.pcd_iou, fusion.py:724-741).  numpy in / numpy out like the
reference; the per-pixel, per-point and per-pair work runs on the ROCm device in fp64.  open3d itself is not needed:
its voxel_down_sample is restated as a device voxel-grid mean (same point set, ascending voxel order instead of open3d's
hash-map order), and where the reference returns an open3d PointCloud (out_o3d=True) this returns one too when open3d
is importable, else a ``PointCloud`` carrying the same ``points`` / ``colors`` arrays.
"""
import ctypes

import numpy as np
import torch

from . import _lib

__all__ = ["depth2fgpcd", "aggr_point_cloud_from_data", "voxel_downsample", "PointCloud", "as_point_cloud", "pcd_iou",
           "init_low_level_memory", "vox_idx_iou", "erode", "fps_pixels", "masked_pixel_fps"]


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("pcd_utils needs the ROCm device; there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def _dbl(values):
    arr = (ctypes.c_double * len(values))(*[float(v) for v in values])
    return arr


def _backproject(depth, mask, cam_params, cam_to_world, bounds, dev):
    """One view -> (points [n,3] float64 tensor, pixel index [n] int32 tensor), ascending pixel order.  depth / mask: numpy
    arrays, or DEVICE tensors (depth any float dtype, mask uint8 / bool) that are used in place."""
    lib = _lib.load()
    H, W = depth.shape
    if isinstance(depth, torch.Tensor):
        d = depth.to(device=dev, dtype=torch.float64).contiguous()
    else:
        d = torch.from_numpy(np.ascontiguousarray(depth, dtype=np.float64)).to(dev)
    if mask is None:
        m = None
    elif isinstance(mask, torch.Tensor):
        m = mask.to(device=dev, dtype=torch.uint8).contiguous()
    else:
        m = torch.from_numpy(np.ascontiguousarray(mask).astype(np.uint8)).to(dev)
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


class PointCloud:
    """What stands in for open3d.geometry.PointCloud when open3d is not installed: `.points` / `.colors` are float64 [n,3]
    arrays (np.asarray(pcd.points) works on both), `+` concatenates like open3d's operator."""

    def __init__(self, points=None, colors=None):
        self.points = np.zeros((0, 3)) if points is None else np.asarray(points, dtype=np.float64)
        self.colors = np.zeros((0, 3)) if colors is None else np.asarray(colors, dtype=np.float64)

    def __add__(self, other):
        return PointCloud(np.concatenate([self.points, np.asarray(other.points)], 0), np.concatenate([self.colors, np.asarray(other.colors)], 0))

    def __len__(self):
        return self.points.shape[0]


def as_point_cloud(points, colors=None):
    """np2o3d (utils/draw_utils.py): an open3d PointCloud when open3d is importable, else a PointCloud (above)."""
    try:
        import open3d as o3d
        pcd = o3d.geometry.PointCloud()
        pcd.points = o3d.utility.Vector3dVector(np.asarray(points, dtype=np.float64))
        if colors is not None:
            pcd.colors = o3d.utility.Vector3dVector(np.asarray(colors, dtype=np.float64))
        return pcd
    except ImportError:
        return PointCloud(points, colors)


def voxel_downsample(pcd, voxel_size, pcd_color=None):
    """Reference utils/draw_utils.py:318-323 (open3d's voxel_down_sample): one point (and colour) per occupied voxel of
    side voxel_size anchored at min_bound - voxel_size/2 = the mean of the voxel's points.  Runs as d3f_voxel_downsample
    on the device: the same SET as open3d's to ~1e-14 m, in ascending voxel order (open3d: the order of its hash map).
    Returns (points, colors) or points, like the reference."""
    pts_np = np.ascontiguousarray(pcd, dtype=np.float64).reshape(-1, 3)
    n = pts_np.shape[0]
    if n == 0:
        return (pts_np, np.zeros((0, 3))) if pcd_color is not None else pts_np
    dev = _device()
    lib = _lib.load()
    pts = torch.from_numpy(pts_np).to(dev)
    col = torch.from_numpy(np.ascontiguousarray(pcd_color, dtype=np.float64).reshape(-1, 3)).to(dev) if pcd_color is not None else None
    out_p = torch.empty((n, 3), dtype=torch.float64, device=dev)
    out_c = torch.empty((n, 3), dtype=torch.float64, device=dev) if col is not None else None
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    ws_bytes = lib.d3f_voxel_downsample_workspace_bytes(n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.d3f_voxel_downsample(_lib.ptr(pts), _lib.ptr(col), n, float(voxel_size), _lib.ptr(out_p), _lib.ptr(out_c),
                                            _lib.ptr(cnt), _lib.ptr(ws), ws_bytes, _lib.current_stream_handle(dev)))
    v = int(cnt.item())
    if pcd_color is not None:
        return out_p[:v].cpu().numpy(), out_c[:v].cpu().numpy()
    return out_p[:v].cpu().numpy()


def aggr_point_cloud_from_data(colors, depths, Ks, poses, downsample=True, masks=None, boundaries=None, out_o3d=True):
    """Reference utils/draw_utils.py:325-413, defaults included: per view depth2fgpcd -> camera-to-world -> boundary crop
    (one device pass with an order-preserving compaction), optional 1-cm voxel-grid mean per view (voxel_downsample above),
    views concatenated.  out_o3d=False: (pcds [n,3], colors [n,3]) numpy arrays; out_o3d=True: a point cloud object
    (as_point_cloud)."""
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
        pts_np, col_np = pts.cpu().numpy(), colors[i].reshape(-1, 3)[pix.cpu().numpy()]
        if downsample:
            pts_np, col_np = voxel_downsample(pts_np, 0.01, col_np)          # draw_utils.py:390-400: radius 0.01, both branches
        pcds.append(pts_np)
        pcd_colors.append(col_np)
    pts_all, col_all = np.concatenate(pcds, axis=0), np.concatenate(pcd_colors, axis=0)
    if out_o3d:
        return as_point_cloud(pts_all, col_all)
    return pts_all, col_all


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


# ---- voxel indices of instance association (reference fusion.py:118-180, 794-799) -----------------------------------
def init_low_level_memory(lower_bound, higher_bound, voxel_size, voxel_num):
    """The six closures of the reference's _init_low_level_memory (fusion.py:118-180), same names and order:
    (pcd_to_voxel, voxel_to_pcd, voxel_to_index, index_to_voxel, pcd_to_index, index_to_pcd).

    pcd_to_voxel / pcd_to_index -- the per-point work merge_instances_from_new_view_vox_ver does on every masked
    cloud (fusion.py:807) -- run on the ROCm device (d3f_pcd_to_index: fp64 floor-divide, numpy's int32 cast and
    int32 wrap-around, bit-exact).  The four pure index conversions are one-line integer formulas on a handful of
    voxels and are evaluated with numpy exactly as written in the reference."""
    lower = np.asarray(lower_bound, dtype=np.float64).reshape(3)
    num = np.asarray(voxel_num).astype(np.int32).reshape(3)
    vs = float(voxel_size)

    def _device_index(pcds, want_voxels):
        pcds = np.asarray(pcds) if not isinstance(pcds, np.ndarray) else pcds
        lead = pcds.shape[:-1]
        assert pcds.shape[-1] == 3
        dev = _device()
        flat = torch.from_numpy(np.ascontiguousarray(pcds, dtype=np.float64).reshape(-1, 3)).to(dev)
        n = flat.shape[0]
        idx = torch.empty(n, dtype=torch.int32, device=dev)
        vox = torch.empty((n, 3), dtype=torch.int32, device=dev) if want_voxels else None
        with torch.cuda.device(dev):
            _lib.check(_lib.load().d3f_pcd_to_index(_lib.ptr(flat), n, _dbl(lower), vs, (ctypes.c_int32 * 3)(*[int(v) for v in num]),
                                                    _lib.ptr(idx), _lib.ptr(vox), _lib.current_stream_handle(dev)))
        if want_voxels:
            return vox.cpu().numpy().reshape(lead + (3,))
        return idx.cpu().numpy().reshape(lead)

    def pcd_to_voxel(pcds):
        return _device_index(pcds, True)

    def voxel_to_pcd(voxels):
        return np.asarray(voxels) * voxel_size + lower_bound

    def voxel_to_index(voxels):
        voxels = np.asarray(voxels)
        return voxels[..., 0] * num[1] * num[2] + voxels[..., 1] * num[2] + voxels[..., 2]

    def index_to_voxel(indexes):
        indexes = np.asarray(indexes)
        voxels = np.zeros(indexes.shape + (3,), dtype=np.int32)
        voxels[..., 2] = indexes % num[2]
        rest = indexes // num[2]
        voxels[..., 1] = rest % num[1]
        voxels[..., 0] = rest // num[1]
        return voxels

    def pcd_to_index(pcds):
        return _device_index(pcds, False)

    def index_to_pcd(indexes):
        return voxel_to_pcd(index_to_voxel(indexes))

    return pcd_to_voxel, voxel_to_pcd, voxel_to_index, index_to_voxel, pcd_to_index, index_to_pcd


def vox_idx_iou(vox_idx_1, vox_idx_2):
    """Fusion.vox_idx_iou (fusion.py:794-799): (|A & B| / |A | B|, len(vox_idx_1) / |A | B|, len(vox_idx_2) / |A | B|)
    with A, B the SETS of the two index arrays (the lengths in the last two ratios are the raw lengths, duplicates
    included, as in the reference).  The set sizes come from one device hash set (d3f_vox_idx_iou); an empty union
    raises ZeroDivisionError like the reference."""
    dev = _device()
    a = torch.from_numpy(np.ascontiguousarray(np.asarray(vox_idx_1).reshape(-1), dtype=np.int32)).to(dev)
    b = torch.from_numpy(np.ascontiguousarray(np.asarray(vox_idx_2).reshape(-1), dtype=np.int32)).to(dev)
    lib = _lib.load()
    ws_bytes = lib.d3f_vox_iou_workspace_bytes(a.numel(), b.numel())
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    counts = torch.empty(2, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.d3f_vox_idx_iou(_lib.ptr(a), a.numel(), _lib.ptr(b), b.numel(), _lib.ptr(counts), _lib.ptr(ws), ws_bytes,
                                       _lib.current_stream_handle(dev)))
    inter, union = (int(v) for v in counts.tolist())
    return inter / union, a.numel() / union, b.numel() / union


def erode(image, kernel, iterations=1):
    """cv2.erode(image, kernel, iterations=1) for an all-ones kernel on an (H,W) uint8 image, as the reference uses it
    on instance masks (fusion.py:1293, 1305, 1561): numpy in, numpy out; runs as d3f_erode on the device."""
    kernel = np.asarray(kernel)
    if kernel.ndim != 2 or not np.all(kernel != 0):
        raise NotImplementedError("only all-ones rectangular structuring elements (np.ones([kh, kw])) are supported")
    if iterations != 1:
        raise NotImplementedError("iterations=1 only (the reference never passes another value)")
    img = np.ascontiguousarray(image)
    if img.dtype != np.uint8 or img.ndim != 2:
        raise TypeError("erode expects an (H,W) uint8 image")
    dev = _device()
    src = torch.from_numpy(img).to(dev)
    dst = torch.empty_like(src)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().d3f_erode(_lib.ptr(src), img.shape[0], img.shape[1], kernel.shape[0], kernel.shape[1],
                                         _lib.ptr(dst), _lib.current_stream_handle(dev)))
    return dst.cpu().numpy()


def fps_pixels(pixel_idx, particle_num, init_idx=-1):
    """fps_np (utils/my_utils.py:478-497) on an (n,2) INTEGER array -- select_features_rand_v2 samples the (row, col)
    indices of an eroded mask with it (fusion.py:1565-1566).  Returns (selected [k,2] int array, index list,
    max remaining distance) like fps_np; init_idx == -1 draws the start with np.random.randint, as the reference."""
    pix = np.ascontiguousarray(pixel_idx)
    assert pix.ndim == 2 and pix.shape[1] == 2 and pix.shape[0] > 0
    if not np.issubdtype(pix.dtype, np.integer):
        raise TypeError("fps_pixels expects integer pixel coordinates")
    n = pix.shape[0]
    start = int(np.random.randint(n)) if init_idx == -1 else int(init_idx)
    k = int(particle_num)       # NOT clamped to n: fps_np keeps appending (index 0 once every distance is 0) and always returns
    dev = _device()             # particle_num points, which select_features_rand_v2 relies on for small eroded masks
    pts = torch.from_numpy(pix.astype(np.int32)).to(dev)
    idx = torch.empty(k, dtype=torch.int64, device=dev)
    maxd = torch.empty(1, dtype=torch.float64, device=dev)
    ws = torch.empty(_lib.load().d3f_fps_pixels_workspace_bytes(n), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().d3f_fps_pixels(_lib.ptr(pts), n, k, start, _lib.ptr(idx), _lib.ptr(maxd), _lib.ptr(ws),
                                              _lib.current_stream_handle(dev)))
    sel = idx.cpu().numpy()
    return pix[sel], sel.tolist(), float(maxd.item())


def masked_pixel_fps(mask_channel, depth, particle_num, depth_lo=0.0, depth_hi=1.5, kernel=(15, 15), init_idx=-1):
    """The pixel side of select_features_rand_v2 for one (instance, camera) (fusion.py:1554-1568) as ONE device pipeline:
    gate (mask != 0 & depth_lo < depth < depth_hi) -> cv2.erode with an all-ones kernel -> row-major nonzero -> fps_np on
    the pixel indices -> the selected pixels and their depths.  mask_channel: (H,W) float32 DEVICE view (any strides, e.g.
    curr_obs_torch['mask'][cam, :, :, i]); depth: (H,W) float32 device tensor.  Host traffic: the nonzero COUNT (fps_np
    draws its start with np.random.randint(count), so the count has to reach the host first) and the result --
    (sel_idx [k,2] int64 rows/cols, sel_depth [k] float32) as numpy arrays.  Raises like fps_np's assert when the eroded
    mask is empty."""
    lib = _lib.load()
    dev = mask_channel.device
    H, W = int(depth.shape[0]), int(depth.shape[1])
    assert mask_channel.shape == (H, W)
    # the mask is stored in Fusion.dtype (float16 under the fp16 storage mode) or installed by the caller as bool / uint8
    # one-hot: the gate reads float32 (the reference does mask.astype(bool): any non-zero counts)
    if mask_channel.dtype != torch.float32:
        mask_channel = (mask_channel != 0).to(torch.float32)
    depth = depth.to(torch.float32).contiguous()
    gated = torch.empty((H, W), dtype=torch.uint8, device=dev)
    eroded = torch.empty_like(gated)
    rc = torch.empty((H * W, 2), dtype=torch.int32, device=dev)
    count = torch.empty(1, dtype=torch.int64, device=dev)
    ws = torch.empty(lib.d3f_backproject_workspace_bytes(H, W), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        st = _lib.current_stream_handle(dev)
        _lib.check(lib.d3f_mask_gate(_lib.ptr(mask_channel), mask_channel.stride(0), mask_channel.stride(1), _lib.ptr(depth), H, W,
                                     float(depth_lo), float(depth_hi), _lib.ptr(gated), st))
        _lib.check(lib.d3f_erode(_lib.ptr(gated), H, W, int(kernel[0]), int(kernel[1]), _lib.ptr(eroded), st))
        _lib.check(lib.d3f_nonzero_pixels(_lib.ptr(eroded), H, W, H * W, _lib.ptr(rc), _lib.ptr(count), _lib.ptr(ws), st))
        n = int(count.item())                                   # the one sync: np.random.randint(n) needs it
        assert n > 0, "fps_np asserts a non-empty point set (the eroded instance mask is empty)"
        start = int(np.random.randint(n)) if init_idx == -1 else int(init_idx)
        k = int(particle_num)
        idx = torch.empty(k, dtype=torch.int64, device=dev)
        dist_ws = torch.empty(lib.d3f_fps_pixels_workspace_bytes(n), dtype=torch.uint8, device=dev)
        _lib.check(lib.d3f_fps_pixels(_lib.ptr(rc), n, k, start, _lib.ptr(idx), None, _lib.ptr(dist_ws), st))
        sel = rc[idx].to(torch.int64)                            # [k,2] rows, cols
        sel_depth = depth[sel[:, 0], sel[:, 1]]
        return sel.cpu().numpy(), sel_depth.cpu().numpy()
