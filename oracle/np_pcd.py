"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's point-cloud helpers (fp64).

  backproject_view : depth2fgpcd (utils/my_utils.py:522-537) + inv(pose) transform + boundary crop of
                     aggr_point_cloud_from_data (utils/draw_utils.py:325-413), one view
  nearest          : one direction of Fusion.pcd_iou's search (fusion.py:731-735)
  pcd_to_voxel / pcd_to_index / vox_idx_iou : the voxel-index IoU of instance association (fusion.py:118-180, 794-799)
  erode_cv2        : cv2.erode with an all-ones kernel.  THIRD-PARTY ARITHMETIC ABSENT HERE: opencv-python (env.yaml:17,
                     unpinned) is not installed in the build container, so its published definition is restated
                     (OpenCV docs, cv::erode: dst(x,y) = min over the element of src(x + x' - anchor.x, y + y' - anchor.y),
                     default anchor = element centre (cols/2, rows/2), default border value +max, i.e. out-of-image samples
                     never lower the minimum) and cross-checked against scipy.ndimage.grey_erosion, an independent
                     implementation with the same window convention (tests/test_oracle_golden.py).  The reference's own
                     call sites are fusion.py:1293 (2x2), :1305 (2x2) and :1561 (15x15).
  fps_int          : fps_np (utils/my_utils.py:478-497) restated for the integer pixel arrays of fusion.py:1565-1566
Pinned against tests/golden/pcd_utils.npz / assoc.npz / select_v2.npz (written by running the reference).
Never imported by d3fields_amd.
"""
import numpy as np


def backproject_view(depth, mask, cam_params, cam_to_world, bounds=None):
    depth = np.asarray(depth, np.float64)
    H, W = depth.shape
    fg = (mask & (depth > 0)) if mask is not None else ((depth > 0) & (depth < 1.5))      # draw_utils.py:345-348
    pix = np.flatnonzero(fg.reshape(-1))                                                   # ascending pixel order
    d = depth.reshape(-1)[pix]
    fx, fy, cx, cy = cam_params
    x = ((pix % W) - cx) * d / fx                                                          # my_utils.py:534-536
    y = ((pix // W) - cy) * d / fy
    T = np.asarray(cam_to_world, np.float64).reshape(4, 4)
    w = np.stack([T[r, 0] * x + T[r, 1] * y + T[r, 2] * d + T[r, 3] for r in range(3)], axis=1)
    if bounds is not None:
        keep = np.ones(len(w), bool)
        for k in range(3):
            keep &= (w[:, k] > bounds[2 * k]) & (w[:, k] < bounds[2 * k + 1])             # draw_utils.py:374-379
        w, pix = w[keep], pix[keep]
    return w, pix


def nearest(a, b, chunk=512):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    md = np.empty(len(a))
    am = np.empty(len(a), np.int64)
    for s in range(0, len(a), chunk):
        d = np.linalg.norm(a[s:s + chunk, None] - b[None], axis=-1)
        md[s:s + chunk] = d.min(axis=1)
        am[s:s + chunk] = d.argmin(axis=1)
    return md, am


def pcd_to_voxel(pcds, lower_bound, voxel_size):
    """fusion.py:119-125"""
    return np.floor((np.asarray(pcds) - lower_bound) / voxel_size).astype(np.int32)


def voxel_to_index(voxels, voxel_num):
    """fusion.py:135-145 (int32 arithmetic of the int32 voxel arrays)"""
    voxels = np.asarray(voxels)
    return voxels[..., 0] * voxel_num[1] * voxel_num[2] + voxels[..., 1] * voxel_num[2] + voxels[..., 2]


def pcd_to_index(pcds, lower_bound, voxel_size, voxel_num):
    """fusion.py:159-164"""
    return voxel_to_index(pcd_to_voxel(pcds, lower_bound, voxel_size), voxel_num)


def vox_idx_iou(vox_idx_1, vox_idx_2):
    """fusion.py:794-799: Python sets; the second and third ratio use the RAW lengths."""
    a, b = set(np.asarray(vox_idx_1).tolist()), set(np.asarray(vox_idx_2).tolist())
    union = len(a | b)
    return len(a & b) / union, len(vox_idx_1) / union, len(vox_idx_2) / union


def erode_cv2(src, kernel, iterations=1):
    """cv2.erode(src, kernel, iterations=1) for an all-ones kernel on a 2-D uint8 image (definition in the header)."""
    kernel = np.asarray(kernel)
    assert kernel.ndim == 2 and np.all(kernel != 0) and iterations == 1
    src = np.asarray(src)
    assert src.ndim == 2 and src.dtype == np.uint8
    kh, kw = kernel.shape
    ay, ax = kh // 2, kw // 2
    H, W = src.shape
    pad = np.full((H + kh - 1, W + kw - 1), 255, np.uint8)
    pad[ay:ay + H, ax:ax + W] = src
    out = np.full((H, W), 255, np.uint8)
    for i in range(kh):
        for j in range(kw):
            out = np.minimum(out, pad[i:i + H, j:j + W])
    return out


def fps_int(pix, particle_num, init_idx):
    """fps_np (utils/my_utils.py:478-497) on an (n,2) integer array: float64 norms, first maximum (np.argmax)."""
    pix = np.asarray(pix)
    idx = [int(init_idx)]
    dist = np.linalg.norm(pix - pix[idx[0]], axis=1)
    while len(idx) < particle_num:
        idx.append(int(dist.argmax()))
        dist = np.minimum(dist, np.linalg.norm(pix - pix[idx[-1]], axis=1))
    return pix[idx], idx, float(dist.max())


def voxel_mean(points, voxel_size, colors=None):
    """open3d 0.17 PointCloud::VoxelDownSample restated (geometry/PointCloud.cpp: voxel_min_bound = min_bound - voxel_size/2,
    voxel index = floor((p - voxel_min_bound) / voxel_size), per voxel the mean of the accumulated points / colours in point
    order).  Returned in ASCENDING voxel-index order (open3d iterates its unordered_map); the SET is open3d's."""
    points = np.asarray(points, dtype=np.float64)
    if points.shape[0] == 0:                                     # an empty cloud stays empty (the loop over its points never runs)
        return points.reshape(0, 3) if colors is None else (points.reshape(0, 3), np.zeros((0, 3)))
    anchor = points.min(axis=0) - voxel_size * 0.5
    idx = np.floor((points - anchor) / voxel_size).astype(np.int64)
    acc = {}
    for i in range(points.shape[0]):
        k = tuple(idx[i])
        a = acc.setdefault(k, [np.zeros(3), np.zeros(3), 0])
        a[0] = a[0] + points[i]
        if colors is not None:
            a[1] = a[1] + colors[i]
        a[2] += 1
    keys = sorted(acc)
    pts = np.stack([acc[k][0] / acc[k][2] for k in keys])
    if colors is None:
        return pts
    return pts, np.stack([acc[k][1] / acc[k][2] for k in keys])
