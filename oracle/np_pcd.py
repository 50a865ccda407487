"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's point-cloud helpers (fp64).

  backproject_view : depth2fgpcd (utils/my_utils.py:522-537) + inv(pose) transform + boundary crop of
                     aggr_point_cloud_from_data (utils/draw_utils.py:325-413), one view
  nearest          : one direction of Fusion.pcd_iou's search (fusion.py:731-735)
Pinned against tests/golden/pcd_utils.npz (written by running the reference).  Never imported by d3fields_amd.
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
