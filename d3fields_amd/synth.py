"""Deterministic synthetic scenes for tests and bench.py.

No datasets or checkpoints are reachable offline, so every workload is generated:
cameras on a ring looking at the origin (same world->camera ``pose=[R|t]`` (V,3,4) and
``K`` (V,3,3) convention as the reference driver, vis_repr.py:57-76), a ray-cast depth
image of a table plane plus a few spheres ("smooth") or white-noise depth with holes
("stress"), and random channel maps standing in for DINOv2 / SAM / colour producers.
"""
import math

import numpy as np
import torch


def ring_cameras(V, H, W, radius=0.8, height=-0.6, phase=0.3):
    """Returns K [V,3,3], pose [V,3,4] (float32 numpy)."""
    K = np.zeros((V, 3, 3), np.float64)
    Rt = np.zeros((V, 3, 4), np.float64)
    f = 600.0 * (W / 640.0)
    for v in range(V):
        a = 2.0 * math.pi * v / V + phase
        c = np.array([radius * math.cos(a), radius * math.sin(a), height])
        z = -c / np.linalg.norm(c)
        up = np.array([0.0, 0.0, -1.0])
        x = np.cross(z, up)
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z], 0)
        Rt[v, :, :3] = R
        Rt[v, :, 3] = -R @ c
        K[v] = [[f, 0.0, W / 2.0], [0.0, f, H / 2.0], [0.0, 0.0, 1.0]]
    return K.astype(np.float32), Rt.astype(np.float32)


_SPHERES = [(-0.15, -0.10, -0.06, 0.06), (0.12, 0.05, -0.05, 0.05), (0.02, -0.22, -0.08, 0.08),
            (0.25, -0.15, -0.04, 0.04)]


def raycast_depth(K, Rt, H, W):
    """Depth (camera z) of the plane z=0 and a few spheres, [V,H,W] float32."""
    V = K.shape[0]
    out = np.zeros((V, H, W), np.float32)
    uu, vv = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    for v in range(V):
        Kv, R, t = K[v].astype(np.float64), Rt[v, :, :3].astype(np.float64), Rt[v, :, 3].astype(np.float64)
        c = -R.T @ t
        dc = np.stack([(uu - Kv[0, 2]) / Kv[0, 0], (vv - Kv[1, 2]) / Kv[1, 1], np.ones_like(uu)], -1)
        dw = dc @ R                                  # rows: R^T d
        best = np.full((H, W), np.inf)
        with np.errstate(divide="ignore", invalid="ignore"):
            tp = -c[2] / dw[..., 2]
        tp[~(tp > 0)] = np.inf
        best = np.minimum(best, tp)
        for (sx, sy, sz, r) in _SPHERES:
            oc = c - np.array([sx, sy, sz])
            a = (dw * dw).sum(-1)
            b = 2.0 * (dw * oc).sum(-1)
            cc = (oc * oc).sum() - r * r
            disc = b * b - 4 * a * cc
            ts = np.where(disc > 0, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), np.inf)
            ts[~(ts > 0)] = np.inf
            best = np.minimum(best, ts)
        best[~np.isfinite(best)] = 0.0
        out[v] = best.astype(np.float32)             # camera-z == ray parameter (d_z = 1)
    return out


def noise_depth(V, H, W, seed=0, lo=0.6, hi=1.2, hole_frac=0.1):
    g = torch.Generator().manual_seed(seed)
    d = torch.rand(V, H, W, generator=g) * (hi - lo) + lo
    holes = torch.rand(V, H, W, generator=g) < hole_frac
    d[holes] = 0.0
    return d.numpy()


def random_map(V, fh, fw, C, seed=1, device="cpu"):
    """Channels-last [V,fh,fw,C] float32 stand-in for a producer's feature map."""
    g = torch.Generator(device=device).manual_seed(seed)
    return torch.randn(V, fh, fw, C, generator=g, device=device, dtype=torch.float32)


def random_onehot_mask(V, H, W, NI, seed=2, device="cpu"):
    g = torch.Generator(device=device).manual_seed(seed)
    idx = torch.randint(0, NI, (V, H, W), generator=g, device=device)
    return torch.nn.functional.one_hot(idx, NI).to(torch.float32)


WORK_BOX = dict(x_lower=-0.4, x_upper=0.4, y_lower=-0.4, y_upper=0.3, z_lower=-0.2, z_upper=0.02)


def random_cloud(N, seed=3, box=WORK_BOX, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    lo = torch.tensor([box["x_lower"], box["y_lower"], box["z_lower"]])
    hi = torch.tensor([box["x_upper"], box["y_upper"], box["z_upper"]])
    return (torch.rand(N, 3, generator=g) * (hi - lo) + lo).to(device)


def make_scene(V=4, H=480, W=640, depth_kind="smooth", seed=0):
    """Returns dict(K, pose, depth) as float32 torch CPU tensors."""
    K, Rt = ring_cameras(V, H, W)
    if depth_kind == "smooth":
        depth = raycast_depth(K, Rt, H, W)
    elif depth_kind == "stress":
        depth = noise_depth(V, H, W, seed=seed)
    else:
        raise ValueError(depth_kind)
    return {"K": torch.from_numpy(K), "pose": torch.from_numpy(Rt), "depth": torch.from_numpy(depth)}


# ---- synthetic multi-view segmentation (stands in for Grounded-SAM's per-view detections) -------------------------------------
_SPHERE_NAMES = ["mug", "mug", "box", "pen"]


def sphere_label_images(K, Rt, depth):
    """[V,H,W] int64: 0 = table plane / nothing, s + 1 = the pixel's surface point lies on sphere s of the ray-cast scene."""
    V, H, W = depth.shape
    out = np.zeros((V, H, W), np.int64)
    uu, vv = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    for v in range(V):
        Kv, R, t = K[v].astype(np.float64), Rt[v, :, :3].astype(np.float64), Rt[v, :, 3].astype(np.float64)
        d = depth[v].astype(np.float64)
        cam = np.stack([(uu - Kv[0, 2]) / Kv[0, 0] * d, (vv - Kv[1, 2]) / Kv[1, 1] * d, d], -1)
        world = (cam - t) @ R
        for s, (sx, sy, sz, r) in enumerate(_SPHERES):
            hit = (np.linalg.norm(world - np.array([sx, sy, sz]), axis=-1) < r + 2e-3) & (d > 0)
            out[v][hit] = s + 1
    return out


def multiview_segmentation(K, Rt, depth, seed=0):
    """Per-view detections in the format the reference keeps in curr_obs_torch (fusion.py:1141-1143; producer:
    utils/grounded_sam.py:404-442): mask_gs[v] bool [n_v,H,W] with the background (complement of the union) first,
    mask_label[v] list of n_v strings starting with 'background', mask_conf[v] float64 [n_v] starting with 1.0.
    The detections follow the scene's spheres (consistent across views) with seeded imperfections: dropped and split
    detections, a 'table' detection, a spurious one, overlaps, shuffled order."""
    r = np.random.default_rng(seed)
    lab = sphere_label_images(K, Rt, depth)
    V, H, W = lab.shape
    yy, xx = np.mgrid[0:H, 0:W]
    gs, labels, confs = [], [], []
    for v in range(V):
        det, names = [], []
        for s in range(len(_SPHERES)):
            m = lab[v] == s + 1
            if m.sum() < 12 or r.random() < 0.12:
                continue
            if r.random() < 0.25:                                       # one object found as two detections
                cut = xx[m].mean()
                parts = [m & (xx < cut + 1), m & (xx >= cut - 1)]
            else:
                parts = [m]
            for q in parts:
                if r.random() < 0.3:                                    # a mask one pixel too wide: overlaps its neighbours
                    q = q | np.roll(q, 1, axis=1) | np.roll(q, 1, axis=0)
                det.append(q); names.append(_SPHERE_NAMES[s])
        plane = (lab[v] == 0) & (depth[v] > 0)
        if r.random() < 0.6:
            y0, x0 = int(r.integers(0, H // 2)), int(r.integers(0, W // 2))
            det.append(plane & (yy >= y0) & (yy < y0 + H // 3) & (xx >= x0) & (xx < x0 + W // 3)); names.append("table")
        if r.random() < 0.4:
            y0, x0 = int(r.integers(0, H - 12)), int(r.integers(0, W - 12))
            det.append(plane & (yy >= y0) & (yy < y0 + 10) & (xx >= x0) & (xx < x0 + 12)); names.append(str(r.choice(_SPHERE_NAMES)))
        order = r.permutation(len(det))
        det = [det[k] for k in order]; names = [names[k] for k in order]
        union = np.zeros((H, W), bool)
        for q in det:
            union |= q
        gs.append(np.stack([~union] + det, axis=0))
        labels.append(["background"] + names)
        confs.append(np.concatenate([np.array([1.0]), r.uniform(0.3, 0.95, len(det)).astype(np.float32)], axis=0))
    return gs, labels, confs
