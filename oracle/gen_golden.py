"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz by RUNNING THE REFERENCE.

Run in the build container (where /root/reference is mounted):

    python oracle/gen_golden.py

The reference publishes no tests or golden vectors of its own, so parity is
pinned by importing its Python here (recipe: oracle/_ref_import.py), feeding
it deterministic synthetic inputs and storing inputs + the reference's outputs.  Only data
is stored -- never reference source.  The fixtures travel to the GPU box; this script and
the reference do not need to.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import _ref_import as R          # noqa: E402
from d3fields_amd import synth               # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def pathological_points(pose, n_each=8, seed=11):
    """Points that exercise |z|<1e-4, behind-camera, far-outside-image and mirror cases."""
    g = np.random.default_rng(seed)
    Rt = pose.numpy().astype(np.float64)
    pts = []
    for v in range(Rt.shape[0]):
        R_, t = Rt[v, :, :3], Rt[v, :, 3]
        c = -R_.T @ t
        zdir = R_[2]
        for _ in range(n_each):
            lateral = R_[0] * g.uniform(-0.2, 0.2) + R_[1] * g.uniform(-0.2, 0.2)
            pts.append(c + lateral + zdir * g.uniform(-9e-5, 9e-5))       # |z| < 1e-4
            pts.append(c + lateral - zdir * g.uniform(0.05, 0.5))         # behind the camera
            pts.append(c + lateral * 20 + zdir * g.uniform(0.2, 1.0))      # far outside image
    pts.append([5.0, 5.0, 5.0])
    pts.append([0.0, 0.0, 0.5])                                           # under the table
    pts.append([0.0, 0.0, 0.0])
    return torch.tensor(np.array(pts), dtype=torch.float32)


def to_np(d):
    return {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in d.items()}


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024))


def scene_case(fusion, name, V, H, W, kind, fmap_hw, C, NI, N, dense_K=False, seed=0):
    sc = synth.make_scene(V, H, W, kind, seed=seed)
    if dense_K:
        g = torch.Generator().manual_seed(77)
        sc["K"] = sc["K"] + (torch.rand(V, 3, 3, generator=g) - 0.5) * 0.02 * sc["K"].abs().clamp(min=1.0)
    feats = synth.random_map(V, fmap_hw[0], fmap_hw[1], C, seed=seed + 1)
    mask = synth.random_onehot_mask(V, H, W, NI, seed=seed + 2)
    gcol = torch.Generator().manual_seed(seed + 5)
    color = torch.randint(0, 256, (V, H, W, 3), generator=gcol).to(torch.float32) / 255.0
    obs = dict(sc)
    obs.update(dino_feats=feats, mask=mask, color_tensor=color)
    f = R.make_reference_fusion(fusion, obs, H, W)
    pts = torch.cat([synth.random_cloud(N, seed=seed + 3), pathological_points(sc["pose"])])
    with torch.no_grad():
        full = f.eval(pts, return_names=["dino_feats", "mask", "color_tensor"], return_inter=True)
        default = f.eval(pts)                                   # default return_names
        empty = f.eval(pts, return_names=[])
        dist_only = f.eval_dist(pts)
        inst = fusion.onehot2instance(full["mask"])
    assert sorted(default.keys()) == ["dino_feats", "dist", "mask", "valid_mask"]
    assert sorted(empty.keys()) == ["dist", "valid_mask"]
    assert torch.equal(default["dino_feats"], full["dino_feats"])
    assert torch.equal(empty["dist"], full["dist"])
    arrays = dict(H=H, W=W, mu=f.mu, pts=pts, K=sc["K"], pose=sc["pose"], depth=sc["depth"],
                  in_dino_feats=feats, in_mask=mask, in_color_tensor=color,
                  dist=full["dist"], valid_mask=full["valid_mask"],
                  dino_feats=full["dino_feats"], mask=full["mask"], color_tensor=full["color_tensor"],
                  dino_feats_inter=full["dino_feats_inter"], mask_inter=full["mask_inter"],
                  color_tensor_inter=full["color_tensor_inter"],
                  evaldist_dist=dist_only["dist"], evaldist_valid_mask=dist_only["valid_mask"],
                  mask_instance=inst)
    save(name, **to_np(arrays))


def batch_case(fusion):
    """batch_eval across 3 chunks with a ragged tail (fusion.py:526-545), N = 130 001."""
    V, H, W = 4, 60, 80
    sc = synth.make_scene(V, H, W, "smooth")
    feats = synth.random_map(V, 6, 8, 2, seed=21)
    mask = synth.random_onehot_mask(V, H, W, 3, seed=22)
    obs = dict(sc)
    obs.update(dino_feats=feats, mask=mask)
    f = R.make_reference_fusion(fusion, obs, H, W)
    N = 130001
    pts = synth.random_cloud(N, seed=23)
    with torch.no_grad():
        out = f.batch_eval(pts, return_names=["dino_feats", "mask"])
        out_default = f.batch_eval(pts)
        out_empty = f.batch_eval(pts, return_names=[])
    assert torch.equal(out_default["mask"], out["mask"])
    assert sorted(out_empty.keys()) == ["dist", "valid_mask"]
    sub = slice(0, N, 13)                      # stored subset; full tensors pinned by sums below
    arrays = dict(H=H, W=W, mu=f.mu, N=N, cloud_seed=23, K=sc["K"], pose=sc["pose"], depth=sc["depth"],
                  in_dino_feats=feats, in_mask=mask, stride=13,
                  pts_sub=pts[sub], dist_sub=out["dist"][sub], dino_feats_sub=out["dino_feats"][sub],
                  mask_sub=out["mask"][sub],
                  valid_bits=np.packbits(out["valid_mask"].numpy()),
                  pts_sum=pts.double().sum(0), dist_sum=out["dist"].double().sum(),
                  dino_feats_sum=out["dino_feats"].double().sum(0), mask_sum=out["mask"].double().sum(0))
    save("batch_eval_130001", **to_np(arrays))


def grid_case(fusion):
    """create_init_grid on the vis_repr.py boundaries/step (vis_repr.py:36-52,88)."""
    b = dict(synth.WORK_BOX)
    for step, tag in [(0.004, "004"), (0.02, "020")]:
        coords, shape = fusion.create_init_grid(b, step)
        n = coords.shape[0]
        sub = slice(0, n, 1009)
        save("init_grid_" + tag, step=step, shape=np.array(shape), n=n,
             bounds=np.array([b["x_lower"], b["x_upper"], b["y_lower"], b["y_upper"], b["z_lower"], b["z_upper"]]),
             head=coords[:128].numpy(), tail=coords[-128:].numpy(), sub=coords[sub].numpy(),
             colsum=coords.double().sum(0).numpy())


def onehot_case(fusion):
    g = torch.Generator().manual_seed(31)
    inst = torch.randint(0, 6, (5, 7, 9), generator=g).to(torch.uint8)
    oh_t = fusion.instance2onehot(inst, 6)
    oh_n = fusion.instance2onehot(inst.numpy(), 6)
    assert np.array_equal(oh_t.numpy(), oh_n)
    soft = torch.rand(300, 6, generator=g)
    soft[5] = 0.0                                      # all-equal row -> index 0
    soft[6, 2] = soft[6, 4] = 2.0                      # tie -> first index
    back_t = fusion.onehot2instance(soft)
    back_n = fusion.onehot2instance(soft.numpy())
    assert np.array_equal(back_t.numpy(), back_n)
    save("onehot", inst=inst.numpy(), NI=6, onehot=oh_t.numpy(), soft=soft.numpy(), soft_inst=back_t.numpy())


def corr_case(corr):
    g = torch.Generator().manual_seed(41)
    B, Hh, Ww, C = 3, 10, 12, 24
    fmap = torch.randn(B, Hh, Ww, C, generator=g)
    tgt = torch.randn(C, generator=g)
    arrays = dict(fmap_bhwc=fmap.numpy(), tgt=tgt.numpy(), scale=0.7)
    fmap_bchw = fmap.permute(0, 3, 1, 2).contiguous()
    for dt in ("l2", "square"):
        arrays["similarity_" + dt] = corr.compute_similarity(fmap.numpy(), tgt.numpy(), 0.7, dist_type=dt)
        arrays["similarity_tensor_" + dt] = corr.compute_similarity_tensor(fmap_bchw, tgt, 0.7, dist_type=dt).numpy()
        arrays["dist_tensor_" + dt] = corr.compute_dist_tensor(fmap_bchw, tgt, dist_type=dt).numpy()
    # 2-D variant of the tensor helpers ([B, C] with no trailing dims)
    flat = torch.randn(50, C, generator=g)
    arrays["flat"] = flat.numpy()
    arrays["flat_similarity_tensor_l2"] = corr.compute_similarity_tensor(flat, tgt, 0.7).numpy()
    arrays["flat_dist_tensor_l2"] = corr.compute_dist_tensor(flat, tgt).numpy()
    B1, B2 = 700, 37
    src = torch.randn(B1, C, generator=g)
    tg = torch.randn(B2, C, generator=g)
    tg[3] = src[100]                                   # an exact match (distance 0)
    tg[4] = src[101] + 1e-4                            # a near match (cancellation stress)
    arrays.update(multi_src=src.numpy(), multi_tgt=tg.numpy(), multi_scale=1.3)
    for dt in ("l2", "square"):
        o = corr.compute_similarity_tensor_multi(src, tg, None, None, 1.3, dist_type=dt)
        arrays["multi_" + dt] = o.numpy()
        arrays["multi_argmax_" + dt] = o.argmax(0).numpy()
    save("corr_utils", **arrays)


def grad_case(fusion):
    """d(sum(feat) + sum(dist))/d pts through the reference's autograd (for the backward row)."""
    V, H, W = 3, 48, 64
    sc = synth.make_scene(V, H, W, "smooth")
    feats = synth.random_map(V, 12, 16, 4, seed=51)
    obs = dict(sc)
    obs.update(dino_feats=feats)
    f = R.make_reference_fusion(fusion, obs, H, W)
    pts = synth.random_cloud(500, seed=53).requires_grad_(True)
    out = f.eval(pts, return_names=["dino_feats"])
    (out["dino_feats"].sum() + out["dist"].sum()).backward()
    save("grad_500", H=H, W=W, mu=f.mu, K=sc["K"].numpy(), pose=sc["pose"].numpy(), depth=sc["depth"].numpy(),
         in_dino_feats=feats.numpy(), pts=pts.detach().numpy(), grad_pts=pts.grad.numpy(),
         dist=out["dist"].detach().numpy(), dino_feats=out["dino_feats"].detach().numpy())


def select_case(fusion):
    """select_features_rand (fusion.py:1418-1475) + fps_np (utils/my_utils.py:478-497) on a 1-cm grid."""
    V, H, W = 4, 96, 128
    sc = synth.make_scene(V, H, W, "smooth")
    feats = synth.random_map(V, 12, 16, 16, seed=61)
    # instance masks with large coherent regions: label = f(pixel column band), 4 instances incl. background 0
    lab = (torch.arange(W)[None, None, :] // 32 + torch.arange(H)[None, :, None] // 48).expand(V, H, W) % 4
    mask = torch.nn.functional.one_hot(lab, 4).to(torch.float32)
    obs = dict(sc)
    obs.update(dino_feats=feats, mask=mask, consensus_mask_label=["background", "a", "b", "c"],
               color=np.zeros((V, H, W, 3), np.uint8))
    f = R.make_reference_fusion(fusion, obs, H, W)
    box = dict(synth.WORK_BOX)
    feats_l, pts_l, _ = f.select_features_rand(box, 24, per_instance=True, res=0.01, init_idx=0)
    grid, shape = fusion.create_init_grid(box, 0.01)
    with torch.no_grad():
        out = f.batch_eval(grid, return_names=["mask"])
    shell = (out["dist"].abs() < 0.005) & out["valid_mask"]
    g = np.random.default_rng(3).normal(size=(5000, 3)).astype(np.float32)
    fp, fi, fd = fusion.fps_np(g, 64, init_idx=17)
    arrays = dict(H=H, W=W, mu=f.mu, K=sc["K"].numpy(), pose=sc["pose"].numpy(), depth=sc["depth"].numpy(),
                  in_dino_feats=feats.numpy(), in_mask=mask.numpy(), res=0.01, N=24,
                  bounds=np.array([box["x_lower"], box["x_upper"], box["y_lower"], box["y_upper"], box["z_lower"], box["z_upper"]]),
                  shell_index=torch.nonzero(shell)[:, 0].numpy(), grid_shape=np.array(shape),
                  grid_dist=out["dist"].numpy(), grid_valid=np.packbits(out["valid_mask"].numpy()),
                  n_inst=len(pts_l), fps_cloud=g, fps_idx=np.array(fi), fps_pts=fp, fps_maxdist=np.float32(fd))
    for i, (a, b) in enumerate(zip(feats_l, pts_l)):
        arrays["sel_feats_%d" % i] = a.numpy()
        arrays["sel_pts_%d" % i] = b
    # select_features_from_pcd (fusion.py:1477-1537) on a jittered copy of the shell points
    cloud = (grid[shell].numpy() + np.random.default_rng(5).normal(0, 0.002, (int(shell.sum()), 3))).astype(np.float32)
    pf, pp, _ = f.select_features_from_pcd(cloud, 16, per_instance=True, init_idx=0)
    arrays.update(pcd_cloud=cloud, pcd_n_inst=len(pp))
    for i, (a, b) in enumerate(zip(pf, pp)):
        arrays["pcd_feats_%d" % i] = a.numpy()
        arrays["pcd_pts_%d" % i] = b
    save("select_features", **arrays)


def pcd_case(fusion):
    """depth2fgpcd / aggr_point_cloud_from_data (numpy outputs) / Fusion.pcd_iou of the reference."""
    V, H, W = 3, 60, 80
    sc = synth.make_scene(V, H, W, "smooth")
    depths = sc["depth"].numpy().astype(np.float64)
    depths[0, 10:20, 30:50] = 0.0                                     # holes
    K = sc["K"].numpy().astype(np.float64)
    pose44 = np.tile(np.eye(4), (V, 1, 1))
    pose44[:, :3] = sc["pose"].numpy().astype(np.float64)
    rng = np.random.default_rng(71)
    colors = rng.integers(0, 256, size=(V, H, W, 3), dtype=np.uint8)
    masks = rng.random((V, H, W)) < 0.6
    box = dict(synth.WORK_BOX)
    a_pts, a_col = fusion.aggr_point_cloud_from_data(colors, depths, K, pose44, downsample=False, masks=masks, boundaries=box, out_o3d=False)
    b_pts, b_col = fusion.aggr_point_cloud_from_data(colors, depths, K, pose44, downsample=False, masks=None, boundaries=None, out_o3d=False)
    fg = fusion.depth2fgpcd(depths[1], masks[1], [K[1, 0, 0], K[1, 1, 1], K[1, 0, 2], K[1, 1, 2]])
    f = fusion.Fusion.__new__(fusion.Fusion)
    p1, p2 = a_pts[::7][:1500], a_pts[3::5][:1200] + 0.002
    iou = f.pcd_iou(p1, p2, 0.005)
    save("pcd_utils", depths=depths, K=K, pose44=pose44, colors=colors, masks=masks,
         bounds=np.array([box[k] for k in ("x_lower", "x_upper", "y_lower", "y_upper", "z_lower", "z_upper")]),
         crop_pts=a_pts, crop_col=a_col, all_pts=b_pts, all_col=b_col, fg_view1=fg, p1=p1, p2=p2,
         iou=np.array(iou[:3], dtype=np.float64), overlap_1=iou[3], overlap_2=iou[4], idx_12=iou[5], idx_21=iou[6])


def smooth_feature_map(V, fh, fw, C, seed=71):
    """Low-frequency sinusoid features (a tracking loss needs a landscape, not white noise)."""
    g = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(fh) / fh, np.arange(fw) / fw, indexing="ij")
    out = np.zeros((V, fh, fw, C), np.float32)
    for v in range(V):
        for c in range(C):
            a, b = g.uniform(0.5, 2.5, 2) * g.choice([-1, 1], 2)
            out[v, :, :, c] = np.sin(2 * np.pi * (a * xx + b * yy) + g.uniform(0, 2 * np.pi))
    return torch.from_numpy(out)


def tracking_case(fusion):
    """Fusion.rigid_tracking (fusion.py:1608-1685) run by the reference on CPU, with pytorch3d's so3_exp_map /
    Transform3d restated (oracle/pytorch3d_restated.py): two instances x 40 keypoints just above the ground
    plane, displaced by a small rigid motion that the 100 Adam steps have to undo."""
    V, H, W = 4, 96, 128
    sc = synth.make_scene(V, H, W, "smooth")
    feats = smooth_feature_map(V, 24, 32, 8)
    obs = dict(sc)
    obs.update(dino_feats=feats)
    f = R.make_reference_fusion(fusion, obs, H, W)
    g = np.random.default_rng(73)
    n = 40
    centres = np.array([[-0.22, 0.15, -0.003], [0.15, 0.20, -0.003]], np.float32)   # clear of the spheres
    true_pts = np.stack([c + np.concatenate([g.uniform(-0.05, 0.05, (n, 2)), np.zeros((n, 1))], 1) for c in centres]).astype(np.float32)
    with torch.no_grad():
        src = f.eval(torch.from_numpy(true_pts.reshape(-1, 3)), return_names=["dino_feats"])
    assert bool(src["valid_mask"].all())
    # displaced start: rotate each instance about its centre by a few degrees around z and shift it
    last = []
    for i, (ang, sh) in enumerate([(4.0, (0.012, -0.009, 0.0)), (-3.0, (-0.008, 0.011, 0.002))]):
        a = np.deg2rad(ang)
        Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32)
        last.append(((true_pts[i] - centres[i]) @ Rz.T + centres[i] + np.array(sh, np.float32)).astype(np.float32))
    info = {"a": {"src_feats": src["dino_feats"][:n]}, "b": {"src_feats": src["dino_feats"][n:]}}
    res = f.rigid_tracking(info, last, dict(synth.WORK_BOX), n)
    match = np.stack(res["match_pts_list"])
    err0 = float(np.abs(np.stack(last) - true_pts).max())
    err1 = float(np.abs(match - true_pts).max())
    print("rigid_tracking: max |start - true| = %.4f m, max |tracked - true| = %.4f m" % (err0, err1))
    save("rigid_tracking", H=H, W=W, mu=f.mu, K=sc["K"].numpy(), pose=sc["pose"].numpy(), depth=sc["depth"].numpy(),
         in_dino_feats=feats.numpy(), src_feats=src["dino_feats"].numpy(), true_pts=true_pts, last_pts=np.stack(last),
         match_pts=match, n=n)


def assoc_case(fusion):
    """The voxel-index IoU of instance association: _init_low_level_memory closures (fusion.py:118-180) and
    Fusion.vox_idx_iou (fusion.py:794-799), run by the reference on clouds inside AND partly outside the box."""
    box = dict(synth.WORK_BOX)
    lower = np.array([box["x_lower"], box["y_lower"], box["z_lower"]])
    higher = np.array([box["x_upper"], box["y_upper"], box["z_upper"]])
    voxel_size = 0.03                                                     # fusion.py:1078
    voxel_num = ((higher - lower) / voxel_size).astype(np.int32)          # fusion.py:1079
    fns = fusion._init_low_level_memory(lower, higher, voxel_size, voxel_num=voxel_num)
    pcd_to_voxel, voxel_to_pcd, voxel_to_index, index_to_voxel, pcd_to_index, index_to_pcd = fns
    g = np.random.default_rng(81)
    pcd = g.uniform(lower - 0.05, higher + 0.05, size=(6000, 3))          # ~25 % of the points lie outside the box
    pcd[:50] = lower + voxel_size * g.integers(0, 5, size=(50, 3))         # exactly on voxel faces
    pcd32 = g.uniform(lower, higher, size=(500, 3)).astype(np.float32)     # float32 input (promoted like numpy does)
    f = fusion.Fusion.__new__(fusion.Fusion)
    a = pcd_to_index(pcd[:2500])
    b = pcd_to_index(pcd[1500:6000] + 0.004)
    pairs = {"ab": (a, b), "aa": (a, a), "disjoint": (a[a % 2 == 0], a[a % 2 == 1]), "one_empty": (a, a[:0]),
             "small": (np.array([5, 5, 5, 7], np.int32), np.array([7, 9], np.int32))}
    arrays = dict(lower=lower, higher=higher, voxel_size=voxel_size, voxel_num=voxel_num, pcd=pcd, pcd32=pcd32,
                  voxels=pcd_to_voxel(pcd), index=pcd_to_index(pcd), index32=pcd_to_index(pcd32),
                  index_of_voxels=voxel_to_index(pcd_to_voxel(pcd)), voxel_of_index=index_to_voxel(pcd_to_index(pcd[100:200])),
                  pcd_of_index=index_to_pcd(pcd_to_index(pcd[100:200])))
    for k, (x, y) in pairs.items():
        arrays["iou_%s_a" % k], arrays["iou_%s_b" % k] = x, y
        arrays["iou_%s" % k] = np.array(f.vox_idx_iou(x, y), dtype=np.float64)
    save("assoc", **arrays)


def select_v2_case(fusion):
    """select_features_rand_v2 (fusion.py:1539-1606) run by the reference (cv2.erode restated, oracle/np_pcd.py):
    15x15 erosion of each instance mask, fps_np on the pixel indices (np.random seeded), back-projection, eval."""
    V, H, W = 4, 96, 128
    sc = synth.make_scene(V, H, W, "smooth")
    feats = synth.random_map(V, 12, 16, 16, seed=91)
    yy, xx = np.mgrid[0:H, 0:W]
    lab = np.zeros((V, H, W), np.int64)
    for v in range(V):                                   # two blobs per view + background, a hole in one of them
        lab[v][(yy - 30 - 2 * v) ** 2 + (xx - 40) ** 2 < 22 ** 2] = 1
        lab[v][(np.abs(yy - 60) < 18) & (np.abs(xx - 92 + 3 * v) < 20)] = 2
        lab[v][(np.abs(yy - 60) < 2) & (np.abs(xx - 92) < 3)] = 0
    mask = torch.nn.functional.one_hot(torch.from_numpy(lab), 3).to(torch.float32)
    labels = ["background", "mug", "box"]
    obs = dict(sc)
    obs.update(dino_feats=feats, mask=mask, mask_label=[labels] * V, consensus_mask_label=labels,
               color=np.zeros((V, H, W, 3), np.uint8))
    f = R.make_reference_fusion(fusion, obs, H, W)
    np.random.seed(1234)
    feats_l, pts_l, _ = f.select_features_rand_v2(dict(synth.WORK_BOX), 32, per_instance=True)
    er = fusion.cv2.erode(((lab[0] == 1) * 255).astype(np.uint8), np.ones([15, 15], np.uint8), iterations=1)
    arrays = dict(H=H, W=W, mu=f.mu, K=sc["K"].numpy(), pose=sc["pose"].numpy(), depth=sc["depth"].numpy(),
                  in_dino_feats=feats.numpy(), in_mask=mask.numpy(), N=32, seed=1234, n_inst=len(pts_l), eroded_v0_i1=er)
    for i, (a, b) in enumerate(zip(feats_l, pts_l)):
        arrays["feats_%d" % i] = a.numpy()
        arrays["pts_%d" % i] = b
    save("select_v2", **arrays)


def masked_pcd_case(fusion):
    """Fusion.extract_masked_pcd / extract_masked_pcd_in_views (fusion.py:1262-1299; downsample=False, numpy outputs) run by
    the reference (cv2.erode restated, oracle/np_pcd.py): OR of instance masks -> 2x2 erosion -> masked back-projection of
    every view -> world frame -> boundary crop."""
    V, H, W = 4, 96, 128
    sc = synth.make_scene(V, H, W, "smooth")
    depth = sc["depth"].clone()
    depth[1, 20:30, 40:70] = 0.0                                     # a hole under an instance
    yy, xx = np.mgrid[0:H, 0:W]
    lab = np.zeros((V, H, W), np.int64)
    for v in range(V):
        lab[v][(yy - 30 - 2 * v) ** 2 + (xx - 40) ** 2 < 22 ** 2] = 1
        lab[v][(np.abs(yy - 60) < 18) & (np.abs(xx - 92 + 3 * v) < 20)] = 2
        lab[v][(np.abs(yy - 80) < 6) & (np.abs(xx - 20) < 9)] = 3
    mask = torch.nn.functional.one_hot(torch.from_numpy(lab), 4).to(torch.float32)
    rng = np.random.default_rng(77)
    color = rng.integers(0, 256, size=(V, H, W, 3), dtype=np.uint8)
    labels = ["background", "mug", "box", "pen"]
    mask_gs = [np.stack([lab[v] == i for i in range(4)], axis=0) for v in range(V)]      # per view [NI,H,W] bool (fusion.py:1141)
    obs = dict(depth=depth, K=sc["K"], pose=sc["pose"])
    obs.update(mask=mask, mask_gs=mask_gs, mask_label=[labels] * V, consensus_mask_label=labels, color=color)
    f = R.make_reference_fusion(fusion, obs, H, W)
    box = dict(synth.WORK_BOX)
    tight = dict(box, x_lower=-0.1, y_upper=0.12)
    arrays = dict(H=H, W=W, K=sc["K"].numpy(), pose=sc["pose"].numpy(), depth=depth.numpy(), in_mask=mask.numpy(), color=color,
                  tight=np.array([tight[k] for k in ("x_lower", "x_upper", "y_lower", "y_upper", "z_lower", "z_upper")]))
    arrays["pcd_12_box"] = f.extract_masked_pcd([1, 2], boundaries=box)
    arrays["pcd_3_none"] = f.extract_masked_pcd([3])
    arrays["pcd_123_tight"] = f.extract_masked_pcd([1, 2, 3], boundaries=tight)
    arrays["pcd_123_none"] = f.extract_masked_pcd([1, 2, 3])              # == the points of get_query_obj_pcd (fusion.py:1301-1311)
    arrays["view2_2_box"] = f.extract_masked_pcd_in_views([2], [2], box, downsample=False)
    arrays["view0_13_tight"] = f.extract_masked_pcd_in_views([1, 3], [0], tight, downsample=False)
    save("masked_pcd", **arrays)


def _instances_summary(prefix, instances, V, arrays):
    """instances_info of the reference (list of dicts, fusion.py:1069-1073) as arrays: per instance its label, the sorted set
    and the raw length of 'vox_idx', 'idx' as a [V] row (-1 = not seen in that view), and 'conf_per_pt' as sorted keys + the
    confidences in append order (ragged, flattened)."""
    arrays[prefix + "_labels"] = np.array([inst["label"] for inst in instances])
    arrays[prefix + "_idx"] = np.array([[inst["idx"].get(v, -1) for v in range(V)] for inst in instances], np.int64).reshape(len(instances), V)
    for k, inst in enumerate(instances):
        arrays["%s_%d_voxset" % (prefix, k)] = np.array(sorted(set(int(x) for x in inst["vox_idx"])), np.int64)
        arrays["%s_%d_voxlen" % (prefix, k)] = np.int64(len(inst["vox_idx"]))
        keys = sorted(int(x) for x in inst["conf_per_pt"])
        arrays["%s_%d_confkeys" % (prefix, k)] = np.array(keys, np.int64)
        arrays["%s_%d_confcount" % (prefix, k)] = np.array([len(inst["conf_per_pt"][x]) for x in keys], np.int64)
        flat = [float(c) for x in keys for c in inst["conf_per_pt"][x]]
        arrays["%s_%d_confvals" % (prefix, k)] = np.array(flat, np.float64)


def align_case(fusion, name, seed, V=4, H=120, W=160, queries=("mug", "box", "pen")):
    """Fusion.align_instance_mask_v3 (fusion.py:1065-1098) run by the reference on synthetic per-view detections
    (d3fields_amd/synth.py:multiview_segmentation): merge_instances_from_new_view_vox_ver per view (fusion.py:801-849),
    filter_instances_vox_ver (:978-1040), reorder_instances (:1042-1050), swap_instance_mask (:1052-1063).  Third-party code the
    image lacks is restated, as for the other caller goldens: cv2.erode (oracle/np_pcd.py:erode_cv2) and open3d 0.17's
    voxel_down_sample behind utils/draw_utils.py:314-323 (oracle/np_pcd.py:voxel_mean; only the SET of 1-cm voxel means matters
    downstream, their order does not).  Stored: the detections, the reference's instances after the merges and after the filter,
    the consensus labels and the (V,H,W) uint8 label image."""
    import copy
    import importlib
    from oracle import np_pcd
    du = importlib.import_module("utils.draw_utils")
    du.voxel_downsample = lambda pcd, voxel_size, pcd_color=None: np_pcd.voxel_mean(pcd, voxel_size, pcd_color)
    sc = synth.make_scene(V, H, W, "smooth")
    gs, labels, confs = synth.multiview_segmentation(sc["K"].numpy(), sc["pose"].numpy(), sc["depth"].numpy(), seed=seed)
    color = np.zeros((V, H, W, 3), np.uint8)              # the association never looks at colours (stored clouds carry none)
    box = dict(synth.WORK_BOX)

    def fresh():
        obs = dict(depth=sc["depth"], K=sc["K"], pose=sc["pose"], color=color, mask_gs=[g.copy() for g in gs],
                   mask_label=[list(x) for x in labels], mask_conf=[c.copy() for c in confs])
        return R.make_reference_fusion(fusion, obs, H, W)

    # the stages one by one (what align_instance_mask_v3 does, fusion.py:1067-1094), to store the intermediate instances
    f = fresh()
    f.iou_threshold = 0.005
    lower = np.array([box["x_lower"], box["y_lower"], box["z_lower"]])
    higher = np.array([box["x_upper"], box["y_upper"], box["z_upper"]])
    f.voxel_num = ((higher - lower) / 0.03).astype(np.int32)
    (f.pcd_to_voxel, f.voxel_to_pcd, f.voxel_to_index, f.index_to_voxel, f.pcd_to_index, f.index_to_pcd) = \
        fusion._init_low_level_memory(lower, higher, 0.03, voxel_num=f.voxel_num)
    arrays = dict(H=H, W=W, V=V, seed=seed, K=sc["K"].numpy(), pose=sc["pose"].numpy(), depth=sc["depth"].numpy(),
                  queries=np.array(list(queries)),
                  bounds=np.array([box[k] for k in ("x_lower", "x_upper", "y_lower", "y_upper", "z_lower", "z_upper")]))
    for v in range(V):
        arrays["mask_gs_%d" % v] = np.packbits(gs[v], axis=None)
        arrays["mask_n_%d" % v] = np.int64(gs[v].shape[0])
        arrays["mask_label_%d" % v] = np.array(labels[v])
        arrays["mask_conf_%d" % v] = confs[v]
    instances = []
    for v in range(V):
        instances = f.merge_instances_from_new_view_vox_ver(instances, v, box)
        arrays["n_after_view_%d" % v] = np.int64(len(instances))
    _instances_summary("merged", copy.deepcopy(instances), V, arrays)
    arrays["n_merged"] = np.int64(len(instances))
    instances = f.filter_instances_vox_ver(instances)
    _instances_summary("filtered", copy.deepcopy(instances), V, arrays)
    arrays["n_filtered"] = np.int64(len(instances))
    # and the whole call
    g = fresh()
    g.align_instance_mask_v3(list(queries), box)
    arrays["consensus_mask_label"] = np.array(g.curr_obs_torch["consensus_mask_label"])
    arrays["mask"] = g.curr_obs_torch["mask"].numpy()
    assert arrays["mask"].dtype == np.uint8 and arrays["mask"].shape == (V, H, W)
    save(name, **arrays)
    print("   ", name, "instances after each view", [int(arrays["n_after_view_%d" % v]) for v in range(V)], "filtered", len(instances),
          "consensus", list(arrays["consensus_mask_label"]), "label histogram", np.bincount(arrays["mask"].reshape(-1)).tolist())


def main():
    torch.set_num_threads(4)
    fusion, corr = R.import_reference()
    scene_case(fusion, "scene_patchres_smooth", V=3, H=48, W=64, kind="smooth", fmap_hw=(12, 16), C=5, NI=8, N=4096)
    scene_case(fusion, "scene_patchres_stress", V=3, H=48, W=64, kind="stress", fmap_hw=(12, 16), C=5, NI=8, N=4096, seed=3)
    scene_case(fusion, "scene_fullres_smooth", V=4, H=48, W=64, kind="smooth", fmap_hw=(48, 64), C=12, NI=4, N=2048, seed=5)
    scene_case(fusion, "scene_denseK_1view", V=1, H=40, W=56, kind="stress", fmap_hw=(7, 9), C=3, NI=2, N=1024, dense_K=True, seed=7)
    scene_case(fusion, "scene_wideC_9views", V=9, H=30, W=40, kind="smooth", fmap_hw=(5, 7), C=70, NI=3, N=500, seed=9)
    batch_case(fusion)
    grid_case(fusion)
    onehot_case(fusion)
    corr_case(corr)
    grad_case(fusion)
    select_case(fusion)
    pcd_case(fusion)
    tracking_case(fusion)
    assoc_case(fusion)
    select_v2_case(fusion)
    masked_pcd_case(fusion)
    align_case(fusion, "align_v3_a", seed=3)
    align_case(fusion, "align_v3_b", seed=8, V=3, queries=("box", "mug"))


if __name__ == "__main__":
    main()
