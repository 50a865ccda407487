"""Seeded sweep over shapes the named tests do not visit: view counts 1..8, odd channel counts, wide maps at patch / half / full
resolution, thin maps riding along or alone, lattices and clouds above the small-batch threshold (so that the window, cell-run
and channel-sliced launches are what runs), non-finite query points.  Every case is compared with the CPU oracle
(oracle/c_oracle.py, pinned to the reference's goldens): 'dist' / 'valid_mask' and every thin map bit for bit, wide maps within
the contract's 1e-5 (folded weights) and bit for bit with Fusion.reference_rounding."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def cpu(x):
    return x.detach().cpu().numpy()


def _case(seed):
    r = np.random.default_rng(1000 + seed)
    V = int(r.integers(1, 9))
    H, W = [(96, 128), (120, 160), (240, 320)][int(r.integers(0, 3))]
    wide = bool(r.integers(0, 4))                                     # one in four: only thin maps
    C = int(r.choice([128, 256, 384, 512, 1024] if r.integers(0, 3) else [3, 20, 96, 100, 130, 200])) if wide else 0
    res = int(r.integers(0, 3))                                       # patch / half / full resolution
    fhw = [(max(H // 10, 2), max(W // 10, 2)), (H // 2, W // 2), (H, W)][res]
    NI = int(r.choice([0, 2, 8]))
    color = bool(r.integers(0, 2))
    if not wide and NI == 0 and not color:
        NI = 8
    lattice = bool(r.integers(0, 3))                                  # two in three
    kind = "smooth" if r.integers(0, 4) else "stress"
    # one in three: the wide map is a channel range of a larger tensor (texel stride > C, first channel at any 4-byte offset)
    view = (int(r.choice([0, 1, 4, 7])), int(r.choice([0, 4, 5, 32]))) if r.integers(0, 3) == 0 else None
    return dict(V=V, H=H, W=W, C=C, fhw=fhw, NI=NI, color=color, lattice=lattice, kind=kind, view=view, seed=seed)


@pytest.mark.parametrize("seed", range(40))
def test_seeded_shape_against_oracle(dev, seed):
    _run(dev, _case(seed), seed)


# shapes the draw above rarely lands on: the pipelined window kernel with eight views, wide patch maps with both thin maps,
# two views on the window kernel's plain loop, a cloud on the cell runs of a 1024-d map
DIRECTED = [
    dict(V=8, H=240, W=320, C=1024, fhw=(24, 32), NI=8, color=True, lattice=True, kind="smooth"),
    dict(V=4, H=240, W=320, C=384, fhw=(24, 32), NI=0, color=False, lattice=True, kind="smooth", view=(4, 12)),      # 16-byte aligned, texel stride 400 floats
    dict(V=4, H=240, W=320, C=384, fhw=(240, 320), NI=8, color=False, lattice=True, kind="smooth", view=(1, 3)),     # 4-byte aligned only
    dict(V=8, H=240, W=320, C=256, fhw=(24, 32), NI=0, color=False, lattice=True, kind="stress"),
    dict(V=4, H=240, W=320, C=1024, fhw=(24, 32), NI=2, color=True, lattice=True, kind="stress"),
    dict(V=2, H=120, W=160, C=128, fhw=(12, 16), NI=8, color=False, lattice=True, kind="smooth"),
    dict(V=4, H=240, W=320, C=1024, fhw=(24, 32), NI=8, color=True, lattice=False, kind="smooth"),
    dict(V=1, H=240, W=320, C=384, fhw=(24, 32), NI=0, color=True, lattice=True, kind="smooth"),
]


@pytest.mark.parametrize("k", range(len(DIRECTED)))
def test_directed_shape_against_oracle(dev, k):
    _run(dev, dict(DIRECTED[k], seed=100 + k), 100 + k)


def _cloud_case(seed):
    """Clouds large enough for the gated window / cell-run pair of launches (>= 262 144 points, round 5): patch-resolution wide map,
    1 ... 8 views, the cloud's density drawn so that the device-side gate lands on either side."""
    r = np.random.default_rng(3000 + seed)
    V = int(r.choice([1, 2, 3, 4, 4, 6, 8]))
    H, W = [(120, 160), (240, 320), (480, 640)][int(r.integers(0, 3))]
    C = int(r.choice([128, 256, 384, 1024]))
    NI = int(r.choice([0, 0, 8]))
    color = bool(r.integers(0, 3) == 0)
    n = int(r.integers(262144, 420000))
    scale = float(r.choice([0.25, 0.5, 1.0, 1.0, 2.0]))                 # dense clouds fit the pool, sparse ones go to the cell runs
    view = (int(r.choice([0, 4])), int(r.choice([0, 32]))) if r.integers(0, 4) == 0 else None
    return dict(V=V, H=H, W=W, C=C, fhw=(max(H // 10, 2), max(W // 10, 2)), NI=NI, color=color, lattice=False,
                kind="smooth" if r.integers(0, 4) else "stress", view=view, cloud_n=n, cloud_scale=scale, seed=seed)


@pytest.mark.parametrize("seed", range(12))
def test_seeded_large_cloud_against_oracle(dev, seed):
    _run(dev, _cloud_case(seed), 200 + seed)


def _run(dev, c, seed):
    from d3fields_amd import Fusion, create_init_grid, synth
    from oracle import c_oracle as O
    V, H, W = c["V"], c["H"], c["W"]
    sc = synth.make_scene(V, H, W, c["kind"])
    maps, names = {}, []
    if c["C"]:
        off, pad = c.get("view") or (0, 0)
        big = synth.random_map(V, c["fhw"][0], c["fhw"][1], off + c["C"] + pad, seed=seed + 1, device=dev)
        maps["dino_feats"] = big[..., off:off + c["C"]]
        names.append("dino_feats")
    if c["NI"]:
        maps["mask"] = synth.random_onehot_mask(V, H, W, c["NI"], seed=seed + 2, device=dev)
        names.append("mask")
    if c["color"]:
        maps["color_tensor"] = torch.rand(V, H, W, 3, generator=torch.Generator(device=dev).manual_seed(seed + 3), device=dev)
        names.append("color_tensor")
    f = Fusion(num_cam=V, device=str(dev))
    f.curr_obs_torch = {k: sc[k].to(dev) for k in ("depth", "K", "pose")}
    f.curr_obs_torch.update(maps)
    f.H, f.W = H, W
    f.record_plans = True
    r = np.random.default_rng(2000 + seed)
    if c["lattice"]:
        dims = [(48, 44, 32), (130, 9, 60), (64, 64, 17), (20, 120, 28)][int(r.integers(0, 4))]       # >= 65536 points
        step = float(r.choice([0.004, 0.0107, 0.02]))
        box = dict(x_lower=-dims[0] * step / 2, x_upper=dims[0] * step / 2 - step / 4, y_lower=-dims[1] * step / 2,
                   y_upper=dims[1] * step / 2 - step / 4, z_lower=-0.2, z_upper=-0.2 + dims[2] * step - step / 4)
        pts_c = create_init_grid(box, step)[0]
        assert pts_c.shape[0] == dims[0] * dims[1] * dims[2]
    else:
        pts_c = synth.random_cloud(c.get("cloud_n", 70001), seed=seed) * c.get("cloud_scale", 1.0)
    bad = r.integers(0, pts_c.shape[0], 6)
    pts_c[bad[0], 0] = float("inf"); pts_c[bad[1], 1] = float("-inf"); pts_c[bad[2], 2] = float("nan")
    pts_c[bad[3]] = torch.tensor([1e30, -1e30, 1e30]); pts_c[bad[4]] = 0.0; pts_c[bad[5], 2] = 1e-30
    pts = pts_c.to(dev)
    with torch.no_grad():
        out = f.batch_eval(pts, return_names=names)
        plan = dict(f.last_plan())
        f.reference_rounding = True
        strict = f.batch_eval(pts, return_names=names)
        f.reference_rounding = False
        sub = torch.from_numpy(r.permutation(pts_c.shape[0])[:4000]).to(dev)
        sub[:6] = torch.from_numpy(bad).to(dev)
        inter = f.eval(pts[sub], return_names=names, return_inter=True)
        dd = f.eval_dist(pts)
        none = f.batch_eval(pts, return_names=[])                                       # the distance-only launch
    ref = O.eval_field(sc["depth"], sc["K"], sc["pose"], pts_c, [maps[k].float().cpu() for k in names])
    ref_i = O.eval_field(sc["depth"], sc["K"], sc["pose"], pts_c[sub.cpu()], [maps[k].float().cpu() for k in names], return_inter=True)
    what = "case %s, plan %s" % (c, {k: plan.get(k) for k in ("kernel", "point_order")})
    for o in (out, strict, none):
        assert np.array_equal(cpu(o["valid_mask"]), ref["valid_mask"].astype(bool)), what
        assert np.array_equal(cpu(o["dist"]), ref["dist"], equal_nan=True), what
    ref_d = O.eval_field(sc["depth"], sc["K"], sc["pose"], pts_c, [], mode="eval_dist")
    assert np.array_equal(cpu(dd["valid_mask"]), ref_d["valid_mask"].astype(bool)), what
    assert np.array_equal(cpu(dd["dist"]), ref_d["dist"], equal_nan=True), what
    for j, k in enumerate(names):
        want = ref["sets"][j]
        fin = np.isfinite(want).all(axis=1)
        wide = k == "dino_feats" and c["C"] * 4 > 256                                     # folded weights on the fast path
        for o in (out, strict):
            got = cpu(o[k])
            assert np.array_equal(np.isfinite(got).all(axis=1), fin), (k, what)
            assert rel_err(got[fin], want[fin]) <= 1e-5, (k, what)                        # (the device's expf is not the host's)
        if not wide:
            assert torch.equal(torch.nan_to_num(out[k], nan=7.0, posinf=8.0, neginf=9.0),
                               torch.nan_to_num(strict[k], nan=7.0, posinf=8.0, neginf=9.0)), (k, what)      # one path for thin maps
        assert np.array_equal(cpu(inter[k + "_inter"]), ref_i["inter"][j], equal_nan=True), (k, what)       # no weights in these: bit for bit
    if c["NI"]:
        from d3fields_amd import onehot2instance
        m = ref["sets"][names.index("mask")]
        top2 = np.sort(m, axis=1)[:, -2:]
        clear = np.isfinite(m).all(axis=1) & (top2[:, 1] - top2[:, 0] > 1e-4 * np.maximum(top2[:, 1], 1e-30))
        assert np.array_equal(cpu(onehot2instance(out["mask"]))[clear], np.argmax(m, axis=1)[clear]), what


@pytest.mark.parametrize("seed", range(16))
def test_seeded_pairwise_against_oracle(dev, seed):
    """utils/corr_utils.py's softmax-normalised descriptor similarity on drawn shapes (rows 1 ... 30 000, columns 1 ... 400, channel
    counts that are and are not multiples of the kernel's 32-channel stage, both distance types, scales over two decades)."""
    from d3fields_amd import corr_utils as cu
    from oracle import c_oracle as O
    r = np.random.default_rng(3000 + seed)
    B1 = int(r.choice([1, 2, 63, 64, 65, 1000, 4097, 30000]))
    B2 = int(r.choice([1, 5, 15, 16, 17, 64, 100, 300, 400]))
    C = int(r.choice([1, 3, 31, 32, 33, 96, 384, 1000]))
    dist_type = "l2" if r.integers(0, 2) else "square"
    scale = float(r.choice([0.05, 0.7, 3.0]))
    g = torch.Generator().manual_seed(seed)
    src = torch.randn(B1, C, generator=g) * float(r.choice([0.1, 1.0]))
    tgt = torch.randn(B2, C, generator=g) * float(r.choice([0.1, 1.0]))
    tgt[B2 // 2] = src[B1 // 3]
    sim = cu.compute_similarity_tensor_multi(src.to(dev), tgt.to(dev), None, None, scale, dist_type)
    ref, am = O.pairwise(src.numpy(), tgt.numpy(), scale, dist_type, return_argmax=True)
    what = dict(B1=B1, B2=B2, C=C, dist_type=dist_type, scale=scale)
    assert sim.shape == (B1, B2), what
    assert rel_err(cpu(sim), ref) <= 1e-5, what
    assert torch.allclose(sim.sum(0), torch.ones(B2, device=dev), atol=1e-4), what
    if dist_type == "l2":
        out, idx = cu.nearest_descriptor(src.to(dev), tgt.to(dev), scale)
        assert rel_err(cpu(out), ref) <= 1e-5, what
        col = ref[:, :]                                     # the arg max is only pinned where the oracle's best row is clear
        best = np.sort(col, axis=0)[-2:] if B1 > 1 else None
        clear = np.ones(B2, bool) if B1 == 1 else (best[1] - best[0] > 1e-4 * np.maximum(best[1], 1e-30))
        assert np.array_equal(cpu(idx)[clear], am[clear]), what
        assert idx[B2 // 2].item() == B1 // 3 or not clear[B2 // 2], what


@pytest.mark.parametrize("seed", range(10))
def test_seeded_gradient_against_torch_port(dev, seed):
    """d3f_eval_backward (gradient with respect to the query points) on drawn shapes, random upstream gradients, against autograd
    through the torch-ops port of the reference's op sequence (oracle/torch_port.py, pinned to the reference's own gradient)."""
    from d3fields_amd import Fusion, synth
    from oracle import torch_port
    r = np.random.default_rng(4000 + seed)
    V = int(r.integers(1, 9))
    H, W = [(48, 64), (96, 128)][int(r.integers(0, 2))]
    C = int(r.choice([1, 7, 32, 100, 384]))
    fhw = [(max(H // 8, 2), max(W // 8, 2)), (H, W)][int(r.integers(0, 2))]
    names = ["dino_feats"] + (["mask"] if r.integers(0, 2) else []) + (["color_tensor"] if r.integers(0, 2) else [])
    kind = "smooth" if r.integers(0, 3) else "stress"
    N = int(r.choice([1, 257, 3000]))
    sc = synth.make_scene(V, H, W, kind)
    maps = {"dino_feats": synth.random_map(V, fhw[0], fhw[1], C, seed=seed + 1), "mask": synth.random_onehot_mask(V, H, W, 5, seed=seed + 2),
            "color_tensor": torch.rand(V, H, W, 3, generator=torch.Generator().manual_seed(seed + 4))}
    f = Fusion(num_cam=V, device=str(dev))
    f.curr_obs_torch = {k: sc[k].to(dev) for k in ("depth", "K", "pose")}
    f.curr_obs_torch.update({k: m.to(dev) for k, m in maps.items()})
    f.H, f.W = H, W
    pts = synth.random_cloud(N, seed=seed + 6)
    gen = torch.Generator().manual_seed(seed + 7)
    w_dist = torch.randn(N, generator=gen)
    w_k = {k: torch.randn(N, maps[k].shape[3], generator=gen) for k in names}
    p_ref = pts.clone().requires_grad_(True)
    obs = dict(sc)
    obs.update(maps)
    o_ref = torch_port.field_query(obs, p_ref, names, H, W)
    ((o_ref["dist"] * w_dist).sum() + sum((o_ref[k] * w_k[k]).sum() for k in names)).backward()
    p_gpu = pts.to(dev).requires_grad_(True)
    o = f.eval(p_gpu, return_names=names)
    ((o["dist"] * w_dist.to(dev)).sum() + sum((o[k] * w_k[k].to(dev)).sum() for k in names)).backward()
    what = dict(V=V, H=H, W=W, C=C, fhw=fhw, names=names, kind=kind, N=N)
    assert rel_err(cpu(p_gpu.grad), p_ref.grad.numpy()) <= 2e-5, what        # GRAD_TOL of test_gpu_parity.py


@pytest.mark.parametrize("seed", range(12))
def test_seeded_integer_callers_against_numpy(dev, seed):
    """The integer steps of instance association and keypoint selection on drawn sizes -- cv2.erode with all-ones kernels of any
    size, farthest-point sampling on pixel coordinates, voxel linearisation, the voxel-set IoU -- against their numpy
    restatements (oracle/np_pcd.py, pinned to the reference's closures by the golden tests): all exact."""
    from d3fields_amd import pcd_utils
    from d3fields_amd.fusion import erode, _init_low_level_memory
    from oracle import np_pcd
    r = np.random.default_rng(5000 + seed)
    # erode
    shape = (int(r.integers(1, 200)), int(r.integers(1, 200)))
    k = (int(r.integers(1, 20)), int(r.integers(1, 20)))
    img = ((r.random(shape) < r.uniform(0.5, 0.99)) * 255).astype(np.uint8)
    if r.integers(0, 2):
        img = r.integers(0, 256, size=shape, dtype=np.uint8)
    kern = np.ones(list(k), np.uint8)
    assert np.array_equal(erode(img, kern, iterations=1), np_pcd.erode_cv2(img, kern)), (shape, k)
    # farthest-point sampling on pixels (ties: the first maximum wins)
    mask = r.random((int(r.integers(2, 120)), int(r.integers(2, 160)))) < r.uniform(0.05, 0.9)
    pix = np.array(mask.nonzero()).T
    if pix.shape[0] >= 1:
        kk, start = int(r.integers(1, 60)), int(r.integers(0, pix.shape[0]))
        sel, idx, md = pcd_utils.fps_pixels(pix, kk, init_idx=start)
        wsel, widx, wmd = np_pcd.fps_int(pix, kk, start)
        assert idx == widx and np.array_equal(sel, wsel) and md == wmd, (pix.shape, kk, start)
    # voxel linearisation (float64 -> int32 casts, wrap-around) and the voxel-set IoU
    n = int(r.choice([1, 1000, 200000]))
    pcd = r.uniform(-3.0, 3.0, size=(n, 3))
    lower, vs = r.uniform(-1.0, 0.0, size=3), float(r.choice([0.001, 0.01, 0.1]))
    num = r.integers(1, 3000, size=3).astype(np.int32)
    with np.errstate(invalid="ignore", over="ignore"):
        want = np_pcd.pcd_to_index(pcd, lower, vs, num)
    assert np.array_equal(_init_low_level_memory(lower, lower + 1, vs, num)[4](pcd), want), (n, vs, num)
    a = r.integers(-50000, 50000, size=int(r.integers(1, 300000))).astype(np.int32)
    b = np.concatenate([a[:: int(r.integers(1, 5))], r.integers(-2**31, 2**31, size=int(r.integers(0, 1000)), dtype=np.int64).astype(np.int32)])
    assert pcd_utils.vox_idx_iou(a, b) == np_pcd.vox_idx_iou(a, b)
