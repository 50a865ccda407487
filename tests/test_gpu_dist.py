"""The distance-only pass (return_names=[], eval_dist; reference fusion.py:396-436, vis_repr.py:93) on its own kernel
(fused_eval_dist_kernel, fuse_direct.hip): 'dist' / 'valid_mask' bit for bit against the CPU oracle for 1..9 views (1..4: one batch in
SGPRs, 5..8: two batches, 9: the branch of fused_eval_kernel), tile tails, and projections built to hit every case in which the
short form of the IEEE division (d3f_device.h: project_point_short) must hand over to the compiler's form: quotients that are
exactly zero (+0 and -0), infinite, NaN, denormal, huge; points on / behind the camera plane."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def cpu(x):
    return x.detach().cpu().numpy()


def _fusion(dev, sc, H, W):
    from d3fields_amd import Fusion
    V = sc["K"].shape[0]
    f = Fusion(num_cam=V, device=str(dev))
    f.curr_obs_torch = {k: sc[k].to(dev) for k in ("depth", "K", "pose")}
    f.H, f.W = H, W
    f.record_plans = True
    return f


def _check(dev, sc, H, W, pts_c, what):
    from oracle import c_oracle as O
    f = _fusion(dev, sc, H, W)
    pts = pts_c.to(dev)
    with torch.no_grad():
        none = f.batch_eval(pts, return_names=[])
        plan = dict(f.last_plan())
        dd = f.eval_dist(pts)
    ref = O.eval_field(sc["depth"], sc["K"], sc["pose"], pts_c, [])
    ref_d = O.eval_field(sc["depth"], sc["K"], sc["pose"], pts_c, [], mode="eval_dist")
    assert np.array_equal(cpu(none["valid_mask"]), ref["valid_mask"].astype(bool)), what
    assert np.array_equal(cpu(none["dist"]), ref["dist"], equal_nan=True), what
    assert np.array_equal(cpu(dd["valid_mask"]), ref_d["valid_mask"].astype(bool)), what
    assert np.array_equal(cpu(dd["dist"]), ref_d["dist"], equal_nan=True), what
    return plan


@pytest.mark.parametrize("V", [1, 2, 3, 4, 5, 6, 7, 8, 9])
@pytest.mark.parametrize("kind", ["smooth", "stress"])
def test_dist_only_every_view_count(dev, V, kind):
    from d3fields_amd import synth
    H, W = 120, 160
    sc = synth.make_scene(V, H, W, kind, seed=V)
    n = 4 * 1024 + 257 + V                                  # several tiles of 256 points and a ragged tail
    pts_c = synth.random_cloud(n, seed=40 + V)
    r = np.random.default_rng(V)
    bad = r.integers(0, n, 8)
    pts_c[bad[0], 0] = float("inf"); pts_c[bad[1], 1] = float("-inf"); pts_c[bad[2], 2] = float("nan")
    pts_c[bad[3]] = torch.tensor([1e30, -1e30, 1e30]); pts_c[bad[4]] = 0.0; pts_c[bad[5], 2] = 1e-30
    pts_c[bad[6]] = torch.tensor([3e38, 3e38, -3e38]); pts_c[bad[7]] = torch.tensor([1e-40, -1e-42, 1e-45])
    plan = _check(dev, sc, H, W, pts_c, "V=%d %s" % (V, kind))
    assert plan["kernel"].startswith("fused_eval_dist_kernel" if V <= 8 else "fused_eval_kernel"), plan


def test_dist_only_big_batch_tiles(dev):
    """>= 2^22 points: 1024-point tiles (four points per lane) and the depth pixels looked up in the tiled copy (4 x 8-pixel tiles,
    fuse_direct.hip: depth_tile_kernel) -- the bench's dist_only geometry at a size the oracle finishes; map sizes that are not
    multiples of the tile."""
    from d3fields_amd import synth
    H, W = 243, 321
    sc = synth.make_scene(4, H, W, "smooth")
    n = (1 << 22) + 1000 + 3
    pts_c = synth.random_cloud(n, seed=5)
    pts_c[7] = torch.tensor([float("nan"), 0.0, 0.1]); pts_c[9] = torch.tensor([1e30, 0.0, 0.1]); pts_c[11] = 0.0
    plan = _check(dev, sc, H, W, pts_c, "big batch")
    assert plan["tile_points"] == 1024 and plan["kernel"].startswith("fused_eval_dist_kernel<0, 4, 6, true"), plan


def test_dist_only_projections_the_short_division_must_hand_over(dev):
    """Cameras at the origin looking down +z with the principal point at pixel (0, 0) (view 0) / in the image (views 1, 2): points on
    the axes project to EXACT zeros (+0 / -0 quotients), points at |z| < 1e-4 take zc = 1e-3 (fusion.py:52-53), huge and tiny
    coordinates give infinite, NaN, denormal and overflowing quotients.  Lane by lane the results must be the reference's."""
    H, W = 48, 64
    K = np.zeros((3, 3, 3), np.float32)
    for v, (fx, cx, cy) in enumerate([(50.0, 0.0, 0.0), (50.0, 31.5, 23.5), (1e-3, 7.0, 9.0)]):
        K[v] = [[fx, 0, cx], [0, fx, cy], [0, 0, 1]]
    pose = np.zeros((3, 3, 4), np.float32)
    pose[:, 0, 0] = pose[:, 1, 1] = pose[:, 2, 2] = 1.0
    pose[2, :, 3] = [0.0, 0.0, 1e-3]
    r = np.random.default_rng(7)
    depth = (0.5 + r.random((3, H, W))).astype(np.float32)
    depth[:, ::7, ::5] = 0.0
    sc = {"K": torch.from_numpy(K), "pose": torch.from_numpy(pose), "depth": torch.from_numpy(depth)}
    special = [0.0, -0.0, 1e-45, -1e-45, 1e-38, 1e-30, -1e-30, 1e-4, 9.9e-5, -9.9e-5, 1e-3, 0.5, 1.0, -1.0, 1e10, -1e10, 1e30, 3e38, -3e38,
               float("inf"), float("-inf"), float("nan")]
    pts = [(x, y, z) for x in special for y in (0.0, -0.0, 0.25, 1e-40, 1e38) for z in special]
    rows = r.random((2048, 3)).astype(np.float32)
    rows[:, 2] += 0.3
    rows[::3, 0] = 0.0                                      # xc == 0 in views 0 / 1: a +-0 quotient among ordinary lanes
    rows[1::5, 1] = -0.0
    pts_c = torch.cat([torch.tensor(pts, dtype=torch.float32), torch.from_numpy(rows)])
    _check(dev, sc, H, W, pts_c, "special projections")


def test_dist_only_short_and_long_divisions_agree_on_a_lattice(dev):
    """The bench's shape at a reduced size: a 1-mm lattice through a ray-cast scene, every point against the oracle."""
    from d3fields_amd import create_init_grid, synth
    H, W = 480, 640
    sc = synth.make_scene(4, H, W, "smooth")
    box = dict(x_lower=-0.2, x_upper=0.2, y_lower=-0.15, y_upper=0.15, z_lower=-0.05, z_upper=0.06)
    pts_c = create_init_grid(box, 0.002)[0]
    _check(dev, sc, H, W, pts_c, "lattice")


@pytest.mark.parametrize("V", [3, 8])
def test_dist_only_tiled_depth_on_a_lattice(dev, V):
    """A 2-mm lattice of 2^22+ points (the shape of vis_repr.py:93's query): tiled lookups, three and eight views, every point against the
    oracle; then the same query without scratch through the C-ABI's row-major path (reorder_points off) -- identical bits."""
    from d3fields_amd import create_init_grid, synth
    H, W = 240, 320
    sc = synth.make_scene(V, H, W, "smooth")
    step = 0.002
    dims = (170, 160, 155)
    box = dict(x_lower=-dims[0] * step / 2, x_upper=dims[0] * step / 2 - step / 4, y_lower=-dims[1] * step / 2,
               y_upper=dims[1] * step / 2 - step / 4, z_lower=-0.05, z_upper=-0.05 + dims[2] * step - step / 4)
    pts_c = create_init_grid(box, step)[0]
    assert pts_c.shape[0] == dims[0] * dims[1] * dims[2] >= 1 << 22
    plan = _check(dev, sc, H, W, pts_c, "lattice V=%d" % V)
    assert plan["kernel"].startswith("fused_eval_dist_kernel<0, %d, 6, true" % (V if V <= 4 else 0)), plan
    f = _fusion(dev, sc, H, W)
    pts = pts_c.to(dev)
    with torch.no_grad():
        a = f.batch_eval(pts, return_names=[])
        f.reorder_points = False
        b = f.batch_eval(pts, return_names=[])
        assert "false" in f.last_plan()["kernel"], f.last_plan()
    assert torch.equal(a["valid_mask"], b["valid_mask"])
    assert torch.equal(torch.nan_to_num(a["dist"], nan=7.0), torch.nan_to_num(b["dist"], nan=7.0))


@pytest.mark.parametrize("V,dims,step", [(4, (170, 160, 155), 0.002), (3, (90, 81, 37), 0.004), (8, (170, 160, 155), 0.002), (9, (60, 50, 41), 0.005)])
def test_grid_shell_fast_pass_keeps_the_same_survivors(dev, V, dims, step):
    """The keypoint pre-filter (select_features_rand, fusion.py:1418-1444) on the distance-only kernel's arithmetic (grid_kernels.hip:
    grid_shell_flag_fast_kernel; >= 2^22 grid points: tiled depth lookups; nine views: the general kernel): the survivors are exactly the
    lattice points the oracle's distance field selects, in ascending flat index."""
    from d3fields_amd import create_init_grid, synth
    from oracle import c_oracle as O
    H, W = 240, 320
    sc = synth.make_scene(V, H, W, "smooth")
    box = dict(x_lower=-dims[0] * step / 2, x_upper=dims[0] * step / 2 - step / 4, y_lower=-dims[1] * step / 2,
               y_upper=dims[1] * step / 2 - step / 4, z_lower=-0.05, z_upper=-0.05 + dims[2] * step - step / 4)
    pts_c = create_init_grid(box, step)[0]
    assert pts_c.shape[0] == dims[0] * dims[1] * dims[2]
    f = _fusion(dev, sc, H, W)
    thr = 2.5 * step
    idx, pts = f.grid_shell(box, step, thr)
    ref = O.eval_field(sc["depth"], sc["K"], sc["pose"], pts_c, [])
    want = np.nonzero(ref["valid_mask"].astype(bool) & (np.abs(ref["dist"]) < np.float32(thr)))[0]
    assert want.size > 1000
    assert np.array_equal(cpu(idx), want)
    assert np.array_equal(cpu(pts), pts_c.numpy()[want])


def test_dist_only_full_size_tiled_equals_row_major_and_oracle(dev):
    """bench.py's dist_only workload at its full size (123.2 M points of the 1-mm grid, 4096-point tiles): the tiled lookups against the
    row-major ones (no scratch) on EVERY point, bit for bit, and 300 000 random points of it against the oracle."""
    import bench
    from oracle import c_oracle as O
    f, pts, names, w, sc = bench.build_workload("dist_only", dev, 0, 1, "grid")
    assert names == [] and pts.shape[0] == w["N"] == 123200000
    f.record_plans = True
    with torch.no_grad():
        a = f.batch_eval(pts, return_names=[])
        plan = dict(f.last_plan())
        f.reorder_points = False
        b = f.batch_eval(pts, return_names=[])
        plan_b = dict(f.last_plan())
    assert plan["kernel"] == "fused_eval_dist_kernel<0, 4, 6, true, false>" and plan["tile_points"] == 4096, plan
    assert plan_b["kernel"] == "fused_eval_dist_kernel<0, 4, 6, false, false>", plan_b
    assert torch.equal(a["valid_mask"], b["valid_mask"])
    assert torch.equal(a["dist"], b["dist"])                      # (finite query points: no NaN in 'dist')
    assert 0.2 < float(a["valid_mask"].float().mean()) < 1.0
    pick = torch.randperm(pts.shape[0], generator=torch.Generator().manual_seed(4))[:300000].to(dev)
    ref = O.eval_field(sc["depth"], sc["K"], sc["pose"], pts[pick].cpu(), [])
    assert np.array_equal(cpu(a["valid_mask"][pick]), ref["valid_mask"].astype(bool))
    assert np.array_equal(cpu(a["dist"][pick]), ref["dist"])
