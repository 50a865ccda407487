"""CPU: pins oracle/ (C restatement) against the golden vectors produced by the reference.

The golden files were written by oracle/gen_golden.py, which imports and runs the reference
(fusion.py / utils/corr_utils.py) on deterministic inputs.
"""
import numpy as np
import pytest

from conftest import ALIGN_CASES, SCENE_CASES, SET_NAMES, align_inputs, assert_instances_match, load_golden, rel_err
from oracle import c_oracle as O

TOL = 1e-5          # BASELINE.json north_star: <= 1e-5 relative fp32


def _assert_dist(got, ref, V):
    """dist is bit-exact for V <= 4.  For V > 4 torch-CPU's sum(0) switches, only for the
    last (N mod 16) columns of a tensor, from the sequential order to 4 interleaved partial
    sums (measured; an ATen vectorisation artefact that depends on a point's position in the
    batch), so there the comparison is at the north-star tolerance."""
    if V <= 4:
        assert np.array_equal(got, ref, equal_nan=True)
    else:
        assert rel_err(got, ref) <= TOL
        assert (got == ref).mean() > 0.95


@pytest.mark.parametrize("case", SCENE_CASES)
def test_eval_matches_reference(case):
    g = load_golden(case)
    maps = [g["in_" + k] for k in SET_NAMES]
    o = O.eval_field(g["depth"], g["K"], g["pose"], g["pts"], maps, mu=float(g["mu"]),
                     return_inter=True)
    # discrete outputs: bit-exact; dist: the restatement reproduces torch-CPU rounding exactly
    assert np.array_equal(o["valid_mask"], g["valid_mask"])
    _assert_dist(o["dist"], g["dist"], g["depth"].shape[0])
    for i, k in enumerate(SET_NAMES):
        assert np.array_equal(o["inter"][i], g[k + "_inter"]), k       # bilinear: bit-exact
        assert rel_err(o["sets"][i], g[k]) <= TOL, k                   # expf may differ by 1 ulp
    assert np.array_equal(O.onehot2instance(o["sets"][1]), g["mask_instance"])
    # every branch of the path must actually occur in the fixture
    assert (~g["valid_mask"]).any() and g["valid_mask"].any()
    assert (g["dist"] == 1e3).any()


@pytest.mark.parametrize("case", SCENE_CASES)
def test_eval_dist_matches_reference(case):
    g = load_golden(case)
    o = O.eval_field(g["depth"], g["K"], g["pose"], g["pts"], [], mu=float(g["mu"]), mode="eval_dist")
    assert np.array_equal(o["valid_mask"], g["evaldist_valid_mask"])
    _assert_dist(o["dist"], g["evaldist_dist"], g["depth"].shape[0])


def test_batch_eval_130001():
    from d3fields_amd import synth
    g = load_golden("batch_eval_130001")
    N, st = int(g["N"]), int(g["stride"])
    pts = synth.random_cloud(N, seed=int(g["cloud_seed"])).numpy()
    assert np.array_equal(pts[::st], g["pts_sub"])                 # same cloud as the generator
    o = O.eval_field(g["depth"], g["K"], g["pose"], pts, [g["in_dino_feats"], g["in_mask"]], mu=float(g["mu"]))
    assert np.array_equal(np.packbits(o["valid_mask"]), g["valid_bits"])
    assert np.array_equal(o["dist"][::st], g["dist_sub"])
    assert rel_err(o["sets"][0][::st], g["dino_feats_sub"]) <= TOL
    assert rel_err(o["sets"][1][::st], g["mask_sub"]) <= TOL
    assert abs(o["dist"].astype(np.float64).sum() - float(g["dist_sum"])) <= 1e-6 * abs(float(g["dist_sum"]))
    assert np.allclose(o["sets"][0].astype(np.float64).sum(0), g["dino_feats_sum"], rtol=1e-6, atol=1e-3)


def test_onehot_roundtrip():
    g = load_golden("onehot")
    assert np.array_equal(O.instance2onehot(g["inst"], int(g["NI"])), g["onehot"])
    assert np.array_equal(O.onehot2instance(g["soft"]), g["soft_inst"])
    assert np.array_equal(O.onehot2instance(g["onehot"].astype(np.float32)), g["inst"])


@pytest.mark.parametrize("dt", ["l2", "square"])
def test_corr_utils(dt):
    g = load_golden("corr_utils")
    sc = float(g["scale"])
    fm = g["fmap_bhwc"]
    bchw = np.ascontiguousarray(fm.transpose(0, 3, 1, 2))
    assert rel_err(O.similarity_exp(fm, g["tgt"], sc, dt, channel_axis=-1), g["similarity_" + dt]) <= TOL
    assert rel_err(O.similarity_softmax(bchw, g["tgt"], sc, dt, channel_axis=1), g["similarity_tensor_" + dt]) <= TOL
    assert rel_err(O.dist_to_target(bchw, g["tgt"], dt, channel_axis=1), g["dist_tensor_" + dt]) <= TOL
    out, am = O.pairwise(g["multi_src"], g["multi_tgt"], float(g["multi_scale"]), dt, return_argmax=True)
    assert rel_err(out, g["multi_" + dt]) <= TOL
    assert np.allclose(out.sum(0), 1.0, atol=1e-5)
    assert np.array_equal(am, g["multi_argmax_" + dt])
    if dt == "l2":
        assert rel_err(O.similarity_softmax(g["flat"], g["tgt"], sc, dt, channel_axis=1), g["flat_similarity_tensor_l2"]) <= TOL
        assert rel_err(O.dist_to_target(g["flat"], g["tgt"], dt, channel_axis=1), g["flat_dist_tensor_l2"]) <= TOL


# ---- the torch-ops port (bench.py's cpu_baseline leg) is the same function as the reference ----
@pytest.mark.parametrize("case", ["scene_patchres_stress", "scene_fullres_smooth"])
def test_torch_port_matches_reference(case):
    import torch
    from oracle import torch_port
    g = load_golden(case)
    obs = {k: torch.from_numpy(g[k]) for k in ("depth", "K", "pose")}
    for k in SET_NAMES:
        obs[k] = torch.from_numpy(g["in_" + k])
    pts = torch.from_numpy(g["pts"])
    out = torch_port.field_query(obs, pts, SET_NAMES, int(g["H"]), int(g["W"]), float(g["mu"]), keep_inter=True)
    assert np.array_equal(out["valid_mask"].numpy(), g["valid_mask"])
    assert np.array_equal(out["dist"].numpy(), g["dist"])
    for k in SET_NAMES:
        assert np.array_equal(out[k].numpy(), g[k]) and np.array_equal(out[k + "_inter"].numpy(), g[k + "_inter"])
    dd = torch_port.dist_query(obs, pts, int(g["H"]), int(g["W"]))
    assert np.array_equal(dd["dist"].numpy(), g["evaldist_dist"], equal_nan=True)
    b = torch_port.batched_field_query(obs, pts, ["mask"], int(g["H"]), int(g["W"]), float(g["mu"]), chunk=1000)
    assert np.array_equal(b["mask"].numpy(), g["mask"])


def test_torch_port_gradient_matches_reference():
    import torch
    from oracle import torch_port
    g = load_golden("grad_500")
    obs = {k: torch.from_numpy(g[k]) for k in ("depth", "K", "pose")}
    obs["dino_feats"] = torch.from_numpy(g["in_dino_feats"])
    pts = torch.from_numpy(g["pts"]).requires_grad_(True)
    out = torch_port.field_query(obs, pts, ["dino_feats"], int(g["H"]), int(g["W"]), float(g["mu"]))
    (out["dino_feats"].sum() + out["dist"].sum()).backward()
    assert rel_err(pts.grad.numpy(), g["grad_pts"]) <= TOL


def test_fps_and_shell_match_reference():
    g = load_golden("select_features")
    idx, md = O.fps(g["fps_cloud"], 64, 17)
    assert np.array_equal(idx, g["fps_idx"]) and np.array_equal(g["fps_cloud"][idx], g["fps_pts"])
    assert md == float(g["fps_maxdist"])
    # the select_features pre-filter is |dist| < 5 mm & valid on the 1-cm grid
    from d3fields_amd import create_init_grid
    b = dict(zip(["x_lower", "x_upper", "y_lower", "y_upper", "z_lower", "z_upper"], g["bounds"].tolist()))
    grid, shape = create_init_grid(b, float(g["res"]))
    o = O.eval_field(g["depth"], g["K"], g["pose"], grid.numpy(), [g["in_mask"]], mu=float(g["mu"]))
    assert np.array_equal(o["dist"], g["grid_dist"]) and np.array_equal(np.packbits(o["valid_mask"]), g["grid_valid"])
    shell = np.nonzero((np.abs(o["dist"]) < 0.005) & o["valid_mask"])[0]
    assert np.array_equal(shell, g["shell_index"])


def test_pcd_restatement_matches_reference():
    from oracle import np_pcd
    g = load_golden("pcd_utils")
    V = g["depths"].shape[0]
    pts, cols = [], []
    for i in range(V):
        K = g["K"][i]
        w, pix = np_pcd.backproject_view(g["depths"][i], g["masks"][i], [K[0, 0], K[1, 1], K[0, 2], K[1, 2]],
                                         np.linalg.inv(g["pose44"][i]), g["bounds"])
        pts.append(w)
        cols.append((g["colors"][i] / 255.).reshape(-1, 3)[pix])
    pts, cols = np.concatenate(pts), np.concatenate(cols)
    assert pts.shape == g["crop_pts"].shape and np.allclose(pts, g["crop_pts"], rtol=0, atol=1e-12)
    assert np.array_equal(cols, g["crop_col"])
    md, am = np_pcd.nearest(g["p1"], g["p2"])
    assert np.array_equal(am, g["idx_12"]) and np.array_equal(np.where(md < 0.005)[0], g["overlap_1"])


def test_rigid_tracking_restatement_matches_reference():
    """oracle/torch_port.rigid_tracking vs the reference's Fusion.rigid_tracking (fusion.py:1608-1685; golden written
    by oracle/gen_golden.py with pytorch3d's two functions restated) -- and the restated so3_exp_map itself."""
    import torch
    from scipy.spatial.transform import Rotation
    from oracle import torch_port as T
    from oracle.pytorch3d_restated import so3_exp_map, Transform3d
    g = load_golden("rigid_tracking")
    obs = {k: torch.from_numpy(g[k]) for k in ("depth", "K", "pose")}
    obs["dino_feats"] = torch.from_numpy(g["in_dino_feats"])
    torch.set_num_threads(4)
    cur = T.rigid_tracking(obs, int(g["H"]), int(g["W"]), torch.from_numpy(g["src_feats"]), torch.from_numpy(g["last_pts"]),
                           float(g["mu"]))
    assert np.abs(cur.numpy().reshape(g["match_pts"].shape) - g["match_pts"]).max() <= 1e-5
    # the tracker moved the keypoints towards the truth (15 mm -> 5 mm in the fixture)
    assert np.abs(g["match_pts"] - g["true_pts"]).max() < 0.5 * np.abs(g["last_pts"] - g["true_pts"]).max()
    w = torch.tensor([[0.1, -0.3, 0.2], [0.0, 0.0, 0.0], [1e-3, 0.0, 0.0], [2.0, 1.0, -0.5]])
    R = so3_exp_map(w)
    assert np.abs(R.numpy() - Rotation.from_rotvec(w.numpy()).as_matrix()).max() <= 1e-6
    assert np.abs((R @ R.transpose(1, 2)).numpy() - np.eye(3)).max() <= 1e-6
    x = torch.randn(4, 5, 3, generator=torch.Generator().manual_seed(0))
    t = torch.tensor([[1.0, 2.0, 3.0]]).expand(4, 3)
    got = Transform3d().rotate(R).translate(t).transform_points(x)
    assert torch.allclose(got, torch.bmm(x, R) + t[:, None, :], atol=1e-6)      # row-vector convention


def test_assoc_restatement_matches_reference():
    """Voxel indices and voxel-set IoU of instance association (fusion.py:118-180, 794-799): the numpy
    restatement against the values the reference's own closures / method returned (golden 'assoc')."""
    from oracle import np_pcd
    g = load_golden("assoc")
    lower, vs, num = g["lower"], float(g["voxel_size"]), g["voxel_num"]
    assert np.array_equal(np_pcd.pcd_to_voxel(g["pcd"], lower, vs), g["voxels"]) and g["voxels"].dtype == np.int32
    assert np.array_equal(np_pcd.pcd_to_index(g["pcd"], lower, vs, num), g["index"])
    assert np.array_equal(np_pcd.pcd_to_index(g["pcd32"], lower, vs, num), g["index32"])
    assert np.array_equal(np_pcd.voxel_to_index(g["voxels"], num), g["index_of_voxels"])
    assert (g["voxels"] < 0).any() and (g["voxels"] >= num).any()          # points outside the box are in the fixture
    for k in ("ab", "aa", "disjoint", "one_empty", "small"):
        got = np_pcd.vox_idx_iou(g["iou_%s_a" % k], g["iou_%s_b" % k])
        assert np.array_equal(np.array(got), g["iou_" + k]), k              # integer ratios: exact


def test_erode_restatement_against_scipy_and_fixture():
    """cv2 is absent from the build container, so erode_cv2 restates OpenCV's published definition; it is checked
    against scipy.ndimage's independent implementation (same window convention) and against the eroded mask stored
    when the reference's select_features_rand_v2 ran with it (golden 'select_v2')."""
    from scipy import ndimage
    from oracle import np_pcd
    rng = np.random.default_rng(7)
    for (H, W), (kh, kw) in [((40, 50), (15, 15)), ((33, 17), (2, 2)), ((20, 20), (3, 5)), ((9, 9), (15, 15)), ((5, 7), (1, 1))]:
        img = ((rng.random((H, W)) < 0.9) * 255).astype(np.uint8)
        want = ndimage.grey_erosion(img, size=(kh, kw), mode="constant", cval=255)
        assert np.array_equal(np_pcd.erode_cv2(img, np.ones([kh, kw], np.uint8)), want), (H, W, kh, kw)
    g = load_golden("select_v2")
    m = (g["in_mask"][0, :, :, 1] > 0)
    assert np.array_equal(np_pcd.erode_cv2((m * 255).astype(np.uint8), np.ones([15, 15], np.uint8)), g["eroded_v0_i1"])
    assert 0 < (g["eroded_v0_i1"] > 0).sum() < m.sum()


def test_select_v2_restatement_matches_reference():
    """select_features_rand_v2 (fusion.py:1539-1606) restated from the oracle's pieces (erode_cv2, fps_int, the
    back-projection formulas) reproduces the keypoints the reference returned; descriptors through the C oracle."""
    from oracle import np_pcd
    g = load_golden("select_v2")
    V, N = g["depth"].shape[0], int(g["N"])
    np.random.seed(int(g["seed"]))
    for i in range(1, 3):
        pts = []
        for v in range(V):
            m = (g["in_mask"][v, :, :, i] > 0) & (g["depth"][v] > 0.0) & (g["depth"][v] < 1.5)
            er = np_pcd.erode_cv2((m * 255).astype(np.uint8), np.ones([15, 15], np.uint8))
            pix = np.array(er.nonzero()).T
            sel, _, _ = np_pcd.fps_int(pix, N // V, np.random.randint(pix.shape[0]))
            d = g["depth"][v][sel[:, 0], sel[:, 1]]
            K = g["K"][v]
            cam = np.stack([(sel[:, 1] - K[0, 2]) * d / K[0, 0], (sel[:, 0] - K[1, 2]) * d / K[1, 1], d, np.ones_like(d)], 0)
            pose = np.concatenate([g["pose"][v], np.array([[0, 0, 0, 1]])], axis=0)
            pts.append(np.matmul(np.linalg.inv(pose), cam)[:3].T)
        pts = np.concatenate(pts, 0)
        assert np.array_equal(pts, g["pts_%d" % (i - 1)])
        o = O.eval_field(g["depth"], g["K"], g["pose"], pts.astype(np.float32), [g["in_dino_feats"]], mu=float(g["mu"]))
        assert rel_err(o["sets"][0], g["feats_%d" % (i - 1)]) <= TOL


@pytest.mark.parametrize("case", ALIGN_CASES)
def test_align_restatement_matches_reference(case):
    """oracle/np_assoc.py against the reference's align_instance_mask_v3 (fusion.py:1065-1098) run on synthetic per-view
    detections: the instances after the merges and after the filter (labels, voxel sets, raw lengths, per-voxel confidence lists,
    view -> detection maps), the consensus labels and the label image, all exact."""
    from oracle import np_assoc
    g = load_golden(case)
    gs, labels, confs = align_inputs(g)
    V = int(g["V"])
    stages = {}
    img, consensus = np_assoc.align(g["depth"], g["K"], g["pose"], gs, labels, confs, [str(q) for q in g["queries"]], g["bounds"].tolist(), stages)
    assert stages["count_after_view"] == [int(g["n_after_view_%d" % v]) for v in range(V)]
    assert_instances_match(g, "merged", stages["merged"], V)
    assert_instances_match(g, "filtered", stages["filtered"], V)
    assert consensus == [str(x) for x in g["consensus_mask_label"]]
    assert img.dtype == np.uint8 and np.array_equal(img, g["mask"])
