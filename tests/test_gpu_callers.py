"""GPU parity of the callers either side of the field query (SURVEY §8f row 4 and the judge-added rows): voxel-index
IoU of instance association, cv2.erode / pixel FPS / select_features_rand_v2, and the text_queries_* state contract
driven end to end (update -> text_queries -> select_features -> tracking) with stub producers.  Every expected value
comes from a golden written by running the reference (oracle/gen_golden.py) or from the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import ALIGN_CASES, align_inputs, assert_instances_match, load_golden, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def cpu(x):
    return x.detach().cpu().numpy()


# ---- voxel indices / voxel-set IoU (fusion.py:118-180, 794-799): integers, bit-exact ---------------------------------
def test_pcd_to_index_matches_reference(dev):
    from d3fields_amd.fusion import _init_low_level_memory
    g = load_golden("assoc")
    fns = _init_low_level_memory(g["lower"], g["higher"], float(g["voxel_size"]), g["voxel_num"])
    pcd_to_voxel, voxel_to_pcd, voxel_to_index, index_to_voxel, pcd_to_index, index_to_pcd = fns
    vox, idx = pcd_to_voxel(g["pcd"]), pcd_to_index(g["pcd"])
    assert vox.dtype == np.int32 and idx.dtype == np.int32
    assert np.array_equal(vox, g["voxels"]) and np.array_equal(idx, g["index"])
    assert np.array_equal(pcd_to_index(g["pcd32"]), g["index32"])                     # float32 clouds are promoted
    assert np.array_equal(pcd_to_index(g["pcd"].reshape(60, 100, 3)), g["index"].reshape(60, 100))   # (..., 3) shapes
    assert np.array_equal(pcd_to_index(g["pcd"][:7].tolist()), g["index"][:7])        # lists, like the reference
    assert np.array_equal(voxel_to_index(g["voxels"]), g["index_of_voxels"])
    assert np.array_equal(index_to_voxel(g["index"][100:200]), g["voxel_of_index"])
    assert np.array_equal(index_to_pcd(g["index"][100:200]), g["pcd_of_index"])
    assert pcd_to_index(np.zeros((0, 3))).shape == (0,)


def test_pcd_to_index_large_and_wraparound(dev):
    """2 M points, huge voxel counts (int32 wrap-around of the linearisation) and non-finite / far-away points
    (numpy's float64 -> int32 cast): equal to the numpy restatement, which the golden test pins."""
    from d3fields_amd.fusion import _init_low_level_memory
    from oracle import np_pcd
    rng = np.random.default_rng(3)
    pcd = rng.uniform(-50.0, 50.0, size=(2_000_000, 3))
    pcd[:4] = [[np.nan, 0, 0], [np.inf, 1, 1], [-1e30, 2, 2], [1e12, -3, 3]]
    lower, vs, num = np.array([-1.0, -2.0, -3.0]), 0.001, np.array([70000, 70000, 70000], np.int32)
    with np.errstate(invalid="ignore", over="ignore"):
        want = np_pcd.pcd_to_index(pcd, lower, vs, num)
    got = _init_low_level_memory(lower, lower + 1, vs, num)[4](pcd)
    assert np.array_equal(got, want)


def test_vox_idx_iou_matches_reference(dev):
    from d3fields_amd import Fusion
    f = Fusion(num_cam=1, device=str(dev))
    g = load_golden("assoc")
    for k in ("ab", "aa", "disjoint", "one_empty", "small"):
        got = f.vox_idx_iou(g["iou_%s_a" % k], g["iou_%s_b" % k])
        assert isinstance(got[0], float) and np.array_equal(np.array(got), g["iou_" + k]), k
    with pytest.raises(ZeroDivisionError):                                             # like the reference on two empty sets
        f.vox_idx_iou(np.zeros(0, np.int32), np.zeros(0, np.int32))


def test_vox_idx_iou_large_random_keys(dev):
    """1.5 M + 1 M indices over the whole int32 range (negative keys, heavy duplication, hash collisions)."""
    from d3fields_amd import pcd_utils
    from oracle import np_pcd
    rng = np.random.default_rng(5)
    a = rng.integers(-2**31, 2**31, size=1_500_000, dtype=np.int64).astype(np.int32)
    b = np.concatenate([a[::3], rng.integers(-1000, 1000, size=500_000).astype(np.int32)])
    a[:100000] = a[0]
    assert pcd_utils.vox_idx_iou(a, b) == np_pcd.vox_idx_iou(a, b)


# ---- cv2.erode / fps_np on pixels / select_features_rand_v2 (fusion.py:1539-1606) ------------------------------------
@pytest.mark.parametrize("shape,k", [((96, 128), (15, 15)), ((480, 640), (15, 15)), ((33, 17), (2, 2)), ((20, 20), (3, 5)),
                                     ((9, 9), (15, 15)), ((5, 7), (1, 1))])
def test_erode_matches_oracle(dev, shape, k):
    from d3fields_amd.fusion import erode
    from oracle import np_pcd
    rng = np.random.default_rng(shape[0] * 31 + k[0])
    img = ((rng.random(shape) < 0.93) * 255).astype(np.uint8)
    img[: shape[0] // 3] = rng.integers(0, 256, size=(shape[0] // 3, shape[1]), dtype=np.uint8)   # grey values: true minimum
    kern = np.ones(list(k), np.uint8)
    assert np.array_equal(erode(img, kern, iterations=1), np_pcd.erode_cv2(img, kern))


def test_erode_matches_reference_fixture(dev):
    from d3fields_amd.fusion import erode
    g = load_golden("select_v2")
    m = ((g["in_mask"][0, :, :, 1] > 0) * 255).astype(np.uint8)
    assert np.array_equal(erode(m, np.ones([15, 15], np.uint8)), g["eroded_v0_i1"])


def test_fps_pixels_matches_oracle(dev):
    from d3fields_amd import pcd_utils
    from oracle import np_pcd
    rng = np.random.default_rng(9)
    mask = rng.random((480, 640)) < 0.3
    pix = np.array(mask.nonzero()).T                                                    # ~92 k pixels, many equal distances
    sel, idx, md = pcd_utils.fps_pixels(pix, 40, init_idx=123)
    wsel, widx, wmd = np_pcd.fps_int(pix, 40, 123)
    assert idx == widx and np.array_equal(sel, wsel) and md == wmd
    line = np.stack([np.zeros(11, np.int64), np.arange(11)], 1)                         # exact ties: first maximum wins
    assert pcd_utils.fps_pixels(line, 3, init_idx=5)[1] == np_pcd.fps_int(line, 3, 5)[1] == [5, 0, 10]
    # more samples than points (a small instance after the 15x15 erosion): fps_np does NOT stop at n, it keeps appending
    # (every distance is 0 by then: argmax = index 0) and returns particle_num points -- so does the kernel
    sel, idx, md = pcd_utils.fps_pixels(line, 15, init_idx=5)
    wsel, widx, wmd = np_pcd.fps_int(line, 15, 5)
    assert len(idx) == 15 and idx == widx and idx[11:] == [0, 0, 0, 0] and np.array_equal(sel, wsel) and md == wmd == 0.0


def test_masked_pixel_fps_pipeline(dev):
    """gate -> erode -> row-major nonzero -> pixel FPS on the device == the reference's numpy sequence (fusion.py:1554-1568),
    including an instance that erodes to fewer pixels than are sampled from it."""
    import torch
    from d3fields_amd import pcd_utils
    from oracle import np_pcd
    rng = np.random.default_rng(4)
    H, W = 120, 160
    depth = rng.uniform(0.4, 1.4, (H, W)).astype(np.float32)
    depth[rng.random((H, W)) < 0.0005] = 0.0              # holes: the gate drops them, the erosion widens them
    depth[22:30, 100:118] = 1.6                             # beyond the 1.5 m gate
    onehot = np.zeros((H, W, 3), np.float32)
    onehot[20:90, 30:120, 1] = 1.0                      # a big instance
    onehot[95:112, 10:28, 2] = 1.0                      # a small one: 17 x 18 pixels, 3 x 4 after the 15 x 15 erosion at best
    depth[95:112, 10:28] = 0.8
    m_dev, d_dev = torch.from_numpy(onehot).to(dev), torch.from_numpy(depth).to(dev)
    for ch, k, start in ((1, 50, 7), (2, 25, 0)):
        gate = onehot[:, :, ch].astype(bool) & (depth > 0.0) & (depth < 1.5)
        er = np_pcd.erode_cv2((gate * 255).astype(np.uint8), np.ones([15, 15], np.uint8))
        pix = np.array(er.nonzero()).T
        assert 0 < pix.shape[0] and (ch == 1 or pix.shape[0] < k)
        wsel, widx, _ = np_pcd.fps_int(pix, k, start)
        sel, z = pcd_utils.masked_pixel_fps(m_dev[:, :, ch], d_dev, k, init_idx=start)
        assert sel.shape == (k, 2) and np.array_equal(sel, wsel)
        assert z.dtype == np.float32 and np.array_equal(z, depth[wsel[:, 0], wsel[:, 1]])
    with pytest.raises(AssertionError):
        pcd_utils.masked_pixel_fps(m_dev[:, :, 0], d_dev, 5)                            # empty mask: fps_np asserts
    # the mask in Fusion.dtype = float16 (the fp16 storage mode), or a caller's bool / uint8 one-hot: same pixels
    wsel = pcd_utils.masked_pixel_fps(m_dev[:, :, 1], d_dev, 50, init_idx=7)[0]
    for other in (m_dev.half(), m_dev.bool(), (m_dev * 255).to(torch.uint8)):
        assert np.array_equal(pcd_utils.masked_pixel_fps(other[:, :, 1], d_dev, 50, init_idx=7)[0], wsel), other.dtype


def _v2_fusion(dev):
    from d3fields_amd import Fusion
    g = load_golden("select_v2")
    V, H, W = g["depth"].shape
    f = Fusion(num_cam=V, device=str(dev), mask_producer=lambda fusion, q, t, b, **kw: {
        "mask": g["in_mask"], "consensus_mask_label": ["background", "mug", "box"]})
    f.mu = float(g["mu"])
    f.update({"color": np.zeros((V, H, W, 3), np.uint8), "depth": g["depth"], "pose": g["pose"], "K": g["K"],
              "dino_feats": g["in_dino_feats"]})
    f.text_queries_for_inst_mask_no_track(["mug", "box"], [0.3, 0.3], None)
    return g, f


def test_select_features_rand_v2_matches_reference(dev):
    g, f = _v2_fusion(dev)
    np.random.seed(int(g["seed"]))
    feats_l, pts_l, imgs = f.select_features_rand_v2(None, int(g["N"]), per_instance=True)
    assert len(pts_l) == int(g["n_inst"]) == len(feats_l) and imgs == []
    for i in range(len(pts_l)):
        assert pts_l[i].dtype == np.float64 and np.array_equal(pts_l[i], g["pts_%d" % i]), i
        assert rel_err(cpu(feats_l[i]), g["feats_%d" % i]) <= TOL


def test_select_features_rand_v2_with_fp16_stored_maps(dev):
    """Fusion(dtype=float16) stores curr_obs_torch['mask'] in half: the pixel pipeline converts instead of asserting float32
    (ADVICE r3); the keypoints are those of the float32 run (pixels, depths and the float64 lift do not depend on the map
    format), the descriptors those of the fp16-stored map."""
    import warnings
    from d3fields_amd import Fusion
    g = load_golden("select_v2")
    V, H, W = g["depth"].shape
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        f = Fusion(num_cam=V, device=str(dev), dtype=torch.float16, mask_producer=lambda fusion, q, t, b, **kw: {
            "mask": g["in_mask"], "consensus_mask_label": ["background", "mug", "box"]})
    f.mu = float(g["mu"])
    f.update({"color": np.zeros((V, H, W, 3), np.uint8), "depth": g["depth"], "pose": g["pose"], "K": g["K"],
              "dino_feats": g["in_dino_feats"]})
    f.text_queries_for_inst_mask_no_track(["mug", "box"], [0.3, 0.3], None)
    assert f.curr_obs_torch["mask"].dtype == torch.float16
    np.random.seed(int(g["seed"]))
    feats_l, pts_l, imgs = f.select_features_rand_v2(None, int(g["N"]), per_instance=True)
    assert len(pts_l) == int(g["n_inst"])
    for i in range(len(pts_l)):
        assert np.array_equal(pts_l[i], g["pts_%d" % i]), i
        assert rel_err(cpu(feats_l[i]), g["feats_%d" % i]) <= 2e-3          # half-precision storage of the descriptors


# ---- text_queries_* state contract (fusion.py:1112-1256) -------------------------------------------------------------
def test_text_queries_no_track_state_contract(dev):
    g, f = _v2_fusion(dev)
    obs = f.curr_obs_torch
    V, H, W = g["depth"].shape
    assert obs["consensus_mask_label"] == ["background", "mug", "box"] and f.get_inst_num() == 3
    assert isinstance(obs["mask_label"], list) and len(obs["mask_label"]) == V and obs["mask_label"][0] == ["background", "mug", "box"]
    assert all(isinstance(x, str) for x in obs["mask_label"][0])
    assert obs["semantic_label"] == ["background", "mug", "box"] and len(obs["mask_conf"]) == V
    assert obs["mask"].shape == (V, H, W, 3) and obs["mask"].dtype == torch.float32 and obs["mask"].is_cuda
    assert np.array_equal(cpu(obs["mask"]), g["in_mask"])
    # an instance that is visible in NO view keeps its channel (the one-hot is sized by the label list, fusion.py:1171)
    lab = g["in_mask"].argmax(-1).astype(np.uint8)
    f.mask_producer = lambda fusion, q, t, b, **kw: {
        "mask": lab, "consensus_mask_label": ["background", "mug", "box", "spoon"],
        "mask_label": [["background", "mug", "mug", "box"]] * V, "mask_conf": [[1.0, 0.9, 0.8, 0.7]] * V}
    f.text_queries_for_inst_mask_no_track(["mug", "box", "spoon"], [0.3] * 3, None, expected_labels=["background", "mug", "box", "spoon"])
    assert f.curr_obs_torch["mask"].shape == (V, H, W, 4) and float(f.curr_obs_torch["mask"][..., 3].sum()) == 0.0
    assert f.curr_obs_torch["semantic_label"] == ["background", "mug", "box"]          # first-occurrence order of view 0
    assert f.curr_obs_torch["mask_conf"][0] == [1.0, 0.9, 0.8, 0.7]
    out = f.eval(torch.zeros(5, 3, device=dev), return_names=["mask"])
    assert out["mask"].shape == (5, 4)
    # contract violations are reported, not guessed around
    f.mask_producer = lambda fusion, q, t, b, **kw: lab
    with pytest.raises(TypeError):
        f.text_queries_for_inst_mask_no_track(["mug"], [0.3], None)
    f.mask_producer = lambda fusion, q, t, b, **kw: {"mask": lab, "consensus_mask_label": ["background"]}
    with pytest.raises(ValueError):
        f.text_queries_for_inst_mask_no_track(["mug"], [0.3], None)


def test_onehot_helpers_accept_cpu_tensors(dev):
    """The reference's helpers take CPU tensors (fusion.py:90-116); here they are converted on the device and
    returned on the caller's device."""
    from d3fields_amd import instance2onehot, onehot2instance
    g = load_golden("onehot")
    inst = torch.from_numpy(g["inst"])
    oh = instance2onehot(inst, int(g["NI"]))
    assert not oh.is_cuda and oh.dtype == torch.bool and np.array_equal(oh.numpy(), g["onehot"])
    back = onehot2instance(torch.from_numpy(g["soft"]))
    assert not back.is_cuda and back.dtype == torch.uint8 and np.array_equal(back.numpy(), g["soft_inst"])
    assert instance2onehot(inst.to(dev), int(g["NI"])).is_cuda


def test_driver_sequence_segment_select_track(dev):
    """The call sequence of the reference's drivers with stub producers (vis_repr.py:81-103, vis_tracking.py:86-131):
    update -> text_queries_for_inst_mask_no_track -> select_features_rand, then per frame
    update -> text_queries_for_inst_mask -> rigid_tracking; keypoints / descriptors / tracked points against the
    goldens the reference itself produced ('select_features', 'rigid_tracking')."""
    from d3fields_amd import Fusion
    gs, gt = load_golden("select_features"), load_golden("rigid_tracking")
    V, H, W = gs["depth"].shape
    labels = ["background", "a", "b", "c"]
    calls = {"producer": 0, "tracker_init": 0, "tracker_step": 0}

    def producer(fusion, queries, thresholds, boundaries, merge_all=False, expected_labels=None, robot_pcd=None):
        calls["producer"] += 1
        assert fusion.curr_obs_torch["color"].shape == (V, H, W, 3)
        return {"mask": torch.from_numpy(gs["in_mask"].argmax(-1).astype(np.uint8)), "consensus_mask_label": labels}

    def tracker(fusion, color, mask):
        calls["tracker_init" if mask is not None else "tracker_step"] += 1
        if mask is not None:
            assert mask.shape == (V, H, W) and mask.dtype == torch.uint8
        return torch.from_numpy(gs["in_mask"])                                          # (V,H,W,NI) one-hot, like xmem_process

    f = Fusion(num_cam=V, device=str(dev), mask_producer=producer, mask_tracker=tracker)
    f.mu = float(gs["mu"])
    box = dict(zip(["x_lower", "x_upper", "y_lower", "y_upper", "z_lower", "z_upper"], gs["bounds"].tolist()))
    color = np.zeros((V, H, W, 3), np.uint8)
    with pytest.raises(RuntimeError):
        f.text_queries_for_inst_mask(labels[1:], [0.3] * 3, box)                       # 'Please call update() first!'
    f.update({"color": color, "depth": gs["depth"], "pose": gs["pose"], "K": gs["K"], "dino_feats": gs["in_dino_feats"]})
    f.text_queries_for_inst_mask_no_track(labels[1:], [0.3] * 3, box)
    feats_l, pts_l, _ = f.select_features_rand(box, int(gs["N"]), per_instance=True, res=float(gs["res"]), init_idx=0)
    assert len(pts_l) == int(gs["n_inst"])
    for i in range(len(pts_l)):
        assert np.array_equal(pts_l[i], gs["sel_pts_%d" % i]) and rel_err(cpu(feats_l[i]), gs["sel_feats_%d" % i]) <= TOL
    # tracking frames on the scene of the 'rigid_tracking' golden (same cameras and image size)
    n = int(gt["n"])
    info = {"a": {"src_feats": gt["src_feats"][:n]}, "b": {"src_feats": gt["src_feats"][n:]}}
    for frame in range(3):
        f.update({"color": color, "depth": gt["depth"], "pose": gt["pose"], "K": gt["K"], "dino_feats": gt["in_dino_feats"]})
        f.text_queries_for_inst_mask(labels[1:], [0.3] * 3, box)
        assert f.curr_obs_torch["mask"].shape == (V, H, W, 4) and f.xmem_first_mask_loaded and f.track_ids == [0, 1, 2, 3]
        res = f.rigid_tracking(info, [p for p in gt["last_pts"]], box, n)
        assert np.abs(np.stack(res["match_pts_list"]) - gt["match_pts"]).max() <= 1e-5, frame
    assert calls == {"producer": 2, "tracker_init": 1, "tracker_step": 2}
    with pytest.raises(NotImplementedError):                                            # fusion.py:1240-1241
        f.text_queries_for_inst_mask(labels[1:], [0.3] * 3, box, use_sam=True)
    f.clear_xmem_memory()
    f.text_queries_for_inst_mask(labels[1:], [0.3] * 3, box)
    assert calls["producer"] == 3 and calls["tracker_init"] == 2


# ---- k nearest descriptors (the north star's "KNN correspondence lookup"; extension of argmax(0)) ---------------------
def _knn_reference(dist, k):
    """numpy: per column the k smallest distances, ties -> lower row, NaN last"""
    d = np.where(np.isnan(dist), np.inf, dist)
    rows = np.arange(d.shape[0])
    idx = np.stack([np.lexsort((rows, d[:, c]))[:k] for c in range(d.shape[1])], axis=1)
    return idx


@pytest.mark.parametrize("B1,B2,C,k", [(700, 37, 24, 5), (100000, 300, 384, 8), (5, 3, 16, 8), (257, 65, 8, 1), (70000, 2, 32, 3)])
def test_knn_descriptors_matches_numpy(dev, B1, B2, C, k):
    from d3fields_amd import corr_utils as cu
    g = torch.Generator().manual_seed(B1 + k)
    src = torch.randn(B1, C, generator=g)
    tgt = torch.randn(B2, C, generator=g)
    src[B1 // 2] = src[0]                                      # exact duplicate rows: ties broken by the lower index
    tgt[0] = src[0]
    sim, idx, val = cu.knn_descriptors(src.to(dev), tgt.to(dev), k, scale=1.3)
    full = cu.compute_similarity_tensor_multi(src.to(dev), tgt.to(dev), None, None, 1.3)
    assert torch.equal(sim, full)                              # the similarity matrix is the reference function's
    dist = cpu(cu._pairwise(src.to(dev), tgt.to(dev), 1.0, "l2", 0, False)[0])    # D3F_SIM_DIST of the same kernel
    want = _knn_reference(dist, k)
    got = cpu(idx)
    kk = min(k, B1)
    assert np.array_equal(got[:kk], want[:kk])
    assert (got[kk:] == -1).all() and np.isnan(cpu(val)[kk:]).all()
    assert np.array_equal(cpu(val)[:kk], np.take_along_axis(cpu(sim), want[:kk], axis=0))
    assert np.array_equal(got[0], cpu(cu.nearest_descriptor(src.to(dev), tgt.to(dev), 1.3)[1]))   # k = 1 == the fused argmax
    assert got[0, 0] == 0 and (B1 < 2 or k < 2 or got[1, 0] == B1 // 2)


def test_knn_descriptors_vs_reference_golden(dev):
    """On the reference's own compute_similarity_tensor_multi output (golden 'corr_utils'): torch.topk of the golden
    similarity gives the same neighbours wherever the golden values are distinct."""
    from d3fields_amd import corr_utils as cu
    g = load_golden("corr_utils")
    src, tg = torch.from_numpy(g["multi_src"]).to(dev), torch.from_numpy(g["multi_tgt"]).to(dev)
    sim, idx, val = cu.knn_descriptors(src, tg, 4, float(g["multi_scale"]))
    ref = torch.from_numpy(g["multi_l2"])
    tv, ti = ref.topk(4, dim=0)
    distinct = (tv[:-1] > tv[1:]).all(0) & (tv[-1] > ref.kthvalue(ref.shape[0] - 4, dim=0).values)
    assert distinct.sum() > 30
    assert torch.equal(idx.cpu()[:, distinct], ti[:, distinct])
    assert rel_err(cpu(val), tv.numpy()) <= TOL


def test_knn_with_nan_rows(dev):
    from d3fields_amd import corr_utils as cu
    src = torch.randn(300, 16, generator=torch.Generator().manual_seed(1))
    src[5] = float("nan")
    tgt = torch.randn(7, 16, generator=torch.Generator().manual_seed(2))
    _, idx, _ = cu.knn_descriptors(src.to(dev), tgt.to(dev), 8)
    assert not (cpu(idx) == 5).any()                            # a NaN distance sorts last


# ---- BASELINE config 5 as a sequence -------------------------------------------------------------
def test_config5_thirty_frame_sequence(dev):
    """4 views x 30 frames: every frame refreshes depth + feature + mask maps (update), queries 100 000 keypoints with
    ['dino_feats', 'mask'] and matches them against 300 reference descriptors (softmax similarity + best match + 4-NN).
    Every frame: instance indices / validity / dist of a 1500-point sample bit-exact against the CPU oracle, features
    <= 1e-5, the fused best match equal to argmax(0) of the similarity and to row 0 of the k-NN lookup."""
    from d3fields_amd import Fusion, synth, onehot2instance, corr_utils as cu
    from oracle import c_oracle as O
    V, H, W, C, NI, N = 4, 480, 640, 384, 8, 100000
    f = Fusion(num_cam=V, device=str(dev))
    ref_desc = torch.randn(300, C, generator=torch.Generator().manual_seed(11)).to(dev)
    keypoints = synth.random_cloud(N, seed=5).to(dev)
    pick = torch.randperm(N, generator=torch.Generator().manual_seed(6))[:1500]
    for frame in range(30):
        sc = synth.make_scene(V, H, W, "smooth" if frame % 2 == 0 else "stress", seed=frame)
        feats = synth.random_map(V, 48, 64, C, seed=100 + frame)
        mask = synth.random_onehot_mask(V, H, W, NI, seed=200 + frame)
        f.update({"color": np.zeros((V, H, W, 3), np.uint8), "depth": sc["depth"].numpy(), "pose": sc["pose"].numpy(),
                  "K": sc["K"].numpy(), "dino_feats": feats})
        f.curr_obs_torch["mask"] = mask.to(dev)
        moved = keypoints + 0.001 * frame                          # a new query tensor every frame (no cached order)
        with torch.no_grad():
            out = f.eval(moved, return_names=["dino_feats", "mask"])
            sim, match = cu.nearest_descriptor(out["dino_feats"], ref_desc, 1.0)
            _, knn, _ = cu.knn_descriptors(out["dino_feats"], ref_desc, 4, 1.0)
        if frame % 5 == 0 or frame == 29:
            ref = O.eval_field(sc["depth"], sc["K"], sc["pose"], moved[pick.to(dev)].cpu(), [feats, mask], mu=f.mu)
            assert np.array_equal(cpu(out["valid_mask"][pick.to(dev)]), ref["valid_mask"]), frame
            assert np.array_equal(cpu(out["dist"][pick.to(dev)]), ref["dist"]), frame
            assert rel_err(cpu(out["dino_feats"][pick.to(dev)]), ref["sets"][0]) <= TOL, frame
            assert np.array_equal(cpu(onehot2instance(out["mask"][pick.to(dev)])), O.onehot2instance(ref["sets"][1])), frame
        assert torch.equal(match, sim.argmax(0)) and torch.equal(match, knn[0]), frame
        assert abs(float(sim.sum(0).mean()) - 1.0) < 1e-4


# ---- masked instance clouds (reference fusion.py:1262-1311) and the voxel-grid mean --------------------------------------
def _masked_fusion(dev):
    from d3fields_amd import Fusion
    g = load_golden("masked_pcd")
    V, H, W = g["depth"].shape
    labels = ["background", "mug", "box", "pen"]
    f = Fusion(num_cam=V, device=str(dev), mask_producer=lambda fusion, q, t, b, **kw: {
        "mask": g["in_mask"], "consensus_mask_label": labels,
        "mask_gs": [np.moveaxis(g["in_mask"][v] > 0, -1, 0) for v in range(V)]})
    f.update({"color": g["color"], "depth": g["depth"], "pose": g["pose"], "K": g["K"], "dino_feats": np.zeros((V, 4, 4, 4), np.float32)})
    f.text_queries_for_inst_mask_no_track(labels[1:], [0.3] * 3, None)
    return g, f


def test_extract_masked_pcd_matches_reference(dev):
    """Fusion.extract_masked_pcd / extract_masked_pcd_in_views(downsample=False) / get_query_obj_pcd against the clouds the
    REFERENCE's methods returned (golden 'masked_pcd'; cv2.erode restated): same points in the same order, 1e-12 m (the
    reference's [4,4] @ [4,n] runs through BLAS, the kernel sums in a fixed order)."""
    from d3fields_amd import synth
    g, f = _masked_fusion(dev)
    box = dict(synth.WORK_BOX)
    tight = dict(zip(("x_lower", "x_upper", "y_lower", "y_upper", "z_lower", "z_upper"), g["tight"].tolist()))
    assert f.get_inst_num() == 4
    for key, got in (("pcd_12_box", f.extract_masked_pcd([1, 2], boundaries=box)), ("pcd_3_none", f.extract_masked_pcd([3])),
                     ("pcd_123_tight", f.extract_masked_pcd([1, 2, 3], boundaries=tight)), ("pcd_123_none", f.extract_masked_pcd([1, 2, 3])),
                     ("view2_2_box", f.extract_masked_pcd_in_views([2], [2], box, downsample=False)),
                     ("view0_13_tight", f.extract_masked_pcd_in_views([1, 3], [0], tight, downsample=False))):
        assert got.dtype == np.float64 and got.shape == g[key].shape, (key, got.shape, g[key].shape)
        assert np.abs(got - g[key]).max() <= 1e-12, key
    q = f.get_query_obj_pcd()
    assert np.abs(np.asarray(q.points) - g["pcd_123_none"]).max() <= 1e-12 and np.asarray(q.colors).shape == g["pcd_123_none"].shape
    with pytest.raises(AssertionError):
        f.extract_masked_pcd_in_views([1], [0, 1], box)


def test_voxel_downsample_matches_open3d_definition(dev):
    """d3f_voxel_downsample against open3d's published VoxelDownSample algorithm restated in numpy (oracle/np_pcd.voxel_mean;
    open3d itself is not installable here): same voxels in ascending voxel order, means to 1e-12; deterministic; and the
    reference's default path extract_masked_pcd_in_views(downsample=True) = that filter over the undownsampled cloud."""
    from d3fields_amd import pcd_utils, synth
    from oracle import np_pcd
    rng = np.random.default_rng(3)
    pts = np.concatenate([rng.normal(0.0, 0.05, (20000, 3)), rng.uniform(-0.3, 0.3, (5000, 3)), np.full((7, 3), 0.1234)])
    col = rng.random(pts.shape)
    want_p, want_c = np_pcd.voxel_mean(pts, 0.01, col)
    got_p, got_c = pcd_utils.voxel_downsample(pts, 0.01, col)
    assert got_p.shape == want_p.shape and np.abs(got_p - want_p).max() <= 1e-12 and np.abs(got_c - want_c).max() <= 1e-12
    again_p, again_c = pcd_utils.voxel_downsample(pts, 0.01, col)
    assert np.array_equal(again_p, got_p) and np.array_equal(again_c, got_c)               # exact sums: deterministic
    only = pcd_utils.voxel_downsample(pts[:100], 0.05)
    assert np.abs(only - np_pcd.voxel_mean(pts[:100], 0.05)).max() <= 1e-12
    assert pcd_utils.voxel_downsample(np.zeros((0, 3)), 0.01).shape == (0, 3)
    g, f = _masked_fusion(dev)
    box = dict(synth.WORK_BOX)
    full = f.extract_masked_pcd_in_views([2], [2], box, downsample=False)
    down = f.extract_masked_pcd_in_views([2], [2], box)                                   # the reference's default: downsample=True
    assert 0 < down.shape[0] < full.shape[0] and np.abs(down - np_pcd.voxel_mean(full, 0.01)).max() <= 1e-12
    cloud = pcd_utils.aggr_point_cloud_from_data(g["color"], g["depth"].astype(np.float64), g["K"].astype(np.float64),
                                                 np.concatenate([g["pose"], np.tile(np.array([[[0, 0, 0, 1.0]]]), (4, 1, 1))], 1).astype(np.float64))
    assert len(np.asarray(cloud.points)) > 0 and np.asarray(cloud.colors).shape == np.asarray(cloud.points).shape   # defaults: downsample, o3d-like


# ---- d3f_track_run (all optimiser steps of a frame in one launch) == the same steps as single launches ----------------
@pytest.mark.parametrize("I,n,V,C", [(1, 64, 4, 384), (3, 20, 2, 128), (5, 33, 8, 512), (16, 32, 4, 64), (2, 7, 3, 48)])
def test_track_run_equals_track_steps(dev, I, n, V, C):
    """One launch for `iters` steps (waves wait for each step's step-tagged parameters inside the kernel) leaves the same
    translations, rotations and keypoints, bit for bit, as `iters` launches of d3f_track_step: one to sixteen instances
    (64 / 32 / 16 / 8 update lanes per instance), up to the 512 keypoints the entry point accepts, one or two channel
    vectors per lane, two to eight views; frame after frame through the captured graph."""
    from d3fields_amd import Fusion, rigid, synth
    H, W = 96, 128
    sc = synth.make_scene(V, H, W, "smooth")
    f = Fusion(num_cam=V, device=str(dev))
    f.curr_obs_torch = {k: sc[k].to(dev) for k in ("depth", "K", "pose")}
    f.curr_obs_torch["dino_feats"] = synth.random_map(V, H // 2, W // 2, C, seed=4, device=dev)
    f.H, f.W = H, W
    g = torch.Generator().manual_seed(I * 100 + n)
    last = synth.random_cloud(I * n, seed=I + n).view(I, n, 3).to(dev)
    with torch.no_grad():
        src = f.eval((last.view(-1, 3) + 0.004).contiguous(), return_names=["dino_feats"])["dino_feats"]
        src = src + 0.01 * torch.randn(src.shape, generator=g).to(dev)
    results = {}
    for loop in (False, True):
        tr = rigid.RigidTracker(f, I, n, iters=12, loop_launch=loop)
        assert tr.single and tr.loop == loop
        for frame in range(2):                                   # the second frame replays the captured graph
            cur, loss = tr.run(f, src, last)
        torch.cuda.synchronize()
        results[loop] = (tr.t_params.clone(), tr.log_r.clone(), cur.clone(), float(loss), tr.state[I * 12:I * 13].clone())
    a, b = results[False], results[True]
    assert torch.equal(a[4], b[4]) and a[4].eq(12.0).all()       # Adam's step counters
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert a[0].abs().max() > 0 and torch.isfinite(a[2]).all()
    assert abs(a[3] - b[3]) <= 1e-5 * max(1.0, abs(a[3]))        # the loss terms are float atomics: order varies


def _golden_tracker_inputs(dev):
    g = load_golden("rigid_tracking")
    from d3fields_amd import Fusion
    f = Fusion(num_cam=g["depth"].shape[0], device=str(dev))
    f.mu = float(g["mu"])
    t = lambda a: torch.as_tensor(np.asarray(a)).to(dev, torch.float32)   # noqa: E731
    f.curr_obs_torch = {"depth": t(g["depth"]), "K": t(g["K"]), "pose": t(g["pose"]), "dino_feats": t(g["in_dino_feats"])}
    f.H, f.W = int(g["H"]), int(g["W"])
    n = int(g["n"])
    info = {"a": {"src_feats": torch.from_numpy(g["src_feats"][:n])}, "b": {"src_feats": torch.from_numpy(g["src_feats"][n:])}}
    return g, f, info, n


def test_track_run_falls_back_when_its_wait_gives_up(dev):
    """d3f_track_run's waves wait for one another inside the kernel; when the bounded wait gives up (a device held by other
    work for seconds) the kernel poisons the loss with NaN.  RigidTracker.run must notice, repeat the frame with one launch per
    step and stay on that form (VERDICT r3 / ADVICE r3: until round 4 the caller silently got undefined poses).  The give-up
    is simulated: the first replay's loss words are overwritten with NaN and the kernel's stall sentinel is left in the scratch
    (ABI 5: the sentinel, not the NaN, is what the tracker looks at -- a NaN that came out of the data must NOT switch the
    tracker to the slow form)."""
    from d3fields_amd import rigid, _lib
    g, f, info, n = _golden_tracker_inputs(dev)
    src = torch.cat([info["a"]["src_feats"], info["b"]["src_feats"]]).to(dev)
    last = torch.from_numpy(np.stack([p for p in g["last_pts"]])).to(dev)
    ref_tr = rigid.RigidTracker(f, 2, n, loop_launch=False)
    want, _ = ref_tr.run(f, src, last)
    tr = rigid.RigidTracker(f, 2, n)
    assert tr.loop
    first, _ = tr.run(f, src, last)                              # captures; a healthy frame
    assert tr.loop and tr.loop_fallbacks == 0 and torch.equal(first, want)
    real_replay, hits = tr.graph.replay, []

    word = int(_lib.load().d3f_track_stall_word(2, n))

    def data_nan():                                              # a NaN loss WITHOUT the sentinel: not a stall
        real_replay()
        tr.loss3.fill_(float("nan"))
    tr.graph.replay = data_nan
    got, _ = tr.run(f, src, last)
    assert tr.loop and tr.loop_fallbacks == 0 and torch.equal(got, want)      # no sentinel: the frame stands, the tracker keeps its form

    def poisoned():
        real_replay()
        if not hits:
            hits.append(1)
            tr.loss3.fill_(float("nan"))
            tr.scratch.view(torch.int32)[word] = _lib.TRACK_STALL_SENTINEL
            tr.t_params.fill_(123.0)                             # ... and the poses are garbage
    tr.graph.replay = poisoned
    got, loss = tr.run(f, src, last)
    assert hits and not tr.loop and tr.loop_fallbacks == 1
    assert torch.equal(got, want) and torch.isfinite(loss)
    again, _ = tr.run(f, src, last)                              # later frames: one launch per step, no more checks
    assert torch.equal(again, want) and tr.loop_fallbacks == 1


def test_rigid_tracking_on_a_busy_device(dev):
    """The golden tracking frame while a second stream keeps every CU busy with long kernels (large GEMMs queued ahead of and
    behind the tracker's launch): whether d3f_track_run gets its waves resident in time or the tracker falls back to per-step
    launches, the keypoints are those of the idle device / the reference's own loop."""
    g, f, info, n = _golden_tracker_inputs(dev)
    idle = np.stack(f.rigid_tracking(info, [p for p in g["last_pts"]], None, n)["match_pts_list"])
    assert np.abs(idle - g["match_pts"]).max() <= 1e-5
    side = torch.cuda.Stream(device=dev)
    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    torch.cuda.synchronize()
    for frame in range(3):
        with torch.cuda.stream(side):
            for _ in range(40):                                  # ~40 x 0.5 ms of all-CU work in flight around the tracker
                c = a @ b
        busy = np.stack(f.rigid_tracking(info, [p for p in g["last_pts"]], None, n)["match_pts_list"])
        assert np.abs(busy - g["match_pts"]).max() <= 1e-5, frame
        assert np.array_equal(busy, idle) or not f._tracker.loop   # same kernel, same results -- unless it had to fall back
    side.synchronize()
    del c


def test_track_run_refuses_what_cannot_be_resident(dev):
    import ctypes
    from d3fields_amd import _lib, rigid, Fusion, synth
    lib = _lib.load()
    cap = lib.d3f_track_run_max_keypoints()              # occupancy of the kernel x CUs of THIS device / 2, at most 512
    assert 64 <= cap <= 512 and (torch.cuda.get_device_properties(dev).multi_processor_count < 256 or cap == 512)
    V, H, W, C = 4, 96, 128, 64
    sc = synth.make_scene(V, H, W, "smooth")
    f = Fusion(num_cam=V, device=str(dev))
    f.curr_obs_torch = {k: sc[k].to(dev) for k in ("depth", "K", "pose")}
    f.curr_obs_torch["dino_feats"] = synth.random_map(V, H // 2, W // 2, C, seed=4, device=dev)
    f.H, f.W = H, W
    tr = rigid.RigidTracker(f, 2, 300)                           # 600 keypoints: one launch per step instead
    assert tr.single and not tr.loop
    tr17 = rigid.RigidTracker(f, 17, 8)                          # 17 instances: likewise
    assert tr17.single and not tr17.loop
    views, keep, _ = f._views(dev)
    fm = f.curr_obs_torch["dino_feats"]
    cm = _lib.ChannelMap(fm.data_ptr(), fm.shape[1], fm.shape[2], fm.shape[3], _lib.DTYPE_F32, fm.stride(0), fm.stride(1), fm.stride(2))
    st = tr.state
    state = _lib.TrackState(_lib.ptr(tr.t_params), _lib.ptr(tr.log_r), _lib.ptr(st[:12]), _lib.ptr(st[12:24]), _lib.ptr(st[24:26]),
                            _lib.ptr(tr.pts), _lib.ptr(tr.loss3), _lib.ptr(tr.scratch))
    rc = lib.d3f_track_run(ctypes.byref(views), ctypes.byref(cm), _lib.ptr(tr.last), 2, 300, _lib.ptr(tr.src), 0.02, 100.0, 1.0, 0.01,
                           0.9, 0.999, 1e-8, 5, ctypes.byref(state), _lib.current_stream_handle(dev))
    assert rc == _lib.ERR_BAD_SHAPE and b"resident" in lib.d3f_last_error()
    assert lib.d3f_track_run(ctypes.byref(views), ctypes.byref(cm), _lib.ptr(tr.last), 2, 300, _lib.ptr(tr.src), 0.02, 100.0, 1.0, 0.01,
                             0.9, 0.999, 1e-8, 0, ctypes.byref(state), _lib.current_stream_handle(dev)) == 0     # no steps: nothing to do


# ---- multi-view instance association (fusion.py:801-1098) -------------------------------------------------------------------
def _align_fusion(dev, g, producer=None):
    from d3fields_amd import Fusion
    V, H, W = int(g["V"]), int(g["H"]), int(g["W"])
    f = Fusion(num_cam=V, device=str(dev), mask_producer=producer)
    f.update({"color": np.zeros((V, H, W, 3), np.uint8), "depth": g["depth"], "pose": g["pose"], "K": g["K"],
              "dino_feats": np.zeros((V, 4, 4, 4), np.float32)})
    return f


@pytest.mark.parametrize("case", ALIGN_CASES)
def test_align_instance_mask_v3_matches_reference(dev, case):
    """Fusion.align_instance_mask_v3 and its stages against what the REFERENCE's methods produced on the same per-view detections
    (goldens align_v3_*: cv2.erode and open3d's voxel_down_sample restated): the instances after the merges and after the filter
    -- labels, voxel sets, raw lengths, per-voxel confidence lists, view -> detection maps -- the consensus labels and the
    (V,H,W) uint8 label images, all exact."""
    from d3fields_amd import association, synth
    g = load_golden(case)
    V = int(g["V"])
    gs, labels, confs = align_inputs(g)
    box = dict(zip(("x_lower", "x_upper", "y_lower", "y_upper", "z_lower", "z_upper"), g["bounds"].tolist()))
    assert box == dict(synth.WORK_BOX)
    queries = [str(q) for q in g["queries"]]
    f = _align_fusion(dev, g)
    f.curr_obs_torch.update(mask_gs=gs, mask_label=labels, mask_conf=confs)
    association.prepare_grid(f, box)
    assert f.voxel_num.dtype == np.int32 and f.iou_threshold == 0.005
    instances = []
    for v in range(V):
        instances = f.merge_instances_from_new_view_vox_ver(instances, v, box)
        assert len(instances) == int(g["n_after_view_%d" % v])
    assert_instances_match(g, "merged", instances, V)
    instances = f.filter_instances_vox_ver(instances)
    assert_instances_match(g, "filtered", instances, V)
    # the whole call, on a fresh object
    f = _align_fusion(dev, g)
    f.curr_obs_torch.update(mask_gs=gs, mask_label=labels, mask_conf=confs)
    f.align_instance_mask_v3(queries, box)
    assert f.curr_obs_torch["consensus_mask_label"] == [str(x) for x in g["consensus_mask_label"]]
    m = f.curr_obs_torch["mask"]
    assert m.dtype == torch.uint8 and m.is_cuda and np.array_equal(cpu(m), g["mask"])
    # ... and through text_queries_for_inst_mask_no_track with a producer that returns what Grounded-SAM returns per view
    f = _align_fusion(dev, g, producer=lambda fusion, q, t, b, **kw: {"mask_gs": gs, "mask_label": labels, "mask_conf": confs})
    f.text_queries_for_inst_mask_no_track(queries, [0.3] * len(queries), box)
    NI = len(g["consensus_mask_label"])
    assert f.get_inst_num() == NI and f.curr_obs_torch["mask"].shape == (V, int(g["H"]), int(g["W"]), NI)
    assert np.array_equal(cpu(f.curr_obs_torch["mask"]).argmax(-1).astype(np.uint8), g["mask"])
    assert float(f.curr_obs_torch["mask"].sum()) == V * int(g["H"]) * int(g["W"])
    first = labels[0]
    assert f.curr_obs_torch["semantic_label"] == [x for k, x in enumerate(first) if x not in first[:k]]
    with pytest.raises(ValueError):
        _align_fusion(dev, g, producer=lambda fusion, q, t, b, **kw: {"mask_gs": gs[:-1], "mask_label": labels, "mask_conf": confs}) \
            .text_queries_for_inst_mask_no_track(queries, [0.3], box)
    # a producer that runs on the GPU hands DEVICE tensors over (np.asarray() of those raised): same consensus, same image
    gs_dev = [torch.as_tensor(np.asarray(x)).to(dev) for x in gs]
    f = _align_fusion(dev, g, producer=lambda fusion, q, t, b, **kw: {"mask_gs": gs_dev, "mask_label": labels, "mask_conf": confs})
    f.text_queries_for_inst_mask_no_track(queries, [0.3] * len(queries), box)
    assert f.curr_obs_torch["consensus_mask_label"] == [str(x) for x in g["consensus_mask_label"]]
    assert np.array_equal(cpu(f.curr_obs_torch["mask"]).argmax(-1).astype(np.uint8), g["mask"])
    # the tracking entry point: first frame = the same association, then the injected tracker (here: it returns what it was given)
    f = _align_fusion(dev, g, producer=lambda fusion, q, t, b, **kw: {"mask_gs": gs, "mask_label": labels, "mask_conf": confs})
    seen = []
    f.mask_tracker = lambda fusion, color, mask: (seen.append(None if mask is None else mask.clone()), f.curr_obs_torch["mask"] if mask is None else mask)[1]
    f.text_queries_for_inst_mask(queries, [0.3] * len(queries), box)
    assert f.xmem_first_mask_loaded and f.track_ids == list(range(NI))
    assert seen[0].dtype == torch.uint8 and np.array_equal(cpu(seen[0]), g["mask"])
    assert np.array_equal(cpu(f.curr_obs_torch["mask"]).argmax(-1).astype(np.uint8), g["mask"])


@pytest.mark.parametrize("seed,V,H,W", [(1, 4, 120, 160), (2, 5, 96, 128), (4, 3, 240, 320), (6, 4, 150, 200), (9, 2, 120, 160)])
def test_align_instance_mask_v3_seeded_against_oracle(dev, seed, V, H, W):
    """The same against the CPU restatement (oracle/np_assoc.py, pinned to the reference by the goldens above) on other seeds,
    view counts and image sizes."""
    from d3fields_amd import Fusion, synth
    from oracle import np_assoc
    sc = synth.make_scene(V, H, W, "smooth")
    K, pose, depth = sc["K"].numpy(), sc["pose"].numpy(), sc["depth"].numpy()
    gs, labels, confs = synth.multiview_segmentation(K, pose, depth, seed=seed)
    box = dict(synth.WORK_BOX)
    bounds = [box[k] for k in ("x_lower", "x_upper", "y_lower", "y_upper", "z_lower", "z_upper")]
    queries = ["pen", "mug", "box"]
    try:
        want_img, want_labels = np_assoc.align(depth, K, pose, [g.copy() for g in gs], labels, confs, queries, bounds)
    except (ZeroDivisionError, IndexError) as e:          # detections the reference itself cannot digest: the same exception here
        want_img, want_labels = type(e), None
    f = Fusion(num_cam=V, device=str(dev))
    f.update({"color": np.zeros((V, H, W, 3), np.uint8), "depth": depth, "pose": pose, "K": K, "dino_feats": np.zeros((V, 4, 4, 4), np.float32)})
    f.curr_obs_torch.update(mask_gs=gs, mask_label=labels, mask_conf=confs)
    if want_labels is None:
        with pytest.raises(want_img):
            f.align_instance_mask_v3(queries, box)
        return
    f.align_instance_mask_v3(queries, box)
    assert f.curr_obs_torch["consensus_mask_label"] == want_labels
    assert np.array_equal(cpu(f.curr_obs_torch["mask"]), want_img)


def test_compose_labels_kernel(dev):
    """d3f_compose_labels: the largest assigned instance index covering a pixel, 0 where none; unassigned detections ignored;
    indices above 255 wrap like numpy's uint8 assignment; pixel counts that are not a multiple of four."""
    import ctypes
    from d3fields_amd import _lib
    lib = _lib.load()
    r = np.random.default_rng(5)
    for n_dets, n_pix in ((1, 7), (9, 1001), (40, 4099), (300, 513)):
        dets = (r.random((n_dets, n_pix)) < 0.2).astype(np.uint8) * r.integers(1, 256, (n_dets, n_pix)).astype(np.uint8)
        owner = r.permutation(n_dets).astype(np.int32)
        owner[r.random(n_dets) < 0.3] = -1
        want = np.zeros(n_pix, np.uint8)
        for k in np.argsort(owner, kind="stable"):
            if owner[k] >= 0:
                want[dets[k] != 0] = np.uint8(owner[k] & 255)
        d, o = torch.from_numpy(dets).to(dev), torch.from_numpy(owner).to(dev)
        out = torch.full((n_pix,), 77, dtype=torch.uint8, device=dev)
        _lib.check(lib.d3f_compose_labels(_lib.ptr(d), n_dets, n_pix, _lib.ptr(o), _lib.ptr(out), _lib.current_stream_handle(dev)))
        assert np.array_equal(cpu(out), want), (n_dets, n_pix)
    assert lib.d3f_compose_labels(None, 0, 0, None, None, None) == 0
    assert lib.d3f_compose_labels(None, 3, 8, None, ctypes.c_void_p(64), None) == _lib.ERR_INVALID_ARG


def test_repr_example_runs(dev, capsys):
    """examples/repr_synthetic.py: vis_repr.py's flow with the association running here; every surface point gets an instance."""
    import importlib.util
    import os
    import sys
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "repr_synthetic.py")
    spec = importlib.util.spec_from_file_location("repr_synthetic", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv, sys.argv = sys.argv, ["repr_synthetic.py", "--step", "0.008"]
    try:
        f, out = mod.main()
    finally:
        sys.argv = argv
    text = capsys.readouterr().out
    assert "association:" in text and "feature query:" in text
    NI = f.get_inst_num()
    assert NI >= 4 and f.curr_obs_torch["consensus_mask_label"][0] == "background"
    assert out["mask"].shape[1] == NI and out["dino_feats"].shape[1] == 384 and out["color_tensor"].shape[1] == 3
    assert bool(out["valid_mask"].all())
    # close() (fusion.py:1704-1712) drops the observation and every buffer kept between calls; the object can be fed again
    f.close()
    assert f.curr_obs_torch == {} and f.mask_producer is None
    with pytest.raises(RuntimeError):
        f.batch_eval(torch.zeros(4, 3, device=dev), return_names=[])
