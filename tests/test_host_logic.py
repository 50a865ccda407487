"""CPU: host-side mirror of the reference interface (no GPU, no compute calls)."""
import numpy as np
import pytest
import torch

from conftest import ALIGN_CASES, align_inputs, assert_instances_match, load_golden
from d3fields_amd import Fusion, corr_utils, create_init_grid, instance2onehot, onehot2instance, sharding, synth


@pytest.mark.parametrize("tag", ["004", "020"])
def test_create_init_grid_matches_reference(tag):
    g = load_golden("init_grid_" + tag)
    b = dict(zip(["x_lower", "x_upper", "y_lower", "y_upper", "z_lower", "z_upper"], g["bounds"].tolist()))
    coords, shape = create_init_grid(b, float(g["step"]))
    assert tuple(shape) == tuple(g["shape"]) and coords.shape == (int(g["n"]), 3) and coords.dtype == torch.float32
    c = coords.numpy()
    assert np.array_equal(c[:128], g["head"]) and np.array_equal(c[-128:], g["tail"])
    assert np.array_equal(c[::1009], g["sub"])
    assert np.allclose(c.astype(np.float64).sum(0), g["colsum"], rtol=0, atol=1e-6)
    if tag == "004":            # vis_repr.py:88 -> 200 x 175 x 55 = 1 925 000 voxels
        assert tuple(shape) == (200, 175, 55)


def test_onehot_numpy_paths_match_reference():
    g = load_golden("onehot")
    assert np.array_equal(instance2onehot(g["inst"], int(g["NI"])), g["onehot"])
    assert np.array_equal(instance2onehot(g["inst"]), g["onehot"][..., :int(g["inst"].max()) + 1])
    assert np.array_equal(onehot2instance(g["soft"]), g["soft_inst"])
    with pytest.raises(NotImplementedError):
        onehot2instance([1, 2, 3])


def test_no_cpu_fallback_anywhere():
    """The product path must fail loudly on CPU tensors instead of computing somewhere else."""
    f = Fusion(num_cam=2)
    with pytest.raises(RuntimeError, match="update"):
        f.eval(torch.zeros(3, 3))
    sc = synth.make_scene(2, 16, 20, "stress")
    f.curr_obs_torch = dict(sc)
    f.H, f.W = 16, 20
    with pytest.raises(RuntimeError, match="no CPU path"):
        f.eval(torch.zeros(3, 3), return_names=[])
    with pytest.raises(RuntimeError, match="no CPU path"):
        f.eval_dist(torch.zeros(3, 3))
    with pytest.raises(RuntimeError, match="no CPU path"):
        f.batch_eval(torch.zeros(3, 3))
    with pytest.raises(AssertionError):
        f.eval(np.zeros((3, 3), np.float32))
    with pytest.raises(AssertionError):
        f.eval(torch.zeros(3, 4))
    with pytest.raises(AssertionError):
        f.eval(torch.zeros(3))
    with pytest.raises(RuntimeError, match="no CPU path"):
        corr_utils.compute_dist_tensor(torch.zeros(2, 4), torch.zeros(4))
    with pytest.raises(RuntimeError, match="no CPU path"):
        corr_utils.compute_similarity_tensor_multi(torch.zeros(2, 4), torch.zeros(3, 4), None, None, 1.0)
    with pytest.raises(AssertionError):
        corr_utils.compute_similarity_tensor_multi(torch.zeros(2, 4), torch.zeros(3, 5), None, None, 1.0)
    with pytest.raises(RuntimeError, match="no CPU path"):
        onehot2instance(torch.zeros(4, 3))
    assert Fusion(num_cam=2, dtype=torch.float16).dtype == torch.float16      # fp16 = storage format of the maps
    with pytest.raises(NotImplementedError):
        Fusion(num_cam=2, dtype=torch.bfloat16)


def test_product_never_imports_the_oracle():
    import os
    import re
    from conftest import ROOT
    pkg = os.path.join(ROOT, "d3fields_amd")
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dp, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), fn
                assert "libd3f_oracle" not in text and "c_oracle" not in text and "torch_port" not in text, fn
                assert not re.search(r"#include\s*[<\"][^>\"]*oracle", text), fn


def test_update_keeps_reference_state_layout():
    calls = {}

    def extractor(color, params):
        calls["params"] = params
        return torch.ones(color.shape[0], params["patch_h"], params["patch_w"], 6)

    f = Fusion(num_cam=3, device="cpu", feature_extractor=extractor)    # host state only; no query is made
    V, H, W = 3, 40, 60
    obs = {"color": np.full((V, H, W, 3), 255, np.uint8), "depth": np.ones((V, H, W), np.float64),
           "pose": np.zeros((V, 3, 4)), "K": np.zeros((V, 3, 3))}
    f.update(obs)
    assert calls["params"] == {"patch_h": 4, "patch_w": 6}          # H//10, W//10 (fusion.py:694-697)
    assert (f.H, f.W, f.num_cam) == (H, W, V)
    o = f.curr_obs_torch
    assert o["dino_feats"].shape == (V, 4, 6, 6) and o["color_tensor"].shape == (V, H, W, 3)
    assert float(o["color_tensor"].max()) == 1.0 and o["depth"].dtype == torch.float32
    assert o["pose"].shape == (V, 3, 4) and o["K"].shape == (V, 3, 3) and o["color"] is obs["color"]


def test_shard_bounds_cover_everything_once():
    for n in [0, 1, 7, 8, 9, 130001]:
        for world in [1, 2, 3, 8]:
            spans = [sharding.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_bench_roofline_numerator_matches_survey(tmp_path):
    """bench.py's algorithmic bytes are SURVEY.md §8d's: N*(12+4+1+4*sum C) + maps read once + 84*V."""
    import bench
    b, per = bench.algorithmic_bytes(bench.WORKLOADS["c2_dense"], 985600)
    assert per == 1553 and b == 985600 * 1553 + 4 * (4 * 480 * 640 + 4 * 480 * 640 * 384) + 84 * 4
    b, per = bench.algorithmic_bytes(bench.WORKLOADS["c3_patch"], 1925000)
    assert per == 1585 and b == 1925000 * 1585 + 4 * (4 * 480 * 640 + 4 * 48 * 64 * 384 + 4 * 480 * 640 * 8) + 336
    assert bench.algorithmic_bytes(bench.WORKLOADS["c4_patch"], 1000000)[1] == 4113
    # the reference's own shape: 1024-d patch maps + 8-instance mask + colours; the distance-only pass
    b, per = bench.algorithmic_bytes(bench.WORKLOADS["ref_patch"], 1925000)
    assert per == 17 + 4 * (1024 + 8 + 3) and b == 1925000 * per + 4 * (4 * 480 * 640 + 4 * 48 * 64 * 1024 + 4 * 480 * 640 * 11) + 336
    assert bench.algorithmic_bytes(bench.WORKLOADS["dist_only"], 123200000) == (123200000 * 17 + 4 * 4 * 480 * 640 + 336, 17)
    # counter traffic is keyed by workload AND point set AND point count AND the fingerprint of the sources the profiled library was
    # built from: the lattice kernel's figure is never printed for a cloud, and a kernel change without a re-profile prints nothing
    import json
    from d3fields_amd import build
    path = str(tmp_path / "traffic.json")
    fp = build.source_fingerprint()
    json.dump({"c2_dense": {"traffic_bytes": 123, "points": 985600, "source": "x", "source_fingerprint": fp},
               "c2_patch_surface": {"traffic_bytes": 5, "points": 70000, "source": "y", "source_fingerprint": fp},
               "c3_dense": {"traffic_bytes": 456, "points": 1925000, "source": "z", "source_fingerprint": "0" * 64},
               "c3_patch": {"traffic_bytes": 789, "points": 1925000, "source": "w"}}, open(path, "w"))
    assert bench.measured_traffic("c2_dense", "grid", 985600, path)[0] == 123
    assert bench.measured_traffic("c2_dense", "grid", 1000, path)[0] is None
    assert bench.measured_traffic("c2_dense", "random", 985600, path)[0] is None
    assert bench.measured_traffic("c2_patch", "surface", 70000, path)[0] == 5
    assert bench.measured_traffic("c3_dense", "grid", 1925000, path)[0] is None          # stamped with other sources
    assert bench.measured_traffic("c3_patch", "grid", 1925000, path)[0] is None          # not stamped at all
    assert bench.measured_traffic("nope", "grid", 1, path) == (None, None, None)


def test_rigid_helpers_match_restated_pytorch3d():
    """d3fields_amd.rigid (product) vs oracle/pytorch3d_restated.py (what the reference's loop ran with) on CPU tensors:
    the exponential map and the row-vector rigid transform are plain torch ops, so they can be checked without a GPU."""
    import torch
    from scipy.spatial.transform import Rotation
    from d3fields_amd import rigid
    from oracle import pytorch3d_restated as p3d
    g = torch.Generator().manual_seed(5)
    w = torch.cat([torch.randn(6, 3, generator=g) * 0.7, torch.zeros(1, 3), torch.tensor([[1e-3, -2e-3, 0.0]])])
    R = rigid.so3_exp_map(w)
    assert torch.allclose(R, p3d.so3_exp_map(w), atol=1e-7)
    assert np.abs(R.numpy() - Rotation.from_rotvec(w.numpy()).as_matrix()).max() <= 1e-6
    x = torch.randn(8, 11, 3, generator=g)
    t = torch.randn(8, 3, generator=g)
    ref = p3d.Transform3d().rotate(R).translate(t).transform_points(x)
    assert torch.allclose(rigid.rigid_transform(x, R, t), ref, atol=1e-6)
    # gradients flow to both parameter sets (the tracker optimises them)
    w.requires_grad_(True)
    t.requires_grad_(True)
    rigid.rigid_transform(x, rigid.so3_exp_map(w), t).sum().backward()
    assert torch.isfinite(w.grad).all() and torch.isfinite(t.grad).all() and float(t.grad.abs().sum()) > 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No libd3fields_hip.so and no hipcc: constructing the product raises -- there is nothing to fall back to."""
    from d3fields_amd import _lib, build
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(build, "LIB_PATH", str(tmp_path / "libd3fields_hip.so"))
    monkeypatch.setattr(build, "_hipcc", lambda: str(tmp_path / "no_such_hipcc"))
    with pytest.raises((OSError, RuntimeError, ImportError)):
        Fusion(num_cam=2)
    monkeypatch.setattr(_lib, "_lib", None)      # the real library loads again afterwards (monkeypatch restores the paths)


def test_stale_library_is_never_loaded_silently(monkeypatch, tmp_path):
    """A libd3fields_hip.so built from other sources than the ones in the tree (content fingerprint, not mtimes) is
    rebuilt when hipcc exists and refused when it does not."""
    import shutil
    from d3fields_amd import _lib, build
    assert not build.is_stale()                              # the library the suite runs on matches csrc/
    fake = tmp_path / "libd3fields_hip.so"
    shutil.copy(build.LIB_PATH, fake)
    (tmp_path / "libd3fields_hip.so.fingerprint").write_text("0" * 64 + "\n")
    monkeypatch.setattr(build, "LIB_PATH", str(fake))
    monkeypatch.setattr(build, "FINGERPRINT_PATH", str(fake) + ".fingerprint")
    assert build.is_stale()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(build, "_hipcc", lambda: str(tmp_path / "no_such_hipcc"))
    with pytest.raises((OSError, RuntimeError, ImportError)):
        _lib.load()
    monkeypatch.setattr(_lib, "_lib", None)


def test_bench_gpus_n_is_a_single_command(monkeypatch):
    """`python bench.py --gpus N` without a launcher around it must not die on plumbing: it re-launches itself as N ranks
    under torch.distributed.run (127.0.0.1 rendezvous), passes its own arguments through and returns the launcher's status."""
    import subprocess
    import sys
    import bench
    cmd = bench.spawn_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], port=29512)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[3:11] == ["--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29512", bench.__file__ if False else cmd[10]]
    assert cmd[10].endswith("bench.py") and cmd[11:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    seen = {}

    def fake_call(c, env=None):
        seen["cmd"], seen["env"] = c, env
        return 3                                            # a rank failed

    monkeypatch.setattr(subprocess, "call", fake_call)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 3                                # the launcher's status is the command's status
    assert seen["cmd"][4:6] == ["--nproc-per-node", "2"] and seen["cmd"][-4:] == ["--gpus", "2", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # free ports are picked per launch
    assert bench.spawn_command(2, [])[9] != "0"
    # --force-dist: ONE self-spawned rank takes the multi-rank path (the RCCL plumbing self-test of a one-GPU box)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1", "--force-dist", "--steps", "3"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 3 and seen["cmd"][4:6] == ["--nproc-per-node", "1"] and "--force-dist" in seen["cmd"]


def test_fusion_float16_warns_about_its_meaning():
    """Fusion(dtype=float16) keeps the reference's constructor argument but means fp16 STORAGE with fp32 arithmetic: say so."""
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        Fusion(num_cam=2, dtype=torch.float16)
    assert any("STORES the channel maps in half" in str(x.message) for x in w)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        Fusion(num_cam=2)
    assert not w


@pytest.mark.parametrize("case", ALIGN_CASES)
def test_association_bookkeeping_matches_reference_without_a_gpu(case, monkeypatch):
    """d3fields_amd/association.py's host side -- which instance a detection joins, which of two overlapping instances keeps a
    voxel, the deletion list, the reorder -- against the reference's instances (goldens align_v3_*), with the device functions it
    calls (masked cloud, voxel index, set sizes, label painting) stood in for by the CPU restatements.  The -m gpu tests run the
    same comparison on the kernels."""
    from d3fields_amd import association, pcd_utils
    from oracle import np_assoc, np_pcd
    g = load_golden(case)
    V, H, W = int(g["V"]), int(g["H"]), int(g["W"])
    gs, labels, confs = align_inputs(g)
    bounds = g["bounds"].tolist()
    box = dict(zip(("x_lower", "x_upper", "y_lower", "y_upper", "z_lower", "z_upper"), bounds))

    def cpu_closures(lower, higher, voxel_size, voxel_num):
        to_index = lambda pcds: np_pcd.pcd_to_index(np.asarray(pcds).reshape(-1, 3), lower, voxel_size, voxel_num)      # noqa: E731
        return None, None, None, None, to_index, None
    monkeypatch.setattr(pcd_utils, "init_low_level_memory", cpu_closures)

    class Stand:
        num_cam, device = V, "cpu"
        curr_obs_torch = {"mask_gs": gs, "mask_label": labels, "mask_conf": confs}

        def extract_masked_pcd_in_views(self, inst, views, boundaries, downsample=True):
            assert len(inst) == 1 and len(views) == 1 and downsample
            v = views[0]
            gate = np_pcd.erode_cv2((gs[v][inst[0]] * 255).astype(np.uint8), np.ones([2, 2], np.uint8)) > 0
            pose44 = np.concatenate([g["pose"][v].astype(np.float64), [[0, 0, 0, 1]]], axis=0)
            K = g["K"][v].astype(np.float64)
            pts, _ = np_pcd.backproject_view(g["depth"][v], gate, [K[0, 0], K[1, 1], K[0, 2], K[1, 2]], np.linalg.inv(pose44), bounds)
            return np_pcd.voxel_mean(pts, 0.01)

        def vox_idx_iou(self, a, b):
            return np_pcd.vox_idx_iou(a, b)

        def merge_instances_from_new_view_vox_ver(self, instances, i, boundaries):
            return association.merge_view(self, instances, i, boundaries)

        def filter_instances_vox_ver(self, instances):
            self.after_merges = __import__("copy").deepcopy(instances)
            out = association.filter_instances(self, instances)
            self.after_filter = __import__("copy").deepcopy(out)
            return out

        def reorder_instances(self, instances, queries):
            return association.reorder(instances, queries)

        def swap_instance_mask(self, instances):
            self.curr_obs_torch["mask"] = np_assoc.label_images(instances, gs)

    f = Stand()
    f.H, f.W = H, W
    instances = association.align(f, [str(q) for q in g["queries"]], box)
    assert f.voxel_num.tolist() == np_assoc.association_grid(bounds)[1].tolist()
    assert_instances_match(g, "merged", f.after_merges, V)
    assert_instances_match(g, "filtered", f.after_filter, V)
    assert [inst["label"] for inst in instances] == f.curr_obs_torch["consensus_mask_label"] == [str(x) for x in g["consensus_mask_label"]]
    assert np.array_equal(f.curr_obs_torch["mask"], g["mask"])


def test_finite_word_ring_keeps_live_verdicts_across_wraps(monkeypatch):
    """Host logic of the finite-check ring, with the device calls stubbed out: a cached verdict keeps its slot however often
    the ring wraps; a slot is taken only after the descriptor validated; a check that cannot be made leaves no cached verdict; once the
    ring has wrapped no batch claims that its words are already zero; close() forgets the ring."""
    import contextlib
    import types
    from d3fields_amd import Fusion, _lib, fusion as fmod

    class _Stream:
        cuda_stream = 0
        def wait_event(self, ev):
            pass

    class _Event:
        def record(self, stream=None):
            pass

    flags_seen = []

    class _FakeLib:
        def d3f_map_check(self, *a):
            return 0
        def d3f_map_check_many(self, descs, views, n, words, zero, stream):
            flags_seen.append(int(zero))
            return 0

    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda dev=None: _Stream())
    monkeypatch.setattr(torch.cuda, "Event", _Event)
    monkeypatch.setattr(torch.cuda, "device", lambda dev: contextlib.nullcontext())
    monkeypatch.setattr(fmod._lib, "current_stream_handle", lambda dev: None)
    f = Fusion(num_cam=2)
    f._lib = _FakeLib()
    depth = torch.ones(2, 8, 8)
    batch = []
    a0 = f._finite_word("depth", depth, batch=batch)
    f._flush_checks(batch, depth.device)
    assert flags_seen == [_lib.CHECK_WORDS_ARE_ZERO] and "depth" in f._finite_cache
    slot = f._word_slot["depth"]
    for k in range(3 * f._WORD_SLOTS):                       # a new map tensor per "frame": a new slot each time, never depth's
        m = torch.ones(2, 4, 4, 8)
        batch = []
        assert f._finite_word("depth", depth, batch=batch) == a0 and not batch       # cache hit: same address, nothing queued
        f._finite_word("dino_feats", m, batch=batch)
        f._flush_checks(batch, m.device)
        assert f._word_slot["depth"] == slot and f._word_slot["dino_feats"] != slot
    assert f._ring_wrapped and flags_seen[-1] == 0            # after a wrap the call clears the words it writes itself
    used = f._next_word
    assert f._finite_word("dino_feats", torch.ones(2, 4, 4, 8, dtype=torch.float64)) is None      # not describable: float64
    assert f._next_word == used and "dino_feats" not in f._finite_cache                          # no slot taken, no verdict left
    f.close()
    assert f._words is None and f._next_word == -1 and not f._ring_wrapped and f._order_ws is None and f._lattice_cache is None


def test_parse_probe_words():
    """The 40 words of d3f_points_probe -> (lattice dims or None, unordered?)."""
    from d3fields_amd import Fusion, _lib
    w = torch.zeros(_lib.PROBE_WORDS, dtype=torch.int32)
    assert Fusion._parse_probe(w) == (None, False)
    w[0], w[1], w[2] = 40, 35, 11
    fl = w[24:36].view(torch.float32)
    fl[0::3] = torch.tensor([1.0, 1.0, 1.0, 1.0]); fl[1::3] = torch.tensor([50.0, 50.0, 50.0, 50.0]); fl[2::3] = torch.tensor([256.0] * 4)
    assert Fusion._parse_probe(w) == ((40, 35, 11), False)
    w[8 + 5] = 1                                             # one sample block contradicts the dims
    assert Fusion._parse_probe(w)[0] is None
    fl[0::3] = torch.tensor([40.0] * 4)                       # consecutive points as far apart as far ones: unordered
    assert Fusion._parse_probe(w) == (None, True)
