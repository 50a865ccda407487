"""GPU parity: the HIP path (through the C ABI) against the golden vectors written by the
reference and against the CPU oracle on seeded inputs.  Run with `-m gpu` on an MI355X.

Tolerances (BASELINE.json north_star): valid_mask / instance indices bit-exact; dist bit-exact
(same IEEE operation sequence as the oracle); fused channels <= 1e-5 relative to
max(|ref|_inf, 1) (expf implementations differ by <= 1 ulp); raw bilinear samples bit-exact.
"""
import os

import numpy as np
import pytest
import torch

from conftest import SCENE_CASES, SET_NAMES, load_golden, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def make_fusion(dev, depth, K, pose, maps, H, W, mu=0.02):
    from d3fields_amd import Fusion
    f = Fusion(num_cam=depth.shape[0], device=str(dev))
    f.mu = mu
    t = lambda a: torch.as_tensor(np.asarray(a)).to(dev, torch.float32)   # noqa: E731
    f.curr_obs_torch = {"depth": t(depth), "K": t(K), "pose": t(pose)}
    for k, m in maps.items():
        f.curr_obs_torch[k] = t(m) if not isinstance(m, torch.Tensor) else m.to(dev)
    f.H, f.W = int(H), int(W)
    return f


def cpu(x):
    return x.detach().cpu().numpy()


def check_dist(got, ref, V):
    if V <= 4:
        assert np.array_equal(got, ref, equal_nan=True)
    else:       # reference sum(0) order is position dependent for V > 4 (see test_oracle_golden)
        assert rel_err(got, ref) <= TOL


# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", SCENE_CASES)
def test_eval_vs_reference_golden(dev, case):
    g = load_golden(case)
    f = make_fusion(dev, g["depth"], g["K"], g["pose"], {k: g["in_" + k] for k in SET_NAMES}, g["H"], g["W"], float(g["mu"]))
    pts = torch.from_numpy(g["pts"]).to(dev)
    V = g["depth"].shape[0]
    with torch.no_grad():
        out = f.eval(pts, return_names=SET_NAMES, return_inter=True)
        default = f.eval(pts)
        empty = f.eval(pts, return_names=[])
        dd = f.eval_dist(pts)
    assert sorted(default.keys()) == ["dino_feats", "dist", "mask", "valid_mask"]
    assert sorted(empty.keys()) == ["dist", "valid_mask"]
    assert out["valid_mask"].dtype == torch.bool and out["dist"].dtype == torch.float32
    assert np.array_equal(cpu(out["valid_mask"]), g["valid_mask"])
    check_dist(cpu(out["dist"]), g["dist"], V)
    assert np.array_equal(cpu(empty["dist"]), cpu(out["dist"]))
    for k in SET_NAMES:
        assert out[k].shape == g[k].shape and out[k + "_inter"].shape == g[k + "_inter"].shape
        assert np.array_equal(cpu(out[k + "_inter"]), g[k + "_inter"]), k
        assert rel_err(cpu(out[k]), g[k]) <= TOL, k
    assert torch.equal(default["dino_feats"], f.eval(pts, return_names=["dino_feats"])["dino_feats"])
    # bit-exact instance indices (fusion.py:1360,1390 apply onehot2instance to the fused mask)
    from d3fields_amd import onehot2instance
    assert np.array_equal(cpu(onehot2instance(out["mask"])), g["mask_instance"])
    assert np.array_equal(cpu(dd["valid_mask"]), g["evaldist_valid_mask"])
    check_dist(cpu(dd["dist"]), g["evaldist_dist"], V)


def test_batch_eval_vs_reference_golden(dev):
    from d3fields_amd import synth
    g = load_golden("batch_eval_130001")
    N, st = int(g["N"]), int(g["stride"])
    f = make_fusion(dev, g["depth"], g["K"], g["pose"], {"dino_feats": g["in_dino_feats"], "mask": g["in_mask"]}, g["H"], g["W"], float(g["mu"]))
    pts = synth.random_cloud(N, seed=int(g["cloud_seed"])).to(dev)
    with torch.no_grad():
        out = f.batch_eval(pts)
        empty = f.batch_eval(pts, return_names=[])
    assert sorted(out.keys()) == ["dino_feats", "dist", "mask", "valid_mask"]
    assert sorted(empty.keys()) == ["dist", "valid_mask"]
    assert out["dino_feats"].shape == (N, 2) and out["mask"].shape == (N, 3)
    assert np.array_equal(np.packbits(cpu(out["valid_mask"])), g["valid_bits"])
    assert np.array_equal(cpu(out["dist"])[::st], g["dist_sub"])
    assert rel_err(cpu(out["dino_feats"])[::st], g["dino_feats_sub"]) <= TOL
    assert rel_err(cpu(out["mask"])[::st], g["mask_sub"]) <= TOL
    assert abs(cpu(out["dist"]).astype(np.float64).sum() - float(g["dist_sum"])) <= 1e-6 * abs(float(g["dist_sum"]))
    assert np.allclose(cpu(out["dino_feats"]).astype(np.float64).sum(0), g["dino_feats_sum"], rtol=1e-6, atol=1e-3)


# ---------------------------------------------------------------------------------------
def oracle_eval(sc, pts, maps, **kw):
    from oracle import c_oracle as O
    return O.eval_field(sc["depth"], sc["K"], sc["pose"], pts, maps, **kw)


@pytest.mark.parametrize("kind,V,H,W,fhw,C,N", [
    ("smooth", 4, 480, 640, (48, 64), 384, 50000),        # BASELINE config 1 (patch-res features)
    ("stress", 4, 480, 640, (48, 64), 384, 20000),
    ("smooth", 4, 120, 160, (120, 160), 384, 20000),      # dense (full-res) features
    ("smooth", 8, 72, 128, (9, 16), 1024, 6000),          # config-4 shape, scaled down
    ("stress", 2, 64, 80, (8, 10), 100, 5000),            # C % 4 == 0, odd vector count
    ("smooth", 3, 64, 80, (8, 10), 6, 5000),              # float2 path
    ("smooth", 3, 64, 80, (8, 10), 7, 5000),              # scalar path
    ("smooth", 17, 32, 40, (4, 5), 12, 3000),             # many views (smaller tiles)
])
def test_eval_vs_oracle_seeded(dev, kind, V, H, W, fhw, C, N):
    from d3fields_amd import synth
    sc = synth.make_scene(V, H, W, kind)
    feats = synth.random_map(V, fhw[0], fhw[1], C, seed=1)
    mask = synth.random_onehot_mask(V, H, W, 8, seed=2)
    color = torch.rand(V, H, W, 3, generator=torch.Generator().manual_seed(4))
    pts = synth.random_cloud(N, seed=3)
    f = make_fusion(dev, sc["depth"], sc["K"], sc["pose"], {"dino_feats": feats, "mask": mask, "color_tensor": color}, H, W)
    with torch.no_grad():
        out = f.eval(pts.to(dev), return_names=SET_NAMES)
    ref = oracle_eval(sc, pts, [feats, mask, color])
    assert np.array_equal(cpu(out["valid_mask"]), ref["valid_mask"])
    assert np.array_equal(cpu(out["dist"]), ref["dist"])
    for i, k in enumerate(SET_NAMES):
        assert rel_err(cpu(out[k]), ref["sets"][i]) <= TOL, k
    from d3fields_amd import onehot2instance
    from oracle import c_oracle as O
    assert np.array_equal(cpu(onehot2instance(out["mask"])), O.onehot2instance(ref["sets"][1]))


@pytest.mark.parametrize("N", [0, 1, 63, 255, 256, 257, 1000])
def test_ragged_sizes(dev, N):
    from d3fields_amd import synth
    V, H, W = 4, 48, 64
    sc = synth.make_scene(V, H, W, "smooth")
    feats = synth.random_map(V, 6, 8, 20, seed=1)
    f = make_fusion(dev, sc["depth"], sc["K"], sc["pose"], {"dino_feats": feats}, H, W)
    pts = synth.random_cloud(N, seed=5)
    with torch.no_grad():
        out = f.eval(pts.to(dev), return_names=["dino_feats"], return_inter=True)
    assert out["dist"].shape == (N,) and out["dino_feats"].shape == (N, 20) and out["dino_feats_inter"].shape == (V, N, 20)
    if N:
        ref = oracle_eval(sc, pts, [feats], return_inter=True)
        assert np.array_equal(cpu(out["dist"]), ref["dist"])
        assert np.array_equal(cpu(out["dino_feats_inter"]), ref["inter"][0])
        assert rel_err(cpu(out["dino_feats"]), ref["sets"][0]) <= TOL


def test_nonfinite_maps_propagate_like_reference(dev):
    """0*NaN through an invalid view must reach the output (fusion.py:385) -- i.e. the
    invalid-view skip is only taken when the maps were verified finite."""
    from d3fields_amd import synth
    V, H, W = 3, 48, 64
    sc = synth.make_scene(V, H, W, "stress")
    feats = synth.random_map(V, 12, 16, 8, seed=1)
    feats[1, 3:9, 4:12, 2] = float("nan")
    feats[2, 0:4, 0:5, 5] = float("inf")
    pts = synth.random_cloud(6000, seed=3)
    f = make_fusion(dev, sc["depth"], sc["K"], sc["pose"], {"dino_feats": feats}, H, W)
    with torch.no_grad():
        out = f.eval(pts.to(dev), return_names=["dino_feats"])
    ref = oracle_eval(sc, pts, [feats])
    got, want = cpu(out["dino_feats"]), ref["sets"][0]
    assert np.isnan(want).any()
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(np.isinf(got), np.isinf(want)) and np.isinf(want).any()
    ok = np.isfinite(want)
    assert rel_err(got[ok], want[ok]) <= TOL
    # and with finite maps the skip changes nothing
    feats2 = torch.nan_to_num(feats, nan=0.5, posinf=1.0)
    f2 = make_fusion(dev, sc["depth"], sc["K"], sc["pose"], {"dino_feats": feats2}, H, W)
    with torch.no_grad():
        o2 = f2.eval(pts.to(dev), return_names=["dino_feats"])
    assert rel_err(cpu(o2["dino_feats"]), oracle_eval(sc, pts, [feats2])["sets"][0]) <= TOL


def test_map_check_words(dev):
    """d3f_map_check (the device-side replacement of torch.isfinite(map).all()): the word is non-zero iff the map holds a NaN /
    Inf -- flat 16-byte path incl. the tail and the very last element, fp16 storage, strided views, unaligned bases; and the
    shim follows in-place torch writes (version counter) without a host sync in the query."""
    import ctypes
    from d3fields_amd import synth, _lib
    lib = _lib.load()
    word = torch.full((4,), 77, dtype=torch.int32, device=dev)
    st = _lib.current_stream_handle(dev)

    def check(t, expect):
        V = t.shape[0]
        desc = _lib.ChannelMap(t.data_ptr(), t.shape[1], t.shape[2], t.shape[3], _lib.DTYPE_F16 if t.dtype == torch.float16 else _lib.DTYPE_F32,
                               t.stride(0), t.stride(1), t.stride(2), None)
        _lib.check(lib.d3f_map_check(ctypes.byref(desc), V, ctypes.c_void_p(word.data_ptr() + 4), st))
        got = word.tolist()
        assert got[0] == 77 and got[2] == 77 and (got[1] != 0) == expect, (got, expect, tuple(t.shape), t.dtype)

    g = torch.Generator().manual_seed(5)
    for dtype in (torch.float32, torch.float16):
        for shape in ((3, 7, 9, 5), (2, 12, 16, 384), (4, 48, 64, 384), (1, 1, 1, 1), (2, 3, 5, 8)):
            base = torch.randn(shape, generator=g).to(dev, dtype)
            check(base, False)
            n = base.numel()
            for pos in sorted({0, n - 1, n // 2, max(n - 3, 0)}):
                for bad in (float("nan"), float("inf"), float("-inf")):
                    t = base.clone()
                    t.view(-1)[pos] = bad
                    check(t, True)
        # a [..., :C] view of a wider buffer (texel stride > C): only the view's elements count
        wide = torch.randn((2, 6, 8, 24), generator=g).to(dev, dtype)
        wide[..., 20] = float("nan")                      # outside the view
        check(wide[..., :16], False)
        wide[1, 5, 7, 15] = float("inf")                  # last element of the view
        check(wide[..., :16], True)
        # unaligned base: the flat path needs 16-byte alignment, this one takes the strided kernel
        flat = torch.randn(2 * 5 * 6 * 12 + 1, generator=g).to(dev, dtype)
        v = flat[1:].view(2, 5, 6, 12)
        check(v, False)
        flat[-1] = float("nan")
        check(v, True)
    # bad arguments come back as status codes
    desc = _lib.ChannelMap(16, 4, 4, 8, 0, 128, 32, 8, None)
    assert lib.d3f_map_check(ctypes.byref(desc), 2, None, st) == _lib.ERR_INVALID_ARG
    assert lib.d3f_map_check(ctypes.byref(desc), 0, ctypes.c_void_p(word.data_ptr()), st) == _lib.ERR_BAD_SHAPE
    desc = _lib.ChannelMap(16, 4, 4, 8, 7, 128, 32, 8, None)
    assert lib.d3f_map_check(ctypes.byref(desc), 2, ctypes.c_void_p(word.data_ptr()), st) == _lib.ERR_BAD_DTYPE

    # the shim: device words instead of torch.isfinite, following in-place writes
    V, H, W = 3, 48, 64
    sc = synth.make_scene(V, H, W, "stress")
    feats = synth.random_map(V, 12, 16, 160, seed=1)
    pts = synth.random_cloud(6000, seed=3)
    f = make_fusion(dev, sc["depth"], sc["K"], sc["pose"], {"dino_feats": feats}, H, W)
    with torch.no_grad():
        a = f.eval(pts.to(dev), return_names=["dino_feats"])
        assert f.maps_are_finite(("depth", "dino_feats"))
        f.curr_obs_torch["dino_feats"][1, 3:9, 4:12, 2] = float("nan")            # in place: torch bumps the version counter
        b = f.eval(pts.to(dev), return_names=["dino_feats"])
        assert not f.maps_are_finite(("depth", "dino_feats"))
    feats_nan = feats.clone()
    feats_nan[1, 3:9, 4:12, 2] = float("nan")
    want = oracle_eval(sc, pts, [feats_nan])["sets"][0]
    got = cpu(b["dino_feats"])
    assert np.isnan(want).any() and np.array_equal(np.isnan(got), np.isnan(want))
    ok = np.isfinite(want)
    assert rel_err(got[ok], want[ok]) <= TOL
    assert rel_err(cpu(a["dino_feats"]), oracle_eval(sc, pts, [feats])["sets"][0]) <= TOL
    # a depth image with a NaN: strict path too (the word of 'depth' is part of every query)
    f2 = make_fusion(dev, sc["depth"], sc["K"], sc["pose"], {"dino_feats": feats}, H, W)
    f2.curr_obs_torch["depth"][0, 5, 7] = float("nan")
    with torch.no_grad():
        f2.eval(pts.to(dev), return_names=["dino_feats"])
    assert not f2.maps_are_finite(("depth",)) and f2.maps_are_finite(("dino_feats",))


def test_map_check_many_and_the_ring_of_words(dev):
    """d3f_map_check_many (ABI 5): depth + every map of a frame in ONE launch -- flat tensors inside the batched kernel, a strided
    one through its own launch inside the call, with and without D3F_CHECK_WORDS_ARE_ZERO -- gives the single-tensor call's
    verdicts; the shim hands every check a fresh slot of a zeroed ring, stays right when the ring wraps (256 checks), and
    `debug_recheck_maps` finds a writer that changed a map behind torch's back."""
    import ctypes
    from d3fields_amd import synth, _lib
    lib = _lib.load()
    st = _lib.current_stream_handle(dev)
    g = torch.Generator().manual_seed(9)
    tensors = [torch.randn((3, 40, 56, 1), generator=g).to(dev),                       # "depth" as a one-channel map
               torch.randn((3, 12, 16, 384), generator=g).to(dev),
               torch.randn((3, 40, 56, 8), generator=g).to(dev).half(),
               torch.randn((3, 6, 8, 24), generator=g).to(dev)[..., :16]]               # a channel range: not one flat block
    for bad_at in (None, 0, 1, 2, 3):
        ts = [t.clone() for t in tensors[:3]]
        ts.append(torch.randn((3, 6, 8, 24), generator=g).to(dev)[..., :16])             # texel stride 24 > C = 16: its own launch
        assert not ts[3].is_contiguous()
        if bad_at is not None:
            ts[bad_at][2, 3, 1, 0] = float("nan")
        n = len(ts)
        descs = (_lib.ChannelMap * n)(*[_lib.ChannelMap(t.data_ptr(), t.shape[1], t.shape[2], t.shape[3], _lib.DTYPE_F16 if t.dtype == torch.float16 else _lib.DTYPE_F32,
                                                        t.stride(0), t.stride(1), t.stride(2), None) for t in ts])
        views = (ctypes.c_int32 * n)(*[t.shape[0] for t in ts])
        for flags, fill in ((0, 55), (_lib.CHECK_WORDS_ARE_ZERO, 0)):
            words = torch.full((8,), fill, dtype=torch.int32, device=dev)
            ptrs = (ctypes.c_void_p * n)(*[words.data_ptr() + 4 * (2 * k) for k in range(n)])
            _lib.check(lib.d3f_map_check_many(descs, views, n, ptrs, flags, st))
            got = words.tolist()
            assert [got[2 * k] != 0 for k in range(n)] == [k == bad_at for k in range(n)], (bad_at, flags, got)
            assert all(got[2 * k + 1] == fill for k in range(n))                       # neighbouring words untouched
    assert lib.d3f_map_check_many(None, None, 0, None, 0, st) == 0
    assert lib.d3f_map_check_many(descs, views, 99, ptrs, 0, st) == _lib.ERR_BAD_SHAPE
    ptrs[1] = None
    assert lib.d3f_map_check_many(descs, views, n, ptrs, 0, st) == _lib.ERR_BAD_LAYOUT

    # the shim: a fresh slot per check; 300 frames wrap the ring of 256 slots; verdicts stay those of the tensors
    V, H, W = 3, 48, 64
    sc = synth.make_scene(V, H, W, "stress")
    feats = synth.random_map(V, 12, 16, 160, seed=1).to(dev)
    pts = synth.random_cloud(3000, seed=3).to(dev)
    f = make_fusion(dev, sc["depth"], sc["K"], sc["pose"], {"dino_feats": feats}, H, W)
    ref = None
    with torch.no_grad():
        for frame in range(300):
            f.curr_obs_torch["dino_feats"] = feats.clone()                              # a NEW tensor object: checked again
            if frame % 50 == 49:
                f.curr_obs_torch["dino_feats"][0, :, :, 0] = float("inf")
            out = f.eval(pts, return_names=["dino_feats"])
            if frame == 0:
                ref = out["dino_feats"].clone()
            if frame % 50 == 49:
                assert not f.maps_are_finite(("depth", "dino_feats")) and not torch.isfinite(out["dino_feats"]).all()
            elif frame % 50 == 0:
                assert f.maps_are_finite(("depth", "dino_feats")) and torch.equal(out["dino_feats"], ref), frame
        # a writer behind torch's back (no version bump): the cached verdict is stale until invalidate_map_checks() -- or debug mode
        m = f.curr_obs_torch["dino_feats"] = feats.clone()
        f.eval(pts, return_names=["dino_feats"])
        assert f.maps_are_finite(("dino_feats",))
        m.data[0, :, :, 0] = float("nan")                 # through .data: the storage changes, m's version counter does not
        assert f.maps_are_finite(("dino_feats",))         # ... so the cached verdict is stale (documented: invalidate_map_checks())
        f.debug_recheck_maps = True
        assert not f.maps_are_finite(("dino_feats",))
        assert torch.isnan(f.eval(pts, return_names=["dino_feats"])["dino_feats"]).any()


def test_ring_wrap_keeps_the_cached_nonfinite_depth(dev):
    """the depth image holds a NaN and stays CACHED (same tensor object) while a new map tensor is checked every frame,
    so the ring of check words wraps with a live non-finite verdict in it.  The wrap must not clear that word (the query would
    take the exact invalid-view skip and drop the 0 * NaN terms the reference keeps) nor hand its slot to another tensor: every
    frame equals frame 0, NaNs included.  Also: a check that cannot be made (unsupported layout) leaves no cached verdict behind."""
    from d3fields_amd import synth
    V, H, W = 3, 48, 64
    sc = synth.make_scene(V, H, W, "smooth")
    depth = sc["depth"].clone()
    depth[1, 10:30, 12:40] = float("nan")
    feats = synth.random_map(V, 12, 16, 160, seed=1).to(dev)
    pts = synth.random_cloud(3000, seed=3).to(dev)
    f = make_fusion(dev, depth, sc["K"], sc["pose"], {"dino_feats": feats}, H, W)
    with torch.no_grad():
        ref = f.eval(pts, return_names=["dino_feats"])
        assert torch.isnan(ref["dino_feats"]).any() and not f.maps_are_finite(("depth",))
        depth_slot = f._word_slot["depth"]
        for frame in range(2 * f._WORD_SLOTS + 10):
            f.curr_obs_torch["dino_feats"] = feats.clone()                              # a NEW tensor object: a new slot every frame
            out = f.eval(pts, return_names=["dino_feats"])
            assert f._word_slot["depth"] == depth_slot and f._word_slot["dino_feats"] != depth_slot, frame
            if frame % 16 == 0 or frame >= f._WORD_SLOTS - 4:
                for k in ("dino_feats", "dist"):
                    assert torch.equal(torch.isnan(out[k]), torch.isnan(ref[k])), (frame, k)
                    assert torch.equal(torch.nan_to_num(out[k]), torch.nan_to_num(ref[k])), (frame, k)
        assert f._ring_wrapped and not f.maps_are_finite(("depth",)) and f.maps_are_finite(("dino_feats",))
        # a tensor the check cannot describe (float64): no word, and the verdict of the tensor checked before is gone
        f.curr_obs_torch["dino_feats"] = feats.clone()
        f.eval(pts, return_names=["dino_feats"])
        assert "dino_feats" in f._finite_cache
        assert f._finite_word("dino_feats", feats.double()) is None and "dino_feats" not in f._finite_cache
    f.close()
    assert f._order_ws is None and f._lattice_cache is None and f._words is None and f._next_word == -1 and not f._ring_wrapped


def test_nonfinite_points(dev):
    from d3fields_amd import synth
    V, H, W = 3, 48, 64
    sc = synth.make_scene(V, H, W, "smooth")
    feats = synth.random_map(V, 12, 16, 8, seed=1)
    pts = synth.random_cloud(600, seed=3)
    pts[5, 0] = float("nan")
    pts[77] = float("inf")
    pts[300, 2] = 1e30
    f = make_fusion(dev, sc["depth"], sc["K"], sc["pose"], {"dino_feats": feats}, H, W)
    with torch.no_grad():
        out = f.eval(pts.to(dev), return_names=["dino_feats"])
    ref = oracle_eval(sc, pts, [feats])
    assert np.array_equal(cpu(out["valid_mask"]), ref["valid_mask"])
    assert np.array_equal(cpu(out["dist"]), ref["dist"], equal_nan=True)
    got, want = cpu(out["dino_feats"]), ref["sets"][0]
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert rel_err(got[~np.isnan(want)], want[~np.isnan(want)]) <= TOL


def test_strided_and_offset_maps(dev):
    """Maps that are views (channel slice, cropped rows) are addressed through their strides."""
    from d3fields_amd import synth
    V, H, W = 4, 48, 64
    sc = synth.make_scene(V, H, W, "smooth")
    big = synth.random_map(V, 14, 18, 24, seed=1).to(dev)
    view = big[:, 1:13, 2:18, 4:16]                       # (V,12,16,12), stride_x 24, 16-B aligned offset
    odd = big[:, :, :, 1:10]                              # C = 9, misaligned -> scalar path
    f = make_fusion(dev, sc["depth"], sc["K"], sc["pose"], {}, H, W)
    f.curr_obs_torch["a"] = view
    f.curr_obs_torch["b"] = odd
    pts = synth.random_cloud(4000, seed=3)
    with torch.no_grad():
        out = f.eval(pts.to(dev), return_names=["a", "b"])
    ref = oracle_eval(sc, pts, [cpu(view), cpu(odd)])
    assert rel_err(cpu(out["a"]), ref["sets"][0]) <= TOL
    assert rel_err(cpu(out["b"]), ref["sets"][1]) <= TOL


# ---------------------------------------------------------------------------------------
# Full BASELINE sizes: size-independent properties (the oracle would take minutes there)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,C,fhw", [(985600, 384, (48, 64)), (1925000, 384, (48, 64)), (500000, 384, (480, 640))])
def test_full_size_properties(dev, N, C, fhw):
    from d3fields_amd import synth, create_init_grid
    V, H, W = 4, 480, 640
    sc = synth.make_scene(V, H, W, "smooth")
    feats = synth.random_map(V, fhw[0], fhw[1], C, seed=1, device=dev)
    mask = synth.random_onehot_mask(V, H, W, 8, seed=2, device=dev)
    f = make_fusion(dev, sc["depth"], sc["K"], sc["pose"], {"dino_feats": feats, "mask": mask}, H, W)
    step = {985600: 0.005, 1925000: 0.004}.get(N)
    pts = (create_init_grid(synth.WORK_BOX, step)[0] if step else synth.random_cloud(N, seed=3)).to(dev)
    assert pts.shape[0] == N
    with torch.no_grad():
        out = f.batch_eval(pts)
        # (1) permutation equivariance, bit-exact: points are independent
        perm = torch.randperm(N, generator=torch.Generator().manual_seed(9)).to(dev)
        sub = perm[:200000]
        o2 = f.eval(pts[sub])
        for k in ("dist", "valid_mask", "dino_feats", "mask"):
            assert torch.equal(o2[k], out[k][sub]), k
        # (2) chunk invariance == batch_eval semantics (60 000-point chunks, ragged tail)
        lo = N - 150001
        parts = [f.eval(pts[i:min(i + 60000, N)]) for i in range(lo, N, 60000)]
        for k in ("dist", "dino_feats"):
            assert torch.equal(torch.cat([p[k] for p in parts]), out[k][lo:]), k
        # (3) sentinel / validity coupling (fusion.py:366-370,386)
        inv = ~out["valid_mask"]
        assert torch.equal(out["dist"] == 1e3, inv)
        assert (out["dino_feats"][inv] == 0).all() and (out["mask"][inv] == 0).all()
        assert (out["dist"][~inv].abs() <= f.mu).all()
        # (4) one-hot mask channels are convex weights: 0 <= sum_c mask <= 1 (weights <= 1, /count)
        ms = out["mask"].sum(1)
        assert (ms >= 0).all() and (ms <= 1 + 1e-5).all()
        # (5) linearity in the map: eval(2*F + 1) == 2*eval(F) + eval(1)
        f.curr_obs_torch["lin"] = feats[..., :64] * 2.0 + 1.0
        f.curr_obs_torch["one"] = torch.ones_like(feats[..., :4])
        f.curr_obs_torch["f64"] = feats[..., :64].contiguous()
        o3 = f.eval(pts[sub], return_names=["lin", "one", "f64"])
        want = 2.0 * o3["f64"] + o3["one"][:, :1]
        assert (o3["lin"] - want).abs().max().item() <= 1e-5 * max(want.abs().max().item(), 1.0)
    # (6) a sample of the full-size result against the oracle
    pick = perm[:3000].cpu()
    ref = oracle_eval(sc, pts[pick.to(dev)].cpu(), [feats.cpu(), mask.cpu()])
    assert np.array_equal(cpu(out["valid_mask"][pick.to(dev)]), ref["valid_mask"])
    assert np.array_equal(cpu(out["dist"][pick.to(dev)]), ref["dist"])
    assert rel_err(cpu(out["dino_feats"][pick.to(dev)]), ref["sets"][0]) <= TOL
    assert rel_err(cpu(out["mask"][pick.to(dev)]), ref["sets"][1]) <= TOL


# ---------------------------------------------------------------------------------------
def test_onehot_helpers(dev):
    from d3fields_amd import instance2onehot, onehot2instance
    g = load_golden("onehot")
    inst = torch.from_numpy(g["inst"]).to(dev)
    oh = instance2onehot(inst, int(g["NI"]))
    assert oh.dtype == torch.bool and np.array_equal(cpu(oh), g["onehot"])
    assert np.array_equal(cpu(onehot2instance(torch.from_numpy(g["soft"]).to(dev))), g["soft_inst"])
    assert np.array_equal(cpu(onehot2instance(oh)), g["inst"])
    nan_row = torch.tensor([[0.1, float("nan"), 5.0], [3.0, 3.0, 1.0]], device=dev)
    assert cpu(onehot2instance(nan_row)).tolist() == torch.argmax(nan_row.cpu(), -1).tolist()


@pytest.mark.parametrize("dt", ["l2", "square"])
def test_corr_utils_vs_reference_golden(dev, dt):
    from d3fields_amd import corr_utils as cu
    g = load_golden("corr_utils")
    sc = float(g["scale"])
    fm = g["fmap_bhwc"]
    bchw = torch.from_numpy(np.ascontiguousarray(fm.transpose(0, 3, 1, 2))).to(dev)
    tgt = torch.from_numpy(g["tgt"]).to(dev)
    sim = cu.compute_similarity(fm, g["tgt"], sc, dist_type=dt)
    assert isinstance(sim, np.ndarray) and rel_err(sim, g["similarity_" + dt]) <= TOL
    assert rel_err(cpu(cu.compute_similarity_tensor(bchw, tgt, sc, dist_type=dt)), g["similarity_tensor_" + dt]) <= TOL
    assert rel_err(cpu(cu.compute_dist_tensor(bchw, tgt, dist_type=dt)), g["dist_tensor_" + dt]) <= TOL
    src, tg = torch.from_numpy(g["multi_src"]).to(dev), torch.from_numpy(g["multi_tgt"]).to(dev)
    out = cu.compute_similarity_tensor_multi(src, tg, None, None, float(g["multi_scale"]), dist_type=dt)
    assert rel_err(cpu(out), g["multi_" + dt]) <= TOL
    assert np.allclose(cpu(out).sum(0), 1.0, atol=1e-5)
    sim2, idx = cu.nearest_descriptor(src, tg, float(g["multi_scale"]), dist_type=dt)
    assert torch.equal(sim2, out) and np.array_equal(cpu(idx), g["multi_argmax_" + dt])
    if dt == "l2":
        flat = torch.from_numpy(g["flat"]).to(dev)
        assert rel_err(cpu(cu.compute_similarity_tensor(flat, tgt, sc)), g["flat_similarity_tensor_l2"]) <= TOL
        assert rel_err(cpu(cu.compute_dist_tensor(flat, tgt)), g["flat_dist_tensor_l2"]) <= TOL
    with pytest.raises(NotImplementedError):
        cu.compute_dist_tensor(bchw, tgt, dist_type="cosine")


@pytest.mark.parametrize("B1,B2,C", [(5000, 300, 384), (1, 1, 3), (257, 65, 33), (100000, 300, 384),
                                     # last column tile with 1 / 2 / 3 live 16-column groups (the fast kernel skips the dead
                                     # ones), a one-tile matrix narrower than a group, many row chunks for the merge
                                     (2000, 80, 64), (2100, 96, 64), (1500, 17, 32), (333, 6, 32), (20000, 44, 32)])
def test_pairwise_vs_oracle(dev, B1, B2, C):
    from d3fields_amd import corr_utils as cu
    from oracle import c_oracle as O
    g = torch.Generator().manual_seed(B1 + C)
    src = torch.randn(B1, C, generator=g)
    tgt = torch.randn(B2, C, generator=g)
    tgt[0] = src[B1 // 2]
    out, idx = cu.nearest_descriptor(src.to(dev), tgt.to(dev), 0.9)
    if B1 <= 20000:
        ref, am = O.pairwise(src.numpy(), tgt.numpy(), 0.9, "l2", return_argmax=True)
        assert rel_err(cpu(out), ref) <= TOL
        assert np.array_equal(cpu(idx), am)
    else:       # full config-5 size: properties
        assert torch.allclose(out.sum(0), torch.ones(B2, device=dev), atol=1e-4)
        assert idx[0].item() == B1 // 2
        assert torch.equal(out.argmax(0), idx)
        sl = slice(40000, 41000)
        d = cu.compute_dist_tensor(src[sl].to(dev), tgt[7].to(dev))
        assert rel_err(cpu(d), O.dist_to_target(src[sl].numpy(), tgt[7].numpy(), "l2", channel_axis=1)) <= TOL


@pytest.mark.parametrize("B1,B2,C,dist_type,scale", [(100000, 300, 384, "l2", 1.0), (20000, 44, 64, "square", 0.05), (4097, 300, 384, "l2", 3.0)])
def test_pairwise_guarded_contraction_near_duplicates(dev, B1, B2, C, dist_type, scale):
    """The guarded contraction form of pairwise_dist (round 6: |a|^2 + |b|^2 - 2 a.b on the fp32 matrix cores; pairs that are not
    at least a quarter of |a|^2 + |b|^2 apart are recomputed in the direct form) at config 5's size, against the oracle, with
    exactly the rows a correspondence lookup exists for: exact duplicates, near matches at graded distances across the guard's
    threshold, two rows tied for a column, a 64 x 64 block of mutually near descriptors (the whole tile falls back to the direct
    form), a column with an offset (everything flagged), and non-finite descriptors (flagged: the reference's propagation)."""
    from d3fields_amd import corr_utils as cu, _lib
    from oracle import c_oracle as O
    g = torch.Generator().manual_seed(B1 + B2)
    src = torch.randn(B1, C, generator=g)
    tgt = torch.randn(B2, C, generator=g)
    typical = float((2 * C) ** 0.5)                             # distance of two unrelated unit-variance descriptors
    eps = [0.0, 1e-4, 1e-2, 0.3, 1.0, 0.2 * typical, 0.45 * typical, 0.5 * typical, 0.55 * typical, 0.8 * typical]
    rows = [(37 * (k + 1)) % B1 for k in range(len(eps))]
    for k, e in enumerate(eps):                                  # column k: a match at distance ~e (0.5 * typical = the guard's threshold)
        noise = torch.randn(C, generator=g)
        tgt[k] = src[rows[k]] + noise * (e / float(noise.norm()))
    tie = B1 // 2
    tgt[12] = src[tie] + 0.1 * torch.randn(C, generator=g)
    src[tie + 1] = tgt[12] + (src[tie] - tgt[12]).flip(0)        # rows tie, tie + 1: the same distance to column 12
    if B2 >= 300:                                                # a dense block: 64 rows (one row tile) x columns 128..191 all mutually near
        base = torch.randn(C, generator=g)
        r0 = min(5056, B1 - 192) // 64 * 64
        src[r0:r0 + 64] = base + 0.05 * torch.randn(64, C, generator=g)
        tgt[128:192] = base + 0.05 * torch.randn(64, C, generator=g)
        tgt[200] = tgt[200] + 40.0                               # an offset column: |b|^2 dwarfs every distance to it
        src[777, 5] = float("inf")
        tgt[201, 7] = float("nan")
    srcd, tgtd = src.to(dev), tgt.to(dev)
    sim, idx = cu.nearest_descriptor(srcd, tgtd, scale, dist_type)
    dist = cu._pairwise(srcd, tgtd, 1.0, dist_type, _lib.SIM_DIST, False)[0]
    ref, am = O.pairwise(src.numpy(), tgt.numpy(), scale, dist_type, return_argmax=True)
    dref = O.pairwise(src.numpy(), tgt.numpy(), 1.0, dist_type, mode="dist")
    got, dgot = cpu(sim), cpu(dist)
    assert np.array_equal(np.isnan(dgot), np.isnan(dref)) and np.array_equal(np.isinf(dgot), np.isinf(dref))
    fin = np.isfinite(dref)
    assert np.abs(dgot[fin] - dref[fin]).max() <= 1e-5 * max(float(dref[fin].max()), 1.0)
    # near pairs take the direct form: as exact as ever, also RELATIVE to themselves (the contraction alone: |a|^2 * 1e-7 absolute)
    near = fin & (dref <= (0.45 * typical if dist_type == "l2" else (0.45 * typical) ** 2))
    assert near.sum() >= len(eps) - 3
    assert np.all(np.abs(dgot[near] - dref[near]) <= 2e-6 * dref[near] + 1e-30)
    assert dgot[rows[0], 0] == 0.0                               # the exact duplicate
    okc = ~np.isnan(ref).any(axis=0)                             # columns the NaN / Inf descriptors did not poison (softmax over rows)
    assert np.array_equal(np.isnan(got).any(axis=0), ~okc)
    assert rel_err(got[:, okc], ref[:, okc]) <= TOL
    assert torch.allclose(sim[:, torch.from_numpy(okc).to(dev)].sum(0), torch.ones(int(okc.sum()), device=dev), atol=1e-4)
    best = np.sort(ref[:, okc], axis=0)[-2:]
    clear = best[1] - best[0] > 1e-4 * np.maximum(best[1], 1e-30)
    assert np.array_equal(cpu(idx)[okc][clear], am[okc][clear])
    for k in range(5):
        assert idx[k].item() == rows[k]


@pytest.mark.parametrize("B1,splits", [(5000, (0, 1700, 1700, 5000)), (257, (0, 256, 257)), (1000, (0, 333, 1000))])
@pytest.mark.parametrize("dist_type", ["l2", "square"])
def test_row_sharded_softmax_steps(dev, B1, splits, dist_type):
    """The three device steps of sharding.sharded_similarity_multi, with the ranks' row blocks (one of them empty)
    processed in one process: rows and global argmax equal the unsharded launch and the oracle."""
    from d3fields_amd import corr_utils as cu
    from d3fields_amd.sharding import _HipSoftmaxKernels as K
    from d3fields_amd import _lib
    from oracle import c_oracle as O
    B2, C = 70, 48
    g = torch.Generator().manual_seed(B1)
    src = torch.randn(B1, C, generator=g)
    tgt = torch.randn(B2, C, generator=g)
    tgt[3] = src[B1 - 1]
    code = {"l2": _lib.DIST_L2, "square": _lib.DIST_SQUARE}[dist_type]
    srcd, tgtd = src.to(dev), tgt.to(dev)
    blocks = [(splits[i], splits[i + 1]) for i in range(len(splits) - 1)]
    local = [K.local(srcd[lo:hi].contiguous(), tgtd, 0.7, code, lo) for lo, hi in blocks]
    parts = torch.stack([st for _, st in local])
    merged, am = K.merge(parts)
    rows = torch.cat([K.apply(o, 0.7, merged) for o, _ in local])
    whole = cu.compute_similarity_tensor_multi(srcd, tgtd, None, None, 0.7, dist_type)
    ref, ref_am = O.pairwise(src.numpy(), tgt.numpy(), 0.7, dist_type, return_argmax=True)
    assert rel_err(cpu(rows), cpu(whole)) <= 1e-6
    assert rel_err(cpu(rows), ref) <= TOL
    assert np.array_equal(cpu(am), ref_am) and am[3].item() == B1 - 1
    assert torch.allclose(rows.sum(0), torch.ones(B2, device=dev), atol=1e-5)


def test_point_order_probe_and_unordered_walk(dev):
    """d3f_point_order_locality separates grids from random clouds; the shim then asks for the Morton walk on small
    maps (D3F_FLAG_UNORDERED_POINTS) and every output stays bit-identical to the caller-order launch."""
    import ctypes
    from d3fields_amd import synth, create_init_grid, _lib
    lib = _lib.load()
    out = torch.empty(2, device=dev)

    def probe(p):
        p = p.to(dev).contiguous()
        _lib.check(lib.d3f_point_order_locality(_lib.ptr(p), p.shape[0], _lib.ptr(out), _lib.current_stream_handle(dev)))
        return out.tolist()

    grid, _ = create_init_grid(synth.WORK_BOX, 0.01)
    near, far = probe(grid)
    assert 0 < near < 0.1 * far
    cloud = synth.random_cloud(200000, seed=3)
    near, far = probe(cloud)
    assert near > 0.5 * far
    bad = cloud.clone()
    bad[::7] = float("nan")
    near, far = probe(bad)
    assert np.isfinite(near) and np.isfinite(far) and near > 0.5 * far
    assert probe(cloud[:1]) == [0.0, 0.0] and probe(cloud[:0]) == [0.0, 0.0]

    V, H, W = 4, 120, 160
    sc = synth.make_scene(V, H, W, "stress")
    maps = {"dino_feats": synth.random_map(V, 12, 16, 96, seed=1), "mask": synth.random_onehot_mask(V, H, W, 5, seed=2)}
    f = make_fusion(dev, sc["depth"], sc["K"], sc["pose"], maps, H, W)
    names = ["dino_feats", "mask"]
    f.detect_lattice = False                    # this test is about the locality probe of NON-lattice inputs
    for pts, unordered in ((cloud, True), (grid[:150000], False)):
        p = pts.to(dev)
        f.detect_point_order = True
        a = f.eval(p, return_names=names)
        assert f._order_cache[1] is unordered
        f.detect_point_order = False
        b = f.eval(p, return_names=names)
        for k in a:
            assert torch.equal(a[k], b[k]), k
    f.detect_point_order = True
    ref = oracle_eval(sc, cloud.numpy(), [maps[k].numpy() for k in names])
    got = f.eval(cloud.to(dev), return_names=names)
    assert np.array_equal(cpu(got["valid_mask"]), ref["valid_mask"].astype(bool))
    assert np.array_equal(cpu(got["dist"]), ref["dist"])
    for s_, k in enumerate(names):
        assert rel_err(cpu(got[k]), ref["sets"][s_]) <= TOL


@pytest.mark.parametrize("use_graph", [False, True, "one launch per step", "five launches", "autograd"])
def test_rigid_tracking_matches_reference(dev, use_graph):
    """Fusion.rigid_tracking (100 Adam steps; eager autograd through d3f_eval / d3f_eval_backward, the closed-form HIP step as
    one launch for the whole frame (d3f_track_run), one per step (d3f_track_step) or five per step, or the autograd step
    graph-replayed) against the keypoints the REFERENCE's loop returned (golden 'rigid_tracking'): measured agreement
    3e-8 m on positions that move 1-15 mm; tolerance 1e-5 m."""
    g = load_golden("rigid_tracking")
    f = make_fusion(dev, g["depth"], g["K"], g["pose"], {"dino_feats": g["in_dino_feats"]}, g["H"], g["W"], float(g["mu"]))
    f.use_hip_graph = bool(use_graph)
    f.fused_tracking = use_graph in (True, "one launch per step", "five launches")     # the closed-form HIP step; "autograd": the autograd step, graph-replayed
    f.single_launch_tracking = use_graph in (True, "one launch per step")
    f.loop_launch_tracking = use_graph is True
    n = int(g["n"])
    info = {"a": {"src_feats": torch.from_numpy(g["src_feats"][:n])}, "b": {"src_feats": torch.from_numpy(g["src_feats"][n:])}}
    res = f.rigid_tracking(info, [p for p in g["last_pts"]], None, n)
    got = np.stack(res["match_pts_list"])
    assert got.shape == g["match_pts"].shape and got.dtype == np.float32
    err = np.abs(got - g["match_pts"]).max()
    assert err <= 1e-5, err
    assert np.abs(got - g["true_pts"]).max() < 0.5 * np.abs(g["last_pts"] - g["true_pts"]).max()
    if use_graph in (True, "one launch per step"):
        assert f._tracker.single, "the one-launch step must be what ran"
        assert f._tracker.loop == (use_graph is True), "one launch for the whole frame / one per step"
    if use_graph:
        # the captured iteration is kept for the sequence: a second frame (other start points, shifted cameras) replays
        # it on fresh inputs and must equal the eager loop; then the first frame again
        tracker = f._tracker
        shifted = [p + np.float32(0.002) for p in g["last_pts"]]
        pose2 = torch.from_numpy(g["pose"]).clone()
        pose2[:, 0, 3] += 0.003
        f.curr_obs_torch["pose"] = pose2.to(dev)
        a = np.stack(f.rigid_tracking(info, shifted, None, n)["match_pts_list"])
        assert f._tracker is tracker
        f.use_hip_graph = False
        b = np.stack(f.rigid_tracking(info, shifted, None, n)["match_pts_list"])
        assert np.abs(a - b).max() <= 1e-5
        f.use_hip_graph = True
        f.curr_obs_torch["pose"] = torch.from_numpy(g["pose"]).to(dev)
        again = np.stack(f.rigid_tracking(info, [p for p in g["last_pts"]], None, n)["match_pts_list"])
        assert np.array_equal(again, got) and f._tracker is tracker


def test_tracking_step_kernels_vs_autograd(dev):
    """The three closed-form kernels of the tracking step against torch autograd on the same expressions: transform,
    loss gradients, and one Adam step of (t, w) from an upstream gradient -- incl. an instance with |w|^2 below the
    clamp (no gradient through the angle) and one far above it."""
    from d3fields_amd import rigid, _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    I, n, C = 4, 37, 48
    last = (torch.randn(I, n, 3, generator=g) * 0.2).to(dev)
    t0 = (torch.randn(I, 3, generator=g) * 0.05).to(dev)
    w0 = torch.tensor([[0.0, 0.0, 0.0], [0.004, -0.003, 0.002], [0.3, -0.2, 0.5], [1.5, 0.4, -0.9]], device=dev)
    stream = _lib.current_stream_handle(dev)

    # 1. transform
    pts = torch.empty(I * n, 3, device=dev)
    norms = torch.empty(2, device=dev)
    _lib.check(lib.d3f_rigid_transform(_lib.ptr(last), I, n, _lib.ptr(t0), _lib.ptr(w0), _lib.ptr(pts), _lib.ptr(norms), stream))
    ref_pts = rigid.rigid_transform(last, rigid.so3_exp_map(w0), t0).reshape(-1, 3)
    assert (pts - ref_pts).abs().max().item() <= 1e-6
    assert abs(norms[0].item() - t0.norm().item()) <= 1e-6 and abs(norms[1].item() - w0.norm().item()) <= 1e-6

    # 2. loss gradients
    N = I * n
    feats = torch.randn(N, C, generator=g).to(dev).requires_grad_(True)
    src = torch.randn(N, C, generator=g).to(dev)
    src[3] = feats[3].detach()                                   # zero difference: norm backward gives 0
    dist = (torch.randn(N, generator=g) * 0.01).to(dev).requires_grad_(True)
    valid = (torch.rand(N, generator=g) > 0.3).to(dev)
    feat_loss = (torch.norm(feats - src, dim=-1) * valid).mean()
    dist_loss = rigid.DIST_W * torch.clamp(dist * valid, min=0).mean()
    (feat_loss + dist_loss).backward()
    gf, gd, loss = torch.empty(N, C, device=dev), torch.empty(N, device=dev), torch.empty(2, device=dev)
    _lib.check(lib.d3f_track_loss_grad(_lib.ptr(feats.detach()), _lib.ptr(src), _lib.ptr(dist.detach()), _lib.ptr(valid), N, C,
                                       rigid.DIST_W, _lib.ptr(gf), _lib.ptr(gd), _lib.ptr(loss), stream))
    assert (gf - feats.grad).abs().max().item() <= 1e-7 and (gd - dist.grad).abs().max().item() <= 1e-7
    assert abs(loss[0].item() - feat_loss.item()) <= 1e-5 and abs(loss[1].item() - dist_loss.item()) <= 1e-6

    # 3. chain rule + regulariser + Adam, two consecutive steps
    t_ref, w_ref = t0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
    opt = torch.optim.Adam([t_ref, w_ref], lr=rigid.LR, betas=(0.9, 0.999))
    t_k, w_k = t0.clone(), w0.clone()
    state = torch.zeros(I * 13, device=dev)
    for it in range(2):
        up = torch.randn(N, 3, generator=g).to(dev)
        opt.zero_grad()
        p = rigid.rigid_transform(last, rigid.so3_exp_map(w_ref), t_ref).reshape(-1, 3)
        ((p * up).sum() + rigid.REG_W * (torch.norm(t_ref) + torch.norm(w_ref))).backward()
        opt.step()
        _lib.check(lib.d3f_rigid_transform(_lib.ptr(last), I, n, _lib.ptr(t_k), _lib.ptr(w_k), _lib.ptr(pts), _lib.ptr(norms), stream))
        _lib.check(lib.d3f_rigid_update(_lib.ptr(last), I, n, _lib.ptr(up), _lib.ptr(t_k), _lib.ptr(w_k), _lib.ptr(state[:I * 6]),
                                        _lib.ptr(state[I * 6:I * 12]), _lib.ptr(state[I * 12:]), _lib.ptr(norms), rigid.REG_W, rigid.LR,
                                        0.9, 0.999, 1e-8, stream))
        assert (t_k - t_ref.detach()).abs().max().item() <= 2e-6, it
        assert (w_k - w_ref.detach()).abs().max().item() <= 2e-6, it
    assert state[I * 12:].tolist() == [2.0] * I


def test_so3_exp_map_and_rigid_transform(dev):
    from scipy.spatial.transform import Rotation
    from d3fields_amd import rigid
    w = torch.tensor([[0.1, -0.3, 0.2], [0.0, 0.0, 0.0], [1e-3, 0.0, 0.0], [2.0, 1.0, -0.5]])
    R = rigid.so3_exp_map(w.to(dev))
    assert np.abs(cpu(R) - Rotation.from_rotvec(w.numpy()).as_matrix()).max() <= 1e-6
    x = torch.randn(4, 5, 3, generator=torch.Generator().manual_seed(0))
    t = torch.tensor([[1.0, 2.0, 3.0]]).expand(4, 3)
    got = rigid.rigid_transform(x.to(dev), R, t.to(dev))
    assert np.abs(cpu(got) - (torch.bmm(x, R.cpu()) + t[:, None, :]).numpy()).max() <= 1e-5


@pytest.mark.parametrize("V,H,W,fhw,C,NI", [(4, 60, 80, (6, 8), 384, 8), (3, 48, 64, (48, 64), 6, 3), (2, 40, 56, (5, 7), 5, 2),
                                             (8, 36, 64, (9, 16), 1024, 0)])
def test_fp16_stored_maps_equal_fp32_query_on_widened_maps(dev, V, H, W, fhw, C, NI):
    """D3F_DTYPE_F16: half is a storage format only.  The query on half maps must equal the query (and the oracle) on
    the same values widened to fp32: raw samples, dist and valid bit for bit, fused rows within TOL of the oracle --
    for 16-B, 8-B and scalar lane mappings, the C = 1024 wide kernel, and fp16 / fp32 maps mixed in one call."""
    from d3fields_amd import synth
    sc = synth.make_scene(V, H, W, "stress")
    feats16 = (synth.random_map(V, fhw[0], fhw[1], C, seed=1) * 2.0).half()
    maps = {"dino_feats": feats16}
    if NI:
        maps["mask"] = synth.random_onehot_mask(V, H, W, NI, seed=2)            # stays fp32: mixed call
    pts = torch.cat([synth.random_cloud(70000 if C <= 384 else 3000, seed=4) * 1.2, torch.from_numpy(
        np.array([[np.nan, 0, 0], [0, 0, 1e30], [0.0, 0.0, 0.0]], np.float32))])
    names = list(maps)
    f16 = make_fusion(dev, sc["depth"], sc["K"], sc["pose"], maps, H, W)
    wide = dict(maps, dino_feats=feats16.float())
    f32 = make_fusion(dev, sc["depth"], sc["K"], sc["pose"], wide, H, W)
    with torch.no_grad():
        a = f16.eval(pts.to(dev), return_names=names, return_inter=True)
        b = f32.eval(pts.to(dev), return_names=names, return_inter=True)
        a2 = f16.batch_eval(pts.to(dev), return_names=names)                     # fast (non-strict) path
        b2 = f32.batch_eval(pts.to(dev), return_names=names)
    assert f16.curr_obs_torch["dino_feats"].dtype == torch.float16 and a["dino_feats"].dtype == torch.float32
    for k in a:
        assert torch.equal(a[k], b[k]) or (torch.isnan(a[k]) == torch.isnan(b[k])).all() and torch.equal(
            torch.nan_to_num(a[k]), torch.nan_to_num(b[k])), k
    for k in a2:
        assert torch.equal(torch.nan_to_num(a2[k]), torch.nan_to_num(b2[k])), k
    sl = slice(0, 3000)
    ref = oracle_eval(sc, pts[sl].numpy(), [wide[k].numpy() for k in names], return_inter=True)
    assert np.array_equal(cpu(a["valid_mask"][sl]), ref["valid_mask"].astype(bool))
    for s_, k in enumerate(names):
        assert np.array_equal(cpu(a[k + "_inter"][:, sl]), ref["inter"][s_], equal_nan=True)
        assert rel_err(cpu(a2[k][sl]), ref["sets"][s_]) <= TOL
    # gradients w.r.t. the points: the backward kernel widens the same way; its dot products over the channels are
    # reduced by 16 instead of 32 lanes, so the sums agree to rounding, not bit for bit
    grads = []
    for f in (f16, f32):
        p = pts[:3000].to(dev).requires_grad_(True)
        o = f.eval(p, return_names=names)
        (sum(o[k].sum() for k in names) + o["dist"].sum()).backward()
        grads.append(p.grad)
    assert torch.equal(torch.isnan(grads[0]), torch.isnan(grads[1]))
    g0, g1 = cpu(torch.nan_to_num(grads[0])), cpu(torch.nan_to_num(grads[1]))
    assert np.abs(g0 - g1).max() <= 2e-5 * max(np.abs(g1).max(), 1.0)


def test_fusion_float16_is_a_storage_format(dev):
    """Fusion(dtype=float16): update() stores the channel maps in half, geometry stays fp32."""
    from d3fields_amd import Fusion
    V, H, W = 2, 40, 50
    from d3fields_amd import synth
    sc = synth.make_scene(V, H, W, "smooth")
    feats = synth.random_map(V, 4, 5, 16, seed=1)
    f = Fusion(num_cam=V, device=str(dev), dtype=torch.float16, feature_extractor=lambda color, params: feats)
    f.update({"color": np.zeros((V, H, W, 3), np.uint8), "depth": sc["depth"].numpy(), "pose": sc["pose"].numpy(), "K": sc["K"].numpy()})
    assert f.curr_obs_torch["dino_feats"].dtype == torch.float16 and f.curr_obs_torch["depth"].dtype == torch.float32
    out = f.eval(synth.random_cloud(500, seed=2).to(dev), return_names=["dino_feats", "color_tensor"])
    assert out["dino_feats"].dtype == torch.float32 and out["dist"].dtype == torch.float32
    with pytest.raises(NotImplementedError):
        Fusion(num_cam=V, device=str(dev), dtype=torch.bfloat16)
    # the grid entry point takes the half map too: same rows as batch_eval of the materialised grid
    from d3fields_amd import create_init_grid
    box = dict(x_lower=-0.2, x_upper=0.2, y_lower=-0.2, y_upper=0.2, z_lower=-0.1, z_upper=0.02)
    g = f.eval_grid(box, 0.02, return_names=["dino_feats"])
    pts, _ = create_init_grid(box, 0.02)
    b = f.batch_eval(pts.to(dev), return_names=["dino_feats"])
    assert torch.equal(g["dino_feats"].reshape(-1, 16), b["dino_feats"]) and torch.equal(g["dist"].reshape(-1), b["dist"])


@pytest.mark.parametrize("seed", range(6))
def test_random_configurations_vs_oracle(dev, seed):
    """Differential sweep: random view counts, image / map sizes, channel counts (16-B, 8-B, scalar lanes), strided map
    views, fp16 storage, truncation distances and tuning flags, with pathological points mixed in -- every output
    against the oracle (valid / dist / raw samples bit-exact for V <= 4, fused rows within TOL)."""
    from d3fields_amd import synth, _lib
    rng = np.random.default_rng(1000 + seed)

    def same_dist(got, ref, V):          # check_dist, tolerant of the NaN / Inf rows the pathological points produce
        assert np.array_equal(np.isfinite(got), np.isfinite(ref)) and np.array_equal(np.isnan(got), np.isnan(ref))
        fin = np.isfinite(ref)
        check_dist(got[fin], ref[fin], V)
        assert np.array_equal(got[~fin & ~np.isnan(ref)], ref[~fin & ~np.isnan(ref)])

    for _ in range(7):
        V = int(rng.integers(1, 7))
        H, W = int(rng.integers(8, 70)), int(rng.integers(8, 90))
        sc = synth.make_scene(V, H, W, "stress" if rng.random() < 0.5 else "smooth", seed=int(rng.integers(100)))
        nm = int(rng.integers(0, 4))
        maps, names = {}, []
        for j in range(nm):
            C = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 12, 16, 33, 64, 96, 384]))
            full = rng.random() < 0.4
            fh, fw = (H, W) if full else (int(rng.integers(1, 12)), int(rng.integers(1, 14)))
            t = synth.random_map(V, fh, fw, C + int(rng.integers(0, 3)) * 4, seed=int(rng.integers(1000)))[..., :C]   # maybe a strided view
            if rng.random() < 0.3:
                t = t.half()
            maps["m%d" % j] = t
            names.append("m%d" % j)
        n = int(rng.integers(1, 5000))
        pts = synth.random_cloud(n, seed=int(rng.integers(1000))) * float(rng.choice([0.5, 1.0, 3.0]))
        if n > 10:
            pts[int(rng.integers(n))] = float("nan")
            pts[int(rng.integers(n)), 2] = float("inf")
            pts[int(rng.integers(n))] = torch.tensor([0.0, 0.0, -0.6])           # at a camera height: |z| tiny in some views
        mu = float(rng.choice([0.02, 0.005, 0.1]))
        f = make_fusion(dev, sc["depth"], sc["K"], sc["pose"], maps, H, W, mu)
        f.tuning_flags = int(rng.choice([0, _lib.TUNE_FORCE_REORDER, _lib.TUNE_NO_REORDER, 3 << 8, 5 << 8]))
        with torch.no_grad():
            got = f.eval(pts.to(dev), return_names=names, return_inter=bool(rng.random() < 0.5))
            gd = f.eval_dist(pts.to(dev))
        wide = [maps[k].float().contiguous().numpy() for k in names]
        want_inter = any(k.endswith("_inter") for k in got)
        ref = oracle_eval(sc, pts.numpy(), wide, mu=mu, return_inter=want_inter)
        tag = "seed %d V=%d %dx%d maps=%s n=%d" % (seed, V, H, W, [tuple(m.shape[1:]) + (str(m.dtype),) for m in maps.values()], n)
        assert np.array_equal(cpu(got["valid_mask"]), ref["valid_mask"].astype(bool)), tag
        same_dist(cpu(got["dist"]), ref["dist"], V)
        for j, k in enumerate(names):
            assert rel_err(np.nan_to_num(cpu(got[k]), posinf=0, neginf=0), np.nan_to_num(ref["sets"][j], posinf=0, neginf=0)) <= TOL, tag
            assert np.array_equal(np.isnan(cpu(got[k])), np.isnan(ref["sets"][j])), tag
            if want_inter:
                assert np.array_equal(cpu(got[k + "_inter"]), ref["inter"][j], equal_nan=True), tag
        rd = oracle_eval(sc, pts.numpy(), [], mu=mu, mode="eval_dist")
        assert np.array_equal(cpu(gd["valid_mask"]), rd["valid_mask"].astype(bool)), tag
        same_dist(cpu(gd["dist"]), rd["dist"], V)


def test_cached_point_order_is_reused_and_harmless(dev):
    """cache_point_order: the Morton order of an unchanged query tensor stays in its scratch (D3F_FLAG_REUSE_POINT_ORDER).
    Same bits with and without the cache, across map updates, in-place changes of the points and a change of n."""
    from d3fields_amd import synth, _lib
    V, H, W = 4, 120, 160
    sc = synth.make_scene(V, H, W, "smooth")
    maps = {"dino_feats": synth.random_map(V, 12, 16, 96, seed=1)}
    f = make_fusion(dev, sc["depth"], sc["K"], sc["pose"], maps, H, W)
    f.tuning_flags = _lib.TUNE_FORCE_REORDER
    pts = synth.random_cloud(120000, seed=4).to(dev)
    with torch.no_grad():
        f.cache_point_order = False
        ref = f.batch_eval(pts, return_names=["dino_feats"])
        f.cache_point_order = True
        a = f.batch_eval(pts, return_names=["dino_feats"])           # builds and keeps the order
        held = f._order_ws
        b = f.batch_eval(pts, return_names=["dino_feats"])           # reuses it
        assert f._order_ws is held and held[2]
        for k in ref:
            assert torch.equal(ref[k], a[k]) and torch.equal(ref[k], b[k]), k
        f.curr_obs_torch["dino_feats"] = synth.random_map(V, 12, 16, 96, seed=9, device=dev)      # new frame, same grid
        c = f.batch_eval(pts, return_names=["dino_feats"])
        f.cache_point_order = False
        c_ref = f.batch_eval(pts, return_names=["dino_feats"])
        assert torch.equal(c["dino_feats"], c_ref["dino_feats"])
        f.cache_point_order = True
        pts.mul_(0.5)                                                 # in-place change: version bump -> rebuilt
        d = f.batch_eval(pts, return_names=["dino_feats"])
        assert f._order_ws is not held
        f.cache_point_order = False
        assert torch.equal(d["dino_feats"], f.batch_eval(pts, return_names=["dino_feats"])["dino_feats"])
        f.cache_point_order = True
        e = f.batch_eval(pts[:70000], return_names=["dino_feats"])   # other n (a view with the same data_ptr)
        assert torch.equal(e["dino_feats"], d["dino_feats"][:70000])
        # a caller passing D3F_FLAG_REUSE_POINT_ORDER with a scratch that holds garbage gets garbage rows, not a device fault
        f.cache_point_order = False
        f.tuning_flags = _lib.TUNE_FORCE_REORDER | _lib.FLAG_REUSE_POINT_ORDER
        torch.randint(0, 255, (64 << 20,), dtype=torch.uint8, device=dev)   # dirty the allocator's blocks
        f.batch_eval(pts, return_names=["dino_feats"])
        torch.cuda.synchronize()
        f.tuning_flags = _lib.TUNE_FORCE_REORDER
        again = f.batch_eval(pts, return_names=["dino_feats"])
        assert torch.equal(again["dino_feats"], d["dino_feats"])


def test_large_batch_indexing(dev):
    """16.8 M + 1 points in one launch (ragged last tile, Morton walk over > 2^24 indices): a strided sample of the rows
    equals the query of just those points, and the oracle."""
    from d3fields_amd import synth
    V, H, W = 4, 60, 80
    sc = synth.make_scene(V, H, W, "smooth")
    maps = {"dino_feats": synth.random_map(V, 6, 8, 4, seed=1)}
    f = make_fusion(dev, sc["depth"], sc["K"], sc["pose"], maps, H, W)
    n = (1 << 24) + 1
    g = torch.Generator(device=dev).manual_seed(5)
    pts = (torch.rand(n, 3, generator=g, device=dev) - 0.5) * torch.tensor([0.9, 0.8, 0.3], device=dev)
    from d3fields_amd import _lib
    for flags in (0, _lib.TUNE_FORCE_REORDER):
        f.tuning_flags = flags
        with torch.no_grad():
            out = f.batch_eval(pts, return_names=["dino_feats"])
            idx = torch.cat([torch.arange(0, n, 4099, device=dev), torch.tensor([n - 1], device=dev)])
            sub = f.eval(pts[idx].contiguous(), return_names=["dino_feats"])
        for k in sub:
            assert torch.equal(out[k][idx], sub[k]), (k, flags)
    ref = oracle_eval(sc, cpu(pts[idx]), [maps["dino_feats"].numpy()])
    assert np.array_equal(cpu(sub["valid_mask"]), ref["valid_mask"].astype(bool)) and np.array_equal(cpu(sub["dist"]), ref["dist"])
    assert rel_err(cpu(sub["dino_feats"]), ref["sets"][0]) <= TOL


def test_limits_max_views_and_max_maps(dev):
    """D3F_MAX_VIEWS = 64 views and D3F_MAX_MAPS = 8 channel maps in one call, against the oracle; one more of either is
    rejected with the documented error."""
    from d3fields_amd import synth
    V, H, W, N = 64, 24, 32, 3000
    sc = synth.make_scene(V, H, W, "stress")
    maps = {"m%d" % j: synth.random_map(V, 3 + j, 4 + j, [1, 2, 3, 4, 8, 12, 16, 40][j], seed=j) for j in range(8)}
    f = make_fusion(dev, sc["depth"], sc["K"], sc["pose"], maps, H, W)
    pts = synth.random_cloud(N, seed=2) * 1.5
    names = list(maps)
    with torch.no_grad():
        got = f.eval(pts.to(dev), return_names=names)
    ref = oracle_eval(sc, pts.numpy(), [maps[k].numpy() for k in names])
    assert np.array_equal(cpu(got["valid_mask"]), ref["valid_mask"].astype(bool))
    check_dist(cpu(got["dist"]), ref["dist"], V)
    for j, k in enumerate(names):
        assert rel_err(cpu(got[k]), ref["sets"][j]) <= TOL, k
    with pytest.raises(ValueError):
        f.curr_obs_torch["m8"] = maps["m0"]
        f.eval(pts.to(dev), return_names=names + ["m8"])
    sc65 = synth.make_scene(65, H, W, "stress")
    f65 = make_fusion(dev, sc65["depth"], sc65["K"], sc65["pose"], {}, H, W)
    from d3fields_amd import _lib
    with pytest.raises(_lib.D3FError) as e:
        f65.eval(pts.to(dev), return_names=[])
    assert e.value.code == _lib.ERR_BAD_SHAPE


def test_c_abi_from_cpp_host(dev, tmp_path):
    """examples/c_abi_demo.cpp: a host program with no Python and no torch drives d3f_eval through include/d3fields_hip.h
    (hipMalloc'd buffers, its own stream, the optional scratch); its inputs and outputs are re-checked with the oracle."""
    import shutil
    import subprocess
    from conftest import ROOT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe, blob = str(tmp_path / "c_abi_demo"), str(tmp_path / "c_abi_demo.bin")
    libdir = os.path.join(ROOT, "d3fields_amd")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_demo.cpp"),
                    "-L", libdir, "-ld3fields_hip", "-Wl,-rpath," + libdir, "-o", exe], check=True, capture_output=True, timeout=600)
    run = subprocess.run([exe, blob], check=True, capture_output=True, timeout=120, text=True)
    assert "valid" in run.stdout
    raw = open(blob, "rb").read()
    V, H, W, fh, fw, C, n, _ = np.frombuffer(raw[:32], np.int32)
    off = [32]

    def take(count, dt=np.float32):
        a = np.frombuffer(raw, dt, count, off[0])
        off[0] += a.nbytes
        return a

    K, pose = take(V * 9).reshape(V, 3, 3), take(V * 12).reshape(V, 3, 4)
    depth, feats = take(V * H * W).reshape(V, H, W), take(V * fh * fw * C).reshape(V, fh, fw, C)
    pts, dist = take(n * 3).reshape(n, 3), take(n)
    valid, fused = take(n, np.uint8), take(n * C).reshape(n, C)
    assert off[0] == len(raw)
    from oracle import c_oracle as O
    ref = O.eval_field(depth, K, pose, pts, [feats])
    assert 0.2 < valid.mean() < 0.95                       # the scene exercises valid and invalid points
    assert np.array_equal(valid.astype(bool), ref["valid_mask"].astype(bool))
    assert np.array_equal(dist, ref["dist"])
    assert rel_err(fused, ref["sets"][0]) <= TOL


def test_fails_loudly_without_gpu_tensors(dev):
    from d3fields_amd import synth
    V, H, W = 2, 32, 40
    sc = synth.make_scene(V, H, W, "smooth")
    f = make_fusion(dev, sc["depth"], sc["K"], sc["pose"], {}, H, W)
    with pytest.raises(RuntimeError):
        f.eval(torch.zeros(4, 3), return_names=[])
    with pytest.raises(KeyError):
        f.eval(torch.zeros(4, 3, device=dev), return_names=["nope"])
    with pytest.raises(AssertionError):
        f.eval(torch.zeros(4, 2, device=dev), return_names=[])


@pytest.mark.parametrize("N", [300, 70000])
def test_point_reordering_changes_nothing(dev, N):
    """The Morton walk (d3f_eval workspace) is a pure performance feature: bit-identical outputs."""
    from d3fields_amd import synth, _lib
    V, H, W = 4, 48, 64
    sc = synth.make_scene(V, H, W, "stress")
    feats = synth.random_map(V, 12, 16, 36, seed=1)
    mask = synth.random_onehot_mask(V, H, W, 4, seed=2)
    f = make_fusion(dev, sc["depth"], sc["K"], sc["pose"], {"dino_feats": feats, "mask": mask}, H, W)
    pts = synth.random_cloud(N, seed=8).to(dev)
    outs = []
    for flags in (_lib.TUNE_NO_REORDER, _lib.TUNE_FORCE_REORDER, _lib.TUNE_FORCE_REORDER | _lib.TUNE_XCD_REMAP | (5 << 8)):
        f.tuning_flags = flags
        f.reorder_points = True
        with torch.no_grad():
            # N=300 is below the shim's own threshold: call with an explicit workspace by lowering it
            import d3fields_amd.fusion as fm
            outs.append(f._run(pts, ["dino_feats", "mask"], True, "eval") if N >= 65536 else _eval_with_workspace(f, pts))
    for o in outs[1:]:
        for k in outs[0]:
            assert torch.equal(o[k], outs[0][k]), k
    ref = oracle_eval(sc, pts.cpu(), [feats, mask])
    assert np.array_equal(cpu(outs[1]["dist"]), ref["dist"])
    assert rel_err(cpu(outs[1]["dino_feats"]), ref["sets"][0]) <= TOL


def _eval_with_workspace(f, pts, names=("dino_feats", "mask"), want_inter=True):
    """Direct C-ABI call with a workspace for a batch below the shim's reorder threshold."""
    import ctypes
    from d3fields_amd import _lib
    lib = _lib.load()
    dev = pts.device
    n = pts.shape[0]
    views, keep, V = f._views(dev)
    names = list(names)
    maps = (_lib.ChannelMap * len(names))()
    fused = (ctypes.c_void_p * len(names))()
    inter = (ctypes.c_void_p * len(names))()
    out = {"dist": torch.empty(n, device=dev), "valid_mask": torch.empty(n, dtype=torch.bool, device=dev)}
    for s, k in enumerate(names):
        m = f.curr_obs_torch[k]
        maps[s] = _lib.ChannelMap(m.data_ptr(), m.shape[1], m.shape[2], m.shape[3], 0, m.stride(0), m.stride(1), m.stride(2))
        out[k] = torch.empty(n, m.shape[3], device=dev)
        fused[s] = out[k].data_ptr()
        if want_inter:
            out[k + "_inter"] = torch.empty(V, n, m.shape[3], device=dev)
            inter[s] = out[k + "_inter"].data_ptr()
    nb = lib.d3f_eval_workspace_bytes(n)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    finite = all(bool(torch.isfinite(f.curr_obs_torch[k]).all()) for k in names + ["depth"])
    flags = int(f.tuning_flags) | (_lib.FLAG_FINITE_MAPS if finite else 0)
    _lib.check(lib.d3f_eval(ctypes.byref(views), _lib.ptr(pts), n, maps, len(names), f.mu, flags, _lib.ptr(out["dist"]),
                            _lib.ptr(out["valid_mask"]), fused, inter if want_inter else None, _lib.ptr(ws), nb,
                            _lib.current_stream_handle(dev)))
    torch.cuda.synchronize()
    return out


# ---------------------------------------------------------------------------------------
# backward: d(loss)/d(pts) as the reference's autograd gives rigid_tracking
# ---------------------------------------------------------------------------------------
GRAD_TOL = 2e-5     # fp32 dot products over C channels in another order than autograd's; rel. to max |grad|


def test_backward_vs_reference_golden(dev):
    g = load_golden("grad_500")
    f = make_fusion(dev, g["depth"], g["K"], g["pose"], {"dino_feats": g["in_dino_feats"]}, g["H"], g["W"], float(g["mu"]))
    pts = torch.from_numpy(g["pts"]).to(dev).requires_grad_(True)
    out = f.eval(pts, return_names=["dino_feats"])
    assert out["dino_feats"].requires_grad and out["dist"].requires_grad and not out["valid_mask"].requires_grad
    (out["dino_feats"].sum() + out["dist"].sum()).backward()
    assert np.array_equal(cpu(out["dist"]), g["dist"])
    assert rel_err(cpu(pts.grad), g["grad_pts"]) <= GRAD_TOL


@pytest.mark.parametrize("kind,V,fhw,C,names", [
    ("smooth", 4, (12, 16), 384, ["dino_feats"]),
    ("stress", 3, (48, 64), 10, ["dino_feats", "mask", "color_tensor"]),
    ("smooth", 9, (6, 8), 7, ["dino_feats", "mask"]),
])
def test_backward_vs_torch_port_autograd(dev, kind, V, fhw, C, names):
    """Random upstream gradients (not just sum()) against autograd through the torch-ops port,
    which is itself pinned to the reference's gradient (test_torch_port_gradient_matches_reference)."""
    from d3fields_amd import synth
    from oracle import torch_port
    H, W, N = 48, 64, 3000
    sc = synth.make_scene(V, H, W, kind)
    maps = {"dino_feats": synth.random_map(V, fhw[0], fhw[1], C, seed=1), "mask": synth.random_onehot_mask(V, H, W, 5, seed=2),
            "color_tensor": torch.rand(V, H, W, 3, generator=torch.Generator().manual_seed(4))}
    f = make_fusion(dev, sc["depth"], sc["K"], sc["pose"], maps, H, W)
    pts = synth.random_cloud(N, seed=6)
    gen = torch.Generator().manual_seed(7)
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
    assert rel_err(cpu(p_gpu.grad), p_ref.grad.numpy()) <= GRAD_TOL
    # only-dist and only-feature losses (None upstream gradients reach the C ABI as NULL)
    p2 = pts.to(dev).requires_grad_(True)
    f.eval(p2, return_names=names)["dist"].sum().backward()
    p3 = pts.clone().requires_grad_(True)
    torch_port.field_query(obs, p3, names, H, W)["dist"].sum().backward()
    assert rel_err(cpu(p2.grad), p3.grad.numpy()) <= GRAD_TOL


def test_rigid_tracking_style_loop(dev):
    """A few Adam steps on a rigid translation, as fusion.py:1643-1665 does, run end to end on the HIP path."""
    from d3fields_amd import synth
    V, H, W = 4, 96, 128
    sc = synth.make_scene(V, H, W, "smooth")
    feats = synth.random_map(V, 12, 16, 32, seed=1)
    f = make_fusion(dev, sc["depth"], sc["K"], sc["pose"], {"dino_feats": feats}, H, W)
    src = synth.random_cloud(400, seed=9).to(dev)
    with torch.no_grad():
        tgt = f.eval(src, return_names=["dino_feats"])["dino_feats"]
    t = torch.tensor([0.004, -0.003, 0.002], device=dev, requires_grad=True)
    opt = torch.optim.Adam([t], lr=5e-4)
    losses = []
    for _ in range(25):
        opt.zero_grad()
        out = f.eval(src + t, return_names=["dino_feats"])
        loss = torch.norm(out["dino_feats"] - tgt, dim=1).mean() + 100 * torch.relu(out["dist"]).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0]
    with pytest.raises(NotImplementedError):
        f.eval(src.clone().requires_grad_(True), return_names=["dino_feats"], return_inter=True)


def test_eval_dist_backward_vs_torch_port(dev):
    from d3fields_amd import synth
    from oracle import torch_port
    V, H, W, N = 5, 48, 64, 4000
    sc = synth.make_scene(V, H, W, "stress")
    f = make_fusion(dev, sc["depth"], sc["K"], sc["pose"], {}, H, W)
    pts = synth.random_cloud(N, seed=6)
    wgt = torch.randn(N, generator=torch.Generator().manual_seed(7))
    p_ref = pts.clone().requires_grad_(True)
    (torch_port.dist_query(dict(sc), p_ref, H, W)["dist"] * wgt).sum().backward()
    p_gpu = pts.to(dev).requires_grad_(True)
    o = f.eval_dist(p_gpu)
    assert o["dist"].requires_grad and not o["valid_mask"].requires_grad
    (o["dist"] * wgt.to(dev)).sum().backward()
    assert rel_err(cpu(p_gpu.grad), p_ref.grad.numpy()) <= GRAD_TOL


# ---------------------------------------------------------------------------------------
# LDS-staged gather (patch-resolution maps on the Morton walk): must be bit-identical to the direct gather
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("C,fhw,N,box_scale,want_inter", [
    (384, (12, 16), 6000, 1.0, False),      # compact tiles: windows fit, one pass
    (384, (12, 16), 6000, 1.0, True),       # + '<k>_inter' (samples every view)
    (1024, (9, 12), 3000, 1.0, False),      # 4 passes of 256 channels
    (128, (24, 32), 4000, 1.0, False),      # one vector per lane
    (384, (12, 16), 300, 1.0, False),       # sparse cloud: windows overflow -> per-view direct fallback
    (100, (6, 8), 2500, 3.0, False),        # odd vector count, many points outside every image
])
def test_morton_walk_paths_are_bit_identical(dev, C, fhw, N, box_scale, want_inter):
    from d3fields_amd import synth, _lib
    V, H, W = 4, 120, 160
    sc = synth.make_scene(V, H, W, "smooth")
    feats = synth.random_map(V, fhw[0], fhw[1], C, seed=1)
    mask = synth.random_onehot_mask(V, H, W, 4, seed=2)
    f = make_fusion(dev, sc["depth"], sc["K"], sc["pose"], {"dino_feats": feats, "mask": mask}, H, W)
    pts = (synth.random_cloud(N, seed=8) * box_scale).to(dev)
    res = {}
    # caller order / Morton walk, each with the planner's fast paths and as the plain direct gather
    for tag, flags in (("direct", _lib.TUNE_NO_REORDER | _lib.TUNE_DIRECT_GATHER), ("staged", _lib.TUNE_FORCE_REORDER), ("unstaged", _lib.TUNE_FORCE_REORDER | _lib.TUNE_DIRECT_GATHER),
                       ("caller", _lib.TUNE_NO_REORDER)):
        f.tuning_flags = flags
        res[tag] = _eval_with_workspace(f, pts, ("dino_feats", "mask"), want_inter)
    for k in res["direct"]:
        assert torch.equal(res["staged"][k], res["direct"][k]), k
        assert torch.equal(res["unstaged"][k], res["direct"][k]), k
        assert torch.equal(res["caller"][k], res["direct"][k]), k
    ref = oracle_eval(sc, pts.cpu(), [feats, mask], return_inter=want_inter)
    assert np.array_equal(cpu(res["staged"]["dist"]), ref["dist"])
    assert rel_err(cpu(res["staged"]["dino_feats"]), ref["sets"][0]) <= TOL
    if want_inter:
        assert np.array_equal(cpu(res["staged"]["dino_feats_inter"]), ref["inter"][0])


def test_morton_walk_nonfinite_map(dev):
    from d3fields_amd import synth, _lib
    V, H, W = 3, 96, 128
    sc = synth.make_scene(V, H, W, "stress")
    feats = synth.random_map(V, 12, 16, 128, seed=1)
    feats[1, 3:9, 4:12, 2] = float("nan")
    mask = synth.random_onehot_mask(V, H, W, 4, seed=2)
    f = make_fusion(dev, sc["depth"], sc["K"], sc["pose"], {"dino_feats": feats, "mask": mask}, H, W)
    pts = synth.random_cloud(5000, seed=3).to(dev)
    f.tuning_flags = _lib.TUNE_FORCE_REORDER
    out = _eval_with_workspace(f, pts, ("dino_feats", "mask"), False)
    ref = oracle_eval(sc, pts.cpu(), [feats, mask])
    got, want = cpu(out["dino_feats"]), ref["sets"][0]
    assert np.isnan(want).any() and np.array_equal(np.isnan(got), np.isnan(want))
    assert rel_err(got[~np.isnan(want)], want[~np.isnan(want)]) <= TOL


# ---------------------------------------------------------------------------------------
# grids, shell pre-filter, farthest point sampling, select_features_rand
# ---------------------------------------------------------------------------------------
def _select_fusion(dev):
    g = load_golden("select_features")
    f = make_fusion(dev, g["depth"], g["K"], g["pose"], {"dino_feats": g["in_dino_feats"]}, g["H"], g["W"], float(g["mu"]))
    # the instance masks arrive the way the reference's drivers deliver them: through text_queries_* (producer injected)
    f.curr_obs_torch["color"] = np.zeros(g["depth"].shape + (3,), np.uint8)
    f.mask_producer = lambda fusion, queries, thresholds, boundaries, **kw: {
        "mask": g["in_mask"].argmax(-1).astype(np.uint8), "consensus_mask_label": ["background", "a", "b", "c"]}
    f.text_queries_for_inst_mask_no_track(["a", "b", "c"], [0.3], None)
    assert np.array_equal(cpu(f.curr_obs_torch["mask"]), g["in_mask"])
    box = dict(zip(["x_lower", "x_upper", "y_lower", "y_upper", "z_lower", "z_upper"], g["bounds"].tolist()))
    return g, f, box


def test_eval_grid_equals_batch_eval_of_materialised_grid(dev):
    from d3fields_amd import create_init_grid
    g, f, box = _select_fusion(dev)
    res = float(g["res"])
    with torch.no_grad():
        a = f.eval_grid(box, res, return_names=["mask", "dino_feats"])
        grid, shape = create_init_grid(box, res)
        b = f.batch_eval(grid.to(dev), return_names=["mask", "dino_feats"])
        d = f.eval_grid(box, res)
    assert tuple(a["grid_shape"]) == tuple(shape) == tuple(g["grid_shape"])
    for k in ("dist", "valid_mask", "mask", "dino_feats"):
        assert torch.equal(a[k], b[k]), k
    assert sorted(d.keys()) == ["dist", "grid_shape", "valid_mask"] and torch.equal(d["dist"], a["dist"])
    assert np.array_equal(cpu(a["dist"]), g["grid_dist"])                       # the reference's own batch_eval
    assert np.array_equal(np.packbits(cpu(a["valid_mask"])), g["grid_valid"])


def test_grid_shell_matches_reference_prefilter(dev):
    from d3fields_amd import create_init_grid
    g, f, box = _select_fusion(dev)
    idx, pts = f.grid_shell(box, float(g["res"]), 0.005)
    assert np.array_equal(cpu(idx), g["shell_index"])
    grid, _ = create_init_grid(box, float(g["res"]))
    assert np.array_equal(cpu(pts), grid.numpy()[g["shell_index"]])
    idx0, pts0 = f.grid_shell(box, float(g["res"]), 0.0)                          # empty shell
    assert idx0.numel() == 0 and pts0.shape == (0, 3)


def test_fps_matches_reference(dev):
    from d3fields_amd import fps
    g = load_golden("select_features")
    p, i, d = fps(g["fps_cloud"], 64, init_idx=17)                               # numpy in / numpy out like fps_np
    assert isinstance(p, np.ndarray) and i == g["fps_idx"].tolist() and np.array_equal(p, g["fps_pts"])
    assert d == float(g["fps_maxdist"])
    pt, it, _ = fps(torch.from_numpy(g["fps_cloud"]).to(dev), 64, init_idx=17)
    assert np.array_equal(cpu(it), g["fps_idx"]) and np.array_equal(cpu(pt), g["fps_pts"])
    big = np.random.default_rng(5).normal(size=(200000, 3)).astype(np.float32)
    from oracle import c_oracle as O
    _, ib, db = fps(big, 50, init_idx=0)
    io, do = O.fps(big, 50, 0)
    assert ib == io.tolist() and db == do
    ps, isel, md = fps(big[:10], 64, init_idx=3)                                  # cloud smaller than the request: fps_np does not
    io, do = O.fps(big[:10], 64, 3)                                               # stop at n, it appends index 0 (distance 0 everywhere)
    assert len(isel) == 64 and isel == io.tolist() and sorted(set(isel)) == list(range(10)) and isel[10:] == [0] * 54 and md == do == 0.0
    # more workgroups than one, a ragged last chunk, exact ties (duplicated points): first maximum wins
    tied = np.tile(big[:1000], (301, 1))[:300017]
    _, it2, dt2 = fps(tied, 40, init_idx=5)
    io2, do2 = O.fps(tied, 40, 5)
    assert it2 == io2.tolist() and dt2 == do2


def test_select_features_rand_matches_reference(dev):
    g, f, box = _select_fusion(dev)
    feats_l, pts_l, imgs = f.select_features_rand(box, int(g["N"]), per_instance=True, res=float(g["res"]), init_idx=0)
    assert len(pts_l) == int(g["n_inst"]) == len(feats_l) and imgs == []
    for i in range(len(pts_l)):
        assert np.array_equal(pts_l[i], g["sel_pts_%d" % i]), i
        assert rel_err(cpu(feats_l[i]), g["sel_feats_%d" % i]) <= TOL


def test_select_features_from_pcd_matches_reference(dev):
    g, f, _ = _select_fusion(dev)
    feats_l, pts_l, imgs = f.select_features_from_pcd(g["pcd_cloud"], 16, per_instance=True, init_idx=0)
    assert len(pts_l) == int(g["pcd_n_inst"]) == len(feats_l) and imgs == []
    for i in range(len(pts_l)):
        assert np.array_equal(pts_l[i], g["pcd_pts_%d" % i]), i
        assert rel_err(cpu(feats_l[i]), g["pcd_feats_%d" % i]) <= TOL
    with pytest.raises(NotImplementedError):
        f.select_features_from_pcd(g["pcd_cloud"], 16, vis=True)


def test_fast_path_against_strict_path(dev):
    """Fast path (device-checked finite maps: invalid-view skip, weight-zero padding, precomputed corner set-up; wide maps
    with the FOLDED weights) against the strict path (every view sampled, value selects, the
    reference's operation order, IEEE '/'), on a full-size feature width:
      * 'dist', 'valid_mask' and the thin map (instance mask) are identical bits;
      * the wide map agrees to a few ulp of its largest term (contract 1e-5 of max|ref|, asserted 2e-6);
      * with Fusion.reference_rounding (D3F_FLAG_REFERENCE_ROUNDING) the wide map is identical bits too."""
    import ctypes
    from d3fields_amd import synth, _lib
    V, H, W = 4, 120, 160
    sc = synth.make_scene(V, H, W, "stress")
    feats = synth.random_map(V, 12, 16, 384, seed=1) * 3.0
    mask = synth.random_onehot_mask(V, H, W, 8, seed=2)
    f = make_fusion(dev, sc["depth"], sc["K"], sc["pose"], {"dino_feats": feats, "mask": mask}, H, W)
    pts = (synth.random_cloud(30000, seed=8) * 1.3).to(dev)
    with torch.no_grad():
        fast = f.eval(pts)                                  # the shim's device words say "finite": fast path
        f.reference_rounding = True
        exact = f.eval(pts)
        f.reference_rounding = False
    assert f.maps_are_finite()
    lib = _lib.load()
    views, keep, _ = f._views(dev)
    names = ["dino_feats", "mask"]
    maps = (_lib.ChannelMap * 2)()
    fused = (ctypes.c_void_p * 2)()
    strict = {"dist": torch.empty_like(fast["dist"]), "valid_mask": torch.empty_like(fast["valid_mask"])}
    for s, k in enumerate(names):
        m = f.curr_obs_torch[k]
        maps[s] = _lib.ChannelMap(m.data_ptr(), m.shape[1], m.shape[2], m.shape[3], 0, m.stride(0), m.stride(1), m.stride(2))
        strict[k] = torch.empty_like(fast[k])
        fused[s] = strict[k].data_ptr()
    _lib.check(lib.d3f_eval(ctypes.byref(views), _lib.ptr(pts), pts.shape[0], maps, 2, f.mu, 0, _lib.ptr(strict["dist"]),
                            _lib.ptr(strict["valid_mask"]), fused, None, None, 0, _lib.current_stream_handle(dev)))
    torch.cuda.synchronize()
    for k in ("dist", "valid_mask", "mask"):
        assert torch.equal(fast[k], strict[k]), k
    for k in fast:
        assert torch.equal(exact[k], strict[k]), ("reference_rounding", k)
    ref = strict["dino_feats"]
    err = float((fast["dino_feats"] - ref).abs().max() / max(float(ref.abs().max()), 1.0))
    assert 0.0 < err <= 2e-6, err            # > 0: the folded form really ran


# ---------------------------------------------------------------------------------------
# point clouds on the mask side: fp64, numpy in / numpy out like the reference
# ---------------------------------------------------------------------------------------
def test_pcd_utils_match_reference(dev):
    from d3fields_amd import pcd_utils
    g = load_golden("pcd_utils")
    box = dict(zip(["x_lower", "x_upper", "y_lower", "y_upper", "z_lower", "z_upper"], g["bounds"].tolist()))
    pts, col = pcd_utils.aggr_point_cloud_from_data(g["colors"], g["depths"], g["K"], g["pose44"], downsample=False,
                                                    masks=g["masks"], boundaries=box, out_o3d=False)
    assert pts.dtype == np.float64 and pts.shape == g["crop_pts"].shape
    assert np.allclose(pts, g["crop_pts"], rtol=0, atol=1e-12) and np.array_equal(col, g["crop_col"])
    pts2, col2 = pcd_utils.aggr_point_cloud_from_data(g["colors"], g["depths"], g["K"], g["pose44"], downsample=False,
                                                      masks=None, boundaries=None, out_o3d=False)
    assert pts2.shape == g["all_pts"].shape and np.allclose(pts2, g["all_pts"], rtol=0, atol=1e-12)
    assert np.array_equal(col2, g["all_col"])
    K = g["K"][1]
    fg = pcd_utils.depth2fgpcd(g["depths"][1], g["masks"][1], [K[0, 0], K[1, 1], K[0, 2], K[1, 2]])
    assert fg.shape == g["fg_view1"].shape and np.allclose(fg, g["fg_view1"], rtol=0, atol=1e-13)
    # the reference's DEFAULT arguments (downsample=True, out_o3d=True): a point-cloud object holding the per-view 1-cm
    # voxel-grid means (open3d's voxel_down_sample restated: tests/test_gpu_callers.py::test_voxel_downsample_...)
    from oracle import np_pcd
    cloud = pcd_utils.aggr_point_cloud_from_data(g["colors"], g["depths"], g["K"], g["pose44"])
    per_view = [pcd_utils.aggr_point_cloud_from_data(g["colors"][v:v + 1], g["depths"][v:v + 1], g["K"][v:v + 1], g["pose44"][v:v + 1],
                                                     downsample=False, out_o3d=False)[0] for v in range(g["depths"].shape[0])]
    want = np.concatenate([np_pcd.voxel_mean(p, 0.01) for p in per_view], axis=0)
    assert np.abs(np.asarray(cloud.points) - want).max() <= 1e-12


def test_pcd_iou_matches_reference(dev):
    from d3fields_amd import Fusion, pcd_utils
    g = load_golden("pcd_utils")
    out = pcd_utils.pcd_iou(g["p1"], g["p2"], 0.005)
    assert np.allclose(np.array(out[:3], dtype=np.float64), g["iou"], rtol=0, atol=1e-15)
    assert np.array_equal(out[3], g["overlap_1"]) and np.array_equal(out[4], g["overlap_2"])
    assert np.array_equal(out[5], g["idx_12"]) and np.array_equal(out[6], g["idx_21"])
    out2 = Fusion(num_cam=1, device=str(dev)).pcd_iou(g["p1"], g["p2"], 0.005)
    assert out2[0] == out[0]
    # larger clouds against the numpy restatement
    from oracle import np_pcd
    rng = np.random.default_rng(2)
    a, b = rng.normal(size=(7000, 3)), rng.normal(size=(5000, 3))
    o = pcd_utils.pcd_iou(a, b, 0.05)
    md, am = np_pcd.nearest(a, b)
    assert np.array_equal(o[5], am) and np.array_equal(o[3], np.where(md < 0.05)[0])


def test_config4_shard_properties(dev):
    """BASELINE config 4, one GPU's shard: 8 views x 720x1280, 72x128x1024 features, 1 000 000 points
    (fused_eval_wide_kernel on the Morton walk): size-independent properties + an oracle sample."""
    from d3fields_amd import synth, sharding
    V, H, W, C, N = 8, 720, 1280, 1024, 1000000
    sc = synth.make_scene(V, H, W, "smooth")
    feats = synth.random_map(V, 72, 128, C, seed=1, device=dev)
    f = make_fusion(dev, sc["depth"], sc["K"], sc["pose"], {"dino_feats": feats}, H, W)
    pts = synth.random_cloud(N, seed=3).to(dev)
    with torch.no_grad():
        out = f.batch_eval(pts, return_names=["dino_feats"])
        # the shard of rank 3 of 8 evaluated alone == the same rows of the full batch (what the all-gather reassembles)
        lo, hi = sharding.shard_bounds(N, 3, 8)
        part = f.batch_eval(pts[lo:hi], return_names=["dino_feats"])
        for k in ("dist", "valid_mask", "dino_feats"):
            assert torch.equal(part[k], out[k][lo:hi]), k
        inv = ~out["valid_mask"]
        assert torch.equal(out["dist"] == 1e3, inv) and (out["dino_feats"][inv] == 0).all()
    pick = torch.randperm(N, generator=torch.Generator().manual_seed(5))[:1500]
    ref = oracle_eval(sc, pts[pick.to(dev)].cpu(), [feats.cpu()])
    assert np.array_equal(cpu(out["valid_mask"][pick.to(dev)]), ref["valid_mask"])
    assert rel_err(cpu(out["dist"][pick.to(dev)]), ref["dist"]) <= TOL          # V = 8: see check_dist
    assert rel_err(cpu(out["dino_feats"][pick.to(dev)]), ref["sets"][0]) <= TOL
