"""GPU parity of the two launch paths added in round 2 -- the closed-form lattice brick walk (grids on large maps) and
the cell-run gather (patch-resolution wide maps) -- and of the BENCH workloads themselves.  Both paths only change the
ORDER in which points are processed / corner texels are fetched, never an arithmetic operation, so the bar is
bit-identity with the caller-order direct gather (torch.equal), which the other GPU tests pin to the oracle and the
reference goldens; a sample of each full-size result is compared with the CPU oracle as well."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def cpu(x):
    return x.detach().cpu().numpy()


def experiments():
    """True when the library was built with D3F_BUILD_EXPERIMENTS=1: the D3F_EXP_* environment knobs select kernel
    variants.  The product build reads no environment variable at all."""
    from d3fields_amd import _lib
    return bool(_lib.load().d3f_build_has_experiments())


class knobs:
    """Selects a launch variant for the duration of the block.  The switch-offs the bit-identity tests use as their
    reference -- D3F_EXP_RUNS=-1 / D3F_EXP_WINDOW=-1 / D3F_EXP_THIN=-1, alone or together -- are the tuning flag
    D3F_TUNE_DIRECT_GATHER (the plain direct gather), which every build honours.  Any other knob is an environment
    variable that only an experiments build reads; the product build then simply runs its default launch."""
    REFERENCE = {"D3F_EXP_RUNS": "-1", "D3F_EXP_WINDOW": "-1", "D3F_EXP_THIN": "-1"}

    def __init__(self, FLAGS=0, **kv):
        self.kv = {k: str(v) for k, v in kv.items()}
        self.flags = int(FLAGS)                  # D3F_TUNE_* bits every build honours (e.g. TUNE_WINDOW_SIDE, TUNE_NO_WINDOW_GATE)
        self.direct = bool(self.kv) and all(self.REFERENCE.get(k) == v for k, v in self.kv.items())

    def __enter__(self):
        from d3fields_amd import Fusion, _lib
        self.old_flags = Fusion.extra_tuning_flags
        Fusion.extra_tuning_flags |= self.flags
        if self.direct:
            Fusion.extra_tuning_flags |= _lib.TUNE_DIRECT_GATHER
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        from d3fields_amd import Fusion
        Fusion.extra_tuning_flags = self.old_flags
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def sweep(variants, keep=2):
    """every variant in an experiments build; the first `keep` (the reference and the default launch) otherwise"""
    return list(variants) if experiments() else list(variants)[:keep]


def fusion_for(dev, V, H, W, maps, kind="smooth"):
    from d3fields_amd import Fusion, synth
    sc = synth.make_scene(V, H, W, kind)
    f = Fusion(num_cam=V, device=str(dev))
    f.curr_obs_torch = {k: sc[k].to(dev) for k in ("depth", "K", "pose")}
    f.curr_obs_torch.update(maps)
    f.H, f.W = H, W
    return f, sc


def oracle_sample(sc, pts, maps, mu=0.02):
    from oracle import c_oracle as O
    return O.eval_field(sc["depth"], sc["K"], sc["pose"], pts, [m.float().cpu() for m in maps], mu=mu)


def box_for(nx, ny, nz, step):
    """boundaries whose create_init_grid has exactly (nx, ny, nz) points around the scene centre"""
    return dict(x_lower=-nx * step / 2, x_upper=nx * step / 2 - step / 4, y_lower=-ny * step / 2, y_upper=ny * step / 2 - step / 4,
                z_lower=-0.2, z_upper=-0.2 + nz * step - step / 4)


# ---- lattice probe --------------------------------------------------------------------------------------------------
def test_lattice_probe(dev):
    from d3fields_amd import Fusion, create_init_grid, synth
    f = Fusion(num_cam=1, device=str(dev))
    f.cache_point_order = False                 # every call below probes (the cache is keyed by storage address)
    s = torch.cuda.current_stream(dev).cuda_stream
    import ctypes
    st = ctypes.c_void_p(s)
    for dims in [(40, 35, 11), (3, 5, 7000), (1, 300, 300), (129, 2, 300), (160, 140, 44)]:
        grid, shape = create_init_grid(box_for(*dims, 0.004), 0.004)
        assert tuple(shape) == dims
        assert f._lattice_dims(grid.to(dev), st) == dims
    grid = create_init_grid(box_for(40, 35, 11, 0.01), 0.01)[0]
    assert f._lattice_dims(synth.random_cloud(100000, seed=1).to(dev), st) is None
    assert f._lattice_dims(grid[torch.randperm(grid.shape[0])].to(dev), st) is None            # shuffled grid
    bent = grid.clone(); bent[7777, 1] += 1e-6
    assert f._lattice_dims(bent.to(dev), st) in (None, (40, 35, 11))                            # spot check: either is harmless
    assert f._lattice_dims(grid[:-1].to(dev), st) is None                                       # ragged
    assert f._lattice_dims(grid[:, [2, 1, 0]].contiguous().to(dev), st) is None                 # x fastest: not this layout
    assert f._lattice_dims(torch.zeros(70000, 3, device=dev), st) is None


def test_points_probe_equals_the_two_probes(dev):
    """d3f_points_probe (ABI 6: lattice + locality in ONE launch, nothing to clear beforehand) gives the verdicts of
    d3f_lattice_probe and d3f_point_order_locality on grids, clouds, a surface-like cloud and degenerate inputs -- also when its
    output buffer holds garbage before the call."""
    import ctypes
    from d3fields_amd import Fusion, create_init_grid, synth, _lib
    lib = _lib.load()
    st = _lib.current_stream_handle(dev)
    grid = create_init_grid(box_for(40, 35, 11, 0.01), 0.01)[0]
    surface = grid[(grid[:, 2] - 0.3 * grid[:, 0]).abs() < 0.004]                 # a sheet of the lattice in flat-index order: local, no lattice
    cases = {"grid": grid, "cloud": synth.random_cloud(100000, seed=1), "shuffled": grid[torch.randperm(grid.shape[0])], "surface": surface,
             "tall": create_init_grid(box_for(3, 5, 7000, 0.004), 0.004)[0], "two": grid[:2], "one": grid[:1], "nan": torch.full((5000, 3), float("nan"))}
    for name, p in cases.items():
        p = p.to(dev).contiguous()
        n = p.shape[0]
        old_l = torch.zeros(4, dtype=torch.int32, device=dev)
        old_o = torch.zeros(2, dtype=torch.float32, device=dev)
        _lib.check(lib.d3f_lattice_probe(_lib.ptr(p), n, _lib.ptr(old_l), st))
        _lib.check(lib.d3f_point_order_locality(_lib.ptr(p), n, _lib.ptr(old_o), st))
        li = old_l.tolist()
        want_dims = tuple(li[:3]) if (li[0] > 0 and li[3] == 0) else None
        near, far = old_o.tolist()
        new = torch.full((_lib.PROBE_WORDS,), 0x7fc00001, dtype=torch.int32, device=dev)        # garbage (NaN bit patterns) beforehand
        _lib.check(lib.d3f_points_probe(_lib.ptr(p), n, _lib.ptr(new), st))
        dims, unordered = Fusion._parse_probe(new.cpu())
        assert dims == want_dims, (name, dims, want_dims)
        assert unordered == bool(near > 0.25 * far), (name, near, far)
    assert Fusion._parse_probe(torch.zeros(_lib.PROBE_WORDS, dtype=torch.int32)) == (None, False)
    assert lib.d3f_points_probe(None, 5, None, st) == _lib.ERR_INVALID_ARG


# ---- lattice brick walk == caller order, bit for bit -------------------------------------------------------------------
@pytest.mark.parametrize("dims,C,mask", [((64, 33, 37), 96, False), ((47, 53, 29), 384, True), ((130, 9, 61), 132, False),
                                         ((2, 2, 20000), 64, True), ((70, 70, 17), 1024, False)])
def test_lattice_walk_is_bit_identical(dev, dims, C, mask):
    from d3fields_amd import create_init_grid, synth, _lib
    V, H, W = 4, 96, 128
    maps = {"dino_feats": synth.random_map(V, H, W, C, seed=1, device=dev)}          # dense maps: texel = pixel
    names = ["dino_feats"]
    if mask:
        maps["mask"] = synth.random_onehot_mask(V, H, W, 8, seed=2, device=dev)
        names.append("mask")
    f, sc = fusion_for(dev, V, H, W, maps)
    box = box_for(*dims, 0.004)
    grid, shape = create_init_grid(box, 0.004)
    assert tuple(shape) == dims and grid.shape[0] >= 65536
    pts = grid.to(dev)
    with torch.no_grad():
        f.tuning_flags = _lib.TUNE_NO_REORDER
        base = f.batch_eval(pts, return_names=names)                                  # caller order, direct gather
        f.tuning_flags = _lib.TUNE_FORCE_REORDER                                      # maps are small here: force the walk
        walk = f.batch_eval(pts, return_names=names)
        viag = f.eval_grid(box, 0.004, return_names=names)                            # axes instead of the point array
        f.detect_lattice = False
        sort = f.batch_eval(pts, return_names=names)                                  # Morton sort of the same points
        f.detect_lattice = True
    for k in ["dist", "valid_mask"] + names:
        assert torch.equal(walk[k], base[k]), k
        assert torch.equal(viag[k], base[k]), k
        assert torch.equal(sort[k], base[k]), k
    pick = torch.randperm(pts.shape[0], generator=torch.Generator().manual_seed(3))[:1500]
    ref = oracle_sample(sc, grid[pick], [maps[k] for k in names])
    assert np.array_equal(cpu(walk["dist"])[pick], ref["dist"]) and np.array_equal(cpu(walk["valid_mask"])[pick], ref["valid_mask"])
    assert rel_err(cpu(walk["dino_feats"])[pick], ref["sets"][0]) <= TOL


def test_lattice_walk_return_inter_and_nonfinite(dev):
    """'<k>_inter' outputs and non-finite maps / points take the strict path on the walk as in caller order."""
    from d3fields_amd import create_init_grid, synth, _lib
    V, H, W = 3, 64, 80
    feats = synth.random_map(V, H, W, 40, seed=1, device=dev)
    feats[1, 20:30, 30:50, 3] = float("nan")
    f, sc = fusion_for(dev, V, H, W, {"dino_feats": feats})
    grid = create_init_grid(box_for(50, 40, 36, 0.006), 0.006)[0]
    grid[12345] = float("nan")
    pts = grid.to(dev)
    with torch.no_grad():
        f.tuning_flags = _lib.TUNE_NO_REORDER
        base = f.eval(pts, return_names=["dino_feats"], return_inter=True)
        f.tuning_flags = _lib.TUNE_FORCE_REORDER
        walk = f.eval(pts, return_names=["dino_feats"], return_inter=True)
    for k in base:
        a, b = walk[k], base[k]
        assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float())), k
    assert torch.isnan(base["dino_feats"]).any()


# ---- cell-run gather == direct gather, bit for bit ---------------------------------------------------------------------
@pytest.mark.parametrize("C,V,fhw,mask,points", [(384, 4, (48, 64), False, "grid"), (384, 4, (48, 64), True, "grid"),
                                                 (512, 8, (24, 32), False, "cloud"), (132, 3, (12, 16), True, "grid"),
                                                 (128, 4, (48, 64), False, "cloud")])
def test_cell_run_gather_is_bit_identical(dev, C, V, fhw, mask, points):
    from d3fields_amd import create_init_grid, synth, _lib
    H, W = 480, 640
    maps = {"dino_feats": synth.random_map(V, fhw[0], fhw[1], C, seed=1, device=dev)}
    names = ["dino_feats"]
    if mask:
        maps["mask"] = synth.random_onehot_mask(V, H, W, 8, seed=2, device=dev)
        names.append("mask")
    f, sc = fusion_for(dev, V, H, W, maps)
    if points == "grid":
        pts_c = create_init_grid(synth.WORK_BOX, 0.0107)[0]                            # 74 x 65 x 20 points
    else:
        pts_c = synth.random_cloud(150001, seed=3)
    pts_c[1000, 1] = float("inf")                                                       # a strict point inside a run
    pts = pts_c.to(dev)
    import ctypes
    plan = _lib.EvalPlan()
    views, keep, _ = f._views(dev)
    m = maps["dino_feats"]
    cm = (_lib.ChannelMap * 1)(_lib.ChannelMap(m.data_ptr(), m.shape[1], m.shape[2], C, 0, m.stride(0), m.stride(1), m.stride(2)))
    _lib.check(f._lib.d3f_eval_plan_query(ctypes.byref(views), pts.shape[0], cm, 1, _lib.FLAG_FINITE_MAPS, 1, 0, ctypes.byref(plan)))
    assert plan.staged[0] > 16, "the cell-run gather must be what runs here"
    variants = (("direct", dict(D3F_EXP_RUNS=-1)), ("auto", dict(D3F_EXP_RUNS=0)), ("u1k8", dict(D3F_EXP_RUNS_U=1)),
                ("u1k8occ4", dict(D3F_EXP_RUNS_U=1, D3F_EXP_RUNS=8, D3F_EXP_RUNS_OCC=4)), ("u1k8occ6", dict(D3F_EXP_RUNS_U=1, D3F_EXP_RUNS=8, D3F_EXP_RUNS_OCC=6)),
                ("u1k4", dict(D3F_EXP_RUNS_U=1, D3F_EXP_RUNS=4)), ("u1k8", dict(D3F_EXP_RUNS_U=1, D3F_EXP_RUNS=8)), ("u3k2", dict(D3F_EXP_RUNS_U=3, D3F_EXP_RUNS=2)),
                ("u3k4", dict(D3F_EXP_RUNS_U=3, D3F_EXP_RUNS=4)), ("u2k4", dict(D3F_EXP_RUNS_U=2, D3F_EXP_RUNS=4)),
                ("u2k8", dict(D3F_EXP_RUNS_U=2, D3F_EXP_RUNS=8)), ("u2k8occ4", dict(D3F_EXP_RUNS_U=2, D3F_EXP_RUNS=8, D3F_EXP_RUNS_OCC=4)))
    variants = sweep(variants)
    with torch.no_grad():
        outs = {}
        for tag, env in variants:
            with knobs(**env):
                outs[tag] = f.batch_eval(pts, return_names=names)
    for tag in [t for t, _ in variants[1:]]:
        for k in outs["direct"]:
            a, b = outs[tag][k], outs["direct"][k]
            assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float())), (tag, k)
    pick = torch.randperm(pts.shape[0], generator=torch.Generator().manual_seed(3))[:1500]
    pick = pick[pick != 1000]
    ref = oracle_sample(sc, pts_c[pick], [maps[k] for k in names])
    assert np.array_equal(cpu(outs["auto"]["dist"])[pick], ref["dist"])
    assert rel_err(cpu(outs["auto"]["dino_feats"])[pick], ref["sets"][0]) <= TOL


def test_cell_run_gather_not_used_when_it_must_not(dev):
    """'<k>_inter', non-finite maps and fp16-stored maps keep the direct gather (the plan says so) and the results
    still agree with the run path's on the same finite fp32 data."""
    from d3fields_amd import synth, _lib
    import ctypes
    V, H, W, C = 4, 480, 640, 384
    feats = synth.random_map(V, 48, 64, C, seed=1, device=dev)
    f, sc = fusion_for(dev, V, H, W, {"dino_feats": feats})
    views, keep, _ = f._views(dev)
    cm = (_lib.ChannelMap * 1)(_lib.ChannelMap(feats.data_ptr(), 48, 64, C, 0, feats.stride(0), feats.stride(1), feats.stride(2)))
    plan = _lib.EvalPlan()
    for flags, inter, want in ((_lib.FLAG_FINITE_MAPS, 0, 16 + 4), (0, 0, 0), (_lib.FLAG_FINITE_MAPS, 1, 0)):
        _lib.check(f._lib.d3f_eval_plan_query(ctypes.byref(views), 200000, cm, 1, flags, 1, inter, ctypes.byref(plan)))
        assert plan.staged[0] == want, (flags, inter)
    _lib.check(f._lib.d3f_eval_plan_query(ctypes.byref(views), 3000, cm, 1, _lib.FLAG_FINITE_MAPS, 1, 0, ctypes.byref(plan)))
    assert plan.staged[0] == 0                                                          # small batches: many small tiles instead
    pts = synth.random_cloud(100000, seed=5).to(dev)
    with torch.no_grad():
        a = f.eval(pts, return_names=["dino_feats"])
        b = f.eval(pts, return_names=["dino_feats"], return_inter=True)      # '<k>_inter': the strict path, reference order
        f.reference_rounding = True
        c = f.eval(pts, return_names=["dino_feats"])
        f.reference_rounding = False
    assert torch.equal(c["dino_feats"], b["dino_feats"])
    assert float((a["dino_feats"] - b["dino_feats"]).abs().max()) <= 2e-6 * max(float(b["dino_feats"].abs().max()), 1.0)   # folded weights


# ---- the BENCH workloads themselves ---------------------------------------------------------------
@pytest.mark.parametrize("workload,points", [("c2_dense", "grid"), ("c3_dense", "grid"), ("c2_patch", "grid"), ("c3_patch", "grid"),
                                             ("c4_patch", "grid"), ("c4_patch", "random"), ("ref_patch", "grid"), ("dist_only", "grid"),
                                             ("c5_track", "grid"), ("c2_patch", "random"), ("c3_patch", "random"), ("ref_patch", "surface"),
                                             ("c3_patch", "surface")])
def test_bench_workload_matches_oracle(dev, workload, points):
    """Exactly what bench.py times (same builder, same launch geometry: lattice walk with 8- / 16-point tiles on the
    dense maps, bricks through LDS texel windows on the patch-resolution maps -- config 4's x-slab of the 8 M-point lattice
    included --, cell runs on config 4's cloud) against the CPU oracle on a 3000-point sample, plus bit-identity with the
    caller-order direct gather on 200 000 points."""
    import bench
    from d3fields_amd import _lib
    f, pts, names, w, sc = bench.build_workload(workload, dev, 0, 1, points)
    if points == "surface":       # the mesh-vertex cloud of vis_repr.py:97-103: a two-voxel shell around the table and the spheres
        assert 30000 < pts.shape[0] < w["N"] // 10
    else:
        assert pts.shape[0] == (w["N"] if points == "grid" else w.get("N_cloud", w["N"]))
    with torch.no_grad():
        f.record_plans = True
        out = f.batch_eval(pts, return_names=names)
        if points == "random" and workload in ("c2_patch", "c3_patch"):      # a dense cloud: the device-side gate opens the window side
            assert f.last_plan()["gated_window"] and f.last_gate()[1], (f.last_plan(), f.last_gate())
        if workload == "c4_patch":                                           # 8 views x 1024 channels: the rows in registers, lattice and cloud
            assert f.last_plan()["family"] == "register-rows" and not f.last_plan()["gated_window"], f.last_plan()
        f.record_plans = False
        sub = torch.randperm(pts.shape[0], generator=torch.Generator().manual_seed(9))[:200000].to(dev)
        f.tuning_flags = _lib.TUNE_NO_REORDER
        with knobs(D3F_EXP_RUNS=-1):
            ref_gpu = f.eval(pts[sub], return_names=names)
    for k in ["dist", "valid_mask"] + names:
        assert torch.equal(out[k][sub], ref_gpu[k]), k
    pick = sub[:3000]
    ref = oracle_sample(sc, pts[pick].cpu(), [f.curr_obs_torch[k] for k in names])
    assert np.array_equal(cpu(out["valid_mask"][pick]), ref["valid_mask"])
    assert np.array_equal(cpu(out["dist"][pick]), ref["dist"])
    for i, k in enumerate(names):
        assert rel_err(cpu(out[k][pick]), ref["sets"][i]) <= TOL, k


def test_bench_workload_c4_dense(dev):
    """Config 4 with DENSE maps (8 x 720 x 1280 x 1024 fp32 = 30.2 GB resident) on its lattice slab: the walk's launch is
    bit-identical to the caller-order direct gather on 100 000 points, and dist / valid_mask of a 3000-point sample equal the
    CPU oracle's (they do not depend on the maps, so the oracle runs with a one-channel stand-in instead of a 30 GB host
    copy).  Skipped on devices with less than 40 GB free."""
    import bench
    from d3fields_amd import _lib, synth
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 40 * 2 ** 30:
        pytest.skip("needs 40 GB of free device memory")
    f, pts, names, w, sc = bench.build_workload("c4_dense", dev, 0, 1)
    f.record_plans = True
    with torch.no_grad():
        out = f.batch_eval(pts, return_names=names)
        assert "brick walk" in f.last_plan()["point_order"], f.last_plan()
        sub = torch.randperm(pts.shape[0], generator=torch.Generator().manual_seed(9))[:100000].to(dev)
        f.tuning_flags = _lib.TUNE_NO_REORDER | _lib.TUNE_DIRECT_GATHER
        ref_gpu = f.eval(pts[sub], return_names=names)
    for k in ["dist", "valid_mask"] + names:
        assert torch.equal(out[k][sub], ref_gpu[k]), k
    pick = sub[:3000]
    ref = oracle_sample(sc, pts[pick].cpu(), [torch.zeros(w["V"], 2, 2, 1)])
    assert np.array_equal(cpu(out["valid_mask"][pick]), ref["valid_mask"])
    assert np.array_equal(cpu(out["dist"][pick]), ref["dist"])
    # the 1024 fused channels of the sample against the torch-ops port of the reference's own op sequence (oracle/torch_port.py,
    # pinned to the imported reference on the CPU), run here on the DEVICE-resident maps: no 30 GB host copy (grid_sample reads
    # the channels-last storage through its permuted view like the reference, fusion.py:373)
    from oracle import torch_port
    obs = {k: f.curr_obs_torch[k] for k in ("depth", "K", "pose", "dino_feats")}
    with torch.no_grad():
        port = torch_port.field_query(obs, pts[pick], names, w["H"], w["W"], mu=f.mu)
    # (torch's DEVICE kernels are not the CPU ones the contract is pinned to -- grid_sample forms its weights differently and
    # divides through reciprocals, DESIGN.md section 2 -- so this leg is an independent implementation at 3e-4, measured 9e-5;
    # the bit-level statements of this workload are the two above and bench.py's `verified` against the C oracle)
    assert float((port["valid_mask"] != out["valid_mask"][pick]).float().mean()) <= 1e-3
    same = cpu(port["valid_mask"] == out["valid_mask"][pick])
    assert rel_err(cpu(out["dino_feats"][pick])[same], cpu(port["dino_feats"])[same]) <= 3e-4
    del f, out, ref_gpu, port, obs
    torch.cuda.empty_cache()


# ---- channel-sliced launch (experiment knob) == default launch, bit for bit -------------------------------------------
@pytest.mark.parametrize("dims,C,mask", [((47, 53, 29), 384, True), ((64, 33, 37), 128, False), ((70, 70, 17), 1024, False),
                                         ((33, 35, 61), 192, True)])
def test_channel_sliced_launch_is_bit_identical(dev, dims, C, mask):
    from d3fields_amd import create_init_grid, synth, _lib
    V, H, W = 4, 96, 128
    maps = {"dino_feats": synth.random_map(V, H, W, C, seed=1, device=dev)}
    names = ["dino_feats"]
    if mask:
        maps["mask"] = synth.random_onehot_mask(V, H, W, 8, seed=2, device=dev)
        names.append("mask")
    f, sc = fusion_for(dev, V, H, W, maps)
    grid = create_init_grid(box_for(*dims, 0.004), 0.004)[0]
    grid[4321, 0] = float("nan")                                                       # a strict point
    pts = grid.to(dev)
    with torch.no_grad():
        f.tuning_flags = _lib.TUNE_NO_REORDER
        base = f.batch_eval(pts, return_names=names)
        f.tuning_flags = _lib.TUNE_FORCE_REORDER
        outs = {"default": f.batch_eval(pts, return_names=names)}
        for sl in ((1, 2, 3) if experiments() else ()):
            if C % (32 << (sl - 1)):
                continue
            for vc in (1, 2, 4):
                with knobs(D3F_EXP_SLICED=sl, D3F_EXP_SLICED_VC=vc):
                    outs[(sl, vc)] = f.batch_eval(pts, return_names=names)
    assert outs
    for sl, o in outs.items():
        for k in ["dist", "valid_mask"] + names:
            a, b = o[k], base[k]
            assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float())), (sl, k)


@pytest.mark.parametrize("n,C,hw,mask", [(70001, 384, (96, 128), True), (131072, 1024, (96, 128), False), (65537, 128, (192, 256), False)])
def test_channel_sliced_launch_on_a_cloud_is_bit_identical(dev, n, C, hw, mask):
    """A random cloud on maps beyond the caches: the Hilbert order feeds the channel-sliced kernel (tiles of 16 / 32 consecutive
    points of the order, one 512-byte slice per workgroup; C = 1024: one slice per XCD).  Same bits as the caller-order
    direct gather, including a strict (NaN) point, a short last tile and the thin map riding along."""
    from d3fields_amd import synth, _lib
    V, (H, W) = 4, hw
    maps = {"dino_feats": synth.random_map(V, H, W, C, seed=1, device=dev)}
    names = ["dino_feats"]
    if mask:
        maps["mask"] = synth.random_onehot_mask(V, H, W, 8, seed=2, device=dev)
        names.append("mask")
    f, sc = fusion_for(dev, V, H, W, maps)
    f.record_plans = True
    pts = synth.random_cloud(n, seed=5).to(dev)
    pts[4321, 1] = float("nan")                                                        # a strict point
    with torch.no_grad():
        out = f.batch_eval(pts, return_names=names)
        plan = f.last_plan()
        assert "Hilbert" in plan["point_order"] and "channel-sliced" in plan["point_order"], plan
        assert plan["kernel"].startswith("fused_eval_sliced_kernel"), plan
        f.tuning_flags = _lib.TUNE_NO_REORDER | _lib.TUNE_DIRECT_GATHER
        base = f.batch_eval(pts, return_names=names)
        assert "sliced" not in f.last_plan()["kernel"]
    for k in ["dist", "valid_mask"] + names:
        a, b = out[k], base[k]
        assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float())), k


# ---- thin maps: views in parallel across lanes == view-sequential gather, bit for bit -------------------------------------
@pytest.mark.parametrize("V,C,N", [(4, 8, 70000), (2, 3, 5000), (3, 5, 130001), (8, 16, 66000), (5, 1, 3000), (4, 12, 90000)])
def test_thin_map_gather_is_bit_identical(dev, V, C, N):
    """gather_map_thin (one lane per (point, view, vector), ordered sum rebuilt with shuffles) against gather_map's
    view-sequential loop (D3F_EXP_THIN=-1): fused rows and '<k>_inter', with strict points (a NaN coordinate, non-finite
    map values) and in the kernels that carry thin maps along (direct, cell runs, windows, channel slices)."""
    from d3fields_amd import create_init_grid, synth
    H, W = 120, 160
    thin = synth.random_map(V, H, W, C, seed=3, device=dev)
    wide = synth.random_map(V, 12, 16, 384, seed=1, device=dev)
    f, sc = fusion_for(dev, V, H, W, {"thin": thin, "dino_feats": wide}, kind="stress")
    pts_c = synth.random_cloud(N, seed=5) * 1.2
    pts_c[N // 2, 2] = float("nan")
    pts = pts_c.to(dev)
    grid = create_init_grid(synth.WORK_BOX, 0.0107)[0].to(dev)
    with torch.no_grad():
        for names, inter, q in ((["thin"], False, pts), (["thin"], True, pts), (["dino_feats", "thin"], False, pts), (["dino_feats", "thin"], False, grid)):
            with knobs(D3F_EXP_THIN=-1):
                ref = f.eval(q, return_names=names, return_inter=inter) if inter else f.batch_eval(q, return_names=names)
            out = f.eval(q, return_names=names, return_inter=inter) if inter else f.batch_eval(q, return_names=names)
            for k in ref:
                a, b = out[k], ref[k]
                assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float())), (names, inter, k)
        bad = thin.clone(); bad[0, 7, 9, 0] = float("inf")                 # non-finite map: everything strict
        f.curr_obs_torch["thin"] = bad
        f.invalidate_map_checks()
        with knobs(D3F_EXP_THIN=-1):
            ref = f.eval(pts, return_names=["thin"])
        out = f.eval(pts, return_names=["thin"])
        assert torch.equal(torch.nan_to_num(out["thin"], nan=7.0, posinf=8.0, neginf=9.0), torch.nan_to_num(ref["thin"], nan=7.0, posinf=8.0, neginf=9.0))


# ---- register rows (fuse_rows.hip: 1024-channel patch maps) == direct gather, bit for bit -------------------------------------------
@pytest.mark.parametrize("V,fhw,thin,points,n", [(8, (36, 64), False, "grid", 0), (8, (36, 64), False, "cloud", 150001), (8, (24, 32), True, "scattered", 70001),
                                                 (4, (24, 32), True, "cloud", 100000), (5, (24, 32), True, "grid", 0), (1, (48, 64), False, "cloud", 66000),
                                                 (4, (48, 64), False, "big cloud", 270001)])
def test_register_rows_are_bit_identical(dev, V, fhw, thin, points, n):
    """fused_eval_rows_kernel against the direct gather on the same points: clipped bricks of a lattice (74 x 65 x 20 is no multiple of
    4 x 4 x 2), the Hilbert order of a cloud, 32 consecutive points of a caller-order cloud, SCATTERED points (every (point, view) its
    own cell: up to 256 cells per workgroup, the cell table read in more than one piece), odd cells (a point paired with itself at
    zero weights), a strict point, points whose corners leave the map, thin maps riding along, 1 / 4 / 5 / 8 views, and -- four views,
    a big cloud -- the rows as the OTHER side of the window kernel's device gate; a sample against the oracle."""
    from d3fields_amd import create_init_grid, synth, _lib
    H, W, C = 480, 640, 1024
    maps = {"dino_feats": synth.random_map(V, fhw[0], fhw[1], C, seed=1, device=dev)}
    names = ["dino_feats"]
    if thin:
        maps["mask"] = synth.random_onehot_mask(V, H, W, 8, seed=2, device=dev)
        maps["color"] = synth.random_map(V, H, W, 3, seed=4, device=dev)
        names += ["mask", "color"]
    f, sc = fusion_for(dev, V, H, W, maps)
    if points == "grid":
        pts_c = create_init_grid(synth.WORK_BOX, 0.0107)[0]                            # 74 x 65 x 20 points
    else:
        pts_c = synth.random_cloud(n, seed=3)
        if points == "cloud" and n <= 100000:                                          # a caller order with locality: sorted along x
            pts_c = pts_c[torch.argsort(pts_c[:, 0])].contiguous()
    pts_c[1000, 1] = float("inf")                                                       # a strict point
    pts = pts_c.to(dev)
    f.record_plans = True
    with torch.no_grad():
        if points == "scattered":
            f.tuning_flags = _lib.TUNE_NO_REORDER                                       # 32 consecutive points of a random cloud
        out = f.batch_eval(pts, return_names=names)
        plan = f.last_plan()
        assert plan["family"] == "register-rows", plan
        if points == "big cloud":
            assert plan["gated_window"], plan                                           # ... behind the gate; both sides, each forced
            with knobs(FLAGS=_lib.TUNE_NO_WINDOW_GATE):
                rows_side = f.batch_eval(pts, return_names=names)
            with knobs(FLAGS=_lib.TUNE_WINDOW_SIDE):
                window_side = f.batch_eval(pts, return_names=names)
            for k in out:
                assert torch.equal(torch.nan_to_num(out[k].float()), torch.nan_to_num(rows_side[k].float())), k
                assert torch.equal(torch.nan_to_num(out[k].float()), torch.nan_to_num(window_side[k].float())), k
        f.record_plans = False
        with knobs(D3F_EXP_RUNS=-1):
            ref = f.batch_eval(pts, return_names=names)
    for k in ref:
        a, b = out[k], ref[k]
        assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float())), k
    assert bool((out["valid_mask"]).any()) and bool((~out["valid_mask"].bool()).any())
    pick = torch.randperm(pts.shape[0], generator=torch.Generator().manual_seed(3))[:1500]
    pick = pick[pick != 1000]
    o = oracle_sample(sc, pts_c[pick], [maps[k] for k in names])
    assert np.array_equal(cpu(out["dist"])[pick], o["dist"]) and np.array_equal(cpu(out["valid_mask"])[pick], o["valid_mask"])
    assert rel_err(cpu(out["dino_feats"])[pick], o["sets"][0]) <= TOL
    for i, k in enumerate(names[1:]):
        assert rel_err(cpu(out[k])[pick], o["sets"][1 + i]) <= TOL, k


# ---- LDS texel windows (experiment knob D3F_EXP_WINDOW) == direct gather, bit for bit -----------------------------------
@pytest.mark.parametrize("C,V,fhw,mask,points", [(384, 4, (48, 64), True, "grid"), (256, 3, (24, 32), False, "cloud"),
                                                 (768, 8, (36, 64), False, "cloud"), (128, 2, (48, 64), True, "grid"),
                                                 (512, 8, (48, 64), True, "grid")])
def test_window_gather_is_bit_identical(dev, C, V, fhw, mask, points):
    """Every (tile, vectors per lane, pool) variant of fused_eval_window_kernel, incl. pools too small for the windows
    (pairs then go direct), clipped bricks (74 x 65 x 20 is no multiple of the brick), a strict point and points whose
    corners leave the map; the plan says the window kernel is what runs."""
    from d3fields_amd import create_init_grid, synth, _lib
    import ctypes
    H, W = 480, 640
    maps = {"dino_feats": synth.random_map(V, fhw[0], fhw[1], C, seed=1, device=dev)}
    names = ["dino_feats"]
    if mask:
        maps["mask"] = synth.random_onehot_mask(V, H, W, 8, seed=2, device=dev)
        names.append("mask")
    f, sc = fusion_for(dev, V, H, W, maps)
    if points == "grid":
        pts_c = create_init_grid(synth.WORK_BOX, 0.0107)[0]                            # 74 x 65 x 20 points
    else:
        pts_c = synth.random_cloud(270001, seed=3)                                      # >= 262 144: the gated pair of launches
    pts_c[1000, 1] = float("inf")                                                       # a strict point
    pts = pts_c.to(dev)
    views, keep, _ = f._views(dev)
    m = maps["dino_feats"]
    cm = (_lib.ChannelMap * 1)(_lib.ChannelMap(m.data_ptr(), m.shape[1], m.shape[2], C, 0, m.stride(0), m.stride(1), m.stride(2)))
    plan = _lib.EvalPlan()
    if points != "grid":
        # product build: a cloud gets BOTH launches behind the device-side gate; TUNE_WINDOW_SIDE opens the window side whatever
        # the probe says (this sparse cloud overflows most pools: the touched-texel pool's direct conversions are exercised)
        _lib.check(f._lib.d3f_eval_plan_query(ctypes.byref(views), pts.shape[0], cm, 1, _lib.FLAG_FINITE_MAPS | _lib.FLAG_UNORDERED_POINTS, 1, 0, ctypes.byref(plan)))
        assert plan.gated_window == 1 and plan.staged[0] >= 16 and 2000 <= plan.reserved2 < 3000, "a cloud of this size takes the gated pair"
        _lib.check(f._lib.d3f_eval_plan_query(ctypes.byref(views), pts.shape[0], cm, 1, _lib.FLAG_FINITE_MAPS | _lib.FLAG_UNORDERED_POINTS | _lib.TUNE_NO_WINDOW_GATE, 1, 0, ctypes.byref(plan)))
        assert plan.gated_window == 0
    with knobs(D3F_EXP_WINDOW=64):
        if points != "grid" and not experiments():
            pass
        elif points == "grid":
            _lib.check(f._lib.d3f_eval_plan_query_lattice(ctypes.byref(views), 74, 65, 20, cm, 1, _lib.FLAG_FINITE_MAPS, 0, ctypes.byref(plan)))
            assert plan.reorder == 2 and plan.workgroups == 19 * 17 * 5
        else:
            _lib.check(f._lib.d3f_eval_plan_query(ctypes.byref(views), pts.shape[0], cm, 1, _lib.FLAG_FINITE_MAPS, 1, 0, ctypes.byref(plan)))
        # (2114 / 2113: four, or three and fewer workgroups per CU -- the pool is sized for ~17 texel slots per view (14 touched
        #  texels per view for a cloud), which up to three views get at four workgroups per CU)
        if points == "grid" or experiments():
            assert plan.staged[0] == 3 and plan.tile_points == 64 and plan.reserved == (2114 if V <= 3 else 2113), "the window kernel must be what runs here"
    variants = [("direct", dict(D3F_EXP_RUNS=-1))]
    for T in (32, 64, 128):
        if (T * (1 if V <= 1 else 2 if V <= 2 else 4 if V <= 4 else 8)) % 64:
            continue
        for U in (1, 2, 3, 4):
            if (C // 128) % U:
                continue
            variants.append(("T%d U%d" % (T, U), dict(D3F_EXP_WINDOW=T, D3F_EXP_WINDOW_U=U)))
    variants += [("T64 occ3", dict(D3F_EXP_WINDOW=64, D3F_EXP_WINDOW_OCC=3)), ("T64 pool 6", dict(D3F_EXP_WINDOW=64, D3F_EXP_WINDOW_POOL=6)),
                 ("T64 pool 2", dict(D3F_EXP_WINDOW=64, D3F_EXP_WINDOW_POOL=2)), ("T64 vc2", dict(D3F_EXP_WINDOW=64, D3F_EXP_WINDOW_VC=2)),
                 ("T64 lpp32", dict(D3F_EXP_WINDOW=64, D3F_EXP_WINDOW_LPP=32)), ("T64 lpp32 occ3", dict(D3F_EXP_WINDOW=64, D3F_EXP_WINDOW_LPP=32, D3F_EXP_WINDOW_OCC=3)),
                 # round 4: the plain view loop instead of the software-pipelined point loop (V = 4 / 8), also at 5 / 6 workgroups per CU
                 ("T64 plain loop", dict(D3F_EXP_WINDOW=64, D3F_EXP_WINDOW_PIPE=-1)), ("T64 plain occ5", dict(D3F_EXP_WINDOW=64, D3F_EXP_WINDOW_PIPE=-1, D3F_EXP_WINDOW_OCC=5)),
                 ("T64 plain occ6", dict(D3F_EXP_WINDOW=64, D3F_EXP_WINDOW_PIPE=-1, D3F_EXP_WINDOW_OCC=6))]
    variants += [("T64 sparse", dict(D3F_EXP_WINDOW=64, D3F_EXP_WINDOW_SPARSE=1)), ("T64 rect", dict(D3F_EXP_WINDOW=64, D3F_EXP_WINDOW_SPARSE=-1)),
                 ("T64 sparse pool 6", dict(D3F_EXP_WINDOW=64, D3F_EXP_WINDOW_SPARSE=1, D3F_EXP_WINDOW_POOL=6)),
                 ("T64 sparse plain loop", dict(D3F_EXP_WINDOW=64, D3F_EXP_WINDOW_SPARSE=1, D3F_EXP_WINDOW_PIPE=-1)),
                 ("T64 contiguous eighths", dict(D3F_EXP_WINDOW=64, D3F_EXP_WINDOW_RR=-1)), ("T64 z-curve", dict(D3F_EXP_WINDOW=64, D3F_EXP_ORDER_MORTON=1))]
    if not experiments():
        variants = [variants[0], ("T64 U1", dict())]                                    # the reference and the default launch
    if points != "grid":                      # every build: the gate's two sides, each forced, and the gate left to the device
        variants += [("window side", dict(FLAGS=_lib.TUNE_WINDOW_SIDE)), ("cell-run side", dict(FLAGS=_lib.TUNE_NO_WINDOW_GATE)), ("gated", dict())]
    with torch.no_grad():
        outs = {}
        for tag, env in variants:
            with knobs(**env):
                outs[tag] = f.batch_eval(pts, return_names=names)
    for tag in [t for t, _ in variants[1:]]:
        for k in outs["direct"]:
            a, b = outs[tag][k], outs["direct"][k]
            assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float())), (tag, k)
    pick = torch.randperm(pts.shape[0], generator=torch.Generator().manual_seed(3))[:1500]
    pick = pick[pick != 1000]
    ref = oracle_sample(sc, pts_c[pick], [maps[k] for k in names])
    assert np.array_equal(cpu(outs["T64 U1"]["dist"])[pick], ref["dist"])
    assert rel_err(cpu(outs["T64 U1"]["dino_feats"])[pick], ref["sets"][0]) <= TOL


@pytest.mark.parametrize("V,hw,fhw,C,scale,expect_window", [(4, (480, 640), (48, 64), 384, 0.6, True), (8, (720, 1280), (72, 128), 256, 1.0, False),
                                                            (4, (480, 640), (48, 64), 128, 0.55, True)])
def test_cloud_gate_follows_the_density(dev, V, hw, fhw, C, scale, expect_window):
    """A cloud of >= 262 144 points in the Hilbert order gets the LDS-window launch AND the cell-run launch behind one device word
    (d3fields_hip.h, ABI 5): the probe counts the sampled 64-point tiles whose touched texels fit the pool -- a dense cloud opens
    the window side, a cloud that is sparse against the texel grid the cell runs -- and whichever side runs, alone, the outputs are
    those of the caller-order direct gather bit for bit (a strict point, a short last tile, a thin map riding along)."""
    from d3fields_amd import synth, _lib
    H, W = hw
    maps = {"dino_feats": synth.random_map(V, fhw[0], fhw[1], C, seed=1, device=dev), "mask": synth.random_onehot_mask(V, H, W, 8, seed=2, device=dev)}
    names = ["dino_feats", "mask"]
    f, sc = fusion_for(dev, V, H, W, maps)
    f.record_plans = True
    pts_c = synth.random_cloud(300001, seed=8) * scale
    pts_c[777, 0] = float("nan")
    pts = pts_c.to(dev)
    with torch.no_grad():
        out = f.batch_eval(pts, return_names=names)
        plan, gate = f.last_plan(), f.last_gate()
        assert plan["gated_window"] and "Hilbert" in plan["point_order"], plan
        assert gate[1] == expect_window and (gate[0] >= _lib.GATE_MIN_FIT) == expect_window, gate
        again = f.batch_eval(pts, return_names=names)                      # cached order (D3F_FLAG_REUSE_POINT_ORDER): the probe runs again
        assert f.last_gate() == gate
        with knobs(FLAGS=_lib.TUNE_NO_WINDOW_GATE):
            runs = f.batch_eval(pts, return_names=names)
            assert not f.last_plan()["gated_window"]
        with knobs(FLAGS=_lib.TUNE_WINDOW_SIDE):
            win = f.batch_eval(pts, return_names=names)
            assert f.last_gate()[1]
        f.tuning_flags = _lib.TUNE_NO_REORDER
        with knobs(D3F_EXP_RUNS=-1):
            base = f.batch_eval(pts, return_names=names)
    for tag, o in (("gated", out), ("again", again), ("runs", runs), ("window", win)):
        for k in ["dist", "valid_mask"] + names:
            a, b = o[k], base[k]
            assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float())), (tag, k)
    pick = torch.randperm(pts.shape[0], generator=torch.Generator().manual_seed(3))[:1500]
    pick = pick[pick != 777]
    ref = oracle_sample(sc, pts_c[pick], [maps[k] for k in names])
    assert np.array_equal(cpu(out["dist"])[pick], ref["dist"])
    assert rel_err(cpu(out["dino_feats"])[pick], ref["sets"][0]) <= TOL


@pytest.mark.parametrize("kind,V,hw,fhw,C,mask", [("window", 4, (480, 640), (48, 64), 384, True), ("window", 8, (480, 640), (36, 64), 1024, False),
                                                  ("window", 3, (480, 640), (24, 32), 256, False), ("sliced", 4, (192, 256), (192, 256), 384, True),
                                                  ("sliced", 2, (240, 320), (240, 320), 1024, False)])
def test_fp16_stored_maps_take_the_window_and_sliced_kernels(dev, kind, V, hw, fhw, C, mask):
    """Round 5: a map STORED in fp16 keeps the fast kernels -- LDS texel windows of 256-byte slices on a lattice over a
    patch-resolution map, 16 lanes x 8 channels per point in the channel-sliced launch over a dense one -- with the halves widened
    inside v_fma_mix_f32: the rows equal those of the fp32 query on the widened map bit for bit (and of the direct gather on the
    fp16 map), including a strict point, clipped bricks and a thin fp32 map riding along."""
    from d3fields_amd import create_init_grid, synth
    H, W = hw
    feats16 = (synth.random_map(V, fhw[0], fhw[1], C, seed=1, device=dev) * 2.0).half()
    maps16 = {"dino_feats": feats16}
    names = ["dino_feats"]
    if mask:
        maps16["mask"] = synth.random_onehot_mask(V, H, W, 8, seed=2, device=dev)
        names.append("mask")
    f16, sc = fusion_for(dev, V, H, W, maps16)
    f32, _ = fusion_for(dev, V, H, W, dict(maps16, dino_feats=feats16.float()))
    pts_c = create_init_grid(synth.WORK_BOX, 0.0107)[0]                                # 74 x 65 x 20 points
    pts_c[1000, 1] = float("inf")                                                       # a strict point
    pts = pts_c.to(dev)
    with torch.no_grad():
        f16.record_plans = True
        a = f16.batch_eval(pts, return_names=names)
        kernel = f16.last_plan()["kernel"]
        assert kernel.startswith("fused_eval_%s_kernel" % kind) and kernel.endswith(", true>"), kernel
        b = f32.batch_eval(pts, return_names=names)
        with knobs(D3F_EXP_RUNS=-1):
            c = f16.batch_eval(pts, return_names=names)
            assert f16.last_plan()["kernel"] == "fused_eval_f16_kernel<0>"
    for tag, o in (("fp32 query on the widened map", b), ("direct gather on the fp16 map", c)):
        for k in ["dist", "valid_mask"] + names:
            x, y = a[k], o[k]
            assert torch.equal(torch.isnan(x), torch.isnan(y)) and torch.equal(torch.nan_to_num(x.float()), torch.nan_to_num(y.float())), (tag, k)
    pick = torch.randperm(pts.shape[0], generator=torch.Generator().manual_seed(3))[:1500]
    pick = pick[pick != 1000]
    ref = oracle_sample(sc, pts_c[pick], [f32.curr_obs_torch[k] for k in names])
    assert np.array_equal(cpu(a["dist"])[pick], ref["dist"])
    assert rel_err(cpu(a["dino_feats"])[pick], ref["sets"][0]) <= TOL


def _hilbert27(q):
    """numpy restatement of order_kernels.hip:hilbert27 (Skilling's transpose form, 9 bits per axis, axis 0 the most significant
    bit of every digit) for the ordering test below"""
    X = [(q[:, k] & 511).astype(np.int64).copy() for k in range(3)]
    Q = 256
    while Q > 1:
        P = Q - 1
        for i in range(3):
            hit = (X[i] & Q) != 0
            X[0] = np.where(hit, X[0] ^ P, X[0])
            t = np.where(hit, 0, (X[0] ^ X[i]) & P)
            X[0] ^= t
            X[i] ^= t
        Q >>= 1
    X[1] ^= X[0]; X[2] ^= X[1]
    t = np.zeros_like(X[0])
    Q = 256
    while Q > 1:
        t = np.where((X[2] & Q) != 0, t ^ (Q - 1), t)
        Q >>= 1
    for i in range(3):
        X[i] ^= t
    key = np.zeros_like(X[0])
    for b in range(8, -1, -1):
        for i in range(3):
            key = (key << 1) | ((X[i] >> b) & 1)
    return key


@pytest.mark.parametrize("n,scale", [(70001, 1.0), (300001, 0.6), (1000003, 1.0), (300001, 3.5)])
def test_point_order_is_the_exact_hilbert_order(dev, n, scale):
    """The order d3f_eval builds for a cloud (order_kernels.hip: counting sort by a prefix of the 27-bit Hilbert key of the
    point's cell on a 512^3 grid over the cloud's box + exact rank inside the counting cell) is a permutation and equals numpy's lexsort by (key, index) -- consecutive cells of the
    curve share a face at every level, which is what makes any 64 consecutive points a compact tile; a clump of > 256 points in one
    counting cell and NaN coordinates keep it a permutation; a cloud wider than 2.04 m falls back to the fixed 4-mm grid."""
    from d3fields_amd import synth
    V, H, W = 4, 96, 128
    f, sc = fusion_for(dev, V, H, W, {"dino_feats": synth.random_map(V, 12, 16, 128, seed=1, device=dev)})
    pts_c = synth.random_cloud(n, seed=11) * scale
    pts_c[5000:6200] = pts_c[5000] + 1e-4 * torch.rand(1200, 3)            # a clump: > 256 points in one counting cell
    pts = pts_c.to(dev)
    with torch.no_grad():
        f.batch_eval(pts, return_names=["dino_feats"])
    torch.cuda.synchronize()
    seg = (n * 4 + 255) // 256 * 256
    order = f._last_ws[2 * seg:2 * seg + 4 * n].view(torch.int32).cpu().numpy().astype(np.int64)
    assert np.array_equal(np.sort(order), np.arange(n)), "the order is not a permutation"
    # the key grid is laid over the cloud's own box: 511 cells along its longest side (order_kernels.hip:key_grid), fp32 as the device
    p32 = pts_c.numpy().astype(np.float32)
    lo = p32.min(axis=0)
    ext = np.float32((p32.max(axis=0) - lo).max())
    boxed = bool(ext <= np.float32(511.0) * np.float32(0.004))
    assert boxed == (scale < 3), ext
    if boxed:
        inv = np.float32(511.0) / ext
        q = np.floor((p32 - lo) * inv).astype(np.int64)
        assert q.min() == 0 and 510 <= q.max() <= 511
    else:                                               # wider than 2.04 m: the fixed 4-mm grid at the origin, keys wrap every 2.048 m
        inv = np.float32(1.0 / np.float32(0.004))
        q = np.floor(p32 * inv).astype(np.int64)
    key = _hilbert27(q)
    bits = 15
    while bits < 20 and (1 << bits) < 2 * n:
        bits += 1
    cell = key >> (27 - bits)                                               # the counting cell (4^3 key cells at 1 M points)
    _, inverse, counts = np.unique(cell, return_inverse=True, return_counts=True)
    clump = counts[inverse] > 256                                           # > 256 points: ranked in aligned pieces of 256 slots
    assert clump[5000:6200].sum() > 256
    assert (np.diff(cell[order]) >= 0).all(), "counting cells are not ascending along the order"
    # exact (key, index) order everywhere but inside the clump's cell
    want = np.lexsort((np.arange(n), key))
    assert np.array_equal(order[~clump[order]], want[~clump[want]])
    # the curve is continuous: consecutive points of the order are close (a Z curve jumps by whole octants)
    step = np.abs(np.diff(q[order], axis=0)).max(axis=1) / float(inv)       # metres
    assert not boxed or np.percentile(step, 99.9) <= (0.048 if n >= 300000 else 0.16), np.percentile(step, 99.9)
    bad = pts_c.clone(); bad[7, 0] = float("nan"); bad[9, 2] = float("inf")
    with torch.no_grad():
        f.batch_eval(bad.to(dev), return_names=["dino_feats"])
    torch.cuda.synchronize()
    order = f._last_ws[2 * seg:2 * seg + 4 * n].view(torch.int32).cpu().numpy().astype(np.int64)
    assert np.array_equal(np.sort(order), np.arange(n))


def test_point_order_of_a_degenerate_cloud(dev):
    """Every point the same (a box of zero extent: the key grid falls back to the fixed one, all points share one counting cell and are
    ranked in pieces of 256) and a cloud without one finite coordinate: the order stays a permutation, the outputs are per point."""
    from d3fields_amd import synth
    V, H, W, n = 4, 96, 128, 100003
    f, sc = fusion_for(dev, V, H, W, {"dino_feats": synth.random_map(V, 12, 16, 128, seed=1, device=dev)})
    f.tuning_flags |= (1 << 14)                                             # D3F_TUNE_FORCE_REORDER: small maps would not be reordered
    seg = (n * 4 + 255) // 256 * 256
    one = synth.random_cloud(8, seed=5)[3:4]
    for pts_c in (one.repeat(n, 1), torch.full((n, 3), float("nan")), torch.full((n, 3), float("inf"))):
        with torch.no_grad():
            out = f.batch_eval(pts_c.to(dev), return_names=["dino_feats"])
        torch.cuda.synchronize()
        order = f._last_ws[2 * seg:2 * seg + 4 * n].view(torch.int32).cpu().numpy().astype(np.int64)
        assert np.array_equal(np.sort(order), np.arange(n)), "the order is not a permutation"
        assert torch.equal(out["dist"], out["dist"][:1].expand(n)) or bool(torch.isnan(out["dist"]).all())
        if torch.isfinite(pts_c).all():
            assert torch.equal(out["dino_feats"], out["dino_feats"][:1].expand(n, -1))
            assert bool(out["valid_mask"].all()) or not bool(out["valid_mask"].any())


def test_window_kernel_with_wrong_lattice_dims_is_still_exact(dev):
    """d3f_eval_lattice promises that ANY dims whose product is n are correct.  The window kernel estimates its texel
    windows from the eight 'corner' slots of a brick -- meaningless for a cloud or a shuffled grid passed with made-up
    dims -- and every pair is checked against its window, so such launches must still be bit-identical to the direct
    gather (they only lose the windows)."""
    from d3fields_amd import create_init_grid, synth
    V, H, W, C = 4, 480, 640, 384
    maps = {"dino_feats": synth.random_map(V, 48, 64, C, seed=1, device=dev), "mask": synth.random_onehot_mask(V, H, W, 8, seed=2, device=dev)}
    names = ["dino_feats", "mask"]
    f, sc = fusion_for(dev, V, H, W, maps)
    f.cache_point_order = True                                                         # _launch asks _lattice_dims
    grid, shape = create_init_grid(synth.WORK_BOX, 0.0107)
    nx, ny, nz = (int(v) for v in shape)
    n = nx * ny * nz
    assert grid.shape[0] == n and n >= 65536
    cases = [(synth.random_cloud(n, seed=4), (nx, ny, nz)),                            # a cloud with a grid's dims
             (grid[torch.randperm(n, generator=torch.Generator().manual_seed(1))], (nx, ny, nz)),   # shuffled grid
             (grid, (nz, ny, nx)), (grid, (n, 1, 1)), (grid, (1, 1, n)), (grid, (nx * ny, 1, nz))]  # wrong factorizations
    for pts_c, dims in cases:
        pts = pts_c.to(dev)
        with torch.no_grad():
            f._lattice_dims = lambda p, st: None
            with knobs(D3F_EXP_RUNS=-1):
                ref = f.batch_eval(pts, return_names=names)
            f._lattice_dims = lambda p, st, d=dims: d
            f.record_plans = True
            out = f.batch_eval(pts, return_names=names)
            assert "window" in f.last_plan()["kernel"], f.last_plan()
        for k in ref:
            assert torch.equal(out[k], ref[k]), (dims, k)


def test_map_order_in_the_call_does_not_matter(dev):
    """return_names=['mask', 'dino_feats'] gets the launch of ['dino_feats', 'mask'] (the library moves the wide map to the
    front for the window / channel-sliced kernels) and the same bits, on a patch-resolution and on a dense feature map."""
    from d3fields_amd import create_init_grid, synth
    V, H, W, C = 4, 480, 640, 384
    for fhw in ((48, 64), (240, 320)):
        maps = {"dino_feats": synth.random_map(V, fhw[0], fhw[1], C, seed=1, device=dev),
                "mask": synth.random_onehot_mask(V, H, W, 8, seed=2, device=dev),
                "color_tensor": synth.random_map(V, H, W, 3, seed=5, device=dev)}
        f, sc = fusion_for(dev, V, H, W, maps)
        pts = create_init_grid(synth.WORK_BOX, 0.0107)[0].to(dev)
        with torch.no_grad():
            f.record_plans = True
            a = f.batch_eval(pts, return_names=["dino_feats", "mask", "color_tensor"])
            ka = f.last_plan()["kernel"]
            b = f.batch_eval(pts, return_names=["color_tensor", "mask", "dino_feats"])
            kb = f.last_plan()["kernel"]
        assert ka == kb and ("window" in ka or "sliced" in ka), (ka, kb)
        for k in a:
            assert torch.equal(a[k], b[k]), (fhw, k)


def test_window_kernel_in_grid_mode(dev):
    """Fusion.eval_grid (axis arrays instead of a point tensor, d3f_eval_grid) takes the window kernel too for a
    patch-resolution wide map; bits equal batch_eval of the materialised grid with windows and cell runs switched off."""
    from d3fields_amd import create_init_grid, synth
    V, H, W, C = 4, 480, 640, 384
    maps = {"dino_feats": synth.random_map(V, 48, 64, C, seed=1, device=dev), "mask": synth.random_onehot_mask(V, H, W, 8, seed=2, device=dev)}
    f, sc = fusion_for(dev, V, H, W, maps)
    box, step = synth.WORK_BOX, 0.0107
    with torch.no_grad():
        a = f.eval_grid(box, step, return_names=["dino_feats", "mask"])
        grid, shape = create_init_grid(box, step)
        assert tuple(a["grid_shape"]) == tuple(shape) and grid.shape[0] >= 65536
        with knobs(D3F_EXP_WINDOW=-1, D3F_EXP_RUNS=-1):
            b = f.batch_eval(grid.to(dev), return_names=["dino_feats", "mask"])
    for k in ("dist", "valid_mask", "dino_feats", "mask"):
        assert torch.equal(a[k], b[k]), k
    import ctypes
    from d3fields_amd import _lib
    views, keep, _ = f._views(dev)
    m = maps["dino_feats"]
    cm = (_lib.ChannelMap * 1)(_lib.ChannelMap(m.data_ptr(), m.shape[1], m.shape[2], C, 0, m.stride(0), m.stride(1), m.stride(2)))
    plan = _lib.EvalPlan()
    nx, ny, nz = (int(v) for v in shape)
    _lib.check(f._lib.d3f_eval_plan_query_lattice(ctypes.byref(views), nx, ny, nz, cm, 1, _lib.FLAG_FINITE_MAPS, 0, ctypes.byref(plan)))
    assert plan.staged[0] == 3


def test_window_kernel_above_4_gib_of_output(dev):
    """2.99 M lattice points x 384 channels = 4.6 GB of fused rows: the window kernel's 32-bit store offsets must give way
    to 64-bit addressing (and the 1.9 M-point workloads below 4 GiB use the 32-bit form: test_bench_workload_matches_oracle)."""
    from d3fields_amd import create_init_grid, synth
    V, H, W, C = 4, 480, 640, 384
    maps = {"dino_feats": synth.random_map(V, 48, 64, C, seed=1, device=dev)}
    f, sc = fusion_for(dev, V, H, W, maps)
    grid, shape = create_init_grid(box_for(144, 144, 144, 0.004), 0.004)
    assert tuple(int(v) for v in shape) == (144, 144, 144) and grid.shape[0] * C * 4 > (1 << 32)
    pts = grid.to(dev)
    with torch.no_grad():
        f.record_plans = True
        out = f.batch_eval(pts, return_names=["dino_feats"])
        assert "window" in f.last_plan()["kernel"]
        with knobs(D3F_EXP_WINDOW=-1, D3F_EXP_RUNS=-1):
            ref = f.batch_eval(pts, return_names=["dino_feats"])
    for k in ref:
        assert torch.equal(out[k], ref[k]), k
    assert float(out["dino_feats"][-1].abs().sum()) >= 0.0 and bool(out["valid_mask"].any())


def test_async_probes_follow_the_data_and_never_change_results(dev):
    """Without the per-tensor cache (bench.py's mode) the shim launches on the verdict of the last FINISHED probes of a
    query of the same size and refreshes it asynchronously: a cloud that follows a grid of the same size is walked with
    the grid's (now meaningless) dims once -- still bit-identical -- and the hint then turns."""
    from d3fields_amd import create_init_grid, synth, _lib
    V, H, W, C = 4, 96, 128, 96
    maps = {"dino_feats": synth.random_map(V, H, W, C, seed=1, device=dev)}
    f, sc = fusion_for(dev, V, H, W, maps)
    f.cache_point_order = False
    grid = create_init_grid(box_for(64, 33, 37, 0.004), 0.004)[0].to(dev)
    n = grid.shape[0]
    cloud = (synth.random_cloud(n, seed=4) * 0.3).to(dev)
    shuffled = grid[torch.randperm(n, generator=torch.Generator().manual_seed(2)).to(dev)].contiguous()
    with torch.no_grad():
        f.tuning_flags = _lib.TUNE_NO_REORDER
        want = {id(t): f.batch_eval(t, return_names=["dino_feats"]) for t in (grid, cloud, shuffled)}
        f.tuning_flags = _lib.TUNE_FORCE_REORDER
        f._hints.clear()
        seen = []
        for t in (grid, grid, cloud, cloud, cloud, shuffled, shuffled, grid, grid):
            got = f.batch_eval(t, return_names=["dino_feats"])
            for k in got:
                assert torch.equal(got[k], want[id(t)][k]), k
            torch.cuda.synchronize(dev)
            f._poll_probes()
            seen.append(f._hints[n][0])
    assert seen[0] == (64, 33, 37) and seen[1] == (64, 33, 37)
    assert seen[2] is None and seen[4] is None and seen[6] is None           # the probes of the cloud / shuffled grid finished
    assert seen[8] == (64, 33, 37)
    assert len(f._pending) == 0


# ---- the output rows' store flavours: same launch x 50, bitwise equal ----------------------------------------------------------
@pytest.mark.parametrize("policy", ["nt", "sc1"])
@pytest.mark.parametrize("workload", ["window", "sliced"])
def test_row_stores_are_deterministic(dev, workload, policy):
    """The fused rows leave as non-temporal stores (round 4): `__builtin_nontemporal_store` in the sliced / generic kernels,
    `global_store_dwordx4 ... sc1 nt` as inline asm in the window kernel (a flagged point's row is stored twice by the same lane
    there, second store wins).  Inline-asm stores are outside the compiler's hazard tracking: round 2 found a VALU write
    scheduled right behind its `... sc1` store tearing dwords of some lanes, nondeterministically, in the two-vectors-per-lane
    window kernel (fixed with `s_nop 1` inside the asm; that form still exists behind D3F_EXP_STORE=1, experiments builds).
    Fifty launches of the window kernel and of the channel-sliced one must be bitwise equal to each other and to the direct
    gather."""
    if policy == "sc1" and not experiments():
        pytest.skip("the sc1 store policy is selectable in experiments builds only (D3F_EXP_STORE=1)")
    with knobs(D3F_EXP_STORE=1 if policy == "sc1" else 0):
        _row_stores_are_deterministic(dev, workload)


def _row_stores_are_deterministic(dev, workload):
    from d3fields_amd import create_init_grid, synth, _lib
    V, H, W = 4, 480, 640
    fhw = (48, 64) if workload == "window" else (240, 320)
    maps = {"dino_feats": synth.random_map(V, fhw[0], fhw[1], 384, seed=1, device=dev)}
    f, sc = fusion_for(dev, V, H, W, maps)
    f.record_plans = True
    pts = create_init_grid(synth.WORK_BOX, 0.0107)[0].to(dev)                            # 74 x 65 x 20 points
    with torch.no_grad():
        if workload == "sliced":
            f.tuning_flags = _lib.TUNE_FORCE_REORDER                                     # the maps are small here: force the walk
        first = f.batch_eval(pts, return_names=["dino_feats"])
        want = "fused_eval_window_kernel<1, 1, 4, 256, 16, 4, false>" if workload == "window" else "fused_eval_sliced_kernel<5, 2, 7>"
        assert f.last_plan()["kernel"] == want, f.last_plan()
        for i in range(50):
            again = f.batch_eval(pts, return_names=["dino_feats"])
            assert torch.equal(again["dino_feats"], first["dino_feats"]), "launch %d differs" % i
        with knobs(D3F_EXP_WINDOW=-1, D3F_EXP_RUNS=-1):
            ref = f.batch_eval(pts, return_names=["dino_feats"])
    assert torch.equal(first["dino_feats"], ref["dino_feats"])
