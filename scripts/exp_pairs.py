"""Round 4 experiment (EXPERIMENTS build): C2-dense as TWO passes over view pairs -- views (0,1), then
(2,3) continuing the stored sums -- each pass walking the points in slabs around its pair's epipolar planes.

    D3F_BUILD_EXPERIMENTS=1 python scripts/exp_pairs.py

The slab orders are computed here with torch (not timed: the question is what the KERNELS gain; a product version would
need a ~0.1 ms counting sort per pass and step).  Kernel times from HIP events around each fused launch."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from d3fields_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
assert lib.d3f_build_has_experiments()
f, pts, names, w, sc = bench.build_workload("c2_dense", dev, 0, 1)
n = pts.shape[0]
with torch.no_grad():
    base = f.batch_eval(pts, return_names=names)
torch.cuda.synchronize()


def slab_order(a, b, thick=1.0, bin_m=0.02):
    pose = sc["pose"].double().to(dev)
    c = [-(pose[v][:, :3].T @ pose[v][:, 3]) for v in range(4)]
    basev = (c[b] - c[a]) / torch.linalg.norm(c[b] - c[a])
    rel = pts.double() - c[a]
    along = rel @ basev
    perp = rel - along[:, None] * basev
    e1 = torch.linalg.cross(basev, torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64, device=dev))
    e1 = e1 / torch.linalg.norm(e1)
    e2 = torch.linalg.cross(basev, e1)
    ang = torch.atan2(perp @ e1, perp @ e2)
    dist_axis = torch.linalg.norm(perp, dim=1)
    slab = torch.floor((ang - ang.min()) / (0.005 * thick / dist_axis.median())).long()
    bb = torch.floor(along / bin_m).long()
    bb = bb - bb.min()
    key = (slab * (int(bb.max()) + 1) + bb).double() * 4.0 + dist_axis          # dist_axis < 4 m
    return torch.argsort(key).to(torch.int32)


def seg(nn):
    return (nn * 4 + 255) // 256 * 256


ws_bytes = lib.d3f_eval_workspace_bytes(n)
views, keep, V = f._views(dev)
m = f.curr_obs_torch["dino_feats"]
maps = (_lib.ChannelMap * 1)(_lib.ChannelMap(m.data_ptr(), m.shape[1], m.shape[2], m.shape[3], 0, m.stride(0), m.stride(1), m.stride(2)))
dist = torch.empty(n, device=dev)
valid = torch.empty(n, dtype=torch.bool, device=dev)
out = torch.empty((n, m.shape[3]), device=dev)
fused = (ctypes.c_void_p * 1)(out.data_ptr())
stream = _lib.current_stream_handle(dev)
FL = _lib.FLAG_FINITE_MAPS | _lib.FLAG_REUSE_POINT_ORDER | _lib.TUNE_FORCE_REORDER


def workspace_with(order):
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=dev)
    ws[2 * seg(n):2 * seg(n) + 4 * n].view(torch.int32).copy_(order)
    return ws


def launch(ws, lo, hi, acc, timed=None):
    os.environ["D3F_EXP_VIEW_LO"], os.environ["D3F_EXP_VIEW_HI"], os.environ["D3F_EXP_VIEW_ACC"] = str(lo), str(hi), str(acc)
    if timed is not None:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); b.record()
        lib.d3f_profile_next_eval(ctypes.c_void_p(a.cuda_event), ctypes.c_void_p(b.cuda_event))
        timed.append((a, b))
    _lib.check(lib.d3f_eval(ctypes.byref(views), _lib.ptr(pts), n, maps, 1, f.mu, FL, _lib.ptr(dist), _lib.ptr(valid), fused, None,
                            _lib.ptr(ws), ws_bytes, stream))


def ms(pairs):
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in pairs)
    return sum(t) / len(t), t[0]


def run(tag, passes, steps=20):
    for _ in range(3):
        for ws, lo, hi, acc in passes:
            launch(ws, lo, hi, acc)
    per = [[] for _ in passes]
    for _ in range(steps):
        for k, (ws, lo, hi, acc) in enumerate(passes):
            launch(ws, lo, hi, acc, per[k])
    t = [ms(p) for p in per]
    same = all(torch.equal(base[k], o) for k, o in (("dist", dist), ("valid_mask", valid), ("dino_feats", out)))
    print("%-64s %s  sum %.3f ms (min %.3f)  bit-identical to the default launch: %s" %
          (tag, " + ".join("%.3f" % a for a, _ in t), sum(a for a, _ in t), sum(b for _, b in t), same), flush=True)


identity = torch.arange(n, dtype=torch.int32, device=dev)
o01 = slab_order(0, 1)
o23 = slab_order(2, 3)
w_id, w01, w23 = workspace_with(identity), workspace_with(o01), workspace_with(o23)
run("one pass, all four views, caller order (z fastest)", [(w_id, 0, 4, 0)])
run("one pass, all four views, slab order of pair (0,1)", [(w01, 0, 4, 0)])
run("two passes (0,1) then (2,3), caller order both", [(w_id, 0, 2, 0), (w_id, 2, 4, 1)])
run("two passes, epipolar slabs of each pair, 1 step thick", [(w01, 0, 2, 0), (w23, 2, 4, 1)])
for thick in (2.0, 4.0):
    a, b = workspace_with(slab_order(0, 1, thick)), workspace_with(slab_order(2, 3, thick))
    run("two passes, epipolar slabs, %.0f steps thick" % thick, [(a, 0, 2, 0), (b, 2, 4, 1)])
for bin_m in (0.01, 0.04):
    a, b = workspace_with(slab_order(0, 1, 1.0, bin_m)), workspace_with(slab_order(2, 3, 1.0, bin_m))
    run("two passes, epipolar slabs, 1 step thick, %.0f-mm bins along the baseline" % (bin_m * 1e3), [(a, 0, 2, 0), (b, 2, 4, 1)])
for k in ("D3F_EXP_VIEW_LO", "D3F_EXP_VIEW_HI", "D3F_EXP_VIEW_ACC"):
    os.environ.pop(k, None)
with torch.no_grad():
    f.batch_eval(pts, return_names=names)
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        f.batch_eval(pts, return_names=names)
    torch.cuda.synchronize()
    print("default launch (lattice brick walk, channel-sliced): %.3f ms per step" % ((time.perf_counter() - t0) / 20 * 1e3))
