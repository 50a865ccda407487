"""How often do neighbouring points of a window-kernel brick share their texel cell?  (open item 0: register reuse of corner
vectors would need the four lane groups of a wave to agree.)  CPU only: python scripts/sim_cell_reuse.py"""
import sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..', '..'))
from d3fields_amd import synth
import bench
def sim(name):
    w = bench.WORKLOADS[name]
    V, H, W = w["V"], w["H"], w["W"]; fh, fw = w["fhw"]
    sc = synth.make_scene(V, H, W)
    K = sc["K"].numpy().astype(np.float64); Rt = sc["pose"].numpy().astype(np.float64)
    step = w["step"]; box = synth.WORK_BOX
    xs = np.arange(box["x_lower"], box["x_upper"], step) + step / 2
    ys = np.arange(box["y_lower"], box["y_upper"], step) + step / 2
    zs = np.arange(box["z_lower"], box["z_upper"], step) + step / 2
    if "slabs" in w: xs = xs[: len(xs) // w["slabs"]]
    rng = np.random.default_rng(0)
    same_all = same_one = tot = 0
    uniq = []
    for _ in range(400):
        bx = rng.integers(0, len(xs) // 4) * 4; by = rng.integers(0, len(ys) // 4) * 4; bz = rng.integers(0, len(zs) // 4) * 4
        X, Y, Z = np.meshgrid(xs[bx:bx + 4], ys[by:by + 4], zs[bz:bz + 4], indexing="ij")       # [x=k, y=wave, z=lane group]
        P = np.stack([X, Y, Z, np.ones_like(X)], -1)
        for v in range(V):
            M = K[v] @ Rt[v][:3]
            q = P @ M.T
            u = q[..., 0] / q[..., 2]; t = q[..., 1] / q[..., 2]
            ix = u / (W - 1) * (fw - 1); iy = t / (H - 1) * (fh - 1)
            cx = np.floor(ix).astype(int); cy = np.floor(iy).astype(int)
            cell = cx * 100000 + cy
            uniq.append(len(np.unique(cell)))
            eq = cell[1:] == cell[:-1]            # k -> k+1, [3, y, z]
            same_one += eq.sum(); tot += eq.size
            same_all += eq.all(axis=2).sum() * 4      # all four z lane groups of a wave
    print("%-10s cells per (brick, view) %.1f; same cell k->k+1: one lane group %.2f, all four of a wave %.2f" % (name, np.mean(uniq), same_one / tot, same_all / tot))
for n in ("c2_patch", "c3_patch", "ref_patch", "c4_patch"): sim(n)
