"""Offline: how many distinct 128-byte lines does a QUAD of lanes (four consecutive points of the caller's order) touch in the nearest-depth lookup of
the distance-only pass, for row-major depth maps and for tiled copies (tile = cols x rows of 32 pixels)?  The texture addresser serves a quad per
cycle when its four lanes fall into one line and a lane per cycle otherwise (what-if builds of session 44/45: 1.75 -> 1.28 / 1.35 ms with coherent
quads), so the mean number of lines per quad is the cost of a lookup.  Geometry: bench.py's workloads (synth ring cameras), lattice steps 1 / 2 / 4 / 5 mm.
    python scripts/notebook/sim_depth_tiles.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from d3fields_amd import synth      # noqa: E402

SHAPES = [(32, 1), (16, 2), (8, 4), (4, 8), (2, 16), (1, 32)]


def lines_per_quad(ix, iy, ok, cols, rows, W):
    tw = (W + cols - 1) // cols
    tile = (iy // rows) * tw + ix // cols
    tile = np.where(ok, tile, -1).reshape(-1, 4)
    s = np.sort(tile, axis=1)
    distinct = 1 + (s[:, 1:] != s[:, :-1]).sum(axis=1)
    return distinct.mean()


def main():
    for V, H, W in ((4, 480, 640), (8, 720, 1280)):
        K, Rt = synth.ring_cameras(V, H, W)
        box = synth.WORK_BOX
        r = np.random.default_rng(0)
        for step in (0.001, 0.002, 0.004, 0.005):
            nz = int(np.ceil((box["z_upper"] - box["z_lower"]) / step))
            ncol = 4000
            x = r.uniform(box["x_lower"], box["x_upper"], ncol)
            y = r.uniform(box["y_lower"], box["y_upper"], ncol)
            z0 = r.integers(0, max(nz - 64, 1), ncol)
            zz = box["z_lower"] + (z0[:, None] + np.arange(64)[None, :]) * step + step / 2
            pts = np.stack([np.repeat(x, 64), np.repeat(y, 64), zz.reshape(-1)], 1)
            out = []
            for cols, rows in SHAPES:
                acc = []
                for v in range(V):
                    M = K[v].astype(np.float64) @ Rt[v].astype(np.float64)
                    c = pts @ M[:, :3].T + M[:, 3]
                    u, w = c[:, 0] / c[:, 2], c[:, 1] / c[:, 2]
                    ix, iy = np.rint(u).astype(np.int64), np.rint(w).astype(np.int64)
                    ok = (ix >= 0) & (ix < W) & (iy >= 0) & (iy < H)
                    acc.append(lines_per_quad(np.clip(ix, 0, W - 1), np.clip(iy, 0, H - 1), ok, cols, rows, W))
                out.append(np.mean(acc))
            print("V=%d %dx%d step %.0f mm: lines per quad " % (V, H, W, step * 1e3) + "  ".join("%dx%d %.2f" % (c, rw, o) for (c, rw), o in zip(SHAPES, out)))


main()
