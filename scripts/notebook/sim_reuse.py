"""Offline reuse analysis of the C2-dense gather (CPU only, numpy): the IDEAL hit rate of a cache that holds exactly
the texels of a window of W consecutive points, for different processing orders of the 985 600-point grid.

    python scripts/sim_reuse.py            # prints the table that DESIGN.md section 5.3 quotes

For every (point, valid view) the kernel requests the four bilinear corner texels (1536 B each at C = 384).  A texel is
re-fetched from beyond the cache unless another point of the same window already touched it, so
hit(W) = 1 - (unique texels per window) / (corner requests per window), summed over the walk.  The per-XCD L2 (4 MiB =
2730 texels of 1536 B) holds a ~512-point window; the 256 MiB Infinity Cache a ~32 k-point window.  Orders compared:
the caller's z-fastest order, the Morton walk of 16-mm cells (round 1), the closed-form blocked lattice walk (round 2),
anisotropic bricks, view-frustum pixel-tile-major orders and an epipolar-slab order.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from d3fields_amd import create_init_grid, synth   # noqa: E402

V, H, W_IMG = 4, 480, 640
sc = synth.make_scene(V, H, W_IMG, "smooth")
pts, shape = create_init_grid(synth.WORK_BOX, 0.005)
pts = pts.numpy().astype(np.float64)
nx, ny, nz = shape
N = pts.shape[0]
K, Rt, depth = sc["K"].numpy().astype(np.float64), sc["pose"].numpy().astype(np.float64), sc["depth"].numpy()
mu = 0.02

ids = np.full((N, V, 4), -1, np.int64)
pix = np.zeros((N, V, 2))
zc_all = np.zeros((N, V))
for v in range(V):
    M = K[v] @ Rt[v]
    cam = pts @ M[:, :3].T + M[:, 3]
    zc = cam[:, 2]
    u, w = cam[:, 0] / zc, cam[:, 1] / zc
    pix[:, v, 0], pix[:, v, 1], zc_all[:, v] = u, w, zc
    rx, ry = np.rint(u).astype(np.int64), np.rint(w).astype(np.int64)
    inb = (rx >= 0) & (rx < W_IMG) & (ry >= 0) & (ry < H)
    d = np.where(inb, depth[v][np.clip(ry, 0, H - 1), np.clip(rx, 0, W_IMG - 1)], 0.0)
    valid = (d > 0) & (d - zc > -mu)
    x0, y0 = np.floor(u).astype(np.int64), np.floor(w).astype(np.int64)
    for c, (dx, dy) in enumerate(((0, 0), (1, 0), (0, 1), (1, 1))):
        x, y = np.clip(x0 + dx, 0, W_IMG - 1), np.clip(y0 + dy, 0, H - 1)
        ids[:, v, c] = np.where(valid, (v * H + y) * W_IMG + x, -1)
ids = ids.reshape(N, V * 4)
total_req = int((ids >= 0).sum())
print("grid %dx%dx%d = %d points, %.1f %% of the (point, view) pairs valid, %.2f M corner requests (%.1f GB at 1536 B)"
      % (nx, ny, nz, N, 100.0 * (ids[:, ::4] >= 0).mean(), total_req / 1e6, total_req * 1536 / 1e9))
print("unique texels touched by the whole grid: %.2f M (%.2f GB) -> ideal hit rate of an infinite cache %.1f %%"
      % (len(np.unique(ids[ids >= 0])) / 1e6, len(np.unique(ids[ids >= 0])) * 1536 / 1e9,
         100.0 * (1 - len(np.unique(ids[ids >= 0])) / total_req)))

ix, iy, iz = np.unravel_index(np.arange(N), (nx, ny, nz))


def spread3(x):
    x = x & 0x3ff
    x = (x | (x << 16)) & 0x030000ff
    x = (x | (x << 8)) & 0x0300f00f
    x = (x | (x << 4)) & 0x030c30c3
    x = (x | (x << 2)) & 0x09249249
    return x


def morton_cells(cell):
    q = np.floor(pts / cell).astype(np.int64)
    key = spread3(q[:, 0] & 127) | (spread3(q[:, 1] & 127) << 1) | (spread3(q[:, 2] & 127) << 2)
    return np.argsort(key, kind="stable")


def bricks(bx, by, bz, inner="rowmajor"):
    key = ((ix // bx) * ((ny + by - 1) // by) + (iy // by)) * ((nz + bz - 1) // bz) + (iz // bz)
    sub = ((ix % bx) * by + (iy % by)) * bz + (iz % bz)
    return np.argsort(key * (bx * by * bz) + sub, kind="stable")


def lattice_walk(t=(2, 2, 2)):
    """the three-level blocked order of fuse_eval.hip (16^3 / 8^3 / 4^3 tiles, clipped), tiles of t points"""
    tx, ty, tz = ix // t[0], iy // t[1], iz // t[2]
    key = np.zeros(N, np.int64)
    ex, ey, ez = (nx + t[0] - 1) // t[0], (ny + t[1] - 1) // t[1], (nz + t[2] - 1) // t[2]
    mult = 1
    # lexicographic key: (macro-brick, sub-brick, mini-brick, tile, point inside tile), each row-major
    comps = []
    for b in (16, 8, 4, 1):
        comps.append((tx // b, ty // b, tz // b))
    key = np.zeros(N, np.int64)
    for lvl, (a, b_, c) in enumerate(comps):
        if lvl == 0:
            loc = (a * (ey // 16 + 1) + b_) * (ez // 16 + 1) + c
        else:
            bb = (16, 8, 4, 1)[lvl - 1] // (16, 8, 4, 1)[lvl]
            loc = ((a % bb) * bb + (b_ % bb)) * bb + (c % bb)
        key = key * 4096 + loc
    sub = ((ix % t[0]) * t[1] + (iy % t[1])) * t[2] + (iz % t[2])
    return np.argsort(key * 64 + sub, kind="stable")


def view_tile_major(v, tile, depth_minor=True):
    """points sorted by the pixel tile of view v they project to, then by depth along the ray"""
    tx_, ty_ = np.floor(pix[:, v, 0] / tile).astype(np.int64), np.floor(pix[:, v, 1] / tile).astype(np.int64)
    tkey = (ty_ - ty_.min()) * (tx_.max() - tx_.min() + 1) + (tx_ - tx_.min())
    order = np.lexsort((zc_all[:, v], tkey)) if depth_minor else np.argsort(tkey, kind="stable")
    return order


def brick_then_view_tile(v, brick=32, tile=8):
    """32 k-point macro-bricks (Infinity-Cache window) walked view-frustum pixel-tile-major inside"""
    bkey = ((ix // brick) * (ny // brick + 1) + (iy // brick)) * (nz // brick + 1) + (iz // brick)
    tx_, ty_ = np.floor(pix[:, v, 0] / tile).astype(np.int64), np.floor(pix[:, v, 1] / tile).astype(np.int64)
    tkey = (ty_ - ty_.min()) * (tx_.max() - tx_.min() + 1) + (tx_ - tx_.min())
    return np.lexsort((zc_all[:, v], tkey, bkey))


def epipolar_slabs(thick=2):
    """planes through the baseline of cameras 0 and 2 (both see such a plane as a line): slab index, then position in it"""
    c = [-(Rt[v][:, :3].T @ Rt[v][:, 3]) for v in range(V)]
    base = (c[2] - c[0]) / np.linalg.norm(c[2] - c[0])
    rel = pts - c[0]
    perp = rel - np.outer(rel @ base, base)
    ang = np.arctan2(perp @ np.cross(base, [0, 0, 1.0]), perp[:, 2])
    slab = np.floor((ang - ang.min()) / (0.005 * thick / 1.0)).astype(np.int64)      # ~ thick grid steps at 1 m
    return np.lexsort((np.linalg.norm(perp, axis=1), rel @ base, slab))


def hit_rates(order, windows):
    seq = ids[order]
    out = []
    for wnd in windows:
        nch = N // wnd
        blk = seq[:nch * wnd].reshape(nch, wnd * V * 4)
        blk = np.sort(blk, axis=1)
        valid = blk >= 0
        uniq = ((blk[:, 1:] != blk[:, :-1]) & valid[:, 1:]).sum() + valid[:, 0].sum()
        out.append(1.0 - uniq / valid.sum())
    return out


WINDOWS = (64, 512, 1024, 2048, 4096, 32768)
orders = [
    ("caller order (z fastest)", np.arange(N)),
    ("Morton, 16-mm cells, stable (round 1)", morton_cells(0.016)),
    ("Morton, 4-mm cells", morton_cells(0.004)),
    ("blocked lattice walk 16/8/4 tiles of 2x2x2 (round 2)", lattice_walk()),
    ("bricks 8x8x8 row-major", bricks(8, 8, 8)),
    ("bricks 4x4x32 (z columns)", bricks(4, 4, 32)),
    ("bricks 16x16x2 (xy slabs)", bricks(16, 16, 2)),
    ("bricks 6x6x14", bricks(6, 6, 14)),
    ("bricks 16x4x8", bricks(16, 4, 8)),
    ("view-0 8x8-pixel tiles x full depth", view_tile_major(0, 8)),
    ("view-0 32x32-pixel tiles x full depth", view_tile_major(0, 32)),
    ("32^3 macro-brick, view-0 8x8-pixel tiles inside", brick_then_view_tile(0, 32, 8)),
    ("32^3 macro-brick, view-0 16x16-pixel tiles inside", brick_then_view_tile(0, 32, 16)),
    ("epipolar slabs of cameras 0/2, 2 steps thick", epipolar_slabs(2)),
    ("epipolar slabs of cameras 0/2, 8 steps thick", epipolar_slabs(8)),
]
print("\nideal hit rate [%%] of a cache holding a window of W consecutive points (4 MiB L2 ~ W = 512 at 1536 B/texel;")
print("with the channels of a texel split over the eight XCDs -- 192 B per texel and L2 -- the same L2 holds W ~ 4096)\n")
print("%-58s" % "order" + "".join("%8d" % w for w in WINDOWS))
for name, order in orders:
    assert len(np.unique(order)) == N
    r = hit_rates(order, WINDOWS)
    print("%-58s" % name + "".join("%8.1f" % (100 * x) for x in r))


# ---- round 4: TWO passes over view PAIRS -- (0,1) then (2,3), the sequential sum order -- each pass
# walking the points in slabs around the pair's epipolar planes: a plane through both camera centres is seen as a line by
# BOTH cameras, so a thin slab of points touches two thin strips of texels instead of four unrelated 2-D patches.  Only the
# pair's own corner requests count in a pass.  The go / no-go bar set by the review: >= 85 % ideal hits at W = 4096.
def epipolar_slabs_pair(a, b, thick, inner="ray"):
    c = [-(Rt[v][:, :3].T @ Rt[v][:, 3]) for v in range(V)]
    base = (c[b] - c[a]) / np.linalg.norm(c[b] - c[a])
    rel = pts - c[a]
    perp = rel - np.outer(rel @ base, base)
    e1 = np.cross(base, [0.0, 0.0, 1.0])
    e1 /= np.linalg.norm(e1)
    e2 = np.cross(base, e1)
    ang = np.arctan2(perp @ e1, perp @ e2)
    dist_axis = np.linalg.norm(perp, axis=1)
    slab = np.floor((ang - ang.min()) / (0.005 * thick / np.median(dist_axis))).astype(np.int64)   # ~ `thick` grid steps at the median distance
    if inner == "ray":          # inside a slab: along the baseline, then away from it
        return np.lexsort((dist_axis, np.floor((rel @ base) / 0.02), slab))
    return np.lexsort((rel @ base, np.floor(dist_axis / 0.02), slab))


def hit_rates_views(order, windows, views):
    cols = np.concatenate([np.arange(v * 4, v * 4 + 4) for v in views])
    seq = ids[order][:, cols]
    out = []
    for wnd in windows:
        nch = N // wnd
        blk = np.sort(seq[:nch * wnd].reshape(nch, wnd * len(cols)), axis=1)
        valid = blk >= 0
        uniq = ((blk[:, 1:] != blk[:, :-1]) & valid[:, 1:]).sum() + valid[:, 0].sum()
        out.append(1.0 - uniq / max(valid.sum(), 1))
    return out


print("\ntwo passes over view pairs: ideal hit rate [%] of the PAIR's corner requests in a window of W consecutive points\n")
print("%-58s" % "pass / order" + "".join("%8d" % w for w in WINDOWS))
walk = lattice_walk()
for pair in ((0, 1), (2, 3), (0, 2), (1, 3)):
    print("%-58s" % ("views %s: blocked lattice walk (today's order)" % (pair,)) + "".join("%8.1f" % (100 * x) for x in hit_rates_views(walk, WINDOWS, pair)))
    for thick in (1, 2, 4, 8):
        for inner in ("ray", "ring"):
            o = epipolar_slabs_pair(pair[0], pair[1], thick, inner)
            print("%-58s" % ("views %s: epipolar slabs, %d steps thick, %s-major inside" % (pair, thick, inner)) +
                  "".join("%8.1f" % (100 * x) for x in hit_rates_views(o, WINDOWS, pair)))
