"""Offline (CPU, numpy) model of the LDS-window kernel's texel windows on CLOUDS, for different point orders and tilings.

    python scripts/sim_cloud_tiles.py [workload ...]

For every 64-point tile of the processing order the window kernel boxes the tile's points, projects the box's eight corners into
every view and stages the texel rectangle they span (fuse_eval.hip, fused_eval_window_body step 2).  A rectangle that does not fit
the pool sends its pairs to the global gather.  This script rebuilds exactly that for a cloud and prints, per order: texels staged
per tile, the share of tiles whose windows overflow a pool of P slots, and the share of valid (point, view) pairs left outside.
Orders: `morton` = round 4's (counting sort by a prefix of the 27-bit Morton key of 4-mm cells, arrival order inside a counting
cell, then every 64 consecutive slots sorted by the full key); `hilbert` = the same pipeline on a 27-bit Hilbert key;
`hilbert_full` = an exact sort by the Hilbert key.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench   # noqa: E402
from d3fields_amd import create_init_grid, synth   # noqa: E402


def spread3(x):
    x = x & 0x3ff
    x = (x | (x << 16)) & 0x030000ff
    x = (x | (x << 8)) & 0x0300f00f
    x = (x | (x << 4)) & 0x030c30c3
    x = (x | (x << 2)) & 0x09249249
    return x


def morton_key(q):
    return spread3(q[:, 0] & 511) | (spread3(q[:, 1] & 511) << 1) | (spread3(q[:, 2] & 511) << 2)


def hilbert_key(q, bits=9):
    """Skilling's transpose algorithm (AIP Conf. Proc. 707, 2004): axes -> Hilbert index, vectorised."""
    X = [(q[:, k] & ((1 << bits) - 1)).astype(np.int64).copy() for k in range(3)]
    M = 1 << (bits - 1)
    Q = M
    while Q > 1:
        P = Q - 1
        for i in range(3):
            hit = (X[i] & Q) != 0
            X[0] = np.where(hit, X[0] ^ P, X[0])
            t = np.where(hit, 0, (X[0] ^ X[i]) & P)
            X[0] ^= t
            X[i] ^= t
        Q >>= 1
    for i in range(1, 3):
        X[i] ^= X[i - 1]
    t = np.zeros_like(X[0])
    Q = M
    while Q > 1:
        t = np.where((X[2] & Q) != 0, t ^ (Q - 1), t)
        Q >>= 1
    for i in range(3):
        X[i] ^= t
    # interleave: bit b of X[0] is the most significant of digit b
    key = np.zeros_like(X[0])
    for b in range(bits - 1, -1, -1):
        for i in range(3):
            key = (key << 1) | ((X[i] >> b) & 1)
    return key


def pipeline_order(key, n, rng):
    """counting sort by key prefix (>= 4 counters per point, 2^15..2^21), random arrival order inside, 64-slot refinement"""
    bits = 15
    while bits < 21 and (1 << bits) < 4 * n:
        bits += 1
    coarse = key >> (27 - bits)
    arrival = rng.permutation(n)
    o = arrival[np.argsort(coarse[arrival], kind="stable")]
    # refinement: each 64-slot window sorted by (key, index)
    pad = (-n) % 64
    oo = np.concatenate([o, np.full(pad, -1, np.int64)]).reshape(-1, 64)
    kk = np.where(oo >= 0, key[np.clip(oo, 0, n - 1)], 1 << 40)
    idx = np.argsort(kk * (1 << 22) + np.where(oo >= 0, oo, 0), axis=1, kind="stable")
    oo = np.take_along_axis(oo, idx, 1).reshape(-1)
    return oo[oo >= 0]


def tile_stats(name, pts, order, V, H, W, fh, fw, K, Rt, depth, mu, pool, T=64, label=""):
    n = len(order)
    nt = n // T
    P = pts[order[: nt * T]].reshape(nt, T, 3)
    lo, hi = P.min(1), P.max(1)
    corners = np.stack([np.where(((np.arange(8) >> k) & 1)[None, :] == 1, hi[:, k:k + 1], lo[:, k:k + 1]) for k in range(3)], -1)   # [nt,8,3]
    tex = np.zeros((nt, V), np.int64)
    inside_pairs = np.zeros(nt)
    valid_pairs = np.zeros(nt)
    run = np.zeros(nt, np.int64)
    for v in range(V):
        M = (K[v] @ Rt[v]).astype(np.float64)
        c = corners @ M[:, :3].T + M[:, 3]
        ix = (c[..., 0] / c[..., 2]) / (W - 1) * (fw - 1)
        iy = (c[..., 1] / c[..., 2]) / (H - 1) * (fh - 1)
        x0 = np.clip(np.floor(ix.min(1) - 1e-3), 0, fw - 1); x1 = np.clip(np.floor(ix.max(1) + 1e-3) + 1, 0, fw - 1)
        y0 = np.clip(np.floor(iy.min(1) - 1e-3), 0, fh - 1); y1 = np.clip(np.floor(iy.max(1) + 1e-3) + 1, 0, fh - 1)
        bw, bh = (x1 - x0 + 1).astype(np.int64), (y1 - y0 + 1).astype(np.int64)
        tex[:, v] = bw * bh
        rows = np.minimum(bw * bh, pool - run) // bw
        rows = np.where(rows < 2, 0, rows)
        # the points themselves
        q = P @ M[:, :3].T + M[:, 3]
        u, w_ = q[..., 0] / q[..., 2], q[..., 1] / q[..., 2]
        rx, ry = np.rint(u).astype(np.int64), np.rint(w_).astype(np.int64)
        inb = (rx >= 0) & (rx < W) & (ry >= 0) & (ry < H)
        d = np.where(inb, depth[v][np.clip(ry, 0, H - 1), np.clip(rx, 0, W - 1)], 0.0)
        valid = (d > 0) & (d - q[..., 2] > -mu)
        px = np.floor(u / (W - 1) * (fw - 1)); py = np.floor(w_ / (H - 1) * (fh - 1))
        inmap = (px >= 0) & (px <= fw - 2) & (py >= 0) & (py <= fh - 2)
        ax = px - x0[:, None]; ay = py - y0[:, None]
        ins = inmap & (ax >= 0) & (ax + 1 < bw[:, None]) & (ay >= 0) & (ay + 1 < rows[:, None])
        inside_pairs += (valid & ins).sum(1)
        valid_pairs += valid.sum(1)
        run += rows * bw
    tot = tex.sum(1)
    print("%-10s %-13s T=%3d pool %3d | texels/tile mean %5.1f p50 %3d p90 %3d p99 %4d | tiles over pool %5.1f %% | valid pairs outside %5.2f %% | tiles with any outside %5.1f %%"
          % (name, label, T, pool, tot.mean(), np.percentile(tot, 50), np.percentile(tot, 90), np.percentile(tot, 99),
             100.0 * (tot > pool).mean(), 100.0 * (1 - inside_pairs.sum() / max(valid_pairs.sum(), 1)),
             100.0 * (inside_pairs < valid_pairs).mean()))


def cloud_of(name, kind):
    w = bench.WORKLOADS[name]
    V, H, W = w["V"], w["H"], w["W"]
    sc = synth.make_scene(V, H, W, "smooth")
    K, Rt, depth = sc["K"].numpy().astype(np.float64), sc["pose"].numpy().astype(np.float64), sc["depth"].numpy()
    if kind == "random":
        pts = synth.random_cloud(w.get("N_cloud", w["N"]), seed=3).numpy().astype(np.float64)
    else:       # surface: grid points with valid & |dist| < step, flat-index order (extract_mesh's vertices, fusion.py:1313-1330)
        g, _ = create_init_grid(synth.WORK_BOX, w["step"])
        g = g.numpy().astype(np.float64)
        dsum = np.zeros(len(g)); cnt = np.zeros(len(g))
        for v in range(V):
            M = K[v] @ Rt[v]
            q = g @ M[:, :3].T + M[:, 3]
            u, w_ = q[:, 0] / q[:, 2], q[:, 1] / q[:, 2]
            rx, ry = np.rint(u).astype(np.int64), np.rint(w_).astype(np.int64)
            inb = (rx >= 0) & (rx < W) & (ry >= 0) & (ry < H)
            d = np.where(inb, depth[v][np.clip(ry, 0, H - 1), np.clip(rx, 0, W - 1)], 0.0)
            dist = d - q[:, 2]
            valid = (d > 0) & (dist > -0.02)
            dsum += np.clip(dist, -0.02, 0.02) * valid; cnt += valid
        dist = np.where(cnt > 0, dsum / (cnt + 1e-6), 1e3)
        pts = g[(cnt > 0) & (np.abs(dist) < w["step"])]
    return w, V, H, W, K, Rt, depth, pts


def main():
    names = sys.argv[1:] or ["c2_patch:random", "c3_patch:surface", "c4_patch:random", "c5_track:random"]
    for spec in names:
        name, kind = spec.split(":")
        w, V, H, W, K, Rt, depth, pts = cloud_of(name, kind)
        fh, fw = w["fhw"]
        n = len(pts)
        q = np.floor(pts / 0.004).astype(np.int64)
        rng = np.random.default_rng(0)
        pool = {4: 80, 8: 136}.get(V, 80)
        print("== %s %s: %d points, %d views, map %dx%d" % (name, kind, n, V, fh, fw))
        mk, hk = morton_key(q), hilbert_key(q)
        orders = {
            "morton": pipeline_order(mk, n, rng),
            "hilbert": pipeline_order(hk, n, rng),
            "hilbert_full": np.argsort(hk, kind="stable"),
            "morton_full": np.argsort(mk, kind="stable"),
        }
        for label, o in orders.items():
            tile_stats(name, pts, o, V, H, W, fh, fw, K, Rt, depth, 0.02, pool, 64, label)
        tile_stats(name, pts, orders["hilbert_full"], V, H, W, fh, fw, K, Rt, depth, 0.02, pool, 32, "hilbert_full")


if __name__ == "__main__":
    main()
