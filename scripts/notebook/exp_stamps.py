"""Phase stamps of the LDS-window kernel (EXPERIMENTS build, D3F_EXP_STAMPS=1): where a workgroup's lifetime goes.

    D3F_BUILD_EXPERIMENTS=1 D3F_EXP_STAMPS=1 python scripts/exp_stamps.py c2_patch [c3_patch c4_patch]

Lane 0 of wave 0 of every 64th workgroup writes s_memtime (shader cycles) at the phase boundaries of
fused_eval_window_body; this prints the mean cycles between consecutive boundaries and the share of the lifetime."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from d3fields_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
assert lib.d3f_build_has_experiments(), "needs the experiments build"
lib.d3f_exp_read_stamps.restype = ctypes.c_int
lib.d3f_exp_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int64]
for wl in sys.argv[1:] or ["c2_patch"]:
    wl, _, pk = wl.partition(":")
    f, pts, names, w, sc = bench.build_workload(wl, dev, 0, 1, pk or "grid")
    f.cache_point_order = False
    with torch.no_grad():
        for _ in range(4):
            f.batch_eval(pts, return_names=names)
    torch.cuda.synchronize()
    nwg = 4096
    buf = np.zeros(nwg * 32, dtype=np.uint64)
    rc = lib.d3f_exp_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
    assert rc == 0, rc
    st = buf.reshape(nwg, 32)
    used = st[:, 0] > 0
    k = int(st[used, 0].max())
    s = st[used][:, 1:1 + k].astype(np.int64)
    s = s[(st[used, 0] == k)]
    d = np.diff(s, axis=1)
    life = (s[:, -1] - s[:, 0]).mean()
    S = (k - 6) // 3
    labels = ["KRt/zero/box corners", "windows", "slot table + DMA(0) issue", "phase A"]
    for sl in range(S):
        labels += ["slice %d: wait pool (DMA, barrier)" % sl, "slice %d: point loop (wave 0)" % sl, "slice %d: barrier (other waves)" % sl]
    labels += ["DMA issue + rows drain (last)"]
    if k == 8:       # the register-rows kernel (fuse_rows.hip)
        print("   cells per brick: mean %.1f, ops per brick: mean %.1f" % (st[used][:, 30].mean(), st[used][:, 31].mean()))
        labels = ["KRt", "phase A (wave 0)", "phase A (barrier)", "ranks, ops, cells", "zero + cell loop", "row stores", "dist/valid, redone points, thin maps"]
    print("%s: %d sampled workgroups, %d stamps each, mean lifetime %.0f cycles" % (wl, s.shape[0], k, life))
    for i in range(d.shape[1]):
        lab = labels[i] if i < len(labels) else "?"
        print("   %-44s %8.0f cycles  %5.1f %%   (median %.0f)" % (lab, d[:, i].mean(), 100 * d[:, i].mean() / life, np.median(d[:, i])))
