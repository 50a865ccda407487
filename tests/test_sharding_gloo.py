"""CPU, world_size 2, gloo: the N>1 plumbing of d3fields_amd.sharding (shard -> evaluate ->
all-gather, ragged tails, partial gathers).  No GPU exists here, so the per-shard evaluator
is injected (the CPU oracle); on the GPU box the same code path calls the HIP kernel."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from d3fields_amd import sharding, synth
        from oracle import c_oracle as O
        V, H, W = 3, 32, 40
        sc = synth.make_scene(V, H, W, "smooth")
        feats = synth.random_map(V, 4, 5, 6, seed=1)
        pts = synth.random_cloud(n, seed=3)

        def evaluator(p, names):
            r = O.eval_field(sc["depth"], sc["K"], sc["pose"], p, [feats], return_inter=True)
            return {"dist": torch.from_numpy(r["dist"]), "valid_mask": torch.from_numpy(r["valid_mask"]),
                    "dino_feats": torch.from_numpy(r["sets"][0]), "dino_feats_inter": torch.from_numpy(r["inter"][0])}

        full = sharding.sharded_eval(None, pts, ["dino_feats"], evaluator=evaluator)
        part = sharding.sharded_eval(None, pts, ["dino_feats"], gather_keys=("dist", "valid_mask"), evaluator=evaluator)
        single = evaluator(pts, None)
        ok = all(torch.equal(full[k], single[k]) for k in single)
        lo, hi = part["local_range"]
        ok = ok and torch.equal(part["dist"], single["dist"]) and "dino_feats" not in part
        ok = ok and torch.equal(part["dino_feats_local"], single["dino_feats"][lo:hi])
        ok = ok and (lo, hi) == sharding.shard_bounds(n, rank, world)
        ok = ok and torch.equal(sharding.shard_points(pts), pts[lo:hi])
        # non-blocking form: collectives in flight until wait(); ragged shards: ONE padded all-gather per key (round 4; rounds 2-3:
        # one broadcast per non-empty rank), whose wait() also moves the rows from the staging buffer into place
        local = evaluator(pts[lo:hi], None)
        counts = [sharding.shard_bounds(n, r, world)[1] - sharding.shard_bounds(n, r, world)[0] for r in range(world)]
        af, works = sharding.all_gather_field(local, keys=("dist", "valid_mask", "dino_feats"), counts=counts, async_op=True)
        ok = ok and len(works) == 3
        for wk in works:
            wk.wait()
        ok = ok and all(torch.equal(af[k], single[k]) for k in af) and af["valid_mask"].dtype == torch.bool
        q.put((rank, bool(ok), full["valid_mask"].dtype == torch.bool, tuple(full["dino_feats_inter"].shape)))
    finally:
        dist.destroy_process_group()


# (world 8: the shape of the first real RCCL run -- eight shards, ragged (1003 = 3 x 126 + 5 x 125) and mostly EMPTY (5 points), every key
# incl. '<k>_inter' [V,n,C] gathered along its point axis)
@pytest.mark.parametrize("n,world", [(1001, 2), (1000, 2), (1, 2), (1000, 3), (2, 3), (1003, 8), (5, 8)])
def test_sharded_eval_world2(n, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok, is_bool, ishape in res:
        assert ok, "rank %d: gathered field differs from the single-process field" % rank
        assert is_bool and ishape == (3, n, 6)


_REC = np.dtype([("m", "<f4"), ("s", "<f4"), ("a", "<i8")])      # d3f_col_stat (include/d3fields_hip.h)


class _NumpySoftmaxKernels:
    """Stand-in for the three HIP steps of sharded_similarity_multi (same record format)."""

    @staticmethod
    def local(src, tgt, scale, code, row_offset):
        from oracle import c_oracle as O
        B1, B2 = src.shape[0], tgt.shape[0]
        rec = np.zeros(B2, _REC)
        if B1 == 0:
            rec["m"], rec["s"], rec["a"] = -np.inf, 0.0, np.iinfo(np.int64).max
            return torch.empty((0, B2)), torch.from_numpy(rec.view(np.uint8).reshape(B2, 16).copy())
        d = O.pairwise(src.numpy(), tgt.numpy(), scale, "l2" if code == 0 else "square", mode="dist")
        lg = (-d * np.float32(scale)).astype(np.float32)
        rec["m"] = lg.max(0)
        rec["s"] = np.exp(lg - rec["m"]).sum(0, dtype=np.float32)
        rec["a"] = lg.argmax(0) + row_offset
        return torch.from_numpy(d), torch.from_numpy(rec.view(np.uint8).reshape(B2, 16).copy())

    @staticmethod
    def merge(parts):
        rec = parts.numpy().reshape(parts.shape[0], parts.shape[1] * 16).view(_REC)      # [P, B2]
        M = rec["m"].max(0)
        with np.errstate(invalid="ignore"):
            w = np.where(np.isneginf(rec["m"]), 0.0, np.exp(rec["m"] - M)).astype(np.float32)
        out = np.zeros(rec.shape[1], _REC)
        out["m"], out["s"] = M, (rec["s"] * w).sum(0, dtype=np.float32)
        out["a"] = np.where(rec["m"] == M, rec["a"], np.iinfo(np.int64).max).min(0)
        return torch.from_numpy(out.view(np.uint8).reshape(-1, 16).copy()), torch.from_numpy(out["a"].copy())

    @staticmethod
    def apply(out, scale, merged):
        rec = merged.numpy().reshape(-1).view(_REC)
        return torch.from_numpy((np.exp(-out.numpy() * np.float32(scale) - rec["m"]) / rec["s"]).astype(np.float32))


class _NumpyTopkKernels:
    """Stand-in for the two HIP steps of sharded_knn_descriptors (d3f_topk_smallest / d3f_topk_merge)."""

    @staticmethod
    def local_topk(dist_local, k):
        d = dist_local.numpy()
        rows, cols = d.shape
        idx = np.full((k, cols), -1, np.int64)
        val = np.full((k, cols), np.nan, np.float32)
        for c in range(cols):
            order = np.lexsort((np.arange(rows), d[:, c]))[:k]            # value ascending, ties -> lower row
            idx[:len(order), c] = order
            val[:len(order), c] = d[order, c]
        return torch.from_numpy(idx), torch.from_numpy(val)

    @staticmethod
    def merge_topk(parts_idx, parts_val, k):
        pi, pv = parts_idx.numpy().reshape(-1, parts_idx.shape[-1]), parts_val.numpy().reshape(-1, parts_val.shape[-1])
        cols = pi.shape[1]
        idx = np.full((k, cols), -1, np.int64)
        val = np.full((k, cols), np.nan, np.float32)
        for c in range(cols):
            live = pi[:, c] >= 0
            order = np.lexsort((pi[live, c], pv[live, c]))[:k]
            idx[:len(order), c] = pi[live, c][order]
            val[:len(order), c] = pv[live, c][order]
        return torch.from_numpy(idx), torch.from_numpy(val)


def _sim_worker(rank, world, port, b1, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from d3fields_amd import sharding
        from oracle import c_oracle as O
        g = torch.Generator().manual_seed(11)
        src = torch.randn(b1, 24, generator=g)
        tgt = torch.cat([src[:1] + 0.01, torch.randn(9, 24, generator=g)])          # column 0 matches row 0
        lo, hi = sharding.shard_bounds(b1, rank, world)
        res = {}
        for dt in ("l2", "square"):
            sim, am = sharding.sharded_similarity_multi(src[lo:hi], tgt, 2.0, dt, kernels=_NumpySoftmaxKernels)
            ref, ref_am = O.pairwise(src.numpy(), tgt.numpy(), 2.0, dt, mode="softmax", return_argmax=True)
            err = float(np.abs(sim.numpy() - ref[lo:hi]).max()) if hi > lo else 0.0
            # the k-NN form: every rank ends with the same global [k,B2] lists = numpy's lexsort over ALL rows
            k = 3
            sim_k, gidx, gval = sharding.sharded_knn_descriptors(src[lo:hi], tgt, k, 2.0, dt, kernels=_NumpySoftmaxKernels, topk=_NumpyTopkKernels)
            dall = O.pairwise(src.numpy(), tgt.numpy(), 2.0, dt, mode="dist")
            want = np.full((k, tgt.shape[0]), -1, np.int64)
            for c in range(tgt.shape[0]):
                order = np.lexsort((np.arange(b1), dall[:, c]))[:k]
                want[:len(order), c] = order
            knn_ok = bool(np.array_equal(gidx.numpy(), want)) and bool(np.array_equal(gidx.numpy()[0], ref_am)) and \
                float(np.abs(sim_k.numpy() - ref[lo:hi]).max() if hi > lo else 0.0) <= 1e-6
            live = want >= 0
            knn_ok = knn_ok and bool(np.allclose(gval.numpy()[live], np.take_along_axis(dall, np.maximum(want, 0), 0)[live], rtol=0, atol=0))
            res[dt] = (tuple(sim.shape), err, bool(np.array_equal(am.numpy(), ref_am)) and knn_ok)
        q.put((rank, res, (lo, hi)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("b1,world", [(257, 2), (7, 2), (1, 2), (1003, 8), (3, 8)])
def test_sharded_similarity_world2(b1, world):
    """softmax(dim=0) of compute_similarity_tensor_multi with the rows split over the ranks: one 16-B record per
    column is exchanged; rows and the global argmax equal the single-process result (rank 1 is empty for b1=1; at
    world 8 with b1=3 five ranks are)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sim_worker, args=(r, world, port, b1, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, r, (lo, hi) in res:
        for dt, (shape, err, am_ok) in r.items():
            assert shape == (hi - lo, 10)
            assert err <= 1e-6, "rank %d %s: similarity rows differ by %g" % (rank, dt, err)
            assert am_ok, "rank %d %s: global argmax differs" % (rank, dt)
