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
        q.put((rank, bool(ok), full["valid_mask"].dtype == torch.bool, tuple(full["dino_feats_inter"].shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [1001, 1000, 1])
def test_sharded_eval_world2(n):
    world = 2
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
