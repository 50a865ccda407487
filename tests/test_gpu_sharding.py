"""GPU, two ranks on ONE device over gloo: the sharded paths of d3fields_amd.sharding running the real HIP kernels
(RCCL itself needs one GPU per rank; the driver's 8-GPU run covers that).  Each rank evaluates its block with
Fusion.batch_eval / the row-sharded softmax steps; gathered results must equal the single-process ones."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from d3fields_amd import Fusion, sharding, synth, corr_utils
        dev = torch.device("cuda:0")
        V, H, W = 4, 120, 160
        sc = synth.make_scene(V, H, W, "stress")
        f = Fusion(num_cam=V, device="cuda:0")
        f.curr_obs_torch = {k: sc[k].to(dev) for k in ("depth", "K", "pose")}
        if rank == 0:       # maps exist on rank 0 only until the broadcast
            f.curr_obs_torch["dino_feats"] = synth.random_map(V, 12, 16, 96, seed=1, device=dev)
            f.curr_obs_torch["mask"] = synth.random_onehot_mask(V, H, W, 5, seed=2, device=dev)
        else:
            f.curr_obs_torch["dino_feats"] = torch.zeros(V, 12, 16, 96, device=dev)
            f.curr_obs_torch["mask"] = torch.zeros(V, H, W, 5, device=dev)
        f.H, f.W = H, W
        sharding.broadcast_observation(f, src=0)
        pts = synth.random_cloud(70001, seed=3).to(dev)          # ragged: 35001 + 35000
        names = ["dino_feats", "mask"]
        with torch.no_grad():
            full = sharding.sharded_eval(f, pts, names)
            part = sharding.sharded_eval(f, pts, names, gather_keys=("dist", "valid_mask"))
            single = f.batch_eval(pts, return_names=names)
        ok = all(torch.equal(full[k], single[k]) for k in single)
        lo, hi = part["local_range"]
        ok = ok and torch.equal(part["dino_feats_local"], single["dino_feats"][lo:hi]) and torch.equal(part["dist"], single["dist"])
        # descriptor rows sharded: softmax over ALL rows with one 16-B record per target column exchanged
        tgt = torch.randn(33, 96, generator=torch.Generator().manual_seed(7)).to(dev)
        tgt[5] = single["dino_feats"][60000]
        sim, am = sharding.sharded_similarity_multi(single["dino_feats"][lo:hi].contiguous(), tgt, 0.8, row_offset=lo)
        ref, ref_am = corr_utils.nearest_descriptor(single["dino_feats"], tgt, 0.8)
        err = float((sim - ref[lo:hi]).abs().max())
        ok_sim = err <= 1e-6 and torch.equal(am, ref_am) and int(am[5]) == 60000
        # ... and the k-NN form: the ranks' [k,B2] lists merged by (distance, global row) == the single-process lookup
        sim_k, gidx, gval = sharding.sharded_knn_descriptors(single["dino_feats"][lo:hi].contiguous(), tgt, 5, 0.8, row_offset=lo)
        _, ref_idx, _ = corr_utils.knn_descriptors(single["dino_feats"], tgt, 5, 0.8)
        ok_sim = ok_sim and torch.equal(gidx, ref_idx) and torch.equal(gidx[0], ref_am) and float((sim_k - ref[lo:hi]).abs().max()) <= 1e-6
        q.put((rank, bool(ok), bool(ok_sim), err))
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_hip_kernels():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, ok, ok_sim, err in res:
        assert ok, "rank %d: sharded field differs from the single-process field" % rank
        assert ok_sim, "rank %d: row-sharded softmax differs (max err %g)" % (rank, err)


def test_bench_eight_ranks_dry_run_and_rccl_device_check():
    """The first real multi-GPU run must not fail for boring reasons: `bench.py --gpus 8 --workload c4_patch
    --gather full` as EIGHT gloo ranks on this one GPU at reduced N, ragged shards -- the launch line, the process group, the
    overlapped full-field gather, the ragged counts and the JSON line are exactly those of the RCCL run; with backend nccl the same
    command must refuse loudly on a node that shows fewer GPUs than ranks instead of wrapping local ranks onto one device."""
    import json
    import subprocess
    K, world = 8192, 8
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--backend", "gloo", "--workload", "c4_patch", "--points",
           "random", "--gather", "full", "--points-per-gpu", str(K), "--ragged", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    counts = [K - k for k in range(world)]
    cfg = line["config"]
    assert line["n_gpus"] == world and cfg["rccl_world_size"] == world and cfg["backend"] == "gloo"
    assert cfg["points_per_rank"] == counts and len(cfg["rank_devices"]) == world
    per_point = 4 + 1 + 4 * 1024                                   # dist + valid_mask + the 1024-d row: every output gathered
    assert set(cfg["gather_keys"]) == {"dist", "valid_mask", "dino_feats"}
    assert cfg["gather_bytes_received_per_rank"] == (sum(counts) - counts[0]) * per_point        # rank 0 receives the other seven shards
    assert line["verified"] is True and line["value"] > 0 and line["value_full_field"] > 0
    assert abs(line["value"] - sum(counts) * line["steps"] / (line["ms_per_step"] * 1e-3 * line["steps"])) <= 1e-6 * line["value"]
    # RCCL: one GPU per rank or nothing
    if torch.cuda.device_count() < 2:
        cmd2 = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-verify",
                "--workload", "c2_patch", "--points-per-gpu", "70000"]
        r2 = subprocess.run(cmd2, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert r2.returncode != 0 and "needs 2 visible GPUs" in (r2.stderr + r2.stdout), (r2.returncode, r2.stderr[-1500:])


def test_bench_line_measures_its_traffic_in_the_run():
    """`python bench.py` (one GPU): the JSON line of the contract with `roofline` and `cpu_baseline`, verified against the oracle, and
    `roofline.traffic` measured IN THE RUN -- two short rocprofv3 PMC passes of the same command on this box (FETCH_SIZE x 2 +
    WRITE_SIZE of the fused kernel).  On patch-resolution maps the traffic is close to the algorithmic bytes; `--traffic off` falls
    back to the committed copy (or null)."""
    import json
    import shutil
    import subprocess
    if shutil.which("rocprofv3") is None and not os.path.exists("/opt/rocm/bin/rocprofv3"):
        pytest.skip("rocprofv3 not installed")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c2_patch", "--steps", "5", "--warmup", "2", "--cpu-sample", "20000"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in line, k
    rf = line["roofline"]
    assert line["verified"] is True and line["n_gpus"] == 1 and line["dtype"] == "f32" and rf["bound"] == "hbm" and rf["peak"] == 8000.0
    if rf["traffic_measured_in_run"]:
        assert 0.9 * rf["algorithmic_bytes_per_launch"] <= rf["traffic"] <= 1.5 * rf["algorithmic_bytes_per_launch"]
    else:       # a harness that profiles this process, or a rocprofv3 that cannot run here: the line says why and falls back
        assert rf["traffic_note"], rf
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0.0 < rf["frac"] < 1.0
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1 and line["cpu_baseline"]["value"] > 0
    r2 = subprocess.run(cmd + ["--traffic", "off", "--no-cpu-baseline", "--no-verify"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r2.returncode == 0, r2.stderr[-2000:]
    line2 = json.loads([ln for ln in r2.stdout.splitlines() if ln.startswith("{")][-1])
    assert line2["roofline"]["traffic_measured_in_run"] is False
