#!/usr/bin/env python
"""bench.py -- fused 3-D query points / second of the d3fields field query on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE pass of the hot path (Fusion.batch_eval == one fused HIP launch) over one
batch of synthetic query points, with every input already resident in HBM.  Rank 0 prints ONE
JSON line (contract in the task statement) carrying `roofline` and, at N=1, `cpu_baseline`.

Workloads (BASELINE.json configs; no datasets/checkpoints offline -> random maps of the same
shape):
  c2_dense  4 views x 480x640 depth, 480x640x384 fp32 feature maps (what "4x(480x640)x384-d"
            in the north star implies), 985 600-point voxel grid (step 5 mm)       [default]
  c2_patch  same, reference-faithful patch-resolution 48x64x384 features (fusion.py:694-697)
  c3_dense / c3_patch  + 480x640x8 one-hot instance mask, 1 925 000-point grid (step 4 mm)
  c4_patch  8 views x 720x1280, 72x128x1024 features, 1 000 000 points per GPU
  c5_track  one tracking frame as SURVEY 8d defines it: NEW depth / feature / mask tensors (the per-frame refresh of
            vis_tracking.py:86: the shim's device-side finite checks run again), eval of 100 000 keypoints (features +
            mask), descriptor correspondence (softmax similarity + argmax); the static-map figure is reported beside it
  ref_patch the reference's own maps and names (vis_repr.py:103): 48x64x1024 DINOv2 patch maps (fusion.py:600,694-697) +
            8-instance mask + colours, return_names=['dino_feats','mask','color_tensor'].  Default points: the full 1 925 000-point
            LATTICE (the shape of vis_repr.py:93's query with the names of :103); `--points surface` = what :97-103 really
            hands to batch_eval: the lattice points with valid_mask & |dist| < step in flat-index order (the vertices
            extract_mesh finds, fusion.py:1313-1330; ~71 k points here), a CLOUD
  dist_only the distance-only pass over the 1-mm grid (return_names=[], vis_repr.py:93 / fusion.py:1420-1428): 123.2 M points on
            fused_eval_dist_kernel (VALU issue of the arithmetic the contract fixes; depth lookups in a tiled copy)
The default workload is c2_dense for EVERY --gpus N (weak scaling: the same per-GPU work at every N, so that the values of
`bench.py --gpus 1/2/4/8` form one curve); `--workload c4_patch --gpus N` is BASELINE.json's eight-GPU configuration, whose N = 1
point is `--gpus 1 --workload c4_patch`.  Every N > 1 line also carries rank 0's single-rank figure of the SAME workload
(`single_rank_points_per_s_same_workload`, measured alone before the group run) and `value_full_field` (the whole field
reassembled on every GPU) beside `value` (compute + the `dist` / `valid_mask` gather).
Multi-GPU: weak scaling -- every rank queries its own shard of N points against replicated
maps; the only exchange is the RCCL all-gather that reassembles the field (`--gather`).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOADS = {
    "c2_dense": dict(V=4, H=480, W=640, C=384, fhw=(480, 640), NI=0, step=0.005, N=985600),
    "c2_patch": dict(V=4, H=480, W=640, C=384, fhw=(48, 64), NI=0, step=0.005, N=985600),
    "c3_dense": dict(V=4, H=480, W=640, C=384, fhw=(480, 640), NI=8, step=0.004, N=1925000),
    "c3_patch": dict(V=4, H=480, W=640, C=384, fhw=(48, 64), NI=8, step=0.004, N=1925000),
    # config 4: an 8 M-point lattice (2.5 mm: 320 x 280 x 88) cut into eight x-slabs of 40 planes, one per GPU (985 600 points
    # each); --points random: a uniform cloud of 1 000 000 points per GPU instead (rounds 1-2 measured only that)
    "c4_patch": dict(V=8, H=720, W=1280, C=1024, fhw=(72, 128), NI=0, step=0.0025, slabs=8, N=985600, N_cloud=1000000),
    # config 2 with the feature maps STORED in fp16 (D3F_DTYPE_F16: widened on load, fp32 arithmetic) -- an extension, not
    # the headline: the reference's own float16 mode computes everything in half
    "c2_dense_f16": dict(V=4, H=480, W=640, C=384, fhw=(480, 640), NI=0, step=0.005, N=985600, f16=True),
    "c2_patch_f16": dict(V=4, H=480, W=640, C=384, fhw=(48, 64), NI=0, step=0.005, N=985600, f16=True),
    # dense variant of config 4: 30.2 GB of feature maps per GPU (3.77 GB per view, just inside the 32-bit texel offsets)
    "c4_dense": dict(V=8, H=720, W=1280, C=1024, fhw=(720, 1280), NI=0, step=0.0025, slabs=8, N=985600, N_cloud=1000000),
    # BASELINE config 5: one tracking frame = Fusion.eval of 100 k keypoints (features + instance mask) followed by the
    # descriptor correspondence of utils/corr_utils.py against 300 reference descriptors (+ fused argmax)
    "c5_track": dict(V=4, H=480, W=640, C=384, fhw=(48, 64), NI=8, step=None, N=100000, corr_refs=300, refresh=True),
    # the reference's own shape: DINOv2 ViT-L patch maps are 1024-d at (H/10, W/10) (fusion.py:600,694-697) and vis_repr.py:103
    # queries features + instance mask + colours in one call
    "ref_patch": dict(V=4, H=480, W=640, C=1024, fhw=(48, 64), NI=8, color=3, step=0.004, N=1925000),
    # the distance-only pass of the meshing / keypoint-selection callers on the 1-mm grid (800 x 700 x 220 points)
    "dist_only": dict(V=4, H=480, W=640, C=0, fhw=(1, 1), NI=0, step=0.001, N=123200000, no_maps=True),
}
# 256 CUs x 4 SIMD-32s: a plain wave64 VALU instruction issues over 2 cycles on CDNA4 (packed-fp32 ones over 4), at the 2.4 GHz
# maximum clock (MI355X_MICROARCH.md, cycle constants) -- an UPPER bound: kernels run at 2.0-2.3 GHz under load (profiles/r6_*)
VALU_PEAK_INST_PER_S = 1024 * 2.4e9 / 2


def algorithmic_bytes(w, n):
    """read every point once, write every output once, read every map once."""
    sumC = w["C"] + w["NI"] + w.get("color", 0)
    per_pt = 12 + 4 + 1 + 4 * sumC
    es = 2 if w.get("f16") else 4                                 # stored bytes per feature channel
    maps = w["V"] * (4 * w["H"] * w["W"] + es * w["fhw"][0] * w["fhw"][1] * w["C"] + 4 * w["H"] * w["W"] * (w["NI"] + w.get("color", 0)))
    return n * per_pt + maps + 84 * w["V"], per_pt


def measured_traffic(workload, points, n, path=None):
    """HBM bytes per launch (and VALU wave instructions, when counted) from the committed rocprofv3 PMC passes
    (profiles/traffic.json), keyed by workload AND point set (`<workload>` = its grid, `<workload>_random` / `_surface` = the
    clouds), only when the profiled launch had the same number of points AND the entry carries the source fingerprint of the
    library that is loaded now; else None -- a figure of another kernel / order is worse than none."""
    try:
        with open(path or os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            e = json.load(fh).get(workload if points == "grid" else workload + "_" + points)
        if not e or int(e.get("points", n)) != int(n):
            return None, None, None
        # the counters belong to the kernels they were collected on: an entry stamped with another source fingerprint than the
        # loaded library's (d3fields_amd/build.py: sha256 over csrc/ + the header + the flags) is stale and is dropped
        from d3fields_amd import build as _build
        if e.get("source_fingerprint") != _build.source_fingerprint():
            return None, None, None
        return e.get("traffic_bytes"), e.get("source"), e.get("valu_insts")
    except (OSError, ValueError, KeyError):
        return None, None, None


def traffic_in_this_run(argv_workload, kernel_hint, timeout_s=150):
    """HBM-side bytes per launch of the dominant fused kernel, measured NOW: this script re-runs itself for a few steps under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, kernel-trace only, as the guide prescribes:
    MI355X_MICROARCH.md, HBM section) with the library that is loaded, on this box, on the same workload and point set.
    traffic = FETCH_SIZE[KB] * 1024 * 2 (gfx950 tallies wide coalesced reads at half size) + WRITE_SIZE[KB] * 1024, averaged over the
    dispatches of the fused kernel with the longest total time.  Returns (bytes, note) or (None, why not)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="d3f_traffic_")
    got = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", out, "-o", "pmc", "--output-format", "csv", "--", sys.executable,
                   os.path.abspath(__file__)] + argv_workload + ["--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-verify", "--traffic", "off"]
            env = dict(os.environ, TMPDIR=tmp)
            try:
                subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=False)
            except subprocess.TimeoutExpired:
                return None, "rocprofv3 pass timed out"
            per = {}
            for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as fh:
                    for r in csv.DictReader(fh):
                        if "fused_eval" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                            per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
            if not per:
                return None, "no %s rows for a fused kernel" % counter
            got[counter] = per
        # the dominant fused kernel: the one the plan names if it is there, else the one with the most counted bytes
        def pick(per):
            for k in per:
                if kernel_hint and kernel_hint.split("<")[0] in k and (kernel_hint.split("<")[-1].split(">")[0].replace(" ", "") in k.replace(" ", "")):
                    return k
            return max(per, key=lambda k: sum(per[k]))
        kf, kw = pick(got["FETCH_SIZE"]), pick(got["WRITE_SIZE"])
        f_kb = sum(got["FETCH_SIZE"][kf]) / len(got["FETCH_SIZE"][kf])
        w_kb = sum(got["WRITE_SIZE"][kw]) / len(got["WRITE_SIZE"][kw])
        return int(f_kb * 1024 * 2 + w_kb * 1024), "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on this box (%d + %d dispatches of %s)" % (
            len(got["FETCH_SIZE"][kf]), len(got["WRITE_SIZE"][kw]), kf.split("(")[0][-70:])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def build_workload(name, dev, rank, world, points="grid"):
    from d3fields_amd import Fusion, create_init_grid, synth
    w = WORKLOADS[name]
    V, H, W = w["V"], w["H"], w["W"]
    sc = synth.make_scene(V, H, W, "smooth")
    f = Fusion(num_cam=V, device=str(dev))
    f.curr_obs_torch = {k: sc[k].to(dev) for k in ("depth", "K", "pose")}
    names = []
    if not w.get("no_maps"):
        f.curr_obs_torch["dino_feats"] = synth.random_map(V, w["fhw"][0], w["fhw"][1], w["C"], seed=1, device=dev)
        if w.get("f16"):
            f.curr_obs_torch["dino_feats"] = f.curr_obs_torch["dino_feats"].half()
        names.append("dino_feats")
    if w["NI"]:
        f.curr_obs_torch["mask"] = synth.random_onehot_mask(V, H, W, w["NI"], seed=2, device=dev)
        names.append("mask")
    if w.get("color"):
        f.curr_obs_torch["color_tensor"] = torch.rand(V, H, W, w["color"], generator=torch.Generator().manual_seed(4)).to(dev)
        names.append("color_tensor")
    f.H, f.W = H, W
    if w["step"] is not None and points == "grid" and w.get("slabs"):
        # one x-slab of the job's lattice per rank (north star: "8M points sharded across 8 GPUs"); fewer ranks than slabs
        # take slabs spread over the box (one GPU: a middle slab, the representative one)
        S = w["slabs"]
        slab = (S // 2 - 1) if world == 1 else (rank * S // world) % S
        box = dict(synth.WORK_BOX)
        width = (box["x_upper"] - box["x_lower"]) / S
        box["x_lower"] = synth.WORK_BOX["x_lower"] + slab * width
        box["x_upper"] = box["x_lower"] + width - w["step"] / 4          # arange(lower, upper, step): exactly width/step planes
        pts, shape = create_init_grid(box, w["step"])
        assert pts.shape[0] == w["N"], (tuple(shape), pts.shape[0], w["N"])
    elif w["step"] is not None and points == "grid":
        # weak scaling: the job's grid is `world` times finer along x (step/world); rank r owns the
        # x-planes congruent to r, i.e. the same box shifted by r*step/world -> N points per rank
        pts, _ = create_init_grid(synth.WORK_BOX, w["step"])
        if world > 1:
            pts[:, 0] += rank * w["step"] / world
    elif points == "surface":
        # what extract_mesh hands to batch_eval (fusion.py:1313-1330, vis_repr.py:97-103): the lattice points on the surface --
        # valid_mask & |dist| < step -- in flat-index order (Fusion.grid_shell: the same pass, compacted on the device)
        if w["step"] is None or w.get("no_maps") or world > 1:
            raise SystemExit("--points surface needs a workload with a lattice and channel maps, on one GPU")
        box = dict(synth.WORK_BOX)
        if w.get("slabs"):
            width = (box["x_upper"] - box["x_lower"]) / w["slabs"]
            box["x_lower"] = synth.WORK_BOX["x_lower"] + (w["slabs"] // 2 - 1) * width
            box["x_upper"] = box["x_lower"] + width - w["step"] / 4
        _, pts = f.grid_shell(box, w["step"], dist_threshold=w["step"])
        pts = pts.contiguous()
    else:       # uniformly random cloud of the same N in the same box: no locality in the caller's order
        pts = synth.random_cloud(w.get("N_cloud", w["N"]), seed=3 + rank)
    return f, pts.to(dev), names, w, sc


def time_steps(fn, steps, dist_on, dev, drain=None):
    """Barrier + synchronize on both sides of exactly `steps` steps; returns wall seconds (max over ranks).
    `drain` completes collectives still in flight from the last step (inside the timed region)."""
    import torch.distributed as dist
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    if drain is not None:
        drain()
    torch.cuda.synchronize(dev)
    if dist_on:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def verify_against_oracle(f, pts, names, w, sc, out, sample=2000):
    """After the timed region: a `sample`-point sample of the output of the timed launch configuration against the CPU
    oracle (oracle/d3f_oracle.c): valid_mask and dist bit-exact, fused channels <= 1e-5 relative (tests/ tolerance)."""
    import numpy as np
    from oracle import c_oracle
    n = pts.shape[0]
    pick = torch.randperm(n, generator=torch.Generator().manual_seed(17))[:sample].to(pts.device)
    ref = c_oracle.eval_field(sc["depth"], sc["K"], sc["pose"], pts[pick].cpu(), [f.curr_obs_torch[k].float().cpu() for k in names],
                              mu=f.mu)
    ok = bool(np.array_equal(out["valid_mask"][pick].cpu().numpy(), ref["valid_mask"]))
    ok = ok and bool(np.array_equal(out["dist"][pick].cpu().numpy(), ref["dist"]))
    worst = 0.0
    for i, k in enumerate(names):
        r = ref["sets"][i]
        err = float(np.abs(out[k][pick].cpu().numpy().astype(np.float64) - r).max() / max(float(np.abs(r).max()), 1.0))
        worst = max(worst, err)
    return ok and worst <= 1e-5, {"sample_points": int(pick.numel()), "max_rel_err_fused": worst,
                                  "dist_and_valid_mask_bit_exact": ok, "tolerance": 1e-5}


def kernel_time_ms(fn, steps, dev):
    """Average device time of one step measured with HIP events recorded on the stream the
    kernel is launched on (torch's current stream: the shim passes exactly that stream)."""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize(dev)
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return sum(ts) / len(ts), ts[len(ts) // 2], ts[0]


def fused_kernel_time_ms(fn, steps, dev):
    """Device time of the dominant kernel alone (fused_eval_kernel), from HIP events the library records
    on the launch stream right around that kernel (d3f_profile_next_eval) -- the figure that must agree
    with rocprofv3's per-kernel average.  Point-ordering kernels of the step are outside this bracket."""
    from d3fields_amd import _lib
    lib = _lib.load()
    pairs = []
    for _ in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); b.record()                      # force creation of the underlying hipEvent_t
        lib.d3f_profile_next_eval(ctypes.c_void_p(a.cuda_event), ctypes.c_void_p(b.cuda_event))
        fn()
        pairs.append((a, b))
    torch.cuda.synchronize(dev)
    ts = sorted(a.elapsed_time(b) for a, b in pairs)
    return sum(ts) / len(ts), ts[len(ts) // 2], ts[0]


def cpu_baseline(sc, w, names, maps_cpu, pts_cpu, budget_pts, threads=0):
    """The torch-ops CPU port of Fusion.batch_eval (oracle/torch_port.py) on the host cores,
    on a bounded sample of the same workload.  Reported beside the GPU number, never a target."""
    from oracle import torch_port
    cores = threads or min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    obs = {k: sc[k] for k in ("depth", "K", "pose")}
    obs.update(maps_cpu)
    n = min(budget_pts, pts_cpu.shape[0])
    idx = torch.linspace(0, pts_cpu.shape[0] - 1, n, dtype=torch.float64).long().clamp_(max=pts_cpu.shape[0] - 1)
    sample = pts_cpu[idx].contiguous()
    times = []
    with torch.no_grad():
        torch_port.batched_field_query(obs, sample[:60000], names, w["H"], w["W"])       # warm-up
        t_all = time.perf_counter()
        while len(times) < 5 and (len(times) < 2 or time.perf_counter() - t_all < 12.0):
            t0 = time.perf_counter()
            torch_port.batched_field_query(obs, sample, names, w["H"], w["W"])
            times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    res = {"value": n / dt, "unit": "points/s", "cores": cores, "kind": "port",
           "sample": "%d points of the same workload and maps through the torch-ops port of batch_eval (60000-pt chunks, "
                     "%d threads; 256 threads measured 27x slower), median of %d runs, %.1f s of CPU time in total"
                     % (n, cores, len(times), sum(times))}
    try:        # second CPU reference point: the scalar C restatement, OpenMP over points
        from oracle import c_oracle
        np_maps = [maps_cpu[k].numpy() for k in names]
        args = (sc["depth"].numpy(), sc["K"].numpy(), sc["pose"].numpy(), sample.numpy(), np_maps)
        c_oracle.eval_field(*args[:3], args[3][:20000], np_maps)
        ct = []
        while len(ct) < 3 and (not ct or sum(ct) < 8.0):
            t0 = time.perf_counter()
            c_oracle.eval_field(*args)
            ct.append(time.perf_counter() - t0)
        res["c_port"] = {"value": n / sorted(ct)[len(ct) // 2], "unit": "points/s", "cores": c_oracle.threads(),
                         "kind": "port", "sample": "same %d points through oracle/d3f_oracle.c (scalar C, OpenMP over "
                         "points), median of %d runs" % (n, len(ct))}
    except Exception as e:                       # the headline baseline above stands on its own
        res["c_port"] = {"error": repr(e)}
    return res


def spawn_command(n, argv, port=None):
    """The launch line of an N-rank run on this node (what the driver itself uses for N > 1)."""
    if port is None:
        import socket
        with socket.socket() as s:                       # a free port: several benches may share a host
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


IPC_FAILURE_EXIT = 77        # a rank's first collective failed: the launcher may retry with HSA_ENABLE_IPC_MODE_LEGACY=0


def spawn_ranks(n, argv):
    """Runs this script as n ranks; returns the launcher's exit status (non-zero if any rank failed).

    The environment is handed on UNCHANGED first.  Only if the ranks report that their first collective failed (exit status
    IPC_FAILURE_EXIT: on hosts whose driver supports dmabuf IPC only, RCCL's hipIpcGetMemHandle fails with 'invalid
    argument' unless HSA_ENABLE_IPC_MODE_LEGACY=0) and the variable is not set already, the launch is repeated once with it
    set; the line then says so (config.ipc_mode_retry)."""
    import subprocess
    import tempfile
    env = dict(os.environ)
    # torch.distributed.run maps every child failure to its own status, so the ranks say WHICH failure it was through a marker
    # file: only "the first collective failed" is retried; a verification mismatch, an OOM or a crash is handed back as it is
    marker = os.path.join(tempfile.mkdtemp(prefix="d3f_bench_"), "ipc_failure")
    env["D3F_BENCH_IPC_MARKER"] = marker
    rc = subprocess.call(spawn_command(n, argv), env=env)
    if rc != 0 and os.path.exists(marker) and "HSA_ENABLE_IPC_MODE_LEGACY" not in env and n > 1:
        os.remove(marker)
        env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        env["D3F_BENCH_IPC_RETRY"] = "1"
        print("bench.py: the first collective of the %d-rank launch failed (status %d); retrying once with HSA_ENABLE_IPC_MODE_LEGACY=0" % (n, rc), file=sys.stderr)
        rc = subprocess.call(spawn_command(n, argv), env=env)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: c2_dense, the configuration the metric is quoted on, for every --gpus N (one curve); c4_patch is the "
                         "configuration BASELINE.json names for eight GPUs (8 views x 720x1280, 1024-d, 8 M points sharded)")
    ap.add_argument("--gather", default="dist", choices=["none", "dist", "full"],
                    help="N>1: what the RCCL all-gather reassembles inside the timed step")
    ap.add_argument("--points", default="grid", choices=["grid", "random", "surface"],
                    help="grid: create_init_grid of the workload (default); random: uniform cloud of the same N in the "
                         "same box (exposes the dependence on the caller's point order); surface: the lattice points with "
                         "valid_mask & |dist| < step in flat-index order -- the mesh-vertex cloud the reference's feature query runs on "
                         "(vis_repr.py:97-103)")
    ap.add_argument("--no-overlap", action="store_true", help="N>1: block on every all-gather instead of overlapping it "
                    "with the next batch's query")
    ap.add_argument("--refresh-maps", action="store_true", help="install NEW depth / map tensors before every step (the per-frame "
                    "update() of a tracking loop: the shim's device-side finite checks, d3f_map_check, run inside the step); "
                    "always on for c5_track")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traffic", default="auto", choices=["auto", "off", "measure"],
                    help="roofline.traffic: auto = measured in this run (two short rocprofv3 PMC passes of this same command, one GPU, ~40 s) "
                         "when rocprofv3 is there, else the committed copy of profiles/traffic.json; off = the committed copy only")
    ap.add_argument("--no-verify", action="store_true", help="skip the post-run oracle check of a 2000-point sample")
    ap.add_argument("--tuning", type=lambda x: int(x, 0), default=0, help="D3F_TUNE_* bits (experiments)")
    ap.add_argument("--cpu-sample", type=int, default=1000000)
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = min(host cores, 64)")
    ap.add_argument("--force-dist", action="store_true", help="run the multi-rank code path even with one rank (self-spawned under "
                                                               "torch.distributed.run): RCCL initialisation, barrier, all-gathers")
    ap.add_argument("--points-per-gpu", type=int, default=0, help="use only the first K points of the rank's point set (0 = the workload's "
                    "own size): the reduced-N dry run of a multi-rank launch (tests/test_gpu_sharding.py); not a benchmark configuration")
    ap.add_argument("--ragged", action="store_true", help="with --points-per-gpu K: rank r takes K - r points (unequal shards: the grouped "
                    "point-to-point form of the all-gather)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; 'gloo' lets the N>1 code path be "
                    "exercised with several ranks on ONE GPU (testing only)")
    args = ap.parse_args()
    if args.workload is None:
        args.workload = "c2_dense"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # --force-dist: the multi-rank code path (process group on RCCL, barrier, max over ranks, overlapped all-gathers) with
    # however many ranks there are, ONE included -- the plumbing self-test a single-GPU box can run
    dist_on = world > 1 or (args.force_dist and "WORLD_SIZE" in os.environ)
    if args.gpus != world and dist_on:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if (args.gpus > 1 or args.force_dist) and not dist_on:
        # `python bench.py --gpus N` is ONE command: it re-launches itself as N ranks (one per GPU) under
        # torch.distributed.run and hands back their exit status; rank 0's JSON line goes to this process's stdout
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    if dist_on and args.backend == "nccl" and world > torch.cuda.device_count():
        # RCCL needs one GPU per rank: wrapping local ranks onto fewer devices would hang or fail inside the first collective
        # (and would not be the run the line claims).  Several ranks on one GPU exist for testing only: --backend gloo.
        raise SystemExit("bench.py: --gpus %d with backend nccl (RCCL) needs %d visible GPUs, this node shows %d (rank %d); "
                         "use --backend gloo to exercise the multi-rank path on fewer devices" % (world, world, torch.cuda.device_count(), rank))
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(args.backend)
        try:                                                 # the first collective: where a wrong IPC mode shows (see spawn_ranks)
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe)
            torch.cuda.synchronize(dev)
            assert int(probe.item()) == world
        except Exception as exc:
            print("bench.py rank %d: first collective failed: %r" % (rank, exc), file=sys.stderr)
            if os.environ.get("D3F_BENCH_IPC_MARKER"):
                open(os.environ["D3F_BENCH_IPC_MARKER"], "w").close()
            os._exit(IPC_FAILURE_EXIT)

    f, pts, names, w, sc = build_workload(args.workload, dev, rank, world, args.points)
    f.tuning_flags = args.tuning
    # The shim can keep what it learned about an unchanged query tensor between calls (its lattice dims, or its Hilbert
    # order): every timed step here starts from scratch instead -- the probe kernels (lattice + locality, ~10 us) or the
    # Hilbert sort are enqueued INSIDE every step; with async_probes the launch follows the verdict of the last FINISHED
    # probes of a query of the same size, so a steady-state step holds NO host sync (only the first query of a size
    # waits once) -- and the cached figure is reported separately below.
    f.cache_point_order = False
    if args.points_per_gpu > 0:
        pts = pts[:max(1, args.points_per_gpu - (rank if args.ragged else 0))].contiguous()
    n = pts.shape[0]
    from d3fields_amd import sharding
    counts = [n] * world                    # points per rank (equal unless --ragged)
    if dist_on:
        import torch.distributed as dist
        dist.all_gather_object(counts, n)
        counts = [int(c) for c in counts]

    corr_src = None
    if w.get("corr_refs"):
        from d3fields_amd import corr_utils
        corr_src = torch.randn(w["corr_refs"], w["C"], generator=torch.Generator().manual_seed(11)).to(dev)

    # every frame REFRESHES the maps (vis_tracking.py:86 calls update() per frame).  The producers are out
    # of scope, so two resident copies of (depth, features, mask) alternate: every step installs tensor OBJECTS the shim has
    # not seen in the previous step, i.e. its per-tensor state (the device-side finite words of d3f_map_check) is rebuilt
    # inside the step exactly as after an update() -- on the stream, without a host sync.
    frames, frame_no = [None], [0]
    if w.get("refresh") or args.refresh_maps:
        frames[0] = [{k: f.curr_obs_torch[k] for k in ["depth"] + names}, {k: f.curr_obs_torch[k].clone() for k in ["depth"] + names}]

    def compute():
        if frames[0] is not None:
            frame_no[0] += 1
            f.curr_obs_torch.update(frames[0][frame_no[0] & 1])
        out = f.batch_eval(pts, return_names=names)
        if corr_src is not None:        # keypoint descriptors vs reference descriptors: softmax similarity + best match
            if dist_on:                 # softmax(dim=0) runs over ALL ranks' keypoints: one 16-B record per reference exchanged
                out["similarity"], out["match"] = sharding.sharded_similarity_multi(out["dino_feats"], corr_src, 1.0,
                                                                                    row_offset=sum(counts[:rank]))
            else:
                out["similarity"], out["match"] = corr_utils.nearest_descriptor(out["dino_feats"], corr_src, 1.0)
        return out

    # N>1: the all-gathers of batches k-1 and k run on RCCL's stream while batch k+1 is queried (two gathers in
    # flight); `drain` waits for what is left inside the timed region.  --no-overlap blocks on every gather instead.
    from collections import deque
    pending, overlap_error = deque(), []
    gather_keys = {}

    def drain(keep=0):
        while len(pending) > keep:
            works, _hold = pending.popleft()
            for wk in works:
                wk.wait()

    def step():
        out = compute()
        if dist_on and args.gather != "none":
            keys = ("dist", "valid_mask") if args.gather == "dist" else tuple(k for k in out if k != "match")
            gather_keys["keys"] = keys
            gather_keys["bytes"] = sharding.gather_bytes(out, keys, counts)
            if args.no_overlap:
                sharding.all_gather_field({k: out[k] for k in keys}, keys=keys, counts=counts)
            else:
                drain(keep=1)                              # gather k-2 must be done before gather k is enqueued
                try:
                    full, works = sharding.all_gather_field({k: out[k] for k in keys}, keys=keys, counts=counts,
                                                            async_op=True)
                except Exception as exc:                   # a backend without async all-gather: block instead (reported)
                    args.no_overlap = True
                    overlap_error.append(repr(exc))
                    sharding.all_gather_field({k: out[k] for k in keys}, keys=keys, counts=counts)
                    return out
                pending.append((works, (out, full)))       # inputs and outputs stay alive until the wait
        return out

    with torch.no_grad():
        # Device-side timings first (HIP events; they also bring clocks, allocator and caches to steady state), then the
        # contract's W untimed warm-up steps immediately followed by the K timed steps.
        compute()
        s_avg, s_med, s_min = kernel_time_ms(compute, max(args.steps, 5), dev)          # whole step on the device
        k_avg, k_med, k_min = fused_kernel_time_ms(compute, max(args.steps, 5), dev)   # dominant kernel only
        extra = {}
        if dist_on:
            # rank 0 alone (the others wait at the barrier): the compute-only step of THIS workload on one GPU, the figure the group's
            # compute-only value divides by
            import torch.distributed as dist
            if rank == 0:
                extra["single_rank_points_per_s_same_workload"] = n * args.steps / time_steps(compute, args.steps, False, dev)
            dist.barrier()
        for _ in range(max(args.warmup, 1)):
            step()
        drain()
        wall = time_steps(step, args.steps, dist_on, dev, drain)
        if not dist_on:
            f.cache_point_order = True          # a static grid queried every frame: the order is built once
            compute(); compute()
            extra["points_per_s_with_cached_point_order"] = n * args.steps / time_steps(compute, args.steps, False, dev)
            extra["cached_point_order_note"] = ("the shim's default: lattice dims / point order of an UNCHANGED query tensor are kept "
                                                "between calls; `value` re-derives them inside every step")
            f.cache_point_order = False
            if frames[0] is not None:           # the same step on maps that do not change (what rounds 1-3 reported for c5)
                keep, frames[0] = frames[0], None
                compute(); compute()
                extra["points_per_s_with_static_maps"] = n * args.steps / time_steps(compute, args.steps, False, dev)
                frames[0] = keep
        if dist_on:
            extra["compute_only_points_per_s"] = sum(counts) * args.steps / time_steps(compute, args.steps, True, dev)
            if args.gather != "full":
                def full():
                    o = compute()
                    sharding.all_gather_field(o, keys=[k for k in o if k != "match"], counts=counts)
                full()
                fs = max(2, args.steps // 4)
                extra["full_field_gather_points_per_s"] = sum(counts) * fs / time_steps(full, fs, True, dev)
            else:
                extra["full_field_gather_points_per_s"] = sum(counts) * args.steps / wall
            # the north star's "reassemble the full field": every output of every point on every GPU
            extra["value_full_field"] = extra["full_field_gather_points_per_s"]
            if rank == 0 and extra.get("single_rank_points_per_s_same_workload"):
                single = extra["single_rank_points_per_s_same_workload"]
                extra["scaling_efficiency"] = {"compute_only": extra["compute_only_points_per_s"] / (world * single),
                                               "with_dist_gather": None, "with_full_field_gather": extra["value_full_field"] / (world * single),
                                               "note": "this run's group figures / (world x rank 0's single-rank compute-only figure of the same workload)"}

        verified, verify_info = None, None
        f.record_plans = True
        out_check = compute()                   # every rank: the c5 step holds a collective (sharded softmax)
        plan = f.last_plan()
        f.record_plans = False
        gate_info = None
        if plan and plan.get("gated_window"):         # a cloud on the gated pair of launches: which side ran?
            g = f.last_gate()
            if g is not None:
                gate_info = {"tiles_that_fit_of_%d" % 128: g[0], "window_side_ran": g[1]}
                if g[1]:
                    plan = dict(plan, **plan["window_side"])
        if rank == 0 and not args.no_verify:      # fp16-stored maps: the oracle runs on the WIDENED maps (the contract)
            verified, verify_info = verify_against_oracle(f, pts, names, w, sc, out_check)

    rank_devices = [torch.cuda.get_device_name(dev)]
    if dist_on:
        import torch.distributed as dist
        names_all = [None] * world
        dist.all_gather_object(names_all, "rank %d: cuda:%d %s" % (rank, dev.index, torch.cuda.get_device_name(dev)))
        rank_devices = names_all
    total_pts = sum(counts) * args.steps
    value = total_pts / wall
    bytes_alg, per_pt = algorithmic_bytes(w, n)
    traffic, traffic_src, valu_insts = measured_traffic(args.workload, args.points if w["step"] is not None else "grid", n)
    traffic_in_run, traffic_note = False, None
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if rank == 0 and world == 1 and args.traffic != "off" and not w.get("no_maps") and not under_profiler:
        # the timed binary, this box, this command: FETCH_SIZE / WRITE_SIZE of the fused kernel from two short profiler passes
        argv_w = ["--workload", args.workload, "--points", args.points] + (["--tuning", str(args.tuning)] if args.tuning else []) + \
                 (["--refresh-maps"] if args.refresh_maps else [])
        torch.cuda.synchronize(dev)
        got, note = traffic_in_this_run(argv_w, (plan or {}).get("kernel"))
        if got is not None:
            traffic, traffic_src, traffic_in_run = got, note, True
        elif args.traffic == "measure":
            raise SystemExit("bench.py --traffic measure: " + note)
        else:
            traffic_note = "not measured in this run (%s): the committed copy, if its fingerprint matches" % note
    elif args.traffic != "off" and under_profiler:
        traffic_note = "not measured in this run (the bench itself runs under a profiler): the committed copy, if its fingerprint matches"
    achieved = bytes_alg / (k_avg * 1e-3) / 1e9
    # SURVEY 8d secondary figure (reported, not graded): bytes the gather requests with zero inter-point reuse
    sumC = w["C"] + w["NI"] + w.get("color", 0)
    b_gather = 12 + w["V"] * (4 + 16 * sumC) + 5 + 4 * sumC
    res = {
        "metric": "fused 3D query-points/sec", "value": value, "unit": "points/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * wall / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": ("f32 (feature maps stored as f16)" if w.get("f16") else "f32"), "data": "synthetic",
        "config": {"workload": "%s: %d views x %dx%d depth, %s%s%s, %d query points per GPU, "
                               "return_names=%s" % (args.workload, w["V"], w["H"], w["W"],
                                                    ("no channel maps" if w.get("no_maps") else "%dx%dx%d fp32 features" % (w["fhw"][0], w["fhw"][1], w["C"])),
                                                    (" + %dx%dx%d one-hot mask" % (w["H"], w["W"], w["NI"])) if w["NI"] else "",
                                                    (" + %dx%dx%d colours" % (w["H"], w["W"], w["color"])) if w.get("color") else "",
                                                    n, names),
                   "maps_refreshed_every_step": bool(w.get("refresh") or args.refresh_maps),
                   "points": ("grid" if (w["step"] is not None and args.points == "grid") else
                              ("surface cloud: lattice points with valid_mask & |dist| < step, flat-index order (vis_repr.py:97-103)"
                               if args.points == "surface" else "random cloud")),
                   "points_per_gpu": n, "points_per_rank": counts, "views": w["V"], "feature_dim": w["C"], "feature_map": list(w["fhw"]),
                   "parallelism": "points sharded x%d, maps replicated" % world,
                   # what torch.distributed itself reports (backend "nccl" is RCCL on ROCm), and where every rank ran
                   "rccl_world_size": (dist.get_world_size() if dist_on else 1), "backend": (args.backend if dist_on else "n/a"),
                   "rank_devices": rank_devices,
                   "hsa_enable_ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                   "ipc_mode_retry": os.environ.get("D3F_BENCH_IPC_RETRY") == "1",
                   "gather": (args.gather if dist_on else "n/a"),
                   "gather_overlap": ((not args.no_overlap) if dist_on else "n/a"),
                   "gather_overlap_error": (overlap_error[0] if overlap_error else None),
                   # bytes every rank RECEIVES per step for the gathered keys, and the time they need at one xGMI link
                   # per peer (153 GB/s each, direct peer exchange: every peer's shard travels on its own link)
                   "gather_keys": (list(gather_keys.get("keys", ())) if dist_on else "n/a"),
                   "gather_bytes_received_per_rank": (gather_keys.get("bytes") if dist_on else "n/a"),
                   "gather_xgmi_floor_ms": ((gather_keys.get("bytes", 0) / max(world - 1, 1)) / 153e9 * 1e3 if dist_on else "n/a")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "traffic_note": traffic_note,
                     "traffic_measured_in_run": traffic_in_run,      # True: PMC passes of this command on this box (traffic_in_this_run); False: the committed copy
                     "kernel": (plan or {}).get("kernel", "fused_eval_kernel<0>"),
                     "kernel_ms_avg": k_avg, "kernel_ms_median": k_med,
                     "kernel_ms_min": k_min, "algorithmic_bytes_per_launch": bytes_alg,
                     "algorithmic_bytes_per_point": per_pt, "kernel_points_per_s": n / (k_avg * 1e-3),
                     "step_device_ms_avg": s_avg, "logical_gather_bytes_per_point": b_gather,
                     "logical_gather_GBps": n * b_gather / (k_avg * 1e-3) / 1e9,
                     "traffic_GBps": (traffic / (k_avg * 1e-3) / 1e9) if traffic else None,
                     # second roof, for the launches that are not memory-bound (dist_only): VALU wave instructions of the committed
                     # counter pass over this run's kernel time, against one PLAIN instruction per SIMD per 2 cycles at 2.4 GHz
                     "valu_issue": ({"insts_per_launch": valu_insts, "insts_per_point": valu_insts / n,
                                     "frac_of_issue_peak": valu_insts / (k_avg * 1e-3) / VALU_PEAK_INST_PER_S,
                                     "peak_inst_per_s": VALU_PEAK_INST_PER_S, "source": traffic_src} if valu_insts else None),
                     "note": "achieved = algorithmic bytes / kernel_ms_avg (HIP events around the "
                     "fused kernel on its launch stream); step_device_ms_avg also covers the per-step lattice probe / Hilbert ordering kernels"},
    }
    res.update(extra)
    res["verified"] = verified
    res["verify"] = verify_info
    if plan:
        res["config"]["point_order"] = plan["point_order"]
        res["config"]["tile_points"] = plan["tile_points"]
    if gate_info:
        res["config"]["device_gate"] = gate_info
        # (VERDICT r5 residual) the library's event pair of a gated cloud brackets the probe kernel and BOTH gated launches -- the side that
        # lost returns at once (~5 us) -- so kernel_ms_* is slightly conservative for the side `roofline.kernel` names
        res["roofline"]["kernel_time_spans"] = "window_gate_probe_kernel + the gated window launch + the gated cell-run launch (the losing side returns at once)"
    if dist_on and res.get("scaling_efficiency"):
        res["scaling_efficiency"]["with_dist_gather"] = value / (world * res["single_rank_points_per_s_same_workload"])
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        maps_cpu = {k: f.curr_obs_torch[k].float().cpu() for k in names}
        res["cpu_baseline"] = cpu_baseline(sc, w, names, maps_cpu, pts.cpu(), args.cpu_sample, args.cpu_threads)
    if rank == 0:
        print(json.dumps(res))
    if dist_on:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
