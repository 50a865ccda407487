"""`Fusion` -- drop-in for the query surface of the reference's ``class Fusion`` (fusion.py:202).

Same method names, argument defaults, dict keys, shapes and dtypes as the reference for
``update / eval / eval_dist / batch_eval / text_queries_for_inst_mask[_no_track]``; the field
query itself runs as one HIP launch through libd3fields_hip.so (include/d3fields_hip.h)
instead of ~20 torch ops.  The 2-D producers (DINOv2, Grounded-SAM, XMem: reference
fusion.py:223-303) are upstream PyTorch-ROCm models and are *injected* as callables; this
module never downloads or builds them.

There is no CPU or torch-op fallback: query points must live on the ROCm device and the
shared library must load, otherwise an exception is raised.
"""
import ctypes

import numpy as np
import torch

from . import _lib

__all__ = ["Fusion", "create_init_grid", "instance2onehot", "onehot2instance", "fps", "_init_low_level_memory", "erode"]

# ----------------------------------------------------------------------------------------
# grid / mask-format helpers (reference fusion.py:79-116)
# ----------------------------------------------------------------------------------------
def _grid_axes(boundaries, step_size):
    """The three axis tensors of the reference grid (fusion.py:82-84): arange(lower, upper, step) + step/2, float32."""
    return [torch.arange(boundaries[ax + "_lower"], boundaries[ax + "_upper"], step_size, dtype=torch.float32) + step_size / 2
            for ax in "xyz"]


def fps(pcd, particle_num, init_idx=-1):
    """Farthest point sampling with the signature and results of the reference's fps_np
    (utils/my_utils.py:478-497): returns (pcd_fps [k,3], fps_idx list, max remaining distance).

    numpy in -> numpy out like the reference; a CUDA tensor in -> (tensor, index tensor, float).  One small launch per
    round over up to 256 workgroups, with numpy's float32 arithmetic and first-maximum tie rule, so for a given init_idx
    the selection is identical to fps_np's.  init_idx == -1 draws the start with np.random.randint, as the reference
    does.  particle_num may exceed the cloud size: fps_np then keeps appending index 0 (all distances are 0) and always
    returns particle_num points -- so does this.
    """
    as_numpy = isinstance(pcd, np.ndarray)
    if as_numpy:
        dev = torch.device("cuda", torch.cuda.current_device())
        pts = torch.from_numpy(np.ascontiguousarray(pcd, dtype=np.float32)).to(dev)
    else:
        if not pcd.is_cuda:
            raise RuntimeError("fps: torch input must be on the ROCm device (no CPU path)")
        dev = pcd.device
        pts = pcd.to(torch.float32).contiguous()
    n = pts.shape[0]
    assert n > 0 and pts.dim() == 2 and pts.shape[1] == 3
    start = int(np.random.randint(n)) if init_idx == -1 else int(init_idx)
    k = int(particle_num)
    idx = torch.empty(k, dtype=torch.int64, device=dev)
    maxd = torch.empty(1, dtype=torch.float32, device=dev)
    ws = torch.empty(_lib.load().d3f_fps_workspace_bytes(n), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().d3f_farthest_point_sampling(_lib.ptr(pts), n, k, start, _lib.ptr(idx), _lib.ptr(maxd),
                                                           _lib.ptr(ws), _lib.current_stream_handle(dev)))
    sel = pts[idx]
    if as_numpy:
        return sel.cpu().numpy(), idx.cpu().tolist(), float(maxd.item())
    return sel, idx, float(maxd.item())


def create_init_grid(boundaries, step_size):
    """Voxel-centre grid, z fastest; returns (coords [N,3] float32, (nx,ny,nz)).

    Same values and order as the reference helper (fusion.py:79-88): per-axis
    ``arange(lower, upper, step) + step/2`` in float32, 'ij' ordering.
    """
    axes = _grid_axes(boundaries, step_size)
    coords = torch.cartesian_prod(*axes)
    shape = torch.Size([a.numel() for a in axes])
    return coords, shape


def _init_low_level_memory(lower_bound, higher_bound, voxel_size, voxel_num):
    """Reference _init_low_level_memory (fusion.py:118-180): the voxel <-> index <-> point closures of instance
    association; pcd_to_voxel / pcd_to_index run on the device (d3fields_amd.pcd_utils.init_low_level_memory)."""
    from . import pcd_utils
    return pcd_utils.init_low_level_memory(lower_bound, higher_bound, voxel_size, voxel_num)


def erode(image, kernel, iterations=1):
    """cv2.erode for the all-ones kernels the reference uses on instance masks (d3fields_amd.pcd_utils.erode)."""
    from . import pcd_utils
    return pcd_utils.erode(image, kernel, iterations)


def _default_device():
    if not torch.cuda.is_available():
        raise RuntimeError("d3fields_amd needs the ROCm device; there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def _as_device_tensor(x, dtype, device):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x.to(device=device, dtype=dtype)


def instance2onehot(instance, N=None):
    """uint8 instance index -> bool one-hot [..., N] (reference fusion.py:90-107).

    numpy in -> numpy out (host data is tiny here); CUDA tensor in -> HIP kernel.
    """
    if N is None:
        N = int(instance.max()) + 1
    if isinstance(instance, np.ndarray):
        assert instance.dtype == np.uint8
        return instance[..., None] == np.arange(N, dtype=np.uint8)
    if isinstance(instance, torch.Tensor):
        assert instance.dtype == torch.uint8
        home = instance.device
        # a CPU tensor is accepted like in the reference: it is moved to the ROCm device, converted by the HIP
        # kernel and moved back (there is still no host implementation)
        inst = (instance if instance.is_cuda else instance.to(_default_device())).contiguous()
        out = torch.empty(inst.shape + (N,), dtype=torch.bool, device=inst.device)
        with torch.cuda.device(inst.device):
            _lib.check(_lib.load().d3f_instance2onehot(_lib.ptr(inst), inst.numel(), N, _lib.ptr(out),
                                                       _lib.current_stream_handle(inst.device)))
        return out.to(home)
    raise NotImplementedError


def onehot2instance(one_hot_mask):
    """[..., N] float/bool (probabilistic or not) -> uint8 argmax (reference fusion.py:109-116)."""
    if isinstance(one_hot_mask, np.ndarray):
        return np.argmax(one_hot_mask, axis=-1).astype(np.uint8)
    if isinstance(one_hot_mask, torch.Tensor):
        home = one_hot_mask.device
        oh = one_hot_mask if one_hot_mask.is_cuda else one_hot_mask.to(_default_device())   # CPU tensors: see instance2onehot
        oh = oh.to(torch.float32).contiguous()
        NI = oh.shape[-1]
        out = torch.empty(oh.shape[:-1], dtype=torch.uint8, device=oh.device)
        with torch.cuda.device(oh.device):
            _lib.check(_lib.load().d3f_onehot2instance(_lib.ptr(oh), out.numel(), NI, _lib.ptr(out),
                                                       _lib.current_stream_handle(oh.device)))
        return out.to(home)
    raise NotImplementedError


class _FieldQueryFn(torch.autograd.Function):
    """Fusion.eval as an autograd node: forward = d3f_eval, backward = d3f_eval_backward (both HIP)."""

    @staticmethod
    def forward(ctx, pts, fusion, names):
        outputs, saved = fusion._launch(pts.detach(), names, False, "eval")
        ctx.fusion, ctx.saved = fusion, saved
        ctx.mark_non_differentiable(outputs["valid_mask"])
        return (outputs["dist"], outputs["valid_mask"]) + tuple(outputs[k] for k in names)

    @staticmethod
    def backward(ctx, grad_dist, _grad_valid, *grad_fused):
        return ctx.fusion._backward(ctx.saved, grad_dist, grad_fused), None, None


class _DistQueryFn(torch.autograd.Function):
    """Fusion.eval_dist as an autograd node (d3f_eval_dist / d3f_eval_dist_backward)."""

    @staticmethod
    def forward(ctx, pts, fusion):
        outputs, _ = fusion._launch(pts.detach(), (), False, "eval_dist")
        views, keep, _ = fusion._views(pts.device)
        ctx.fusion, ctx.saved = fusion, (pts.detach().contiguous(), keep)
        ctx.mark_non_differentiable(outputs["valid_mask"])
        return outputs["dist"], outputs["valid_mask"]

    @staticmethod
    def backward(ctx, grad_dist, _grad_valid):
        pts_c, keep = ctx.saved
        dev = pts_c.device
        n, V = pts_c.shape[0], keep[0].shape[0]
        views = _lib.Views(V, keep[0].shape[1], keep[0].shape[2], _lib.ptr(keep[0]), _lib.ptr(keep[1]), _lib.ptr(keep[2]))
        grad_pts = torch.empty((n, 3), dtype=torch.float32, device=dev)
        gd = grad_dist.to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            _lib.check(ctx.fusion._lib.d3f_eval_dist_backward(ctypes.byref(views), _lib.ptr(pts_c), n, _lib.ptr(gd),
                                                              _lib.ptr(grad_pts), _lib.current_stream_handle(dev)))
        return grad_pts, None


# ----------------------------------------------------------------------------------------
class Fusion:
    """Multi-view 3-D descriptor field (query side).

    Parameters mirror the reference constructor (fusion.py:203); the keyword-only arguments
    inject the upstream producers the reference hard-wires:

    feature_extractor(color[V,H,W,3] uint8 ndarray, params{'patch_h','patch_w'}) -> (V,ph,pw,C) tensor
        stands in for extract_features / DINOv2 (fusion.py:593-629)
    mask_producer(fusion, queries, thresholds, boundaries, merge_all=, expected_labels=, robot_pcd=) -> dict
        {'mask': (V,H,W) uint8 consensus labels or (V,H,W,NI) one-hot, 'consensus_mask_label': [str] * NI,
         optional 'mask_label', 'mask_conf', 'mask_gs'}: Grounded-SAM + instance association (fusion.py:1112-1171)
    mask_tracker(fusion, color, mask_or_None) -> (V,H,W,NI) one-hot or (V,H,W) uint8 labels: XMem's
        xmem_process (fusion.py:631-684); mask is the consensus label image on the first frame, None afterwards

    dtype: torch.float32 (the reference default) or torch.float16.  NOTE: float16 here is NOT the reference's
    half mode (which computes the projection and everything else in half, fusion.py:203,227,709-712): it is a
    STORAGE format of the channel maps only; all arithmetic stays fp32 (see __init__).
    """

    extra_tuning_flags = 0      # D3F_TUNE_* bits OR-ed into every launch of every object (tests: _lib.TUNE_DIRECT_GATHER)

    def __init__(self, num_cam, feat_backbone="dinov2", device="cuda:0", dtype=torch.float32, *,
                 feature_extractor=None, mask_producer=None, mask_tracker=None):
        # dtype=torch.float16: the reference then runs EVERYTHING in half (fusion.py:203,227,709-712), whose projection
        # arithmetic is off by whole pixels (its features differ from its own fp32 run by > 1.0 on the synthetic
        # scenes).  Here float16 is a STORAGE format of the channel maps only (half the texel traffic): depth / K /
        # pose / points and every operation of the query stay fp32, and the result equals the fp32 query on the
        # widened maps bit for bit (D3F_DTYPE_F16, include/d3fields_hip.h).
        if dtype not in (torch.float32, torch.float16):
            raise NotImplementedError("channel maps are stored as float32 or float16; got %s" % dtype)
        if dtype == torch.float16:
            import warnings
            warnings.warn("d3fields_amd.Fusion(dtype=torch.float16) STORES the channel maps in half and computes in float32 (the "
                          "query equals the float32 query on the widened maps bit for bit); the reference's float16 mode "
                          "computes everything in half (fusion.py:203,227,709-712) and gives different numbers.", stacklevel=2)
        self.device = torch.device(device)
        self.dtype = dtype
        self.mu = 0.02                          # reference fusion.py:208
        self.curr_obs_torch = {}
        self.H = -1
        self.W = -1
        self.num_cam = num_cam
        self.feat_backbone = feat_backbone
        self.feature_extractor = feature_extractor
        self.mask_producer = mask_producer
        self.mask_tracker = mask_tracker
        self.track_ids = [0]                    # fusion.py:303
        self.xmem_first_mask_loaded = False     # fusion.py:302
        self._finite_cache = {}                 # key -> (weakref(tensor), signature, stream, event): checked by d3f_map_check
        self._words = None                      # device words of the checks (one per key), allocated on first use
        self._word_slot = {}                    # key -> slot of its newest check;  _slot_key: slot -> the key that took it last
        self._slot_key = []
        self._next_word, self._ring_wrapped = -1, False
        self._finite_override = None            # True: D3F_FLAG_FINITE_MAPS (the caller vouches); False: strict path
        self.debug_recheck_maps = False         # True: d3f_map_check runs before EVERY query (no per-tensor cache): finds writers that
                                                # change a map behind torch's back without invalidate_map_checks()
        self.reference_rounding = False         # True: D3F_FLAG_REFERENCE_ROUNDING -- wide maps in the reference's operation order too
        self._tracker = None                    # rigid_tracking: the captured iteration of the current sequence
        self.tuning_flags = 0                   # D3F_TUNE_* bits (experiments; results do not depend on them)
        self.reorder_points = True              # hand the library scratch so it may walk points in Hilbert order
        self.use_hip_graph = True               # rigid_tracking: capture the optimiser iteration in a HIP graph
        self.fused_tracking = True              # ... and run it as five HIP launches (track_kernels.hip) instead of autograd
        self.graph_whole_tracking_loop = True   # ... with all 100 steps in ONE graph (False: one step replayed 100 times)
        self.single_launch_tracking = True      # ... each step ONE launch (d3f_track_step) where the descriptor map allows it
        self.loop_launch_tracking = True        # ... and all steps of a frame ONE launch (d3f_track_run) up to 512 keypoints
        self.detect_point_order = True          # probe new query tensors for locality (one host sync each, cached)
        self._order_cache = None
        self.cache_point_order = True           # keep the Hilbert order of an unchanged query tensor (a grid queried every
        self._order_ws = None                   # frame) in its scratch and skip the ~0.12 ms re-sort
        self._lattice_cache = None
        self.async_probes = True                # without a per-tensor cache hit: probe asynchronously, launch on the previous verdict
        self._hints = {}                        # n -> [lattice dims or None, unordered?]: what the last finished probes of a
        self._pending = []                      #      query of that size said; _pending: probes still in flight
        self._pinned = []                       # reusable pinned host buffers for the probe results
        self._last_plan = None
        self.record_plans = False               # last_plan(): query the launch plan of every eval (bench.py, tests)
        self.detect_lattice = True              # probe new query tensors for create_init_grid's layout (brick walk, no sort)
        self._lib = _lib.load()                 # fail at construction if the HIP library is missing

    # ---- observation state (reference fusion.py:686-714) --------------------------------
    def update(self, obs):
        """obs: 'color' (V,H,W,3) uint8, 'depth' (V,H,W), 'pose' (V,3,4), 'K' (V,3,3) numpy arrays.

        Optional extension: obs['dino_feats'] (V,ph,pw,C) supplies precomputed features when no
        feature_extractor was injected.
        """
        color = obs["color"]
        self.num_cam = color.shape[0]
        if self.feature_extractor is not None:
            params = {"patch_h": color.shape[1] // 10, "patch_w": color.shape[2] // 10}
            self.curr_obs_torch["dino_feats"] = _as_device_tensor(
                self.feature_extractor(color, params), self.dtype, self.device)
        elif "dino_feats" in obs:
            self.curr_obs_torch["dino_feats"] = _as_device_tensor(obs["dino_feats"], self.dtype, self.device)
        self.curr_obs_torch["color"] = color
        self.curr_obs_torch["color_tensor"] = _as_device_tensor(color, self.dtype, self.device) / 255.0
        for k in ("depth", "pose", "K"):                 # geometry is always fp32 (see __init__)
            self.curr_obs_torch[k] = _as_device_tensor(obs[k], torch.float32, self.device)
        _, self.H, self.W = obs["depth"].shape
        self._finite_cache.clear()

    # ---- the hot path ---------------------------------------------------------------------
    def _check_query(self, pts):
        if len(self.curr_obs_torch) == 0:
            # the reference prints this and calls exit() (fusion.py:313-317); a library raises
            raise RuntimeError("Please call update() first!")
        assert type(pts) == torch.Tensor
        assert len(pts.shape) == 2
        assert pts.shape[1] == 3
        if not pts.is_cuda:
            raise RuntimeError("Fusion.eval: pts must be on the ROCm device (%s); there is no CPU path" % self.device)
        if pts.dtype != torch.float32:
            raise TypeError("Fusion.eval: pts must be float32, got %s" % pts.dtype)

    def _views(self, dev):
        obs = self.curr_obs_torch
        depth, K, pose = obs["depth"], obs["K"], obs["pose"]
        for name, t in (("depth", depth), ("K", K), ("pose", pose)):
            if t.device != dev or t.dtype != torch.float32:
                raise RuntimeError("curr_obs_torch[%r] must be float32 on %s (is %s on %s)" % (name, dev, t.dtype, t.device))
        V = depth.shape[0]
        if tuple(depth.shape[1:]) != (self.H, self.W):
            raise RuntimeError("depth is %s but Fusion.H,W = %d,%d" % (tuple(depth.shape), self.H, self.W))
        pose34 = pose[:, :3, :]                 # the docstring of the reference says (K,4,4); its drivers pass (K,3,4)
        keep = [depth.contiguous(), K.contiguous(), pose34.contiguous()]
        return _lib.Views(V, self.H, self.W, _lib.ptr(keep[0]), _lib.ptr(keep[1]), _lib.ptr(keep[2])), keep, V

    # ---- "this tensor holds only finite values", established ON THE DEVICE ---------------------------------------------
    # The kernels may skip the views that are invalid for a point only when every operand is finite (0 * NaN has to
    # propagate like in the reference, fusion.py:385).  Up to round 3 the shim asked torch: isfinite(t).all().item() per new
    # tensor = five ATen kernels, a bool temporary and a HOST SYNC (2.1 ms per 1.9 GB map against a 1.5 ms query).  Now
    # d3f_map_check streams the tensor once on the caller's stream and leaves a device word; the queries carry the words
    # of depth and maps (d3f_views.depth_nonfinite, d3f_channel_map.nonfinite) and decide on the device.  No sync, no ATen
    # kernel, capturable in a HIP graph.
    # Every CHECK takes the next slot of a ring: a query still in flight on another stream, or a captured HIP graph
    # that baked a word's address in, keeps reading the verdict of ITS tensor while a newer tensor of the same name is checked
    # into another slot; a slot is rewritten only _WORD_SLOTS checks later (a captured graph older than that must be re-captured,
    # or contain its own check).
    _WORD_SLOTS = 256

    def _finite_word(self, key, t, checked=None, batch=None):
        """Address (int) of the device word d3f_map_check wrote for tensor `t` (cached per key on the tensor OBJECT and its
        version counter; a new tensor allocated at the address of a checked one is re-checked), or None when no word can be
        had (then the query takes the strict path: same results).  `checked`: the tensor the kernel should read instead
        of `t` (a contiguous copy the caller just made; not cached).  `batch`: a list -- the check is queued there instead of
        launched, and _flush_checks(batch) enqueues ONE d3f_map_check_many for everything a query needs."""
        dev = t.device
        if self._words is None or self._words.device != dev:
            if torch.cuda.is_current_stream_capturing():
                return None                     # no allocation inside a HIP-graph capture
            self._words = torch.zeros(self._WORD_SLOTS, dtype=torch.int32, device=dev)
            self._word_slot = {}
            self._slot_key = [None] * self._WORD_SLOTS
            self._next_word, self._ring_wrapped = -1, False
            self._finite_cache.clear()
        stream = torch.cuda.current_stream(dev)
        # keyed on the tensor OBJECT (weak reference) and its version counter.  Writers that bypass torch (custom kernels
        # writing into the map in place) must call invalidate_map_checks() -- a stale "finite" verdict would skip 0*NaN
        # terms the reference keeps.
        sig = (t._version, tuple(t.shape), t.data_ptr())
        hit = self._finite_cache.get(key)
        if checked is None and hit is not None and hit[0]() is t and hit[1] == sig and not self.debug_recheck_maps:
            if hit[2] != stream.cuda_stream:
                stream.wait_event(hit[3])       # checked on another stream: order this one behind it
            return self._words.data_ptr() + 4 * self._word_slot[key]
        # every path below makes (or fails to make) a NEW verdict for `key`: the cached one must not outlive this call (ADVICE r5:
        # a `return None` used to leave it behind, pointing at a slot that was already handed on)
        self._finite_cache.pop(key, None)
        src = t if checked is None else checked
        if src.dim() == 3:                      # depth (V,H,W): a one-channel map
            desc = _lib.ChannelMap(src.data_ptr(), src.shape[1], src.shape[2], 1, _lib.DTYPE_F32, src.stride(0), src.stride(1),
                                   max(src.stride(2), 1), None)
            ok = src.dtype == torch.float32 and src.stride(2) >= 1
        else:
            desc = _lib.ChannelMap(src.data_ptr(), src.shape[1], src.shape[2], src.shape[3],
                                   _lib.DTYPE_F16 if src.dtype == torch.float16 else _lib.DTYPE_F32,
                                   src.stride(0), src.stride(1), src.stride(2), None)
            ok = src.dtype in (torch.float32, torch.float16) and src.stride(3) == 1 and src.stride(2) >= src.shape[3]
        if not ok or min(src.stride()) < 0:
            return None
        # the slot, taken only now that the descriptor is valid: the next one of the ring that no LIVE cached verdict owns.  A
        # wrap never zeroes the ring (that also cleared the words of tensors the SAME query had already resolved as
        # cache hits -- a cached non-finite depth read as finite -- and of queries still in flight on other streams): once the
        # ring has wrapped, d3f_map_check_many clears exactly the words it writes (_flush_checks stops passing
        # CHECK_WORDS_ARE_ZERO), and a live verdict keeps its slot until its own tensor is re-checked.
        slot = self._next_word
        for _ in range(self._WORD_SLOTS):
            slot = (slot + 1) % self._WORD_SLOTS
            if slot == 0 and self._next_word >= 0:
                self._ring_wrapped = True
            owner = self._slot_key[slot]
            if owner is None or self._word_slot.get(owner) != slot or owner not in self._finite_cache:
                break
        self._next_word = slot
        self._slot_key[slot] = key
        self._word_slot[key] = slot
        addr = self._words.data_ptr() + 4 * slot
        cacheable = checked is None and not torch.cuda.is_current_stream_capturing()
        if batch is not None:
            import weakref
            batch.append([desc, int(src.shape[0]), addr, src, (key, weakref.ref(t), sig) if cacheable else None])
            return addr
        with torch.cuda.device(dev):
            _lib.check(self._lib.d3f_map_check(ctypes.byref(desc), src.shape[0], ctypes.c_void_p(addr), _lib.current_stream_handle(dev)))
        if cacheable:
            import weakref
            ev = torch.cuda.Event()
            ev.record(stream)
            self._finite_cache[key] = (weakref.ref(t), sig, stream.cuda_stream, ev)
        return addr

    def _flush_checks(self, batch, dev):
        """ONE d3f_map_check_many for the checks _finite_word queued (depth + every map of a query whose tensors are new: the
        per-frame refresh of a tracking loop was six launches)."""
        if not batch:
            return
        n = len(batch)
        descs = (_lib.ChannelMap * n)(*[b[0] for b in batch])
        views = (ctypes.c_int32 * n)(*[b[1] for b in batch])
        words = (ctypes.c_void_p * n)(*[b[2] for b in batch])
        # a captured graph replays the launch on whatever the words hold then: let the call clear them itself there
        # (so does a ring that has wrapped: its slots hold the verdicts of earlier tensors, and nothing zeroes live words)
        zero = 0 if (torch.cuda.is_current_stream_capturing() or self._ring_wrapped) else _lib.CHECK_WORDS_ARE_ZERO
        stream = torch.cuda.current_stream(dev)
        with torch.cuda.device(dev):            # (before the first wrap every check takes a slot that is still zero from the allocation)
            _lib.check(self._lib.d3f_map_check_many(descs, views, n, words, zero, _lib.current_stream_handle(dev)))
        ev = None
        for b in batch:
            if b[4] is not None:
                if ev is None:
                    ev = torch.cuda.Event()
                    ev.record(stream)
                key, ref, sig = b[4]
                self._finite_cache[key] = (ref, sig, stream.cuda_stream, ev)
        del batch[:]

    def maps_are_finite(self, names=("depth", "dino_feats", "mask", "color_tensor")):
        """Host-side verdict of the device words (ONE host sync; diagnostics and tests only -- no query needs it)."""
        bad = False
        for k in names:
            t = self.curr_obs_torch.get(k)
            if isinstance(t, torch.Tensor):
                addr = self._finite_word(k, t)
                if addr is None:
                    return False
                bad = bad or bool(self._words[(addr - self._words.data_ptr()) // 4].item())
        return not bad

    def invalidate_map_checks(self):
        """Forget the device-side 'this map holds only finite values' verdicts: call after writing into a curr_obs_torch
        tensor in place from outside torch (the next query re-checks the tensors it reads, ~0.35 ms per 1.9 GB)."""
        self._finite_cache.clear()

    def _probe_now(self, pts_c, stream):
        """(lattice dims or None, unordered?) of a query tensor from ONE d3f_points_probe launch and ONE host sync (round 6; rounds
        3-5: d3f_lattice_probe and d3f_point_order_locality, a launch and a sync each)."""
        out = torch.empty(_lib.PROBE_WORDS, dtype=torch.int32, device=pts_c.device)
        _lib.check(self._lib.d3f_points_probe(_lib.ptr(pts_c), pts_c.shape[0], _lib.ptr(out), stream))
        return self._parse_probe(out.cpu())

    def _is_unordered(self, pts_c, stream):
        """The locality verdict of d3f_points_probe on a NEW query tensor (cached by storage / version / length, so a grid queried
        repeatedly is probed once): True when consecutive points are no closer than points half the batch apart,
        i.e. a shuffled or random cloud.  Only steers D3F_FLAG_UNORDERED_POINTS / D3F_FLAG_LOCAL_POINTS -- a stale answer costs
        time, never correctness."""
        sig = (pts_c.data_ptr(), pts_c._version, pts_c.shape[0])
        hit = self._order_cache
        if hit is None or hit[0] != sig:
            if torch.cuda.is_current_stream_capturing():
                return False                    # no host sync inside a HIP-graph capture
            dims, unordered = self._probe_now(pts_c, stream)
            hit = (sig, unordered)
            self._order_cache = hit
            if self.cache_point_order:
                self._lattice_cache = (sig, dims)
        return hit[1]

    def _lattice_dims(self, pts_c, stream):
        """(nx, ny, nz) when the query tensor is a z-fastest lattice -- the materialised create_init_grid output the
        reference's drivers hand to batch_eval (vis_repr.py:88-93) -- else None.  d3f_points_probe on a NEW query
        tensor (one host sync; cached by storage / version / length when cache_point_order is on; the same launch also
        answers _is_unordered).  The dims only select the walk order of d3f_eval_lattice, which reads every coordinate from
        the tensor itself: a stale answer costs time, never correctness."""
        sig = (pts_c.data_ptr(), pts_c._version, pts_c.shape[0])
        hit = self._lattice_cache if self.cache_point_order else None
        if hit is None or hit[0] != sig:
            if torch.cuda.is_current_stream_capturing():
                return None                     # no host sync inside a HIP-graph capture
            dims, unordered = self._probe_now(pts_c, stream)
            hit = (sig, dims)
            self._lattice_cache = hit
            self._order_cache = (sig, unordered)
        return hit[1]

    # ---- probes without a host sync ------------------------------------------------------------------------
    # Both probes (is the tensor create_init_grid's lattice? has the caller's order any locality?) only choose the ORDER
    # in which points are processed; no result depends on them (d3f_eval_lattice is correct for any dims whose product
    # is n).  So a query whose tensor is not in the per-tensor cache does not wait for its own probes: they are enqueued,
    # their results land in pinned host memory, and the launch uses the verdict of the most recent FINISHED probes of
    # a query of the same size (per-frame grids / clouds repeat their layout).  Only the first query of a size waits.
    @staticmethod
    def _parse_probe(words):
        """(lattice dims or None, unordered?) from the D3F_PROBE_WORDS words of d3f_points_probe (a CPU int32 tensor)."""
        li = words[:24].tolist()
        fl = words[24:36].view(torch.float32).tolist()
        dims = tuple(li[:3]) if (li[0] > 0 and not any(li[8:24])) else None
        near, far, cnt = sum(fl[0::3]), sum(fl[1::3]), sum(fl[2::3])
        return dims, bool(cnt > 0 and near > 0.25 * far)

    def _poll_probes(self):
        still = []
        for n, out_host, ev in self._pending:
            if ev.query():
                dims, unordered = self._parse_probe(out_host)
                self._hints.pop(n, None)                           # (re-)insert as the most recent entry
                self._hints[n] = [dims, unordered]
                while len(self._hints) > 64:                        # bounded: sizes come and go in long-running trackers
                    self._hints.pop(next(iter(self._hints)))
                self._pinned.append(out_host)
            else:
                still.append((n, out_host, ev))
        self._pending = still

    def _enqueue_probes(self, pts_c, stream):
        """d3f_points_probe (lattice + locality, ONE launch, nothing to clear) on the current stream; results -> pinned host memory,
        asynchronously"""
        dev = pts_c.device
        n = pts_c.shape[0]
        out = torch.empty(_lib.PROBE_WORDS, dtype=torch.int32, device=dev)
        _lib.check(self._lib.d3f_points_probe(_lib.ptr(pts_c), n, _lib.ptr(out), stream))
        host = self._pinned.pop() if self._pinned else torch.empty(_lib.PROBE_WORDS, dtype=torch.int32).pin_memory()
        host.copy_(out, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self._pending.append((n, host, ev))
        return ev

    def _order_hint(self, pts_c, stream):
        """(lattice dims or None, unordered?) for this query without waiting for its own probes (see above)"""
        n = pts_c.shape[0]
        self._poll_probes()
        if len(self._pending) < 8:                                       # bounded queue: a stalled stream must not pile up probes
            ev = self._enqueue_probes(pts_c, stream)
        else:
            ev = None
        if n not in self._hints:
            if ev is None:
                ev = self._pending[-1][2]
            ev.synchronize()                                             # first query of this size: wait once
            self._poll_probes()
        return tuple(self._hints.get(n, [None, False]))

    def _query_flags(self):
        return ((_lib.FLAG_FINITE_MAPS if self._finite_override else 0) | (_lib.FLAG_REFERENCE_ROUNDING if self.reference_rounding else 0) |
                int(self.tuning_flags) | int(Fusion.extra_tuning_flags))

    def last_plan(self):
        """What the last eval / batch_eval launched (for bench.py and tests): kernel entry point, tile size and how the
        points were ordered -- from d3f_eval_plan_query on the same shapes and flags."""
        return self._last_plan

    def _record_plan(self, views, n, maps, n_maps, flags, have_ws, want_inter, lattice):
        plan = _lib.EvalPlan()
        if lattice is not None:
            rc = self._lib.d3f_eval_plan_query_lattice(ctypes.byref(views), lattice[0], lattice[1], lattice[2], maps, n_maps, flags,
                                                       1 if want_inter else 0, ctypes.byref(plan))
        else:
            rc = self._lib.d3f_eval_plan_query(ctypes.byref(views), n, maps, n_maps, flags, 1 if have_ws else 0,
                                               1 if want_inter else 0, ctypes.byref(plan))
        if rc != 0:
            return
        runs = any(plan.staged[s] >= 16 for s in range(n_maps))
        wide = any(plan.vectors_per_lane[s] == -4 for s in range(n_maps))
        f16 = any(maps[s].dtype == _lib.DTYPE_F16 for s in range(n_maps))
        kernel = ("fused_eval_f16_kernel<0>" if f16 else "fused_eval_wide_kernel<0>" if wide else "fused_eval_kernel<0>")
        if n_maps == 0 and int(plan.reorder) == 0 and int(views.V) <= 8:
            # the distance-only pass (fuse_direct.hip): <mode, view count (0: five to eight), waves per SIMD, depth maps tiled first?, points from grid axes?>
            tiled = have_ws and n >= (1 << 22)
            kernel = "fused_eval_dist_kernel<0, %d, %d, %s, false>" % (int(views.V) if int(views.V) <= 4 else 0, 8 if int(views.V) <= 2 else 6, "true" if tiled else "false")
        window = 2000 <= plan.reserved < 3000
        if window:
            r = plan.reserved - 2000
            w0 = [s for s in range(n_maps) if plan.staged[s] == 3][0]           # the windowed map (any position in the call)
            # (the last template argument: the view count as a compile-time constant, 4 / 8 -> software-pipelined point loop)
            vfix = int(views.V) if (int(views.V) in (4, 8) and int(plan.tile_points) == 64 and plan.lanes_per_point[w0] == 16) else 0
            # (third argument: the register budget the variant is built for -- 4 waves per SIMD for every 16-lane variant; the plan's
            #  last digit is the workgroups per CU the POOL is sized for)
            kernel = "fused_eval_window_kernel<%d, %d, %d, 256, %d, %d, %s%s>" % (r // 100, r // 10 % 10, 4 if plan.lanes_per_point[w0] == 16 else r % 10, plan.lanes_per_point[w0], vfix,
                                                                                "false" if lattice is not None else "true",
                                                                                ", true" if maps[w0].dtype == _lib.DTYPE_F16 else "")
        elif any(plan.staged[s] == 5 for s in range(n_maps)):                  # the rows of a 32-point brick in registers (fuse_rows.hip)
            kernel = "fused_eval_rows_kernel"
        elif plan.reserved >= 100:
            lg, vc = (plan.reserved - 100) // 10, (plan.reserved - 100) % 10
            kernel = "fused_eval_sliced_kernel<%d, %d, %d%s>" % (lg, vc, {1: 8, 2: 7, 4: 5}.get(vc, 5), ", true" if f16 else "")
        elif runs and not f16 and not wide:
            s0 = [s for s in range(n_maps) if plan.staged[s] >= 16][0]
            kernel = "fused_eval_runs_kernel<0, %d, %d, %d>" % (plan.vectors_per_lane[s0], plan.staged[s0] - 16, plan.reserved)
        sliced = 100 <= plan.reserved < 200
        order = {2: "closed-form brick walk of the lattice (no keys, no sort)", 1: "Hilbert-cell order (512^3 key grid over the cloud's box, counting sort by a key prefix + exact rank inside, hand-written)",
                 0: "caller order"}[int(plan.reorder)]
        order += "; channel-sliced over the XCDs" if sliced else ""
        if window:
            order += "; %d-point bricks through texel windows in LDS" % int(plan.tile_points)
        elif kernel == "fused_eval_rows_kernel":
            order += "; 32-point bricks, their fused rows in registers, walked cell by cell"
        elif runs:
            order += "; cell runs of %d consecutive points" % (max(plan.staged[s] for s in range(n_maps)) - 16)
        self._last_plan = {"kernel": kernel, "tile_points": int(plan.tile_points), "point_order": order,
                           "workgroups": int(plan.workgroups), "lattice": lattice, "gated_window": bool(plan.gated_window),
                           "family": (self._lib.d3f_plan_family_name(int(plan.family)) or b"?").decode()}
        if plan.gated_window:
            # a cloud on the gated pair of launches (ABI 5): the fields above describe the cell-run side; the window side is the
            # sparse-pool window kernel on 64-point tiles of the same order.  last_gate() says which one ran.
            r = int(plan.reserved2) - 2000
            self._last_plan["window_side"] = {
                "kernel": "fused_eval_window_kernel<%d, %d, 4, 256, 16, %d, true>" % (r // 100, r // 10 % 10, int(views.V) if int(views.V) in (4, 8) else 0),
                "tile_points": 64,
                "point_order": order.split(";")[0] + "; 64-point tiles through touched-texel windows in LDS (device-gated against the cell runs)"}

    def last_gate(self):
        """After a query of a cloud that got the gated pair of launches (last_plan()['gated_window']): (tiles of the probe's sample
        that fit the window kernel's pool, True if the window side ran).  ONE host sync; diagnostics (bench.py, tests) only."""
        ws = getattr(self, "_last_ws", None)
        if ws is None or getattr(self, "_last_ws_n", 0) <= 0:
            return None
        off = int(self._lib.d3f_eval_gate_offset(self._last_ws_n))
        fit = int(ws[off:off + 4].view(torch.int32).item())
        forced = bool(self._query_flags() & _lib.TUNE_WINDOW_SIDE)
        return fit, bool(forced or fit >= _lib.GATE_MIN_FIT)

    def _run(self, pts, return_names, return_inter, mode):
        self._check_query(pts)
        if pts.requires_grad and torch.is_grad_enabled():
            # autograd consumer: rigid_tracking back-propagates through eval (fusion.py:1650-1665)
            if return_inter:
                raise NotImplementedError("gradients through '<k>_inter' outputs are not implemented; "
                                          "detach pts or use torch.no_grad()")
            if mode == "eval_dist":
                flat = _DistQueryFn.apply(pts, self)
                return {"dist": flat[0], "valid_mask": flat[1]}
            names = list(return_names)
            flat = _FieldQueryFn.apply(pts, self, names)
            out = {"dist": flat[0], "valid_mask": flat[1]}
            out.update(zip(names, flat[2:]))
            return out
        return self._launch(pts, return_names, return_inter, mode)[0]

    def _launch(self, pts, return_names, return_inter, mode):
        """Enqueues the forward kernel; returns (outputs, tensors the launch read)."""
        dev = pts.device
        lib = self._lib
        n = pts.shape[0]
        pts_c = pts.detach().contiguous()
        views, keep, V = self._views(dev)
        dist = torch.empty(n, dtype=torch.float32, device=dev)
        valid = torch.empty(n, dtype=torch.bool, device=dev)
        outputs = {"dist": dist, "valid_mask": valid}
        with torch.cuda.device(dev):
            stream = _lib.current_stream_handle(dev)
            if mode == "eval_dist":
                _lib.check(lib.d3f_eval_dist(ctypes.byref(views), _lib.ptr(pts_c), n, _lib.ptr(dist), _lib.ptr(valid), stream))
                return outputs, None
            names = list(return_names)
            if len(names) > _lib.MAX_MAPS:
                raise ValueError("at most %d return_names per call" % _lib.MAX_MAPS)
            maps = (_lib.ChannelMap * max(len(names), 1))()
            fused = (ctypes.c_void_p * max(len(names), 1))()
            inter = (ctypes.c_void_p * max(len(names), 1))()
            # finiteness of depth and maps: device words written by d3f_map_check when a tensor is new (no host sync), keyed
            # on the caller's tensor objects; _finite_override (RigidTracker's private observation) replaces them by a fixed flag
            words = self._finite_override is None and bool(names)
            checks = []                                    # new tensors of this query: ONE d3f_map_check_many below
            if words:
                views.depth_nonfinite = self._finite_word("depth", self.curr_obs_torch["depth"], batch=checks)
            used_maps = []
            for s, k in enumerate(names):
                m = self.curr_obs_torch[k]                 # KeyError for unknown names, like the reference
                if not isinstance(m, torch.Tensor) or m.dim() != 4 or m.shape[0] != V:
                    raise ValueError("curr_obs_torch[%r] must be a (V,h,w,C) tensor" % k)
                if m.device != dev or m.dtype not in (torch.float32, torch.float16):
                    raise RuntimeError("curr_obs_torch[%r] must be float32 or float16 on %s" % (k, dev))
                m_caller = m
                if m.stride(3) != 1:                           # a NEW tensor on every call: checked every call, never cached
                    m = m.contiguous()
                    keep.append(m)
                word = self._finite_word(k, m_caller, None if m is m_caller else m, batch=checks) if words else None
                used_maps.append(m)
                C = m.shape[3]
                o = torch.empty((n, C), dtype=torch.float32, device=dev)
                outputs[k] = o
                maps[s] = _lib.ChannelMap(m.data_ptr(), m.shape[1], m.shape[2], C,
                                          _lib.DTYPE_F16 if m.dtype == torch.float16 else _lib.DTYPE_F32,
                                          m.stride(0), m.stride(1), m.stride(2), word)
                fused[s] = o.data_ptr()
                if return_inter:
                    it = torch.empty((V, n, C), dtype=torch.float32, device=dev)
                    outputs[k + "_inter"] = it
                    inter[s] = it.data_ptr()
            self._flush_checks(checks, dev)
            flags = self._query_flags()
            ws, ws_bytes = None, 0
            dims, hinted_unordered = None, None
            if self.reorder_points and names and n >= 65536 and not torch.cuda.is_current_stream_capturing():
                if self.async_probes and not self.cache_point_order:
                    # no per-tensor cache: probes run asynchronously, the launch follows the last finished verdict
                    dims, hinted_unordered = self._order_hint(pts_c, stream)
                    if not self.detect_lattice:
                        dims = None
                elif self.detect_lattice:
                    dims = self._lattice_dims(pts_c, stream)
            if dims is not None:
                # a regular grid: closed-form brick walk on large maps / column runs on patch-resolution maps; no scratch
                if self.record_plans:
                    self._record_plan(views, n, maps, len(names), flags, False, return_inter, dims)
                _lib.check(lib.d3f_eval_lattice(ctypes.byref(views), _lib.ptr(pts_c), dims[0], dims[1], dims[2], maps, len(names),
                                                self.mu, flags, _lib.ptr(dist), _lib.ptr(valid), fused,
                                                inter if return_inter else None, stream))
                return outputs, (pts_c, keep[0], keep[1], keep[2], used_maps)
            if self.reorder_points and names and n >= 65536:
                map_bytes = sum(m.numel() * m.element_size() for m in used_maps)
                small = map_bytes <= (64 << 20)
                # the verdict of the locality probe matters in two places: small maps are reordered only for an UNORDERED cloud, and
                # a small cloud (below the window kernel's 262 144 points) on maps inside the Infinity Cache keeps a LOCAL caller order
                # (the 71 k surface points of vis_repr.py:97-103 in flat-index order: five ordering launches around a 120-us query)
                keep_local = (not small) and n < 262144 and map_bytes <= (256 << 20)
                if (small or keep_local) and self.detect_point_order:
                    unordered = hinted_unordered if hinted_unordered is not None else self._is_unordered(pts_c, stream)
                    if small and unordered:
                        flags |= _lib.FLAG_UNORDERED_POINTS     # larger maps are walked in Hilbert order anyway
                    if keep_local and not unordered:
                        flags |= _lib.FLAG_LOCAL_POINTS
                ws_bytes = lib.d3f_eval_workspace_bytes(n)
                sig = (pts_c.data_ptr(), pts_c._version, n, int(stream.value or 0))   # per stream: the order is written asynchronously
                held = self._order_ws if self.cache_point_order else None
                plan = None
                if self.cache_point_order:
                    plan = _lib.EvalPlan()
                    _lib.check(lib.d3f_eval_plan_query(ctypes.byref(views), n, maps, len(names), flags, 1, 1 if return_inter else 0,
                                                       ctypes.byref(plan)))
                if held is not None and held[0] == sig and held[2] and plan.reorder:
                    ws = held[1]                                            # same points as last time: the order is still there
                    flags |= _lib.FLAG_REUSE_POINT_ORDER
                else:
                    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)   # torch's caching allocator: no hipMalloc per call
                    if self.cache_point_order:
                        self._order_ws = (sig, ws, bool(plan.reorder))      # filled by this call iff the library reorders
            if not names:
                # the distance-only pass over a big batch looks the depth pixels up in a tiled copy (d3f_eval_dist_workspace_bytes: 0 below 2^22 points)
                ws_bytes = int(lib.d3f_eval_dist_workspace_bytes(ctypes.byref(views), n)) if self.reorder_points else 0
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
            if self.record_plans:
                self._record_plan(views, n, maps, len(names), flags, ws is not None, return_inter, None)
            self._last_ws, self._last_ws_n = ws, n
            _lib.check(lib.d3f_eval(ctypes.byref(views), _lib.ptr(pts_c), n, maps, len(names), self.mu, flags,
                                    _lib.ptr(dist), _lib.ptr(valid), fused, inter if return_inter else None,
                                    _lib.ptr(ws), ws_bytes, stream))
        return outputs, (pts_c, keep[0], keep[1], keep[2], used_maps)

    def _backward(self, saved, grad_dist, grad_fused):
        """d3f_eval_backward on the tensors the forward launch read."""
        pts_c, depth, K, pose, used_maps = saved
        dev = pts_c.device
        n, V = pts_c.shape[0], depth.shape[0]
        views = _lib.Views(V, depth.shape[1], depth.shape[2], _lib.ptr(depth), _lib.ptr(K), _lib.ptr(pose))
        grad_pts = torch.empty((n, 3), dtype=torch.float32, device=dev)
        nm = len(used_maps)
        maps = (_lib.ChannelMap * max(nm, 1))()
        gptr = (ctypes.c_void_p * max(nm, 1))()
        hold = []
        for s, m in enumerate(used_maps):
            maps[s] = _lib.ChannelMap(m.data_ptr(), m.shape[1], m.shape[2], m.shape[3],
                                      _lib.DTYPE_F16 if m.dtype == torch.float16 else _lib.DTYPE_F32,
                                      m.stride(0), m.stride(1), m.stride(2))
            g = grad_fused[s]
            if g is not None:
                g = g.to(torch.float32).contiguous()
                hold.append(g)
                gptr[s] = g.data_ptr()
        gd = grad_dist.to(torch.float32).contiguous() if grad_dist is not None else None
        with torch.cuda.device(dev):
            _lib.check(self._lib.d3f_eval_backward(ctypes.byref(views), _lib.ptr(pts_c), n, maps, nm, self.mu, _lib.ptr(gd),
                                                   gptr, _lib.ptr(grad_pts), _lib.current_stream_handle(dev)))
        return grad_pts

    def eval(self, pts, return_names=["dino_feats", "mask"], return_inter=False):
        """Reference Fusion.eval (fusion.py:305-394).

        pts (N,3) float32 in the world frame.  Returns 'dist' (N), 'valid_mask' (N) bool and, per
        name in return_names, the fused (N,C) channels; with return_inter also '<name>_inter' (V,N,C).
        """
        return self._run(pts, return_names, return_inter, "eval")

    def eval_dist(self, pts):
        """Reference Fusion.eval_dist (fusion.py:396-436): unclamped mean signed distance."""
        return self._run(pts, (), False, "eval_dist")

    def batch_eval(self, pts, return_names=["dino_feats", "mask"]):
        """Reference Fusion.batch_eval (fusion.py:526-545).

        The reference walks 60 000-point chunks only to bound its [V,N,C] temporaries; the fused
        kernel has none, so the whole batch is one launch with the same concatenated result.
        """
        return self._run(pts, return_names, False, "eval")

    # ---- regular grids and keypoint selection ---------------------------
    def _grid(self, boundaries, step_size):
        axes = [a.to(self.device) for a in _grid_axes(boundaries, step_size)]
        g = _lib.Grid(_lib.ptr(axes[0]), _lib.ptr(axes[1]), _lib.ptr(axes[2]), axes[0].numel(), axes[1].numel(), axes[2].numel(), 0)
        return g, axes

    def eval_grid(self, boundaries, step_size, return_names=[]):
        """batch_eval(create_init_grid(boundaries, step_size)[0].to(device), return_names) without ever
        materialising the grid (the reference's first pass, vis_repr.py:88-93): the kernel generates each
        voxel centre from the axis arrays.  Same dict as batch_eval, in flat grid order, plus 'grid_shape'."""
        if len(self.curr_obs_torch) == 0:
            raise RuntimeError("Please call update() first!")
        dev = self.device
        lib = self._lib
        grid, axes = self._grid(boundaries, step_size)
        n = grid.nx * grid.ny * grid.nz
        views, keep, V = self._views(dev)
        names = list(return_names)
        dist = torch.empty(n, dtype=torch.float32, device=dev)
        valid = torch.empty(n, dtype=torch.bool, device=dev)
        out = {"dist": dist, "valid_mask": valid, "grid_shape": torch.Size([grid.nx, grid.ny, grid.nz])}
        maps = (_lib.ChannelMap * max(len(names), 1))()
        fused = (ctypes.c_void_p * max(len(names), 1))()
        words = self._finite_override is None and bool(names)
        if words:
            views.depth_nonfinite = self._finite_word("depth", self.curr_obs_torch["depth"])
        for s, k in enumerate(names):
            m = self.curr_obs_torch[k]
            if m.device != dev or m.dtype not in (torch.float32, torch.float16):
                raise RuntimeError("curr_obs_torch[%r] must be float32 or float16 on %s" % (k, dev))
            m_caller = m
            if m.stride(3) != 1:
                m = m.contiguous()
                keep.append(m)
            word = self._finite_word(k, m_caller, None if m is m_caller else m) if words else None
            out[k] = torch.empty((n, m.shape[3]), dtype=torch.float32, device=dev)
            maps[s] = _lib.ChannelMap(m.data_ptr(), m.shape[1], m.shape[2], m.shape[3],
                                      _lib.DTYPE_F16 if m.dtype == torch.float16 else _lib.DTYPE_F32,
                                      m.stride(0), m.stride(1), m.stride(2), word)
            fused[s] = out[k].data_ptr()
        flags = self._query_flags()
        with torch.cuda.device(dev):
            _lib.check(lib.d3f_eval_grid(ctypes.byref(views), ctypes.byref(grid), maps, len(names), self.mu, flags,
                                         _lib.ptr(dist), _lib.ptr(valid), fused, _lib.current_stream_handle(dev)))
        return out

    def grid_shell(self, boundaries, step_size, dist_threshold=0.005):
        """Flat indices (ascending) and coordinates of the grid points with valid_mask & |dist| < dist_threshold:
        the pre-filter of select_features_* (fusion.py:1430,1444) fused into the grid pass, so that no
        per-point tensor of the ~1e8-point grid is ever written."""
        if len(self.curr_obs_torch) == 0:
            raise RuntimeError("Please call update() first!")
        dev = self.device
        grid, axes = self._grid(boundaries, step_size)
        n = grid.nx * grid.ny * grid.nz
        views, keep, V = self._views(dev)
        count = torch.zeros(1, dtype=torch.int64, device=dev)
        capacity = max(1 << 16, n // 16)
        ws_bytes = self._lib.d3f_grid_shell_workspace_bytes(ctypes.byref(grid))
        # (+ room for the tiled copy of the depth maps a big grid's lookups go to: include/d3fields_hip.h, d3f_grid_shell)
        ws_bytes = (ws_bytes + 255) // 256 * 256 + int(self._lib.d3f_eval_dist_workspace_bytes(ctypes.byref(views), grid.nx * grid.ny * grid.nz))
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
        while True:
            idx = torch.empty(capacity, dtype=torch.int64, device=dev)
            with torch.cuda.device(dev):
                _lib.check(self._lib.d3f_grid_shell(ctypes.byref(views), ctypes.byref(grid), self.mu, float(dist_threshold),
                                                    capacity, _lib.ptr(idx), _lib.ptr(count), _lib.ptr(ws), ws_bytes,
                                                    _lib.current_stream_handle(dev)))
            found = int(count.item())
            if found <= capacity:
                break
            capacity = found                      # rare: the shell is thicker than 1/16 of the grid -> one exact re-run
        idx = idx[:found]                         # already ascending: the reference's boolean-mask order
        iz = idx % grid.nz
        ixy = idx // grid.nz
        pts = torch.stack((axes[0][ixy // grid.ny], axes[1][ixy % grid.ny], axes[2][iz]), dim=1)
        return idx, pts

    def select_features_rand(self, boundaries, N, per_instance=False, res=None, init_idx=-1):
        """Reference Fusion.select_features_rand (fusion.py:1418-1475): N farthest-point-sampled keypoints per
        instance on the object surfaces, with their descriptors.  Returns (src_feats_list, src_pts_list,
        img_list); img_list (cv2 debug renderings in the reference) is always empty here."""
        res = 0.001 if res is None else res
        dist_threshold = 0.005
        label = self.curr_obs_torch["consensus_mask_label"]
        with torch.no_grad():
            _, shell_pts = self.grid_shell(boundaries, res, dist_threshold)
            mask = self.eval(shell_pts, return_names=["mask"])["mask"] if shell_pts.shape[0] else shell_pts.new_zeros((0, len(label)))
            mask = mask / (mask.sum(dim=1, keepdim=True) + 1e-7)
            src_feats_list, src_pts_list = [], []
            last_label = label[0]
            for i in range(1, len(label)):
                if label[i] == last_label and not per_instance:
                    continue
                masked_pts = shell_pts[mask[:, i] > 0.6]
                sample_pts, _, _ = fps(masked_pts, N, init_idx=init_idx)
                src_feats_list.append(self.eval(sample_pts)["dino_feats"])
                src_pts_list.append(sample_pts.cpu().numpy())
                last_label = label[i]
        return src_feats_list, src_pts_list, []

    def select_features_from_pcd(self, pcd, N, per_instance=False, init_idx=-1, vis=False):
        """Reference Fusion.select_features_from_pcd (fusion.py:1477-1537): like select_features_rand, but the
        candidates are the points of `pcd` ((M,3) numpy array) instead of a voxel grid: keep |dist| < 5 mm & valid &
        normalised mask of the instance > 0.6, farthest-point-sample N of them, query their descriptors.  Instances
        without a candidate are skipped, as in the reference.  Returns (src_feats_list, src_pts_list, img_list);
        the cv2 debug rendering (vis=True) is upstream tooling and is not provided."""
        if vis:
            raise NotImplementedError("vis=True draws keypoints with cv2 in the reference; not part of the field query")
        dist_threshold = 0.005
        dev = torch.device(self.device)
        label = self.curr_obs_torch["consensus_mask_label"]
        with torch.no_grad():
            pcd_t = _as_device_tensor(np.asarray(pcd), torch.float32, dev)
            out = self.batch_eval(pcd_t, return_names=["mask"])
            near = (out["dist"].abs() < dist_threshold) & out["valid_mask"]
            mask = out["mask"] / (out["mask"].sum(dim=1, keepdim=True) + 1e-7)
            src_feats_list, src_pts_list = [], []
            last_label = label[0]
            for i in range(1, len(label)):
                if label[i] == last_label and not per_instance:
                    continue
                masked_pts = pcd_t[(mask[:, i] > 0.6) & near]
                if masked_pts.shape[0] == 0:
                    continue
                sample_pts, _, _ = fps(masked_pts, N, init_idx=init_idx)
                src_feats_list.append(self.eval(sample_pts)["dino_feats"])
                src_pts_list.append(sample_pts.cpu().numpy())
                last_label = label[i]
        return src_feats_list, src_pts_list, []

    def rigid_tracking(self, src_feat_info, last_match_pts_list, boundaries, rand_ptcl_num):
        """Per-instance SE(3) tracking of keypoints (reference fusion.py:1608-1685): 100 Adam steps (lr 0.01) on
        translation + axis-angle parameters, loss = masked descriptor distance + 100 * positive distance + parameter
        norms, gradients through `eval` on the HIP backward kernel.  Same arguments and return value as the
        reference ({'match_pts_list': [rand_ptcl_num,3] numpy array per instance}); `boundaries` only feeds the
        reference's disabled out-of-bounds term (weight 0, fusion.py:1618,1663) and is accepted and ignored.
        `self.use_hip_graph` replays the iteration as one HIP graph (d3fields_amd/rigid.py)."""
        from . import rigid
        dev = torch.device(self.device)
        src_feats = torch.cat([_as_device_tensor(src_feat_info[k]["src_feats"], torch.float32, dev) for k in src_feat_info.keys()],
                              dim=0)
        num_instance = len(last_match_pts_list)
        last_np = np.stack([np.asarray(p) for p in last_match_pts_list], axis=0)
        assert last_np.shape[:2] == (num_instance, rand_ptcl_num)
        last = torch.from_numpy(last_np).to(dev, dtype=torch.float32)
        if self.use_hip_graph:
            # one capture per sequence: the graph is kept while instances / keypoints / views / map sizes stay the same
            key = rigid.RigidTracker.signature(self, num_instance, rand_ptcl_num)
            if (self._tracker is None or self._tracker.key != key or self._tracker.fused != self.fused_tracking or
                    self._tracker.whole_loop != (self.graph_whole_tracking_loop and self.fused_tracking) or
                    self._tracker.single_requested != self.single_launch_tracking or
                    self._tracker.loop_requested != self.loop_launch_tracking):
                self._tracker = rigid.RigidTracker(self, num_instance, rand_ptcl_num, fused=self.fused_tracking,
                                                   whole_loop=self.graph_whole_tracking_loop, single_launch=self.single_launch_tracking,
                                                   loop_launch=self.loop_launch_tracking)
            cur, _ = self._tracker.run(self, src_feats, last)
        else:
            cur, _ = rigid.track_rigid(self, src_feats, last, use_graph=False)
        cur = cur.cpu().numpy()
        return {"match_pts_list": [cur[i * rand_ptcl_num:(i + 1) * rand_ptcl_num] for i in range(num_instance)]}

    def pcd_iou(self, pcd_1, pcd_2, threshold):
        """Reference Fusion.pcd_iou (fusion.py:724-741); see d3fields_amd.pcd_utils.pcd_iou."""
        from . import pcd_utils
        return pcd_utils.pcd_iou(pcd_1, pcd_2, threshold)

    def vox_idx_iou(self, vox_idx_1, vox_idx_2):
        """Reference Fusion.vox_idx_iou (fusion.py:794-799); see d3fields_amd.pcd_utils.vox_idx_iou."""
        from . import pcd_utils
        return pcd_utils.vox_idx_iou(vox_idx_1, vox_idx_2)

    def select_features_rand_v2(self, boundaries, N, per_instance=False):
        """Reference Fusion.select_features_rand_v2 (fusion.py:1539-1606): for every instance (every mask channel whose
        label differs from its predecessor's, or all of them with per_instance) and camera, erode the instance mask
        (15x15, valid depth only), farthest-point-sample N // num_cam of its PIXELS, lift them to world points with the
        camera's depth and pose, and query their descriptors.  Returns (src_feats_list, src_pts_list, img_list); img_list
        (cv2 keypoint renderings in the reference) is always empty.

        Per (instance, camera) the gate, the erosion, the row-major nonzero, the pixel FPS and the gather of the selected
        pixels' depths are one device pipeline (pcd_utils.masked_pixel_fps: the nonzero count and the k selected pixels are
        the only host traffic -- the count because fps_np seeds itself with np.random.randint(count), so the reference's
        random stream is reproduced draw for draw); the lift of those k pixels is float64 on the host exactly as the
        reference writes it (a 4x4 inverse and a [4,k] product), which keeps the keypoints bit-identical to its own."""
        from . import pcd_utils
        per_cam = N // self.num_cam
        obs = self.curr_obs_torch
        labels = obs["mask_label"][0]
        depth = obs["depth"].float()
        intrinsics = obs["K"].detach().cpu().numpy()
        cam_to_world = [np.linalg.inv(np.concatenate([p[:3], np.array([[0, 0, 0, 1]])], axis=0))
                        for p in obs["pose"].detach().cpu().numpy()]                       # fusion.py:1552, 1573
        feats_out, pts_out = [], []
        previous = labels[0]
        for inst in range(1, len(labels)):
            if labels[inst] == previous and not per_instance:
                continue
            world = []
            for cam in range(self.num_cam):
                pix, z = pcd_utils.masked_pixel_fps(obs["mask"][cam, :, :, inst], depth[cam], per_cam)   # fusion.py:1554-1568
                fx, fy, cx, cy = intrinsics[cam][0, 0], intrinsics[cam][1, 1], intrinsics[cam][0, 2], intrinsics[cam][1, 2]
                lifted = np.zeros([per_cam, 3])                                           # float64, like the reference
                lifted[:, 0] = (pix[:, 1] - cx) * z / fx
                lifted[:, 1] = (pix[:, 0] - cy) * z / fy
                lifted[:, 2] = z
                homog = np.concatenate([lifted, np.ones([per_cam, 1])], axis=-1).T
                world.append(np.matmul(cam_to_world[cam], homog)[:3].T)
            cloud = np.concatenate(world, axis=0)
            pts_out.append(cloud)
            feats_out.append(self.eval(torch.from_numpy(cloud).to(self.device, torch.float32))["dino_feats"])
            previous = labels[inst]
        return feats_out, pts_out, []

    # ---- masked point clouds of instances (reference fusion.py:1258-1311) -----------------------------------------------
    def get_inst_num(self):
        """fusion.py:1258-1260 (the background counts)."""
        return len(self.curr_obs_torch["consensus_mask_label"])

    def _masked_clouds(self, sel_mask, views, boundaries, downsample):
        """sel_mask [len(views),H,W] bool DEVICE tensor -> 2x2 cv2.erode per view (d3f_erode) -> masked back-projection to the
        world frame + boundary crop (d3f_backproject_view) per view, concatenated in view order: what the reference does with
        cv2 + aggr_point_cloud_from_data(..., masks=sel_mask, out_o3d=False) (fusion.py:1271-1278).  Masks, depth and the
        compaction stay on the device; the clouds come back as float64 numpy arrays like the reference's."""
        from . import pcd_utils
        lib = _lib.load()
        dev = sel_mask.device
        H, W = self.H, self.W
        obs = self.curr_obs_torch
        K = obs["K"].detach().cpu().numpy()
        pose = obs["pose"].detach().cpu().numpy()
        bounds = None if boundaries is None else [boundaries[k] for k in ("x_lower", "x_upper", "y_lower", "y_upper", "z_lower", "z_upper")]
        colors = np.asarray(obs["color"])
        gate = (sel_mask.to(torch.uint8) * 255).contiguous()
        pts_all, col_all = [], []
        for j, v in enumerate(views):
            eroded = torch.empty((H, W), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.d3f_erode(_lib.ptr(gate[j]), H, W, 2, 2, _lib.ptr(eroded), _lib.current_stream_handle(dev)))
            pose44 = np.concatenate([pose[v][:3], np.array([[0, 0, 0, 1]])], axis=0)          # fusion.py:1274-1276
            cam = [K[v][0, 0], K[v][1, 1], K[v][0, 2], K[v][1, 2]]
            pts, pix = pcd_utils._backproject(obs["depth"][v], eroded, cam, np.linalg.inv(pose44), bounds, dev)
            pts_np = pts.cpu().numpy()
            col_np = (colors[v] / 255.).reshape(-1, 3)[pix.cpu().numpy()]
            if downsample:
                pts_np, col_np = pcd_utils.voxel_downsample(pts_np, 0.01, col_np)               # draw_utils.py:396-400
            pts_all.append(pts_np)
            col_all.append(col_np)
        return np.concatenate(pts_all, axis=0), np.concatenate(col_all, axis=0)

    def extract_masked_pcd(self, inst_idx_ls, boundaries=None):
        """Reference fusion.py:1262-1279: the world-frame points of the instances `inst_idx_ls` (OR of their mask channels,
        2x2-eroded, valid depth) over all views; float64 [n,3] in the reference's order (views, then ascending pixels)."""
        mask = self.curr_obs_torch["mask"]
        sel = (mask[..., list(inst_idx_ls)] != 0).any(dim=-1) if len(inst_idx_ls) else torch.zeros(mask.shape[:3], dtype=torch.bool, device=mask.device)
        return self._masked_clouds(sel, range(self.num_cam), boundaries, False)[0]

    def extract_masked_pcd_in_views(self, inst_idx_ls, view_idx_ls, boundaries, downsample=True):
        """Reference fusion.py:1281-1299 (exactly one view, taken from the per-view Grounded-SAM masks 'mask_gs').
        downsample=True (the reference's default) is a 1-cm voxel grid mean like open3d's voxel_down_sample: the same SET
        of points as open3d's to rounding, in ascending voxel order (open3d's order is that of its hash map)."""
        assert len(view_idx_ls) == 1
        dev = torch.device(self.device)
        from .association import _device_tensor
        gs = [_device_tensor(self.curr_obs_torch["mask_gs"][v], dev) for v in view_idx_ls]      # [NI,H,W] each
        sel = torch.stack([(g[list(inst_idx_ls)] != 0).any(dim=0) for g in gs], dim=0)
        return self._masked_clouds(sel, list(view_idx_ls), boundaries, downsample)[0]

    # ---- multi-view instance association (reference fusion.py:801-1098; the work: d3fields_amd/association.py) ----------------
    def merge_instances_from_new_view_vox_ver(self, instances_info, i, boundaries):
        """Reference fusion.py:801-849: every detection of view i joins the best-overlapping instance of its label (voxel-index IoU
        > 0.2 on the 3-cm grid align_instance_mask_v3 set up) or founds a new one."""
        from . import association
        return association.merge_view(self, instances_info, i, boundaries)

    def del_partial_vox_idx(self, instance_info, vox_idx):
        """Reference fusion.py:860-868."""
        from . import association
        return association.drop_voxels(instance_info, vox_idx)

    def filter_instances_vox_ver(self, instances_info):
        """Reference fusion.py:978-1040: overlapping instances give their shared voxels to the one that saw them from more views
        (or with the higher mean confidence); 'table' and emptied instances are removed."""
        from . import association
        return association.filter_instances(self, instances_info)

    def reorder_instances(self, instances_info, query_texts):
        """Reference fusion.py:1042-1050: background first, then the instances in the order of the query texts."""
        from . import association
        return association.reorder(instances_info, query_texts)

    def swap_instance_mask(self, instances_info):
        """Reference fusion.py:1052-1063: curr_obs_torch['mask'] = (V,H,W) uint8 consensus label images (d3f_compose_labels)."""
        from . import association
        association.paint_label_images(self, instances_info)

    def align_instance_mask_v3(self, queries, boundaries, expected_labels=None):
        """Reference fusion.py:1065-1098: from the per-view detections in curr_obs_torch ('mask_gs', 'mask_label', 'mask_conf') to
        the consensus label images 'mask' and 'consensus_mask_label'."""
        from . import association
        association.align(self, queries, boundaries, expected_labels)

    def get_query_obj_pcd(self):
        """Reference fusion.py:1301-1311: the cloud of every non-background instance.  The reference returns an open3d
        PointCloud (aggr_point_cloud_from_data's default out_o3d=True); with open3d importable so does this, otherwise a
        pcd_utils.PointCloud carrying the same .points / .colors arrays."""
        from . import pcd_utils
        mask = self.curr_obs_torch["mask"]
        sel = mask[..., 1:].sum(dim=-1) > 0
        pts, col = self._masked_clouds(sel, range(self.num_cam), None, False)
        return pcd_utils.as_point_cloud(pts, col)

    # ---- instance masks: upstream producers (reference fusion.py:1112-1256) ----------------
    # The reference hard-wires Grounded-SAM and XMem (xmem_process, fusion.py:631-684).  They stay upstream models here and are
    # injected.  A mask_producer may return the finished consensus (below) or only Grounded-SAM's per-view detections
    # {'mask_gs', 'mask_label', 'mask_conf'} -- then align_instance_mask_v3 (fusion.py:1067-1098) runs here, as in the reference:
    #
    #   mask_producer(fusion, queries, thresholds, boundaries, merge_all=False, expected_labels=None, robot_pcd=None)
    #       -> dict with
    #          'mask'                 (V,H,W) uint8 consensus instance index per pixel, or (V,H,W,NI) one-hot
    #                                 (what align_instance_mask_v3 leaves in curr_obs_torch['mask'], fusion.py:1055-1065)
    #          'consensus_mask_label' list of NI label strings, entry 0 == 'background'            (fusion.py:1096)
    #          optional 'mask_label' (per view: list of label strings), 'mask_conf', 'mask_gs'      (fusion.py:1141-1143)
    #   mask_tracker(fusion, color (V,H,W,3) uint8, mask (V,H,W) uint8 tensor or None)
    #       -> (V,H,W,NI) one-hot / probabilities, or (V,H,W) uint8 labels: xmem_process's contract (fusion.py:631-684);
    #          mask is the consensus label image on the first frame and None on every later frame.
    def _store_detections(self, produced, queries, boundaries, expected_labels):
        """The producer returned what Grounded-SAM returns per view (utils/grounded_sam.py:404-442: masks [n_v,H,W] with the
        background first, labels, confidences): stored as the reference stores them (fusion.py:1141-1145) and associated across
        the views by align_instance_mask_v3, as in the reference (fusion.py:1170).  Returns the (V,H,W) uint8 label image."""
        gs, labels, confs = produced["mask_gs"], [list(l) for l in produced["mask_label"]], produced["mask_conf"]
        if not (len(gs) == len(labels) == len(confs) == self.num_cam):
            raise ValueError("'mask_gs', 'mask_label' and 'mask_conf' need one entry per view (%d)" % self.num_cam)
        for v in range(self.num_cam):
            shape = tuple(np.asarray(gs[v]).shape) if not isinstance(gs[v], torch.Tensor) else tuple(gs[v].shape)
            if len(shape) != 3 or shape[1:] != (self.H, self.W) or shape[0] != len(labels[v]) or shape[0] != len(confs[v]):
                raise ValueError("view %d: 'mask_gs' %s does not match %d labels / %d confidences at %dx%d"
                                 % (v, shape, len(labels[v]), len(confs[v]), self.H, self.W))
        self.curr_obs_torch["mask_gs"] = gs
        self.curr_obs_torch["mask_label"] = labels
        self.curr_obs_torch["mask_conf"] = confs
        _, first = np.unique(labels[0], return_index=True)                                       # fusion.py:1144-1145
        self.curr_obs_torch["semantic_label"] = list(np.array(labels[0])[np.sort(first)])
        self.align_instance_mask_v3(queries, boundaries, expected_labels)
        return self.curr_obs_torch["mask"]

    def _store_segmentation(self, produced, queries=None, boundaries=None, expected_labels=None):
        """Writes what text_queries_* write after Grounded-SAM + align_instance_mask_v3 (fusion.py:1141-1145, 1096):
        'mask_gs', 'mask_label' (list per view of label strings), 'mask_conf', 'semantic_label',
        'consensus_mask_label'.  Returns the (V,H,W) uint8 consensus label image."""
        raw = isinstance(produced, dict) and "mask" not in produced and all(k in produced for k in ("mask_gs", "mask_label", "mask_conf"))
        if raw:
            return self._store_detections(produced, queries, boundaries, expected_labels)
        if not isinstance(produced, dict) or "mask" not in produced or "consensus_mask_label" not in produced:
            raise TypeError("mask_producer must return a dict with 'mask' and 'consensus_mask_label' "
                            "(optionally 'mask_label', 'mask_conf', 'mask_gs') or, for the reference's own association, "
                            "only the per-view detections 'mask_gs', 'mask_label', 'mask_conf'; see INTEGRATION.md")
        consensus = [str(x) for x in produced["consensus_mask_label"]]
        NI = len(consensus)
        m = produced["mask"]
        if isinstance(m, np.ndarray):
            m = torch.from_numpy(m)
        m = m.to(self.device)
        if m.dim() == 4:                                           # one-hot / probabilities -> label image
            if m.shape[-1] != NI:
                raise ValueError("'mask' has %d channels but 'consensus_mask_label' names %d instances" % (m.shape[-1], NI))
            m = onehot2instance(m)
        if m.dim() != 3 or tuple(m.shape[1:]) != (self.H, self.W):
            raise ValueError("'mask' must be (V,%d,%d) labels or (V,%d,%d,NI) one-hot, got %s" % (self.H, self.W, self.H, self.W, tuple(m.shape)))
        label_img = m.to(torch.uint8).contiguous()
        if label_img.numel() and int(label_img.max().item()) >= NI:
            raise ValueError("'mask' holds instance index %d but 'consensus_mask_label' names only %d instances"
                             % (int(label_img.max().item()), NI))
        V = label_img.shape[0]
        labels = produced.get("mask_label")
        if labels is None:      # the reference's (disabled) assumption: every view saw every instance (fusion.py:1154-1166)
            labels = [list(consensus) for _ in range(V)]
        labels = [list(l) for l in labels]
        self.curr_obs_torch["mask_gs"] = produced.get("mask_gs")
        self.curr_obs_torch["mask_label"] = labels
        self.curr_obs_torch["mask_conf"] = produced.get("mask_conf", [[1.0] * len(l) for l in labels])
        _, first = np.unique(labels[0], return_index=True)                                       # fusion.py:1144-1145
        self.curr_obs_torch["semantic_label"] = list(np.array(labels[0])[np.sort(first)])
        self.curr_obs_torch["consensus_mask_label"] = consensus
        return label_img

    def _set_mask(self, onehot):
        self.curr_obs_torch["mask"] = onehot.to(device=self.device, dtype=self.dtype).contiguous()
        self._finite_cache.pop("mask", None)

    def _tracked_mask(self, label_img):
        """xmem_process stand-in (fusion.py:631-684): injected tracker -> one-hot (V,H,W,len(track_ids))."""
        out = self.mask_tracker(self, self.curr_obs_torch["color"], label_img)
        if isinstance(out, np.ndarray):
            out = torch.from_numpy(out)
        out = out.to(self.device)
        # validate against the ids this call WOULD install, commit the tracker state only once the output is accepted:
        # a tracker that fails on the first frame must leave the object in "no first mask yet" (the reference reaches
        # the tracking-only branch only after a successful first frame, fusion.py:1240)
        track_ids = list(range(len(self.curr_obs_torch["consensus_mask_label"]))) if label_img is not None else self.track_ids   # fusion.py:657
        if out.dim() == 3:
            out = instance2onehot(out.to(torch.uint8).contiguous(), len(track_ids))               # fusion.py:683
        if out.dim() != 4 or out.shape[-1] != len(track_ids):
            raise ValueError("mask_tracker must return (V,H,W,%d) one-hot or (V,H,W) labels, got %s" % (len(track_ids), tuple(out.shape)))
        if label_img is not None:
            self.xmem_first_mask_loaded = True                                                    # fusion.py:663-665
            self.track_ids = track_ids
        return out

    def text_queries_for_inst_mask_no_track(self, queries, thresholds, boundaries, merge_all=False, expected_labels=None,
                                            robot_pcd=None):
        """Reference fusion.py:1112-1171: segment every view, align the instances across views, store
        'mask_label' / 'mask_conf' / 'semantic_label' / 'consensus_mask_label' and 'mask' as a one-hot
        (V,H,W,len(consensus_mask_label)) tensor of Fusion.dtype.  The segmentation is injected; the
        association runs here (align_instance_mask_v3) when the producer returns raw per-view detections, else the producer's is taken."""
        if "color" not in self.curr_obs_torch:
            raise RuntimeError("Please call update() first!")
        if self.mask_producer is None:
            raise RuntimeError("no mask_producer was injected (Grounded-SAM is an upstream PyTorch-ROCm producer)")
        produced = self.mask_producer(self, queries, thresholds, boundaries, merge_all=merge_all, expected_labels=expected_labels,
                                      robot_pcd=robot_pcd)
        aligned_here = isinstance(produced, dict) and "mask" not in produced
        label_img = self._store_segmentation(produced, queries, boundaries, expected_labels)
        consensus = self.curr_obs_torch["consensus_mask_label"]
        if expected_labels is not None and consensus != expected_labels and not aligned_here:
            print("consensus mask label", consensus)                                              # fusion.py:1097-1098
        self._set_mask(instance2onehot(label_img, len(consensus)))                                # fusion.py:1171

    def text_queries_for_inst_mask(self, queries, thresholds, boundaries, use_sam=False, merge_all=False, expected_labels=None,
                                   robot_pcd=None):
        """Reference fusion.py:1173-1256: the first call segments + aligns (as _no_track) and initialises the tracker
        with the consensus label image; every later call only tracks (XMem, injected).  use_sam=True after the first
        frame raises NotImplementedError, as in the reference (fusion.py:1240-1241)."""
        if "color" not in self.curr_obs_torch:
            raise RuntimeError("Please call update() first!")        # the reference prints this and calls exit()
        if self.mask_tracker is None:
            raise RuntimeError("no mask_tracker was injected (XMem is an upstream PyTorch-ROCm producer); "
                               "use text_queries_for_inst_mask_no_track for single frames")
        if not self.xmem_first_mask_loaded:
            if self.mask_producer is None:
                raise RuntimeError("no mask_producer was injected (Grounded-SAM is an upstream PyTorch-ROCm producer)")
            label_img = self._store_segmentation(self.mask_producer(self, queries, thresholds, boundaries, merge_all=merge_all,
                                                                    expected_labels=expected_labels, robot_pcd=robot_pcd),
                                                 queries, boundaries, expected_labels)
            self._set_mask(self._tracked_mask(label_img))                                         # fusion.py:1237
        elif not use_sam:
            self._set_mask(self._tracked_mask(None))                                              # fusion.py:1239
        else:
            raise NotImplementedError

    def clear_xmem_memory(self):
        """fusion.py:1698-1702: the next text_queries_for_inst_mask call segments again (the injected tracker
        owns its own memory; give it a `clear_memory()` attribute to have it called here)."""
        if hasattr(self.mask_tracker, "clear_memory"):
            self.mask_tracker.clear_memory()
        self.xmem_first_mask_loaded = False

    def close(self):
        """Reference fusion.py:1704-1712 (drops the observation and the models): here the observation, the injected producers
        and every device buffer the object kept between calls (workspaces, cached point order, finite-check words)."""
        self.curr_obs_torch = {}
        self.feature_extractor = self.mask_producer = self.mask_tracker = None
        self._finite_cache, self._word_slot, self._words = {}, {}, None
        self._slot_key, self._next_word, self._ring_wrapped = [], -1, False       # (they describe the buffer that was just dropped)
        self._order_cache = self._tracker = None
        self._order_ws = self._lattice_cache = None     # ~20 bytes per point + an 8 MiB table per cached query / the lattice verdict
        self._last_ws, self._last_ws_n = None, 0
