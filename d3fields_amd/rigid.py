"""Per-instance rigid tracking on the HIP field query (reference: Fusion.rigid_tracking, fusion.py:1608-1685).

The reference optimises one SE(3) transform per instance with 100 Adam steps; every step is one
`Fusion.eval` of the transformed keypoints with autograd back to the 6 pose parameters.  Here the
query and its gradient are the HIP kernels (`d3f_eval`, `d3f_eval_backward`); the few 3x3 / [I,3]
tensor ops around them (exponential map, loss terms, Adam) are torch device ops.  An iteration is
~25 small launches, i.e. launch-bound, so the whole iteration is captured once in a HIP graph and
replayed (`use_graph=True`): same kernels, same order, one host call per iteration.

pytorch3d (reference env pins 0.7.5, env.yaml:14) is neither available on ROCm images nor needed:
the two functions the reference calls are a dozen lines each and are restated below.
"""
import torch

__all__ = ["so3_exp_map", "rigid_transform", "track_rigid", "RigidTracker"]

LR, ITERS, REG_W, DIST_W = 0.01, 100, 1.0, 100.0       # fusion.py:1613-1617


def so3_exp_map(log_rot, eps=1e-4):
    """[I,3] axis-angle -> [I,3,3] rotation, Rodrigues' formula with the angle clamped at sqrt(eps)
    (what pytorch3d.transforms.so3.so3_exp_map computes; fusion.py:1649)."""
    x, y, z = log_rot[:, 0], log_rot[:, 1], log_rot[:, 2]
    o = torch.zeros_like(x)
    skew = torch.stack([o, -z, y, z, o, -x, -y, x, o], dim=1).view(-1, 3, 3)
    theta = torch.clamp((log_rot * log_rot).sum(1), eps).sqrt()
    a = (theta.sin() / theta)[:, None, None]
    b = ((1.0 - theta.cos()) / (theta * theta))[:, None, None]
    eye = torch.eye(3, dtype=log_rot.dtype, device=log_rot.device)[None]
    return a * skew + b * torch.bmm(skew, skew) + eye


def rigid_transform(points, rot, trans):
    """pytorch3d's Transform3d().rotate(R).translate(t).transform_points(p) (fusion.py:1650-1651):
    row-vector convention, p' = p @ R + t per instance.  points [I,n,3], rot [I,3,3], trans [I,3]."""
    return torch.bmm(points, rot) + trans[:, None, :]


def _iteration(fusion, last, src_feats, t_params, log_r, opt):
    cur = rigid_transform(last, so3_exp_map(log_r), t_params).reshape(-1, 3)
    out = fusion.eval(cur, return_names=["dino_feats"])
    valid = out["valid_mask"]
    feat_loss = (torch.norm(out["dino_feats"] - src_feats, dim=-1) * valid).mean()
    dist_loss = DIST_W * torch.clamp(out["dist"] * valid, min=0).mean()
    reg_loss = REG_W * (torch.norm(t_params) + torch.norm(log_r))
    loss = feat_loss + dist_loss + reg_loss
    opt.zero_grad(set_to_none=False)
    loss.backward()
    opt.step()
    return cur, loss


def track_rigid(fusion, src_feats, last_match_pts, use_graph=True, iters=ITERS, lr=LR):
    """src_feats [I*n,C] and last_match_pts [I,n,3] on the device -> (current keypoints [I*n,3] as evaluated in
    the last iteration -- what the reference returns --, last loss)."""
    dev = last_match_pts.device
    num_inst = last_match_pts.shape[0]
    t_params = torch.zeros(num_inst, 3, device=dev, requires_grad=True)
    log_r = torch.zeros(num_inst, 3, device=dev, requires_grad=True)
    opt = torch.optim.Adam([t_params, log_r], lr=lr, betas=(0.9, 0.999), capturable=bool(use_graph))
    if not use_graph:
        cur = loss = None
        for _ in range(iters):
            cur, loss = _iteration(fusion, last_match_pts, src_feats, t_params, log_r, opt)
        return cur.detach(), loss.detach()

    # warm-up on a side stream (allocator pools, Adam state, the shim's caches), then rewind to the initial state
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(2):
            _iteration(fusion, last_match_pts, src_feats, t_params, log_r, opt)
    torch.cuda.current_stream(dev).wait_stream(side)
    with torch.no_grad():
        t_params.zero_()
        log_r.zero_()
        for st in opt.state.values():
            for v in st.values():
                if isinstance(v, torch.Tensor):
                    v.zero_()
    graph = torch.cuda.CUDAGraph()          # a hipGraph on ROCm
    with torch.cuda.graph(graph):
        cur, loss = _iteration(fusion, last_match_pts, src_feats, t_params, log_r, opt)
    for _ in range(iters):
        graph.replay()
    return cur.detach().clone(), loss.detach().clone()


class RigidTracker:
    """The tracking iteration captured ONCE and replayed for every frame of a sequence.

    track_rigid(use_graph=True) warms up and captures per call (~15 of its ~39 ms per frame).  A sequence keeps its shapes
    -- instances, keypoints, views, map sizes -- so this object owns static device buffers (a private observation the
    captured kernels point at, the keypoints, the source descriptors, the pose parameters and Adam's state); per frame
    it copies the new observation and inputs into them, rewinds parameters and optimiser state to the reference's
    initial values and replays the graph `iters` times.  The captured query runs without D3F_FLAG_FINITE_MAPS
    (the flag would be baked into the graph; results are identical either way).

    fused=True (default): the step is closed-form HIP (csrc/track_kernels.hip: exponential map, transform, query,
    loss gradients, backward of the query, chain rule and Adam) instead of torch autograd's ~90 launches -- as ONE launch
    per step (d3f_track_step: a wave per keypoint, the last wave to finish steps Adam; single_launch=True, the default
    whenever the descriptor map is fp32 with <= 512 channels and <= 8 views; with loop_launch=True and at most 512 keypoints
    ALL steps of a frame are one launch, d3f_track_run, whose waves wait for each step's update inside the kernel) or as five (d3f_rigid_transform, d3f_eval,
    d3f_track_loss_grad, d3f_eval_backward, d3f_rigid_update).  fused=False replays the autograd step."""

    def __init__(self, fusion, num_inst, n, iters=ITERS, lr=LR, fused=True, whole_loop=True, single_launch=True, loop_launch=True):
        from .fusion import Fusion
        dev = torch.device(fusion.device)
        obs = fusion.curr_obs_torch
        self.key = self.signature(fusion, num_inst, n)
        self.iters, self.lr, self.fused = iters, lr, fused
        self.single = False
        self.loop = False
        self.single_requested = bool(single_launch)
        self.loop_requested = bool(loop_launch)
        # whole_loop: ALL `iters` steps are captured into one HIP graph (5 x iters kernel nodes, one launch per frame)
        # instead of one step replayed `iters` times (a graph launch per step)
        self.whole_loop = bool(whole_loop) and fused
        self.num_inst, self.n = num_inst, n
        self.shadow = Fusion(num_cam=fusion.num_cam, device=str(dev), dtype=fusion.dtype)
        self.shadow.H, self.shadow.W, self.shadow.mu = fusion.H, fusion.W, fusion.mu
        self.shadow.curr_obs_torch = {k: torch.empty_like(obs[k]) for k in ("depth", "K", "pose", "dino_feats")}
        self.shadow._finite_override = False
        C = obs["dino_feats"].shape[3]
        self.last = torch.empty(num_inst, n, 3, device=dev)
        self.src = torch.empty(num_inst * n, C, device=dev)
        self.t_params = torch.zeros(num_inst, 3, device=dev, requires_grad=not fused)
        self.log_r = torch.zeros(num_inst, 3, device=dev, requires_grad=not fused)
        if fused:
            self.state = torch.zeros(num_inst * 13 + 4, device=dev)      # Adam m [I,6], v [I,6], step [I], norms [2], loss [2]
            self.pts = torch.empty(num_inst * n, 3, device=dev)
            self.grad_feats = torch.empty(num_inst * n, C, device=dev)
            self.grad_dist = torch.empty(num_inst * n, device=dev)
            self.opt = None
            feats = obs["dino_feats"]
            # (the same layout conditions d3f_track_step checks: a view with odd strides or an unaligned base takes the
            # five-launch step instead of failing inside the graph capture)
            self.single = (bool(single_launch) and feats.dtype == torch.float32 and C % 4 == 0 and C <= 512 and fusion.num_cam <= 8 and
                           feats.stride(3) == 1 and all(feats.stride(d) % 4 == 0 for d in (0, 1, 2)) and feats.data_ptr() % 16 == 0)
            if self.single:
                from . import _lib
                self.loss3 = torch.zeros(3, device=dev)
                self.scratch = torch.zeros(_lib.load().d3f_track_step_scratch_bytes(num_inst, n) // 4, device=dev)
                # loop_launch: all `iters` steps in ONE launch (d3f_track_run; the steps wait for one another inside the
                # kernel, so every keypoint's wave must be resident)
                self.loop = bool(loop_launch) and self.whole_loop and num_inst * n <= _lib.load().d3f_track_run_max_keypoints() and num_inst <= 16
        else:
            self.opt = torch.optim.Adam([self.t_params, self.log_r], lr=lr, betas=(0.9, 0.999), capturable=True)
        self.graph = None
        self.cur = self.loss = None
        self.loop_fallbacks = 0                  # frames d3f_track_run gave up on (NaN loss) and the per-step launches repeated

    @staticmethod
    def signature(fusion, num_inst, n):
        o = fusion.curr_obs_torch
        return (num_inst, n, float(fusion.mu), fusion.H, fusion.W) + tuple(
            (tuple(o[k].shape), o[k].dtype) for k in ("depth", "K", "pose", "dino_feats"))

    def _rewind(self):
        with torch.no_grad():
            self.t_params.zero_()
            self.log_r.zero_()
            if self.fused:
                self.state.zero_()
                if self.single:
                    self.scratch.zero_()
            else:
                for st in self.opt.state.values():
                    for v in st.values():
                        if isinstance(v, torch.Tensor):
                            v.zero_()

    def _fused_iteration(self, iters=1):
        """transform -> d3f_eval -> loss gradients -> d3f_eval_backward -> chain rule + Adam: five launches, no autograd
        (single: one launch per step, d3f_track_step, or one for `iters` steps, d3f_track_run)."""
        from . import _lib
        import ctypes
        lib, dev = _lib.load(), self.last.device
        I, n, N = self.num_inst, self.n, self.num_inst * self.n
        st = self.state
        m, v, step, norms, loss = st[:I * 6], st[I * 6:I * 12], st[I * 12:I * 13], st[I * 13:I * 13 + 2], st[I * 13 + 2:I * 13 + 4]
        if self.single:
            with torch.cuda.device(dev), torch.no_grad():
                stream = _lib.current_stream_handle(dev)
                views, keep, V = self.shadow._views(dev)
                fm = self.shadow.curr_obs_torch["dino_feats"]
                cm = _lib.ChannelMap(fm.data_ptr(), fm.shape[1], fm.shape[2], fm.shape[3], _lib.DTYPE_F32, fm.stride(0), fm.stride(1), fm.stride(2))
                state = _lib.TrackState(_lib.ptr(self.t_params), _lib.ptr(self.log_r), _lib.ptr(m), _lib.ptr(v), _lib.ptr(step),
                                        _lib.ptr(self.pts), _lib.ptr(self.loss3), _lib.ptr(self.scratch))
                if iters == 1:
                    _lib.check(lib.d3f_track_step(ctypes.byref(views), ctypes.byref(cm), _lib.ptr(self.last), I, n, _lib.ptr(self.src),
                                                  float(self.shadow.mu), DIST_W, REG_W, self.lr, 0.9, 0.999, 1e-8, ctypes.byref(state), stream))
                else:
                    _lib.check(lib.d3f_track_run(ctypes.byref(views), ctypes.byref(cm), _lib.ptr(self.last), I, n, _lib.ptr(self.src),
                                                 float(self.shadow.mu), DIST_W, REG_W, self.lr, 0.9, 0.999, 1e-8, int(iters),
                                                 ctypes.byref(state), stream))
            return self.pts, self.loss3.sum()
        assert iters == 1
        with torch.cuda.device(dev), torch.no_grad():
            stream = _lib.current_stream_handle(dev)
            _lib.check(lib.d3f_rigid_transform(_lib.ptr(self.last), I, n, _lib.ptr(self.t_params), _lib.ptr(self.log_r),
                                               _lib.ptr(self.pts), _lib.ptr(norms), stream))
            out, saved = self.shadow._launch(self.pts, ["dino_feats"], False, "eval")
            _lib.check(lib.d3f_track_loss_grad(_lib.ptr(out["dino_feats"]), _lib.ptr(self.src), _lib.ptr(out["dist"]),
                                               _lib.ptr(out["valid_mask"]), N, self.src.shape[1], DIST_W, _lib.ptr(self.grad_feats),
                                               _lib.ptr(self.grad_dist), _lib.ptr(loss), stream))
            grad_pts = self.shadow._backward(saved, self.grad_dist, [self.grad_feats])
            _lib.check(lib.d3f_rigid_update(_lib.ptr(self.last), I, n, _lib.ptr(grad_pts), _lib.ptr(self.t_params), _lib.ptr(self.log_r),
                                            _lib.ptr(m), _lib.ptr(v), _lib.ptr(step), _lib.ptr(norms), REG_W, self.lr, 0.9, 0.999, 1e-8,
                                            stream))
        # total loss of this step as the reference forms it: feature + distance + regulariser (norms are pre-update)
        return self.pts, loss[0] + loss[1] + REG_W * (norms[0] + norms[1])

    def _step(self, iters=1):
        if self.fused:
            return self._fused_iteration(iters)
        assert iters == 1
        return _iteration(self.shadow, self.last, self.src, self.t_params, self.log_r, self.opt)

    def run(self, fusion, src_feats, last_match_pts):
        dev = self.last.device
        with torch.no_grad():
            for k, t in self.shadow.curr_obs_torch.items():
                t.copy_(fusion.curr_obs_torch[k])
            self.last.copy_(last_match_pts)
            self.src.copy_(src_feats)
        if self.graph is None:
            self._rewind()                       # (a fallback after a failed d3f_track_run: its scratch words are stale)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(2):
                    self._step()
            torch.cuda.current_stream(dev).wait_stream(side)
            self._rewind()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                if self.loop:
                    self.cur, self.loss = self._step(self.iters)
                else:
                    for _ in range(self.iters if self.whole_loop else 1):
                        self.cur, self.loss = self._step()
        self._rewind()
        for _ in range(1 if self.whole_loop else self.iters):
            self.graph.replay()
        cur, loss = self.cur.detach().clone(), self.loss.detach().clone()
        if self.loop:
            # d3f_track_run's waves wait for one another INSIDE the kernel; with the device held by other work for seconds
            # the bounded wait gives up, poisons the loss with NaN (the poses are then undefined) and leaves a sentinel word in
            # the scratch.  One host sync per frame -- the caller (Fusion.rigid_tracking) copies the keypoints to the host right
            # away anyway -- reads that word: a stall repeats the frame with one launch per step, which cannot stall, and the
            # tracker stays on that form; a NaN that came out of the DATA (no sentinel) is the caller's result as it is.
            from . import _lib
            I, n = self.last.shape[0], self.last.shape[1]
            word = int(_lib.load().d3f_track_stall_word(I, n))
            if int(self.scratch.view(torch.int32)[word].item()) == _lib.TRACK_STALL_SENTINEL:
                self.loop = False
                self.graph = None
                self.loop_fallbacks += 1
                return self.run(fusion, src_feats, last_match_pts)
        return cur, loss
