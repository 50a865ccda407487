"""TEST INFRASTRUCTURE ONLY -- torch-ops CPU port of the reference's field query.

The reference's "CPU path" *is* a sequence of ATen ops (bmm -> grid_sample(nearest) ->
elementwise -> grid_sample(bilinear) -> broadcast multiplies -> sum(0), in 60 000-point
chunks; fusion.py:32-77, 305-394, 526-545).  On the GPU box the reference itself is absent,
so bench.py's `cpu_baseline` leg times this port of the same op sequence on the host cores
(kind = "port"), and tests use it for autograd gradients.  Validated against the imported
reference in tests/test_oracle_golden.py::test_torch_port_*.  Never imported by d3fields_amd.
"""
import torch
import torch.nn.functional as F

CHUNK = 60000          # fusion.py:527


def _pixel_coords(pts, pose, K):
    """fusion.py:32-55: returns uv [V,N,2], ok [V,N], z [V,N]."""
    n = pts.shape[0]
    homog = torch.cat((pts, pts.new_ones(n, 1)), dim=1)                      # [N,4]
    bottom = pts.new_zeros(pose.shape[0], 1, 4)
    bottom[..., 3] = 1.0
    T = torch.cat((K @ pose, bottom), dim=1)                                 # [V,4,4]
    cam = (T.unsqueeze(1) @ homog.view(1, n, 4, 1)).squeeze(-1)[..., :3]     # [V,N,3]
    z = cam[..., 2:3]
    degenerate = z.abs() < 1e-4
    z[degenerate] = 1e-3
    return cam[..., :2] / z, ~degenerate[..., 0], z[..., 0]


def _sample(chw_maps, uv, H, W, mode):
    """fusion.py:57-77 with align_corners=True and zeros padding -> [V,N,C]."""
    gx = uv[..., 0] / (W - 1) * 2 - 1
    gy = uv[..., 1] / (H - 1) * 2 - 1
    grid = torch.stack((gx, gy), dim=-1).unsqueeze(1)                        # [V,1,N,2]
    got = F.grid_sample(chw_maps, grid, mode=mode, padding_mode="zeros", align_corners=True)
    return got.squeeze(2).transpose(1, 2)


def field_query(obs, pts, names, H, W, mu=0.02, keep_inter=False):
    """One un-chunked Fusion.eval (fusion.py:305-394) in torch ops."""
    uv, ok, z = _pixel_coords(pts, obs["pose"], obs["K"])
    seen = _sample(obs["depth"].unsqueeze(1), uv, H, W, "nearest")[..., 0]
    sd = seen - z
    live = (seen > 0.0) & ok & (sd > -mu)
    weight = torch.exp(torch.clamp(mu - sd.abs(), max=0) / mu)
    livef = live.float()
    count = livef.sum(0)
    dist = (sd.clamp(-mu, mu) * livef).sum(0) / (count + 1e-6)
    empty = count == 0
    dist[empty] = 1e3
    out = {"dist": dist, "valid_mask": ~empty}
    for k in names:
        per_view = _sample(obs[k].permute(0, 3, 1, 2), uv, H, W, "bilinear")
        fused = (per_view * livef.unsqueeze(-1) * weight.unsqueeze(-1)).sum(0) / (count.unsqueeze(-1) + 1e-6)
        fused[empty] = 0.0
        out[k] = fused
        if keep_inter:
            out[k + "_inter"] = per_view
    return out


def dist_query(obs, pts, H, W):
    """Fusion.eval_dist (fusion.py:396-436) in torch ops."""
    uv, ok, z = _pixel_coords(pts, obs["pose"], obs["K"])
    seen = _sample(obs["depth"].unsqueeze(1), uv, H, W, "nearest")[..., 0]
    livef = ((seen > 0.0) & ok).float()
    count = livef.sum(0)
    return {"dist": ((seen - z) * livef).sum(0) / (count + 1e-6), "valid_mask": count != 0}


def batched_field_query(obs, pts, names, H, W, mu=0.02, chunk=CHUNK):
    """Fusion.batch_eval (fusion.py:526-545): chunk loop + cat."""
    pieces = [field_query(obs, pts[i:i + chunk], names, H, W, mu) for i in range(0, pts.shape[0], chunk)]
    return {k: torch.cat([p[k] for p in pieces], dim=0) for k in pieces[0]} if pieces else {}


def rigid_tracking(obs, H, W, src_feats, last_match_pts, mu=0.02, iters=100, lr=0.01, reg_w=1.0, dist_w=100.0):
    """Fusion.rigid_tracking (fusion.py:1608-1685) in torch ops on the CPU: per-instance translation + axis-angle,
    Adam, loss = masked descriptor distance + dist_w * positive distance + parameter norms.
    last_match_pts [I,n,3] -> keypoints [I*n,3] as evaluated in the last iteration (what the reference returns).
    pytorch3d's so3_exp_map / Transform3d as restated in oracle/pytorch3d_restated.py."""
    from oracle.pytorch3d_restated import so3_exp_map
    num_inst = last_match_pts.shape[0]
    t_params = torch.zeros(num_inst, 3, requires_grad=True)
    log_r = torch.zeros(num_inst, 3, requires_grad=True)
    opt = torch.optim.Adam([t_params, log_r], lr=lr, betas=(0.9, 0.999))
    cur = None
    for _ in range(iters):
        rot = so3_exp_map(log_r)
        cur = (torch.bmm(last_match_pts, rot) + t_params[:, None, :]).reshape(-1, 3)
        out = field_query(obs, cur, ["dino_feats"], H, W, mu)
        live = out["valid_mask"]
        loss = ((torch.norm(out["dino_feats"] - src_feats, dim=-1) * live).mean()
                + dist_w * torch.clamp(out["dist"] * live, min=0).mean()
                + reg_w * (torch.norm(t_params) + torch.norm(log_r)))
        opt.zero_grad()
        loss.backward()
        opt.step()
    return cur.detach()
