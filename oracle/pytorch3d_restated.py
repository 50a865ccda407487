"""TEST INFRASTRUCTURE ONLY -- the two pytorch3d entry points Fusion.rigid_tracking uses
(fusion.py:1627-1628, 1649-1651), restated so that the REFERENCE's tracking loop can run in this
container and write a golden fixture (oracle/gen_golden.py: rigid_tracking case).

Third-party dependency absent from /root/reference and from this image: pytorch3d, pinned to 0.7.5 by the
reference (env.yaml:14).  Published algorithm of that version, restated from its documentation:

* ``so3_exp_map(log_rot, eps=1e-4)`` (pytorch3d/transforms/so3.py): Rodrigues' formula
      theta = sqrt(clamp(|w|^2, min=eps));  K = hat(w)
      R = I + sin(theta)/theta * K + (1 - cos(theta))/theta^2 * K@K
  with hat(w) = [[0,-z,y],[z,0,-x],[-y,x,0]].
* ``Transform3d().rotate(R).translate(t).transform_points(p)`` (pytorch3d/transforms/transform3d.py):
  ROW-vector convention -- 4x4 matrices act from the right, rotate puts R in the upper-left 3x3 block,
  translate puts t in the last ROW, composition multiplies in call order, and the result is divided by the
  homogeneous coordinate:  p' = ([p, 1] @ (M_rot @ M_trans))[:3] / w   (= p @ R + t, w = 1).

No golden vectors of pytorch3d itself are available offline: this restatement is pinned only by the
analytic properties checked in tests/test_oracle_golden.py (orthonormality, small-angle limit, agreement
with scipy's Rotation.from_rotvec; the row-vector convention of transform_points).
"""
import torch


def hat(v):
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    o = torch.zeros_like(x)
    return torch.stack([torch.stack([o, -z, y], 1), torch.stack([z, o, -x], 1), torch.stack([-y, x, o], 1)], 1)


def so3_exp_map(log_rot, eps=0.0001):
    nrms = (log_rot * log_rot).sum(1)
    ang = torch.clamp(nrms, eps).sqrt()
    inv = 1.0 / ang
    fac1 = inv * ang.sin()
    fac2 = inv * inv * (1.0 - ang.cos())
    K = hat(log_rot)
    KK = torch.bmm(K, K)
    eye = torch.eye(3, dtype=log_rot.dtype, device=log_rot.device)[None]
    return fac1[:, None, None] * K + fac2[:, None, None] * KK + eye


class Transform3d:
    def __init__(self, dtype=torch.float32, device="cpu", matrix=None):
        self.dtype, self.device = dtype, device
        self._m = torch.eye(4, dtype=dtype, device=device)[None] if matrix is None else matrix

    def _compose(self, m):
        return Transform3d(self.dtype, self.device, torch.matmul(self._m, m))

    def rotate(self, R):
        R = R.to(device=self.device, dtype=self.dtype)
        m = torch.eye(4, dtype=self.dtype, device=self.device).repeat(R.shape[0], 1, 1)
        m = m.clone()
        m[:, :3, :3] = R
        return self._compose(m)

    def translate(self, t):
        t = t.to(device=self.device, dtype=self.dtype)
        m = torch.eye(4, dtype=self.dtype, device=self.device).repeat(t.shape[0], 1, 1)
        m = m.clone()
        m[:, 3, :3] = t
        return self._compose(m)

    def get_matrix(self):
        return self._m

    def transform_points(self, points):
        ones = torch.ones(points.shape[:-1] + (1,), dtype=points.dtype, device=points.device)
        out = torch.matmul(torch.cat([points, ones], -1), self._m)
        return out[..., :3] / out[..., 3:]
