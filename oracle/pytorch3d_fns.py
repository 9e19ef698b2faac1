"""Oracle: the two pytorch3d==0.7.5 functions the reference's two-view bundle adjustment uses.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference calls ``pytorch3d.transforms.so3.hat`` and
``pytorch3d.transforms.se3_exp_map`` at ``pose_optimization/two_view/bundle_adjust_gauss_newton_2_view.py:64,193``;
pytorch3d is a pinned third-party dependency (``requirements.txt:36``) that is neither on disk nor installed, so the
published algorithm is restated (Rodrigues formula and the SE(3) V-matrix, with pytorch3d's ``eps = 1e-4`` clamp of the
squared rotation angle).  **Parity unpinned at this boundary.**  ``se3_exp_map`` returns pytorch3d's row-vector
convention (the transposed 4x4), which is why the reference applies ``.permute(0, 2, 1)`` to it.
"""
import torch


def hat(v):
    x, y, z = v[..., 0], v[..., 1], v[..., 2]
    o = torch.zeros_like(x)
    return torch.stack([o, -z, y, z, o, -x, -y, x, o], dim=-1).reshape(v.shape[:-1] + (3, 3))


def se3_exp_map(log_transform, eps=1e-4):
    """log_transform [N,6] = (translation part v, rotation part w)."""
    v, w = log_transform[:, :3], log_transform[:, 3:]
    nrms = (w * w).sum(1)
    th = torch.clamp(nrms, eps).sqrt()
    K = hat(w)
    K2 = K @ K
    eye = torch.eye(3, dtype=w.dtype)[None]
    R = (th.sin() / th)[:, None, None] * K + ((1 - th.cos()) / th ** 2)[:, None, None] * K2 + eye
    V = eye + ((1 - th.cos()) / th ** 2)[:, None, None] * K + ((th - th.sin()) / th ** 3)[:, None, None] * K2
    T = (V @ v[:, :, None])[:, :, 0]
    out = torch.zeros(w.shape[0], 4, 4, dtype=w.dtype)
    out[:, :3, :3] = R
    out[:, :3, 3] = T
    out[:, 3, 3] = 1.0
    return out.permute(0, 2, 1)
