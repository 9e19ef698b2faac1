"""Oracle: confidence-weighted eight-point relative pose (two views), torch-CPU.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates
``pose_optimization/two_view/estimate_relative_pose.py`` (function by function, cited
below) and ``pose_optimization/two_view/compute_pose_error.py``; the kornia calls go to
``oracle/kornia_fns.py``.  Pinned against the reference's own file (imported in the build
container by ``tests/golden/make_golden.py``) through ``tests/golden/w8pt_*.npz``.

Runs in the dtype of its inputs: fp32 = what the reference executes; fp64 = the "truth"
leg the HIP kernels (fp64 Gram / Jacobi) are held to more tightly.

Documented deviations from the reference (results identical where the reference works):
* E5  row scaling ``w[:, :, None] * X`` instead of ``diag_embed(w) @ X`` (no B x N x N).
* E7  ``compute_translation_error_as_angle(reduce=False)`` keeps the batch shape (invalid
      entries -> 0) instead of boolean-mask indexing that breaks ``choose_closest``.
"""
import torch

from . import kornia_fns as K


def normalize(kpts, intr):
    """estimate_relative_pose.py:9-14 - pixel -> normalised camera coordinates."""
    fx, fy = intr[..., 0, 0], intr[..., 1, 1]
    cx, cy = intr[..., 0, 2], intr[..., 1, 2]
    x = (kpts[..., 0] - cx.unsqueeze(-1)) / fx.unsqueeze(-1)
    y = (kpts[..., 1] - cy.unsqueeze(-1)) / fy.unsqueeze(-1)
    return torch.stack([x, y], dim=-1)


def get_kpts(data, result, id0, id1):
    """estimate_relative_pose.py:16-31 - fixed-shape gather; -1 wraps to the last keypoint."""
    if f"keypoints{id0}" in data:
        k0, k1 = data[f"keypoints{id0}"], data[f"keypoints{id1}"]
    else:
        k0, k1 = data[f"keypoints{id0}_{id0}_{id1}"], data[f"keypoints{id1}_{id0}_{id1}"]
    matches = result[f"matches{id0}_{id0}_{id1}"]
    bs, n0, _ = k0.shape
    conf = (matches >= 0).to(k0.dtype).unsqueeze(-1) * result[f"conf_scores_{id0}_{id1}"]
    bidx = torch.arange(bs).unsqueeze(-1).expand(bs, n0)
    return k0, k1[bidx, matches], data[f"intr{id0}"], data[f"intr{id1}"], conf


def design_matrix(p1n, p2n):
    """estimate_relative_pose.py:56-65 - rows [x2x1, x2y1, x2, y2x1, y2y1, y2, x1, y1, 1]."""
    x1, y1 = p1n[..., 0:1], p1n[..., 1:2]
    x2, y2 = p2n[..., 0:1], p2n[..., 1:2]
    return torch.cat([x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, torch.ones_like(x1)], dim=-1)


def find_fundamental(points1, points2, weights):
    """estimate_relative_pose.py:34-82 - weighted normalised 8-point, rank-2, /F22."""
    if points1.shape != points2.shape:
        raise AssertionError(points1.shape, points2.shape)
    if not (weights.dim() == 2 and weights.shape[1] == points1.shape[1]):
        raise AssertionError(weights.shape)
    p1n, T1 = K.normalize_points(points1)
    p2n, T2 = K.normalize_points(points2)
    X = weights.unsqueeze(-1) * design_matrix(p1n, p2n)  # weights multiply ROWS (E1)
    _, _, V = K.svd(X)
    F = V[..., -1].reshape(-1, 3, 3)
    U, S, V = K.svd(F)
    S = S * S.new_tensor([1.0, 1.0, 0.0])
    F = U @ (torch.diag_embed(S) @ V.transpose(-2, -1))
    F = T2.transpose(-2, -1) @ (F @ T1)
    return K.normalize_transformation(F)


def compute_rotation_error(T0, T1, reduce=True):
    """compute_pose_error.py:3-12 - geodesic angle of R0^T R1."""
    R = T0[..., :3, :3].transpose(-1, -2) @ T1[..., :3, :3]
    cos_a = (R.diagonal(dim1=-1, dim2=-2).sum(-1) - 1.0) / 2.0
    a = torch.arccos(cos_a.clamp(-1.0, 1.0)).abs()
    return a.mean() if reduce else a


def compute_translation_error_as_angle(T0, T1, reduce=True, keep_shape=False):
    """compute_pose_error.py:14-22 - angle between translation directions.

    Like the reference only the entries whose norm product exceeds 1e-6 count: reduce=True is their mean,
    reduce=False returns exactly those entries (shape [n_valid]).  keep_shape=True (not in the reference) returns
    [B] with 0 for the skipped entries - what the choose_closest loop below needs to stay batched.
    """
    t0, t1 = T0[..., :3, 3], T1[..., :3, 3]
    n = t0.norm(dim=-1) * t1.norm(dim=-1)
    valid = n > 1e-6
    cos = ((t0 * t1).sum(-1) / torch.where(valid, n, torch.ones_like(n))).clamp(-1.0, 1.0)
    err = torch.arccos(cos).abs()
    if keep_shape:
        return torch.where(valid, err, torch.zeros_like(err))
    return err[valid].mean() if reduce else err[valid]


def pose_from_Rt(R, t):
    T = torch.eye(4, dtype=R.dtype).unsqueeze(0).repeat(R.shape[0], 1, 1)
    T[:, :3, :3] = R
    T[:, :3, 3] = t.reshape(-1, 3)
    return T


def estimate_relative_pose_w8pt(kpts0, kpts1, intr0, intr1, confidence, choose_closest=False, T_021=None,
                                determine_inliers=False):
    """estimate_relative_pose.py:84-128."""
    if kpts0.shape[1] < 8:
        return None, None
    confidence = confidence / (confidence.sum(dim=1, keepdim=True) + 1e-6)
    k0n, k1n = normalize(kpts0, intr0), normalize(kpts1, intr1)
    bs = intr0.shape[0]
    eye = torch.eye(3, dtype=kpts0.dtype).unsqueeze(0)
    w = confidence.squeeze(-1) if confidence.dim() == 3 else confidence
    Fs = find_fundamental(k0n, k1n, w)
    if choose_closest:
        Rs, ts = K.motion_from_essential(Fs)
        best = torch.full((bs,), 1e6, dtype=kpts0.dtype)
        T = torch.eye(4, dtype=kpts0.dtype).unsqueeze(0).repeat(bs, 1, 1)
        for c in range(4):
            cand = pose_from_Rt(Rs[:, c], ts[:, c])
            err = compute_rotation_error(cand, T_021, reduce=False) + \
                compute_translation_error_as_angle(cand, T_021, keep_shape=True)
            upd = err < best
            best = torch.where(upd, err, best)
            T = torch.where(upd[:, None, None], cand, T)
    else:
        R, t, _ = K.motion_from_essential_choose_solution(Fs, eye, eye, k0n, k1n, mask=None)
        T = pose_from_Rt(R, t)
    P0 = torch.eye(4, dtype=kpts0.dtype).unsqueeze(0).repeat(bs, 1, 1)[:, :3, :]
    X = K.triangulate_points(P0, T[:, :3, :], k0n, k1n)
    depth0 = X[..., -1]
    depth1 = K.depth_from_point(T[:, :3, :3], T[:, :3, 3:], X)
    pos_depth = (depth0 > 0.0) & (depth1 > 0.0)
    inliers = None
    extra = {}
    if determine_inliers:
        err = K.symmetrical_epipolar_distance(k0n, k1n, Fs).sqrt()
        thr = 3.0 / ((intr0[:, 0, 0] + intr0[:, 1, 1] + intr1[:, 0, 0] + intr1[:, 1, 1]) / 4.0)
        inliers = pos_depth & (err <= thr.unsqueeze(-1))
        extra = {"epi_err": err, "epi_thr": thr.unsqueeze(-1)}  # for the tests' decision margins, not in the reference
    info = {"kpts0_norm": k0n, "kpts1_norm": k1n, "confidence": confidence, "inliers": inliers,
            "pos_depth_mask": pos_depth, "F": Fs, "depth0": depth0, "depth1": depth1, **extra}
    return T, info


def run_weighted_8_point(data, result, id0, id1, choose_closest=False, target_T_021=None):
    """estimate_relative_pose.py:130-136."""
    key = f"matches{id0}_{id0}_{id1}"
    if key in result and result[key].shape[1] != 0:
        k0, k1, i0, i1, conf = get_kpts(data, result, id0, id1)
        return estimate_relative_pose_w8pt(k0, k1, i0, i1, conf, choose_closest=choose_closest, T_021=target_T_021)
    return None, None
