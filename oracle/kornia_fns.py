"""Oracle: the seven kornia==0.7.0 functions the reference's pose path imports.

TEST INFRASTRUCTURE (see oracle/__init__.py).

The reference imports them at ``pose_optimization/two_view/estimate_relative_pose.py:2-4``;
kornia is a pinned third-party dependency (``requirements.txt:20``, kornia==0.7.0) whose
source is NOT under /root/reference and is not installed in the build image, so the
published algorithms are restated here (Hartley & Zisserman: normalised DLT, essential
decomposition, linear triangulation, symmetric epipolar distance).  **Parity unpinned at
this boundary**: nothing in the reference tests these functions.  The same module is what
``tests/golden/make_golden.py`` offers to the reference file as its ``kornia`` names when
it generates the w8pt golden vectors.

One deliberate choice: kornia 0.7.0 selects the cheirality winner with
``Rs[:, idx][:, 0, 0]`` (idx [B,1]); for B>1 that expression applies sample 0's choice to
every sample.  The reference only reaches this path at B=1 (``eval_pairs.py:248``,
``bundle_adjust_io.py:13-20``) where it equals the per-sample arg-max.  We implement the
per-sample arg-max (first maximum wins, as ``torch.max``), identical at every reference
call site.
"""
import torch

EPS = 1e-8


def svd(x):
    """``torch.svd`` convention: returns U, S, V with x = U diag(S) V^T."""
    U, S, Vh = torch.linalg.svd(x, full_matrices=False)
    return U, S, Vh.transpose(-2, -1)


def convert_points_from_homogeneous(p, eps=EPS):
    z = p[..., -1:]
    scale = torch.where(z.abs() > eps, 1.0 / (z + eps), torch.ones_like(z))
    return scale * p[..., :-1]


def convert_points_to_homogeneous(p):
    return torch.cat([p, torch.ones_like(p[..., :1])], dim=-1)


def transform_points(T, p):
    ph = convert_points_to_homogeneous(p)
    return convert_points_from_homogeneous(ph @ T.transpose(-2, -1))


def normalize_points(points, eps=EPS):
    """Hartley normalisation: zero mean, mean distance sqrt(2).  [B,N,2] -> ([B,N,2],[B,3,3])."""
    mean = points.mean(dim=1, keepdim=True)
    scale = (points - mean).norm(dim=-1, p=2).mean(dim=-1)
    scale = torch.sqrt(torch.tensor(2.0, dtype=points.dtype)) / (scale + eps)
    one, zero = torch.ones_like(scale), torch.zeros_like(scale)
    T = torch.stack(
        [scale, zero, -scale * mean[..., 0, 0], zero, scale, -scale * mean[..., 0, 1], zero, zero, one], dim=-1
    ).view(-1, 3, 3)
    return transform_points(T, points), T


def normalize_transformation(M, eps=EPS):
    nv = M[..., -1:, -1:]
    return torch.where(nv.abs() > eps, M / (nv + eps), M)


def decompose_essential_matrix(E):
    U, _, V = svd(E)
    Vt = V.transpose(-2, -1)
    mask = torch.ones_like(E)
    mask[..., -1:] *= -1.0
    U = torch.where((torch.det(U) < 0.0)[..., None, None], U * mask, U)
    Vt = torch.where((torch.det(Vt) < 0.0)[..., None, None], Vt * mask.transpose(-2, -1), Vt)
    W = E.new_tensor([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    R1 = U @ W @ Vt
    R2 = U @ W.transpose(-2, -1) @ Vt
    return R1, R2, U[..., -1:]


def motion_from_essential(E):
    R1, R2, t = decompose_essential_matrix(E)
    return torch.stack([R1, R1, R2, R2], dim=-3), torch.stack([t, -t, t, -t], dim=-3)


def triangulate_points(P1, P2, x1, x2):
    """Linear (DLT) triangulation; P [*,3,4], x [*,N,2] -> [*,N,3]."""
    shape = max(x1.shape, x2.shape)
    A = torch.zeros(shape[:-1] + (4, 4), dtype=x1.dtype)
    for i in range(4):
        A[..., 0, i] = x1[..., 0] * P1[..., 2:3, i] - P1[..., 0:1, i]
        A[..., 1, i] = x1[..., 1] * P1[..., 2:3, i] - P1[..., 1:2, i]
        A[..., 2, i] = x2[..., 0] * P2[..., 2:3, i] - P2[..., 0:1, i]
        A[..., 3, i] = x2[..., 1] * P2[..., 2:3, i] - P2[..., 1:2, i]
    _, _, V = svd(A)
    return convert_points_from_homogeneous(V[..., -1])


def depth_from_point(R, t, X):
    return (R @ X.transpose(-2, -1))[..., 2, :] + t[..., 2, :]


def projection_from_KRt(K, R, t):
    return K @ torch.cat([R, t], dim=-1)


def motion_from_essential_choose_solution(E, K1, K2, x1, x2, mask=None):
    """4 candidates, keep the one with most points in front of both cameras."""
    Rs, ts = motion_from_essential(E)
    B = E.shape[0]
    R1 = torch.eye(3, dtype=E.dtype)[None, None].expand(B, 4, -1, -1)
    t1 = torch.zeros(3, 1, dtype=E.dtype)[None, None].expand(B, 4, -1, -1)
    K1 = K1[:, None].expand(B, 4, -1, -1)
    K2 = K2[:, None].expand(B, 4, -1, -1)
    P1 = projection_from_KRt(K1, R1, t1)
    P2 = projection_from_KRt(K2, Rs, ts)
    X = triangulate_points(P1, P2, x1[:, None].expand(-1, 4, -1, -1), x2[:, None].expand(-1, 4, -1, -1))
    d1 = depth_from_point(R1, t1, X)
    d2 = depth_from_point(Rs, ts, X)
    ok = (d1 > 0.0) & (d2 > 0.0)
    if mask is not None:
        ok = ok & mask.unsqueeze(1)
    idx = torch.max(ok.sum(-1), dim=-1)[1]
    ar = torch.arange(B)
    return Rs[ar, idx], ts[ar, idx], X[ar, idx]


def symmetrical_epipolar_distance(p1, p2, F, squared=True, eps=EPS):
    if p1.shape[-1] == 2:
        p1 = convert_points_to_homogeneous(p1)
    if p2.shape[-1] == 2:
        p2 = convert_points_to_homogeneous(p2)
    l1in2 = p1 @ F.transpose(-2, -1)
    l2in1 = p2 @ F
    num = (p2 * l1in2).sum(dim=-1).pow(2)
    den_inv = 1.0 / l1in2[..., :2].norm(2, dim=-1).pow(2) + 1.0 / l2in1[..., :2].norm(2, dim=-1).pow(2)
    out = num * den_inv
    return out if squared else (out + eps).sqrt()
