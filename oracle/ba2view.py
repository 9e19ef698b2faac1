"""Oracle: two-view Levenberg-Marquardt bundle adjustment (SURVEY.md 8(f) "next" row 1).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates
``pose_optimization/two_view/bundle_adjust_gauss_newton_2_view.py`` (``Observations`` :10-48, ``fill_J`` :50-67,
``compute_A_b`` :69-99, ``BundleAdjustGaussNewton2View.run`` :127-201) and ``run_bundle_adjust_2_view``
(``estimate_relative_pose.py:138-144``) sample by sample with the same dense normal equations: camera 0 fixed at the
identity, 6 pose + 3M point unknowns, Jacobi-preconditioned damped system ``D^-1 J^T J + lambda I`` solved densely, update
ALWAYS applied, best-residual pose kept, lambda /= 3.5 on improvement else *= 1.5, n_iterations + 1 residual
evaluations.  Pinned by tests/golden/ba2view_reference.npz (the reference's own class run in the build container).
"""
import torch

from . import kornia_fns as K
from .pytorch3d_fns import hat, se3_exp_map


def _normal_equations(extr1, pts, x0, x1, c):
    """One sample.  extr1 [4,4]; pts [M,3]; x0,x1 [M,2]; c [M] (normalised weights).  Dense J^T J, -J^T r, |r|^2."""
    M = pts.shape[0]
    dt = pts.dtype
    eye3 = torch.eye(3, dtype=dt)

    def proj(Ap):
        J = torch.zeros(M, 2, 3, dtype=dt)
        J[:, 0, 0] = 1.0 / Ap[:, 2]
        J[:, 0, 2] = -Ap[:, 0] / Ap[:, 2] ** 2
        J[:, 1, 1] = 1.0 / Ap[:, 2]
        J[:, 1, 2] = -Ap[:, 1] / Ap[:, 2] ** 2
        return Ap[:, :2] / Ap[:, 2:3], J

    Ap0 = pts
    Ap1 = pts @ extr1[:3, :3].T + extr1[:3, 3]
    pi0, Jp0 = proj(Ap0)
    pi1, Jp1 = proj(Ap1)
    n_unk = 6 + 3 * M
    J = torch.zeros(4 * M, n_unk, dtype=dt)  # observation order of the reference: all cam-0 rows, then all cam-1 rows
    r = torch.zeros(4 * M, dtype=dt)
    for i in range(M):
        J[2 * i:2 * i + 2, 6 + 3 * i:9 + 3 * i] = c[i] * Jp0[i]
        J[2 * M + 2 * i:2 * M + 2 * i + 2, 6 + 3 * i:9 + 3 * i] = c[i] * (Jp1[i] @ extr1[:3, :3])
        J[2 * M + 2 * i:2 * M + 2 * i + 2, :6] = c[i] * (Jp1[i] @ torch.cat([eye3, -hat(Ap1[i])], 1))
        r[2 * i:2 * i + 2] = c[i] * (pi0[i] - x0[i])
        r[2 * M + 2 * i:2 * M + 2 * i + 2] = c[i] * (pi1[i] - x1[i])
    return J.T @ J, -(J.T @ r), (r ** 2).sum()


def run_bundle_adjust_2_view(kpts0_norm, kpts1_norm, confidence, init_T021, n_iterations, lm_increase=1.5, lm_decrease=3.5):
    """Returns (refined T_021 of the valid samples [n_valid,4,4], valid_batch [B] bool) like the reference."""
    conf = confidence.squeeze(-1) if confidence.dim() == 3 else confidence
    B = kpts0_norm.shape[0]
    dt = kpts0_norm.dtype
    valid = conf > 0.0
    valid_batch = valid.sum(-1) > 6
    out = []
    for b in range(B):
        if not bool(valid_batch[b]):
            continue
        m = valid[b]
        x0, x1, c = kpts0_norm[b][m], kpts1_norm[b][m], conf[b][m]
        c = c / (0.5 * (2 * c.sum()).clamp(min=1e-6))  # each match is two observations (:45-48)
        extr1 = init_T021[b].clone().to(dt)
        P0 = torch.eye(4, dtype=dt)[:3]
        pts = K.triangulate_points(P0[None], extr1[None, :3], x0[None], x1[None])[0]
        lam = 0.1
        best_r, best = None, extr1.clone()
        for it in range(n_iterations + 1):
            A, bvec, rn = _normal_equations(extr1, pts, x0, x1, c)
            if it == 0:
                best_r, best = rn, extr1.clone()
            else:
                if rn < best_r:
                    best_r, best = rn, extr1.clone()
                    lam = lam / lm_decrease
                else:
                    lam = lam * lm_increase
            if it == n_iterations:
                break
            d = torch.diagonal(A)
            if bool((d > 0).all()):
                inv = 1.0 / d.clamp(min=1e-12)
                A = inv[:, None] * A
                bvec = inv * bvec
            A = A + torch.eye(A.shape[0], dtype=dt) * lam
            LU, piv, info = torch.linalg.lu_factor_ex(A)
            if int(info) != 0:
                continue
            dx = torch.linalg.lu_solve(LU, piv, bvec[:, None])[:, 0]
            delta = se3_exp_map(dx[None, :6]).permute(0, 2, 1)[0]
            extr1 = delta @ extr1
            pts = pts + dx[6:].view(-1, 3)
        out.append(best)
    res = torch.stack(out) if out else torch.zeros(0, 4, 4, dtype=dt)
    return res, valid_batch
