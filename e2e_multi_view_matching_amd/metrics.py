"""Pose error in degrees and AUC@thresholds - host-side metric of the evaluation scripts.

Mirrors ``models.models.utils.{compute_pose_error, pose_auc}`` that ``eval_pairs.py:16``
imports from the absent submodule and uses at ``eval_pairs.py:263-270`` (upstream SuperGlue
``models/utils.py`` semantics).  Runs on B floats per evaluation - not part of the device
hot path; the per-pair angles themselves come from ``e2emv_pose_errors``.
"""
import numpy as np


def _angle_mat(R1, R2):
    c = np.clip((np.trace(R1.T @ R2) - 1) / 2, -1.0, 1.0)
    return np.rad2deg(np.abs(np.arccos(c)))


def _angle_vec(a, b):
    n = np.linalg.norm(a) * np.linalg.norm(b)
    return np.rad2deg(np.arccos(np.clip(np.dot(a, b) / n, -1.0, 1.0)))


def compute_pose_error(T_0to1, R, t):
    err_t = _angle_vec(t, T_0to1[:3, 3])
    return np.minimum(err_t, 180 - err_t), _angle_mat(R, T_0to1[:3, :3])


def pose_auc(errors, thresholds):
    e = np.sort(np.asarray(errors, dtype=np.float64))
    rec = (np.arange(len(e)) + 1) / len(e)
    e, rec = np.r_[0.0, e], np.r_[0.0, rec]
    trap = getattr(np, "trapezoid", None) or np.trapz
    out = []
    for thr in thresholds:
        k = np.searchsorted(e, thr)
        out.append(trap(np.r_[rec[:k], rec[k - 1]], x=np.r_[e[:k], thr]) / thr)
    return out


def pair_errors_deg(rot_rad, transl_rad):
    """max(err_R, min(err_t, 180 - err_t)) per pair from device angle errors (``eval_pairs.py:263-266``)."""
    er = np.rad2deg(np.asarray(rot_rad, dtype=np.float64))
    et = np.rad2deg(np.asarray(transl_rad, dtype=np.float64))
    return np.maximum(er, np.minimum(et, 180 - et))


# ---- image-plane rotation helpers of upstream's ``models/utils.py`` (imported by ``eval_pairs.py:16`` for the ScanNet
# in-plane rotations; host-side, numpy; restated from upstream's published semantics - the submodule is absent) ----
_QUARTER_TURNS_DEG = (0.0, 270.0, 180.0, 90.0)  # rot = number of clockwise quarter turns of the image


def rotate_pose_inplane(i_T_w, rot):
    """World-to-camera pose after rotating the image by ``rot`` quarter turns about the optical axis."""
    a = np.deg2rad(_QUARTER_TURNS_DEG[rot])
    Rz = np.eye(4, dtype=np.float32)
    Rz[0, 0], Rz[0, 1], Rz[1, 0], Rz[1, 1] = np.cos(a), -np.sin(a), np.sin(a), np.cos(a)
    return Rz @ i_T_w


def rotate_intrinsics(K, image_shape, rot):
    """3x3 intrinsics after rotating the image by ``rot`` quarter turns; ``image_shape`` = shape AFTER the rotation."""
    if not 0 <= rot <= 3:
        raise AssertionError(rot)
    h, w = image_shape[:2][::-1 if (rot % 2) else 1]
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    focal, centre = {1: ((fy, fx), (cy, w - 1 - cx)), 2: ((fx, fy), (w - 1 - cx, h - 1 - cy)),
                     3: ((fy, fx), (h - 1 - cy, cx)), 0: ((fy, fx), (h - 1 - cy, cx))}[rot % 4]
    return np.array([[focal[0], 0.0, centre[0]], [0.0, focal[1], centre[1]], [0.0, 0.0, 1.0]], dtype=K.dtype)
