"""Oracle: pose error in degrees and pose AUC (numpy).

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference imports ``pose_auc`` and
``compute_pose_error`` from the absent submodule's ``models/utils.py`` (``eval_pairs.py:16``)
and uses them at ``eval_pairs.py:263-270``; this restates the published upstream SuperGlue
``models/utils.py`` semantics.  Checked in tests against hand-computed
trapezoids.
"""
import numpy as np


_trapz = getattr(np, "trapezoid", None) or np.trapz


def angle_error_mat(R1, R2):
    cos = (np.trace(np.dot(R1.T, R2)) - 1) / 2
    cos = np.clip(cos, -1.0, 1.0)
    return np.rad2deg(np.abs(np.arccos(cos)))


def angle_error_vec(v1, v2):
    n = np.linalg.norm(v1) * np.linalg.norm(v2)
    return np.rad2deg(np.arccos(np.clip(np.dot(v1, v2) / n, -1.0, 1.0)))


def compute_pose_error(T_0to1, R, t):
    R_gt, t_gt = T_0to1[:3, :3], T_0to1[:3, 3]
    err_t = angle_error_vec(t, t_gt)
    err_t = np.minimum(err_t, 180 - err_t)  # sign ambiguity of E
    return err_t, angle_error_mat(R, R_gt)


def pose_auc(errors, thresholds):
    order = np.argsort(errors)
    errors = np.array(errors, dtype=np.float64)[order]
    recall = (np.arange(len(errors)) + 1) / len(errors)
    errors = np.r_[0.0, errors]
    recall = np.r_[0.0, recall]
    aucs = []
    for t in thresholds:
        last = np.searchsorted(errors, t)
        r = np.r_[recall[:last], recall[last - 1]]
        e = np.r_[errors[:last], t]
        aucs.append(_trapz(r, x=e) / t)
    return aucs


def pair_errors(T_pred, T_gt):
    """max(err_t, err_R) in degrees per pair (``eval_pairs.py:266``); inf if T_pred is None."""
    out = []
    for p, g in zip(T_pred, T_gt):
        if p is None or not np.all(np.isfinite(p)):
            out.append(np.inf)
            continue
        et, er = compute_pose_error(np.asarray(g, np.float64), np.asarray(p, np.float64)[:3, :3],
                                    np.asarray(p, np.float64)[:3, 3])
        out.append(max(et, er))
    return np.array(out)
