"""Drop-in for ``pose_optimization/two_view/estimate_relative_pose.py`` and
``compute_pose_error.py`` on top of libe2emv.so.

Same names, argument order, keyword names, ``info`` keys and ``None`` conventions as the
reference (``estimate_relative_pose.py:9-14, 16-31, 84-136``; ``compute_pose_error.py:3-22``);
the arithmetic runs in the HIP kernels of ``csrc/pose.hip`` (fp64 Gram / Jacobi instead of
the reference's library SVDs).  Inference only (no autograd).  No CPU fallback.
"""
import ctypes

import torch

from . import _lib


def _dev_of(*tensors):
    for t in tensors:
        if t is not None and t.is_cuda:
            return t.device
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None


def _prep(t, dev):
    return t.to(dev, torch.float32).contiguous()


def normalize(kpts, intr):
    """``normalize`` (:9-14): pixel -> camera coordinates; a trivial elementwise op kept in
    torch for callers that use it stand-alone (``eval_pairs.py:240-241``)."""
    fx, fy, cx, cy = intr[..., 0, 0], intr[..., 1, 1], intr[..., 0, 2], intr[..., 1, 2]
    out = torch.zeros_like(kpts)
    out[..., 0] = (kpts[..., 0] - cx.unsqueeze(-1)) / fx.unsqueeze(-1)
    out[..., 1] = (kpts[..., 1] - cy.unsqueeze(-1)) / fy.unsqueeze(-1)
    return out


def get_kpts(data, result, id0, id1):
    """``get_kpts`` (:16-31) through ``e2emv_gather_matched`` (index -1 wraps to the last
    keypoint and gets weight 0, exactly like the reference's fancy indexing)."""
    if "keypoints" + str(id0) in data:
        k0, k1 = data["keypoints" + str(id0)], data["keypoints" + str(id1)]
    else:
        k0, k1 = data["keypoints{}_{}_{}".format(id0, id0, id1)], data["keypoints{}_{}_{}".format(id1, id0, id1)]
    matches = result["matches{}_{}_{}".format(id0, id0, id1)]
    conf = result["conf_scores_{}_{}".format(id0, id1)]
    dev = _dev_of(matches, k0)
    ctx = _lib.context(dev)
    k0d, k1d = _prep(k0, dev), _prep(k1, dev)
    m = matches.to(dev, torch.int64).contiguous()
    c = _prep(conf.reshape(conf.shape[0], -1), dev)
    B, N0 = m.shape
    k1g = torch.empty((B, N0, 2), dtype=torch.float32, device=dev)
    cout = torch.empty((B, N0), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        ctx.call("e2emv_gather_matched", B, N0, k1d.shape[1], _lib.ptr(k1d), _lib.ptr(m), _lib.ptr(c), _lib.ptr(k1g),
                 _lib.ptr(cout), _lib.stream_ptr(dev))
    return k0d, k1g, data["intr" + str(id0)], data["intr" + str(id1)], cout.unsqueeze(-1)


def estimate_relative_pose_w8pt(kpts0, kpts1, intr0, intr1, confidence, choose_closest=False, T_021=None,
                                determine_inliers=False):
    """``estimate_relative_pose_w8pt`` (:84-128).  Returns ``(T_021 [B,4,4], info)`` or
    ``(None, None)`` when fewer than 8 correspondences are given (:85-86)."""
    if kpts0.shape[1] < 8:
        return None, None
    if kpts0.shape != kpts1.shape:
        raise AssertionError(kpts0.shape, kpts1.shape)
    dev = _dev_of(kpts0, kpts1, intr0, confidence)
    if dev is None:
        raise RuntimeError("estimate_relative_pose_w8pt needs an MI355X (no CPU fallback)")
    ctx = _lib.context(dev)
    B, N = kpts0.shape[:2]
    conf_shape = confidence.shape
    conf2 = confidence.reshape(B, -1)
    if conf2.shape[1] != N:
        raise AssertionError(conf_shape)
    k0, k1, cf = _prep(kpts0, dev), _prep(kpts1, dev), _prep(conf2, dev)
    K0, K1 = _prep(intr0, dev), _prep(intr1, dev)
    kdim = K0.shape[-1]
    if K0.dim() == 2:
        K0, K1 = K0.unsqueeze(0), K1.unsqueeze(0)
    if K0.shape[-2:] != (kdim, kdim) or K1.shape != K0.shape or K0.shape[0] not in (1, B):
        raise AssertionError(K0.shape, K1.shape)
    Tg = _prep(T_021, dev) if (choose_closest and T_021 is not None) else None
    if choose_closest and Tg is None:
        raise ValueError("choose_closest=True needs T_021")
    T = torch.empty((B, 4, 4), dtype=torch.float32, device=dev)
    k0n, k1n = torch.empty_like(k0), torch.empty_like(k1)
    cfn = torch.empty((B, N), dtype=torch.float32, device=dev)
    inl = torch.empty((B, N), dtype=torch.uint8, device=dev) if determine_inliers else None
    pos = torch.empty((B, N), dtype=torch.uint8, device=dev)
    F = torch.empty((B, 3, 3), dtype=torch.float32, device=dev)
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        ctx.call("e2emv_w8pt", B, N, _lib.ptr(k0), _lib.ptr(k1), _lib.ptr(K0), _lib.ptr(K1), kdim, K0.shape[0], _lib.ptr(cf),
                 1 if choose_closest else 0, _lib.ptr(Tg), 1 if determine_inliers else 0, _lib.ptr(T), _lib.ptr(k0n),
                 _lib.ptr(k1n), _lib.ptr(cfn), _lib.ptr(inl), _lib.ptr(pos), _lib.ptr(F), _lib.ptr(status),
                 _lib.stream_ptr(dev))
    info = {"kpts0_norm": k0n, "kpts1_norm": k1n, "confidence": cfn.reshape(conf_shape),
            "inliers": inl.bool() if inl is not None else None, "pos_depth_mask": pos.bool(),
            "F": F, "status": status}
    return T, info


def run_weighted_8_point(data, result, id0, id1, choose_closest=False, target_T_021=None):
    """``run_weighted_8_point`` (:130-136)."""
    match_key = "matches{}_{}_{}".format(id0, id0, id1)
    if match_key in result and result[match_key].shape[1] != 0:
        kpts0, kpts1, intr0, intr1, confidence = get_kpts(data, result, id0, id1)
        return estimate_relative_pose_w8pt(kpts0, kpts1, intr0, intr1, confidence, choose_closest=choose_closest,
                                           T_021=target_T_021)
    return None, None


def pose_errors(T0, T1):
    """Per-sample (rotation, translation-direction) angle errors in radians on the device."""
    dev = _dev_of(T0, T1)
    ctx = _lib.context(dev)
    a, b = _prep(T0, dev), _prep(T1, dev)
    B = a.shape[0]
    rot = torch.empty((B,), dtype=torch.float32, device=dev)
    tr = torch.empty((B,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        ctx.call("e2emv_pose_errors", B, _lib.ptr(a), _lib.ptr(b), _lib.ptr(rot), _lib.ptr(tr), _lib.stream_ptr(dev))
    return rot, tr


def compute_rotation_error(T0, T1, reduce=True):
    """``compute_rotation_error`` (compute_pose_error.py:3-12)."""
    rot, _ = pose_errors(T0, T1)
    return rot.mean() if reduce else rot


def compute_translation_error_as_angle(T0, T1, reduce=True):
    """``compute_translation_error_as_angle`` (compute_pose_error.py:14-22); ``reduce=True``
    averages over the entries whose norm product exceeds 1e-6 like the reference."""
    _, tr = pose_errors(T0, T1)
    if not reduce:
        return tr
    n = torch.linalg.norm(T0[..., :3, 3], dim=-1) * torch.linalg.norm(T1[..., :3, 3], dim=-1)
    valid = (n > 1e-6).to(tr.device)
    return tr[valid].mean()


def run_bundle_adjust_2_view(kpts0_norm, kpts1_norm, confidence, init_T021, n_iterations, check_lu_info_strict=False,
                             check_precond_strict=False):
    """``run_bundle_adjust_2_view`` (estimate_relative_pose.py:138-144): returns ``(refined T_021 of the valid samples
    [n_valid,4,4], valid_batch [B] bool)`` so that ``pred_T021[valid] = refined`` works as at ``eval_pairs.py:252-255``.
    The two ``*_strict`` flags only matter for singular systems, which the fp64 Schur solve reports the same way."""
    dev = _dev_of(kpts0_norm, kpts1_norm, init_T021)
    ctx = _lib.context(dev)
    B, N = kpts0_norm.shape[:2]
    k0, k1 = _prep(kpts0_norm, dev), _prep(kpts1_norm, dev)
    cf = _prep(confidence.reshape(B, -1), dev)
    Ti = _prep(init_T021, dev)
    To = torch.empty((B, 4, 4), dtype=torch.float32, device=dev)
    valid = torch.empty((B,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        ctx.call("e2emv_ba_2view", B, N, _lib.ptr(k0), _lib.ptr(k1), _lib.ptr(cf), _lib.ptr(Ti), int(n_iterations),
                 _lib.ptr(To), _lib.ptr(valid), _lib.stream_ptr(dev))
    vb = valid.bool()
    return To[vb], vb
