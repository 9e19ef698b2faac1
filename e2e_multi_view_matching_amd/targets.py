"""Validation targets and match loss on the device: the two data-parallel N^2 steps of the reference's validation pass,
``helpers.compute_gt_matches_of_image_pair`` (:121-203) and ``helpers.compute_match_loss`` (:228-241), on top of
libe2emv.so (``csrc/gtmatch.hip``).  The reference's own ``helpers.compute_gt_matches`` / ``helpers.run_matcher`` stay
the callers (they only loop over the pairs of a tuple); ``gt_matches_for_tuple`` is the batched equivalent of that loop
for device-resident data.  Inference/validation only: no autograd."""
import torch

from . import _lib
from .pose import _dev_of, _prep


def compute_gt_matches_of_image_pair(kpts0, kpts1, K0, K1, T0to1, depth0, depth1, max_matched_reproj_err,
                                     min_unmatched_reproj_err):
    """-> (indices [B,2,N+1] int64, weights [B,2,N+1] f32), exactly the pair ``helpers.py:203`` returns."""
    if kpts0.shape != kpts1.shape:
        raise AssertionError(kpts0.shape, kpts1.shape)
    dev = _dev_of(kpts0, depth0)
    ctx = _lib.context(dev)
    B, N = kpts0.shape[:2]
    k0, k1 = _prep(kpts0, dev), _prep(kpts1, dev)
    Ka, Kb, T = _prep(K0, dev), _prep(K1, dev), _prep(T0to1, dev)
    if Ka.shape[-2:] != (4, 4) or T.shape[-2:] != (4, 4):
        raise AssertionError("intrinsics and pose must be 4x4 (MatchingDataset layout)")
    d0, d1 = _prep(depth0, dev), _prep(depth1, dev)
    H, W = d0.shape[-2:]
    idx = torch.empty((B, 2, N + 1), dtype=torch.int64, device=dev)
    w = torch.empty((B, 2, N + 1), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        ctx.call("e2emv_gt_matches", B, N, _lib.ptr(k0), _lib.ptr(k1), _lib.ptr(Ka), _lib.ptr(Kb), _lib.ptr(T), _lib.ptr(d0),
                 _lib.ptr(d1), H, W, float(max_matched_reproj_err), float(min_unmatched_reproj_err), _lib.ptr(idx), _lib.ptr(w),
                 _lib.stream_ptr(dev))
    return idx, w


def relative_pose(pose_from, pose_to):
    """``inv(pose_to) @ pose_from`` per batch element on the device (the relative pose ``helpers.py:219, 254`` builds with
    torch ops): [B,4,4] x [B,4,4] -> [B,4,4]."""
    dev = _dev_of(pose_from, pose_to)
    ctx = _lib.context(dev)
    a, b = _prep(pose_from, dev), _prep(pose_to, dev)
    if a.shape != b.shape or a.shape[-2:] != (4, 4):
        raise AssertionError(a.shape, b.shape)
    out = torch.empty_like(a)
    with torch.cuda.device(dev):
        ctx.call("e2emv_relative_pose", a.shape[0], _lib.ptr(a), _lib.ptr(b), _lib.ptr(out), _lib.stream_ptr(dev))
    return out


def gt_matches_for_tuple(data, max_matched_reproj_err, min_unmatched_reproj_err, n_images=None):
    """Ground-truth match targets of every pair of a device-resident tuple: returns
    ``{(k, m): (indices, weights)}`` for k < m, the values ``helpers.compute_gt_matches`` stores under
    ``gt_indices_k_m`` / ``gt_weights_k_m``.  Does not touch ``data``."""
    if n_images is None:
        n_images = len(data["ids"])
    rel = {(k, m): relative_pose(data["pose%d" % k], data["pose%d" % m]) for m in range(n_images) for k in range(m)}
    return {km: compute_gt_matches_of_image_pair(data["keypoints%d" % km[0]], data["keypoints%d" % km[1]],
                                                 data["intr%d" % km[0]], data["intr%d" % km[1]], T_km,
                                                 data["depth%d" % km[0]], data["depth%d" % km[1]],
                                                 max_matched_reproj_err, min_unmatched_reproj_err)
            for km, T_km in rel.items()}


def compute_match_loss(log_p, gt_indices_0_1, gt_weights_0_1):
    """``helpers.compute_match_loss``: scalar tensor (on the device)."""
    dev = _dev_of(log_p)
    ctx = _lib.context(dev)
    lp = _prep(log_p, dev)
    B, ft = lp.shape[:2]
    idx = gt_indices_0_1.to(dev, torch.int64).contiguous()
    w = _prep(gt_weights_0_1, dev)
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        ctx.call("e2emv_match_loss", B, ft - 1, _lib.ptr(lp), _lib.ptr(idx), _lib.ptr(w), _lib.ptr(loss), _lib.stream_ptr(dev))
    return loss[0]
