"""Validation targets and loss on the device: drop-ins for ``helpers.compute_gt_matches_of_image_pair`` (:121-203),
``helpers.compute_gt_matches`` (:215-226), ``helpers.compute_match_loss`` (:228-241) and the forward-only part of
``helpers.run_matcher`` (:243-260) on top of libe2emv.so (``csrc/gtmatch.hip``).  Inference/validation only: no autograd."""
import torch

from . import _lib
from .pose import _dev_of, _prep, compute_rotation_error, compute_translation_error_as_angle, run_weighted_8_point


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


def compute_gt_matches(opt, data):
    """``helpers.compute_gt_matches``: fills ``gt_indices_k_m`` / ``gt_weights_k_m`` for every pair of the tuple and pops
    the depth maps, like the reference."""
    T = len(data["ids"])
    for m in range(T):
        for k in range(m):
            T_k2m = torch.linalg.inv(data["pose" + str(m)]) @ data["pose" + str(k)]
            data["gt_indices_{}_{}".format(k, m)], data["gt_weights_{}_{}".format(k, m)] = compute_gt_matches_of_image_pair(
                data["keypoints" + str(k)], data["keypoints" + str(m)], data["intr" + str(k)], data["intr" + str(m)], T_k2m,
                data["depth" + str(k)], data["depth" + str(m)], opt.match_reproj_err, opt.unmatch_reproj_err)
    for m in range(T):
        data.pop("depth" + str(m))


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


def run_matcher(opt, data, matcher):
    """Forward-only ``helpers.run_matcher`` (:243-260): matcher -> match loss per pair (+ pose losses when
    ``opt.pose_loss``).  ``matcher`` may be wrapped in DataParallel/DDP (``.module``) like in the reference."""
    T = len(data["ids"])
    inner = getattr(matcher, "module", matcher)
    inner.config["full_output"] = opt.pose_loss
    result = matcher(data)
    dev = result["scores_0_1"].device
    match_loss = torch.zeros(1, device=dev)
    rot_loss = torch.zeros(1, device=dev)
    transl_loss = torch.zeros(1, device=dev)
    for id1 in range(T):
        for id0 in range(id1):
            match_loss = match_loss + compute_match_loss(result["scores_{}_{}".format(id0, id1)],
                                                         data["gt_indices_{}_{}".format(id0, id1)],
                                                         data["gt_weights_{}_{}".format(id0, id1)])
            if opt.pose_loss:
                target = torch.linalg.inv(data["pose{}".format(id1)]) @ data["pose{}".format(id0)]
                pred, _ = run_weighted_8_point(data, result, id0, id1, choose_closest=True, target_T_021=target)
                rot_loss = rot_loss + compute_rotation_error(pred, target)
                transl_loss = transl_loss + compute_translation_error_as_angle(pred, target)
    return {"match_loss": match_loss, "rot_loss": rot_loss, "transl_loss": transl_loss}, result
