"""Drop-in for ``pose_optimization/two_view/estimate_relative_pose.py`` and
``compute_pose_error.py`` on top of libe2emv.so.

Same names, argument order, keyword names, ``info`` keys and ``None`` conventions as the
reference (``estimate_relative_pose.py:9-14, 16-31, 84-136``; ``compute_pose_error.py:3-22``);
the arithmetic runs in the HIP kernels of ``csrc/pose.hip`` (fp64 Gram / Jacobi instead of
the reference's library SVDs).  The pose is differentiable with respect to the confidences (training, the pose loss:
``_W8ptPose`` / ``e2emv_w8pt_backward``); keypoints and the candidate choice carry no gradient.  No CPU fallback.
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


def _intr_batch(K, B):
    """[k,k] or [1|B,k,k] intrinsics -> (contiguous [nb,k,k], k, nb)."""
    if K.dim() == 2:
        K = K.unsqueeze(0)
    kdim = K.shape[-1]
    if K.shape[-2:] != (kdim, kdim) or kdim not in (3, 4) or K.shape[0] not in (1, B):
        raise AssertionError(K.shape)
    return K, kdim, K.shape[0]


def normalize(kpts, intr):
    """``normalize`` (:9-14): pixel -> camera coordinates (``e2emv_normalize_kpts``; stand-alone callers:
    ``eval_pairs.py:240-241``)."""
    dev = _dev_of(kpts, intr)
    ctx = _lib.context(dev)
    squeeze = kpts.dim() == 2
    k = _prep(kpts.unsqueeze(0) if squeeze else kpts, dev)
    B, N = k.shape[:2]
    K, kdim, nb = _intr_batch(_prep(intr, dev), B)
    out = torch.empty_like(k)
    if B * N:
        with torch.cuda.device(dev):
            ctx.call("e2emv_normalize_kpts", B, N, _lib.ptr(k), _lib.ptr(K), kdim, nb, _lib.ptr(out), _lib.stream_ptr(dev))
    return out[0] if squeeze else out


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
    if torch.is_grad_enabled() and conf.requires_grad:
        cout, k1g = _MaskedConfidence.apply(conf, ctx, dev, k1d, m)
        return k0d, k1g, data["intr" + str(id0)], data["intr" + str(id1)], cout.unsqueeze(-1)
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
    args = (ctx, dev, B, N, k0, k1, K0, K1, kdim, Tg, bool(choose_closest), bool(determine_inliers), conf_shape)
    if torch.is_grad_enabled() and confidence.requires_grad:
        # training (pose loss, helpers.py:253-258): T carries the graph back to the confidences (csrc/pose.hip, w8pt_backward)
        holder = []
        T = _W8ptPose.apply(conf2.to(dev, torch.float32), args, holder)
        return T, holder[0]
    return _w8pt_call(cf, *args)


def _w8pt_call(cf, ctx, dev, B, N, k0, k1, K0, K1, kdim, Tg, choose_closest, determine_inliers, conf_shape):
    T = torch.empty((B, 4, 4), dtype=torch.float32, device=dev)
    k0n, k1n = torch.empty_like(k0), torch.empty_like(k1)
    cfn = torch.empty((B, N), dtype=torch.float32, device=dev)
    # (the kernels write the masks as 0 / 1 bytes - the storage format of torch.bool: no conversion kernel behind the call)
    inl = torch.empty((B, N), dtype=torch.bool, device=dev) if determine_inliers else None
    pos = torch.empty((B, N), dtype=torch.bool, device=dev)
    F = torch.empty((B, 3, 3), dtype=torch.float32, device=dev)
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        ctx.call("e2emv_w8pt", B, N, _lib.ptr(k0), _lib.ptr(k1), _lib.ptr(K0), _lib.ptr(K1), kdim, K0.shape[0], _lib.ptr(cf),
                 1 if choose_closest else 0, _lib.ptr(Tg), 1 if determine_inliers else 0, _lib.ptr(T), _lib.ptr(k0n),
                 _lib.ptr(k1n), _lib.ptr(cfn), _lib.ptr(inl), _lib.ptr(pos), _lib.ptr(F), _lib.ptr(status),
                 _lib.stream_ptr(dev))
    info = {"kpts0_norm": k0n, "kpts1_norm": k1n, "confidence": cfn.reshape(conf_shape),
            "inliers": inl, "pos_depth_mask": pos,
            "F": F, "status": status}
    return T, info


class _W8ptPose(torch.autograd.Function):
    """T = w8pt(confidence): ``e2emv_w8pt`` forward, ``e2emv_w8pt_backward`` for dLoss/dconfidence (the keypoints and the
    candidate choice carry no gradient).  ``holder`` receives the ``info`` dict of the forward."""

    @staticmethod
    def forward(fctx, conf, args, holder):
        cf = conf.contiguous()
        T, info = _w8pt_call(cf, *args)
        holder.append(info)
        fctx.args = args
        fctx.save_for_backward(cf, T, info["kpts0_norm"], info["kpts1_norm"])
        return T

    @staticmethod
    def backward(fctx, gT):
        cf, T, k0n, k1n = fctx.saved_tensors
        ctx, dev, B, N = fctx.args[:4]
        g = gT.to(torch.float32).contiguous()
        gconf = torch.empty((B, N), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            ctx.call("e2emv_w8pt_backward", B, N, _lib.ptr(k0n), _lib.ptr(k1n), _lib.ptr(cf), _lib.ptr(T), _lib.ptr(g), _lib.ptr(gconf),
                     _lib.stream_ptr(dev))
        return gconf, None, None


class _MaskedConfidence(torch.autograd.Function):
    """``confidence = (matches >= 0) * conf_scores`` of ``get_kpts`` (:27-29) with the gather of image 1's keypoints:
    forward ``e2emv_gather_matched``, backward ``e2emv_apply_mask``."""

    @staticmethod
    def forward(fctx, conf, ctx, dev, k1d, m):
        B, N0 = m.shape
        c = conf.reshape(B, -1).to(dev, torch.float32).contiguous()
        k1g = torch.empty((B, N0, 2), dtype=torch.float32, device=dev)
        cout = torch.empty((B, N0), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            ctx.call("e2emv_gather_matched", B, N0, k1d.shape[1], _lib.ptr(k1d), _lib.ptr(m), _lib.ptr(c), _lib.ptr(k1g),
                     _lib.ptr(cout), _lib.stream_ptr(dev))
        fctx.misc = (ctx, dev, m, conf.shape)
        fctx.mark_non_differentiable(k1g)
        return cout, k1g

    @staticmethod
    def backward(fctx, gcout, _gk):
        ctx, dev, m, shape = fctx.misc
        g = gcout.to(torch.float32).contiguous()
        return mask_confidence(g, m >= 0).reshape(shape), None, None, None, None


def run_weighted_8_point(data, result, id0, id1, choose_closest=False, target_T_021=None):
    """``run_weighted_8_point`` (:130-136)."""
    match_key = "matches{}_{}_{}".format(id0, id0, id1)
    if match_key in result and result[match_key].shape[1] != 0:
        kpts0, kpts1, intr0, intr1, confidence = get_kpts(data, result, id0, id1)
        return estimate_relative_pose_w8pt(kpts0, kpts1, intr0, intr1, confidence, choose_closest=choose_closest,
                                           T_021=target_T_021)
    return None, None


def run_weighted_8_point_tuple(data, result, choose_closest=False, targets=None, determine_inliers=False):
    """Weighted 8-point pose of EVERY pair of the tuple in one batched solve (``e2emv_w8pt_tuple``): what the reference
    does by calling ``run_weighted_8_point`` inside its pair loops (``helpers.py:250-258``, ``bundle_adjust_io.py:62-100``),
    with 4 kernel launches per tuple batch instead of 4 per pair.  Returns ``{(id0, id1): (T_021 [B,4,4], info)}`` with
    the per-pair values of ``estimate_relative_pose_w8pt`` (views into the batched outputs) and ``(None, None)`` for a pair
    whose matches are missing.  ``targets``: ``{(id0, id1): T_021 [B,4,4]}`` when ``choose_closest``.  Images with
    different keypoint counts (or per-pair keypoint keys) go through the per-pair call."""
    T = 0
    while "keypoints" + str(T) in data:
        T += 1
    pairs = [(i, j) for j in range(T) for i in range(j)]
    mkeys = ["matches{}_{}_{}".format(i, i, j) for i, j in pairs]
    shapes = {tuple(data["keypoints" + str(t)].shape) for t in range(T)}
    batched = T >= 2 and len(shapes) == 1 and all(k in result for k in mkeys) and next(iter(shapes))[1] >= 8
    if batched and choose_closest and (targets is None or any(p not in targets for p in pairs)):
        raise ValueError("choose_closest=True needs a target T_021 for every pair")
    if not batched:
        out = {}
        for (i, j) in pairs:
            tgt = targets[(i, j)] if (choose_closest and targets is not None) else None
            out[(i, j)] = run_weighted_8_point(data, result, i, j, choose_closest=choose_closest, target_T_021=tgt)
        return out
    dev = _dev_of(result[mkeys[0]], data["keypoints0"])
    ctx = _lib.context(dev)
    B, N = data["keypoints0"].shape[:2]
    P = len(pairs)
    kpts = [_prep(data["keypoints" + str(t)], dev) for t in range(T)]
    intr = [_intr_batch(_prep(data["intr" + str(t)], dev), B) for t in range(T)]
    kdim, nb = intr[0][1], intr[0][2]
    if any(k[1:] != (kdim, nb) for k in intr):
        raise AssertionError("intrinsics of a tuple must share their layout")
    intr = [k[0] for k in intr]
    matches = [result[k].to(dev, torch.int64).contiguous() for k in mkeys]
    conf_shapes = [result["conf_scores_{}_{}".format(i, j)].shape for i, j in pairs]
    conf = [_prep(result["conf_scores_{}_{}".format(i, j)].reshape(B, -1), dev) for i, j in pairs]
    if any(c.shape != (B, N) for c in conf) or any(m.shape != (B, N) for m in matches):
        raise AssertionError("matches / conf_scores must be [B, N]")
    tg = [_prep(targets[p], dev) for p in pairs] if choose_closest else [None] * P
    Tout = torch.empty((P, B, 4, 4), dtype=torch.float32, device=dev)
    k0n = torch.empty((P, B, N, 2), dtype=torch.float32, device=dev)
    k1n = torch.empty_like(k0n)
    cfn = torch.empty((P, B, N), dtype=torch.float32, device=dev)
    inl = torch.empty((P, B, N), dtype=torch.bool, device=dev) if determine_inliers else None  # (0 / 1 bytes, see above)
    pos = torch.empty((P, B, N), dtype=torch.bool, device=dev)
    F = torch.empty((P, B, 3, 3), dtype=torch.float32, device=dev)
    status = torch.empty((P, B), dtype=torch.int32, device=dev)
    keep, args = [], []
    for lst in (kpts, intr, matches, conf, tg):
        pa, arr = _lib.ptr_array(lst)
        keep.append(arr)
        args.append(pa)
    with torch.cuda.device(dev):
        ctx.call("e2emv_w8pt_tuple", B, T, N, args[0], args[1], kdim, nb, args[2], args[3], 1 if choose_closest else 0, args[4],
                 1 if determine_inliers else 0, _lib.ptr(Tout), _lib.ptr(k0n), _lib.ptr(k1n), _lib.ptr(cfn), _lib.ptr(inl),
                 _lib.ptr(pos), _lib.ptr(F), _lib.ptr(status), _lib.stream_ptr(dev))
    posb, inlb = pos, inl
    out = {}
    for q, p in enumerate(pairs):
        info = {"kpts0_norm": k0n[q], "kpts1_norm": k1n[q], "confidence": cfn[q].reshape(conf_shapes[q]),
                "inliers": inlb[q] if inlb is not None else None, "pos_depth_mask": posb[q], "F": F[q], "status": status[q]}
        out[p] = (Tout[q], info)
    return out


def _pose_error_buffers(T0, T1, means):
    dev = _dev_of(T0, T1)
    ctx = _lib.context(dev)
    a, b = _prep(T0, dev), _prep(T1, dev)
    B = a.shape[0]
    rot = torch.empty((B,), dtype=torch.float32, device=dev)
    tr = torch.empty((B,), dtype=torch.float32, device=dev)
    valid = torch.empty((B,), dtype=torch.bool, device=dev)  # (0 / 1 bytes)
    m2 = torch.empty((2,), dtype=torch.float32, device=dev) if means else None
    with torch.cuda.device(dev):
        ctx.call("e2emv_pose_error_means", B, _lib.ptr(a), _lib.ptr(b), _lib.ptr(rot), _lib.ptr(tr), _lib.ptr(valid),
                 _lib.ptr(m2), _lib.stream_ptr(dev))
    return rot, tr, valid, m2


def pose_errors(T0, T1):
    """Per-sample (rotation, translation-direction) angle errors in radians on the device, fixed shape [B] each
    (entries whose translation norm product is <= 1e-6 read 0)."""
    rot, tr, _, _ = _pose_error_buffers(T0, T1, means=False)
    return rot, tr


class _PoseErrors(torch.autograd.Function):
    """(rot [B], transl [B], valid [B]) = pose errors of T0 against T1 with the gradient w.r.t. T0 (``e2emv_pose_errors_backward``):
    the two error functions as the pose LOSS of ``helpers.run_matcher`` (helpers.py:256-258)."""

    @staticmethod
    def forward(fctx, T0, T1):
        rot, tr, valid, _ = _pose_error_buffers(T0, T1, means=False)
        fctx.save_for_backward(_prep(T0, rot.device), _prep(T1, rot.device))
        fctx.mark_non_differentiable(valid)
        return rot, tr, valid

    @staticmethod
    def backward(fctx, g_rot, g_tr, _gv):
        a, b = fctx.saved_tensors
        dev, B = a.device, a.shape[0]
        ctx = _lib.context(dev)
        gr = (torch.zeros(B, device=dev) if g_rot is None else g_rot).to(torch.float32).contiguous()
        gt = (torch.zeros(B, device=dev) if g_tr is None else g_tr).to(torch.float32).contiguous()
        gT = torch.empty((B, 4, 4), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            ctx.call("e2emv_pose_errors_backward", B, _lib.ptr(a), _lib.ptr(b), _lib.ptr(gr), _lib.ptr(gt), _lib.ptr(gT), _lib.stream_ptr(dev))
        return gT, None


def _wants_grad(T0):
    return torch.is_grad_enabled() and torch.is_tensor(T0) and T0.requires_grad


def compute_rotation_error(T0, T1, reduce=True):
    """``compute_rotation_error`` (compute_pose_error.py:3-12); the mean is reduced on the device.  Differentiable with respect to
    ``T0`` when it carries a graph (the rotation loss of ``helpers.run_matcher``)."""
    if _wants_grad(T0):
        rot, _, _ = _PoseErrors.apply(T0, T1)
        return rot.mean() if reduce else rot
    rot, _, _, m2 = _pose_error_buffers(T0, T1, means=reduce)
    return m2[0] if reduce else rot


def compute_translation_error_as_angle(T0, T1, reduce=True):
    """``compute_translation_error_as_angle`` (compute_pose_error.py:14-22): only the entries whose norm product
    exceeds 1e-6 count - ``reduce=True`` averages over them on the device, ``reduce=False`` returns exactly those
    entries (shape [n_valid], like the reference's boolean indexing).  Differentiable with respect to ``T0`` like the above."""
    if _wants_grad(T0):
        _, tr, valid = _PoseErrors.apply(T0, T1)
        sel = tr[valid]
        return sel.mean() if reduce else sel
    _, tr, valid, m2 = _pose_error_buffers(T0, T1, means=reduce)
    return m2[1] if reduce else tr[valid]


def mask_confidence(confidence, mask):
    """``confidence[~mask] = 0`` as a new tensor, on the device (``e2emv_apply_mask``): the step between the weighted 8-point
    solve and the two-view bundle adjustment (``eval_pairs.py:250-251``, ``bundle_adjust_io.py:18-19``)."""
    dev = _dev_of(confidence, mask)
    ctx = _lib.context(dev)
    c = _prep(confidence, dev)
    m = mask.to(dev).reshape(c.shape).to(torch.uint8).contiguous() if mask.dtype != torch.uint8 else mask.to(dev).reshape(c.shape).contiguous()
    out = torch.empty_like(c)
    if c.numel():
        with torch.cuda.device(dev):
            ctx.call("e2emv_apply_mask", c.numel(), _lib.ptr(c), _lib.ptr(m), _lib.ptr(out), _lib.stream_ptr(dev))
    return out


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
