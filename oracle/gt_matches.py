"""Oracle: ground-truth match targets and the match loss (SURVEY.md 8(f) "next" row 2), torch-CPU.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates ``helpers.py``: ``transform_kpts`` :114-118,
``compute_gt_matches_of_image_pair`` :121-203, ``set_weight`` :205-213, ``compute_match_loss`` :228-241.
Pinned by tests/golden/gt_matches_reference.npz (the reference's own helpers.py imported in the build container).

Quirks kept on purpose: keypoints are truncated to integer pixels (``.long()``) and those INTEGER coordinates enter the
reprojection errors; arg-min ties take the first index; the dustbin slot N of the target vectors keeps index -1 and gets
the un-match weight; index -1 in the loss addresses the LAST column (the dustbin) through negative indexing (E11).
"""
import torch


def transform_kpts(kpts, d, K0, K1, T_021):
    """helpers.py:114-118.  kpts [B,N,2], d [B,N,1], K/T [B,1,4,4] -> depth [B,N,1], reprojected kpts [B,N,2]."""
    h = torch.cat((kpts * d, d, torch.ones_like(d)), dim=-1).unsqueeze(-1)
    p = K1 @ T_021 @ torch.linalg.inv(K0) @ h
    depth = p[..., 2, :]
    return depth, p[..., :2, 0] / depth


def compute_gt_matches_of_image_pair(kpts0, kpts1, K0, K1, T0to1, depth0, depth1, max_matched_reproj_err,
                                     min_unmatched_reproj_err):
    """helpers.py:121-203 -> (indices [B,2,N+1] int64, weights [B,2,N+1] f32)."""
    bs, n, _ = kpts0.shape
    bidx = torch.arange(bs).unsqueeze(-1).expand(bs, n)
    k0, k1 = kpts0.long(), kpts1.long()
    d0 = depth0[bidx, k0[..., 1], k0[..., 0]].unsqueeze(-1)
    d1 = depth1[bidx, k1[..., 1], k1[..., 0]].unsqueeze(-1)
    K0u, K1u, Tu = K0.unsqueeze(1), K1.unsqueeze(1), T0to1.unsqueeze(1)
    dep01, k0to1 = transform_kpts(k0, d0, K0u, K1u, Tu)
    dep10, k1to0 = transform_kpts(k1, d1, K1u, K0u, torch.linalg.inv(Tu))
    err = torch.sqrt(((k1to0.unsqueeze(2) - k0.unsqueeze(1)) ** 2).sum(3)).transpose(1, 2)
    err = err + torch.sqrt(((k0to1.unsqueeze(2) - k1.unsqueeze(1)) ** 2).sum(3))
    err = err / 2.0  # [B, N0, N1]
    row_min, col_min = torch.argmin(err, dim=2), torch.argmin(err, dim=1)
    idx0 = torch.full((bs, n + 1), -1, dtype=torch.int64)
    idx1 = torch.full((bs, n + 1), -1, dtype=torch.int64)
    w0 = torch.zeros(bs, n + 1)
    w1 = torch.zeros(bs, n + 1)
    i0s = torch.arange(n).unsqueeze(0).expand(bs, n)
    i1s = row_min
    d0s, d1s = d0.squeeze(-1), d1.squeeze(-1)
    e_sel = err[bidx, i0s, i1s]
    md1 = d1s[bidx, i1s]
    valid_d0, valid_d1 = d0s > 1e-6, md1 > 1e-6
    rel01 = (dep01.squeeze(-1) - md1).abs() / md1
    rel10 = (dep10.squeeze(-1)[bidx, i1s] - d0s).abs() / d0s
    match = (col_min[bidx, i1s] == i0s) & (e_sel <= max_matched_reproj_err) & valid_d0 & valid_d1
    match = match & (rel01 < 0.1) & (rel10 < 0.1)
    idx0[:, :-1][match] = i1s[match]
    idx1[bidx[match], i1s[match]] = i0s[match]
    match_count = match.sum(1)
    drop = (~match) & ((~valid_d0) | (~valid_d1) | (e_sel <= min_unmatched_reproj_err))
    w0[:, :-1][drop] = -1
    drop_count = drop.sum(1)
    j1s = torch.arange(n).unsqueeze(0).expand(bs, n)
    j0s = col_min
    no_match = idx1[:, :-1] == -1
    md0 = d0s[bidx, j0s]
    invalid = (~(md0 > 1e-6)) | (~(d1s > 1e-6))
    drop1 = no_match & (invalid | (err[bidx, j0s, j1s] <= min_unmatched_reproj_err))
    w1[:, :-1][drop1] = -1
    drop_count = drop_count + drop1.sum(1)
    mw = 2.0 * match_count / (2.0 * torch.full_like(match_count, n) - drop_count)
    uw = 0.5 / (1.0 - mw)
    mw = 0.5 / mw
    bad = ~(mw.isfinite() & uw.isfinite())
    mw[bad] = 0.0
    uw[bad] = 0.0
    for w, idx in ((w0, idx0), (w1, idx1)):  # set_weight (:205-213)
        reset = w == -1
        unm = (~reset) & (idx == -1)
        mat = (~reset) & (idx != -1)
        w[reset] = 0.0
        w[unm] = uw.unsqueeze(-1).expand_as(w)[unm]
        w[mat] = mw.unsqueeze(-1).expand_as(w)[mat]
    return torch.stack((idx0, idx1), 1), torch.stack((w0, w1), 1)


def compute_match_loss(log_p, gt_indices, gt_weights):
    """helpers.py:228-241: negative log-likelihood of the targets, index -1 = dustbin column."""
    bs, ft, _ = log_p.shape
    i0, i1 = gt_indices[:, 0].reshape(bs * ft), gt_indices[:, 1].reshape(bs * ft)
    w0, w1 = gt_weights[:, 0].reshape(bs * ft), gt_weights[:, 1].reshape(bs * ft)
    r = torch.arange(bs * ft)
    l0 = -log_p.reshape(bs * ft, ft)[r, i0]
    l1 = -log_p.transpose(1, 2).reshape(bs * ft, ft)[r, i1]
    return (torch.dot(l0, w0) + torch.dot(l1, w1)) / bs
