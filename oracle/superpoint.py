"""ORACLE (test infrastructure only - never imported by the product path): torch-CPU restatement of the SuperPoint
front-end the reference runs before the matcher.

The reference imports ``models.models.superpoint.SuperPoint`` from an ABSENT, un-vendored submodule (`.gitmodules:1-3`);
its call sites are ``helpers.py:73-96`` (``run_super_point``: ``super_point({"image": [batch, ...]})`` -> dict of lists)
and the constructor configs at ``train.py:335-341``, ``eval_pairs.py:197-202``, ``eval_multi_view.py:135-140`` (keys
``nms_radius, keypoint_threshold, max_keypoints, remove_borders, fill_with_random_keypoints``).  The algorithm is upstream
magicleap SuperPoint (DeTone et al. 2018; SuperGluePretrainedNetwork ``models/superpoint.py``): VGG-style encoder
(conv1a..conv4b, 3 max-pools), detector head convPa/convPb -> softmax over 65 -> drop dustbin -> 8x8 depth-to-space ->
``simple_nms`` -> threshold -> border removal -> top-k; descriptor head convDa/convDb -> L2 normalise -> bilinear
``grid_sample(align_corners=True)`` at the keypoints -> L2 normalise.

PINNING: against the HuggingFace port of upstream (``transformers`` ``models/superpoint/modeling_superpoint.py``) run in the
build container with seeded random weights -> ``tests/golden/superpoint_hf.npz`` (tests/golden/make_golden.py).  The
fork-only ``fill_with_random_keypoints`` option has no visible source: parity unpinned for it (defined in DESIGN.md).
Ties in the top-k are broken by the lower pixel index (torch.topk leaves the order of equal scores unspecified).
"""
import torch
import torch.nn.functional as F

LAYERS = [("conv1a", 1, 64, 3), ("conv1b", 64, 64, 3), ("conv2a", 64, 64, 3), ("conv2b", 64, 64, 3), ("conv3a", 64, 128, 3),
          ("conv3b", 128, 128, 3), ("conv4a", 128, 128, 3), ("conv4b", 128, 128, 3), ("convPa", 128, 256, 3), ("convPb", 256, 65, 1),
          ("convDa", 128, 256, 3), ("convDb", 256, 256, 1)]


def seeded_state(seed=0, gain=1.0):
    """Random weights in upstream's parameter names; Kaiming-like scale so activations stay O(1) through the stack."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, cin, cout, k in LAYERS:
        sd[name + ".weight"] = torch.randn(cout, cin, k, k, generator=g) * (gain * (2.0 / (cin * k * k)) ** 0.5)
        sd[name + ".bias"] = torch.randn(cout, generator=g) * 0.05
    return sd


def simple_nms(scores, nms_radius):
    def max_pool(x):
        return F.max_pool2d(x, kernel_size=nms_radius * 2 + 1, stride=1, padding=nms_radius)
    zeros = torch.zeros_like(scores)
    max_mask = scores == max_pool(scores)
    for _ in range(2):
        supp_mask = max_pool(max_mask.float()) > 0
        supp_scores = torch.where(supp_mask, zeros, scores)
        new_max_mask = supp_scores == max_pool(supp_scores)
        max_mask = max_mask | (new_max_mask & (~supp_mask))
    return torch.where(max_mask, scores, zeros)


def dense_maps(sd, image):
    """image [B,1,H,W] -> (score map after NMS input [B,H,W] i.e. BEFORE nms, dense descriptors [B,256,H/8,W/8] unnormalised)."""
    c = lambda x, n, pad: F.conv2d(x, sd[n + ".weight"], sd[n + ".bias"], padding=pad)  # noqa: E731
    x = F.relu(c(image, "conv1a", 1)); x = F.relu(c(x, "conv1b", 1)); x = F.max_pool2d(x, 2, 2)
    x = F.relu(c(x, "conv2a", 1)); x = F.relu(c(x, "conv2b", 1)); x = F.max_pool2d(x, 2, 2)
    x = F.relu(c(x, "conv3a", 1)); x = F.relu(c(x, "conv3b", 1)); x = F.max_pool2d(x, 2, 2)
    x = F.relu(c(x, "conv4a", 1)); x = F.relu(c(x, "conv4b", 1))
    s = c(F.relu(c(x, "convPa", 1)), "convPb", 0)
    s = F.softmax(s, 1)[:, :-1]
    b, _, h, w = s.shape
    s = s.permute(0, 2, 3, 1).reshape(b, h, w, 8, 8).permute(0, 1, 3, 2, 4).reshape(b, h * 8, w * 8)
    d = c(F.relu(c(x, "convDa", 1)), "convDb", 0)
    return s, d


def sample_descriptors(keypoints, descriptors, s=8):
    """keypoints [1,N,2] (x, y), descriptors [1,C,h,w] (already L2-normalised) -> [1,C,N] normalised."""
    b, c, h, w = descriptors.shape
    keypoints = keypoints - s / 2 + 0.5
    keypoints = keypoints / torch.tensor([(w * s - s / 2 - 0.5), (h * s - s / 2 - 0.5)]).to(keypoints)[None]
    keypoints = keypoints * 2 - 1
    d = F.grid_sample(descriptors, keypoints.view(b, 1, -1, 2), mode="bilinear", align_corners=True)
    return F.normalize(d.reshape(b, c, -1), p=2, dim=1)


def select_keypoints(score_map, threshold, border, max_keypoints):
    """One image's NMS-ed score map [H,W] -> (keypoints [n,2] as (y, x) int64, scores [n]) in the reference order:
    row-major when nothing is cut, score-descending (ties: lower pixel index first) when top-k applies."""
    H, W = score_map.shape
    kp = torch.nonzero(score_map > threshold)
    sc = score_map[tuple(kp.t())]
    m = (kp[:, 0] >= border) & (kp[:, 0] < H - border) & (kp[:, 1] >= border) & (kp[:, 1] < W - border)
    kp, sc = kp[m], sc[m]
    if max_keypoints >= 0 and max_keypoints < len(kp):
        order = torch.argsort(sc, descending=True, stable=True)[:max_keypoints]
        kp, sc = kp[order], sc[order]
    return kp, sc


def forward(sd, image, nms_radius=4, keypoint_threshold=0.005, max_keypoints=-1, remove_borders=4):
    """Upstream SuperPoint.forward: returns dict of lists (one entry per image): keypoints [n,2] (x, y) float,
    scores [n], descriptors [256, n]; plus the dense NMS-ed score maps for the tests."""
    s, d = dense_maps(sd, image)
    s = simple_nms(s, nms_radius)
    d = F.normalize(d, p=2, dim=1)
    out = {"keypoints": [], "scores": [], "descriptors": [], "score_map": s}
    for b in range(image.shape[0]):
        kp, sc = select_keypoints(s[b], keypoint_threshold, remove_borders, max_keypoints)
        kxy = torch.flip(kp, [1]).float()
        out["keypoints"].append(kxy)
        out["scores"].append(sc)
        out["descriptors"].append(sample_descriptors(kxy[None], d[b][None], 8)[0])
    return out
