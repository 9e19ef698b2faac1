"""Seeded random-shape sweep (-m gpu): Sinkhorn, match extraction, the weighted 8-point solve, the GNN building blocks and the
whole matcher on ragged / tiny / odd sizes against the oracle.  Complements the hand-picked cases of the other test files:
the shapes here are drawn, not chosen, so that boundary handling (N % 4, N % 16, N < 64, M != N, single keypoints) is
exercised broadly.  Same bars: scores 1e-4, indices bit-exact, pose 1e-4."""
import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("split_always")]


def _rng(seed):
    return np.random.default_rng(seed)


@pytest.mark.parametrize("seed", range(8))
def test_sinkhorn_and_matches_random_shapes(gpu, seed):
    import e2e_multi_view_matching_amd as E
    from oracle.sinkhorn import extract_matches, log_optimal_transport
    r = _rng(seed)
    for _ in range(4):
        B, M, N = int(r.integers(1, 4)), int(r.integers(1, 330)), int(r.integers(1, 330))
        iters = int(r.choice([0, 1, 7, 30]))
        g = torch.Generator().manual_seed(seed * 100 + M)
        s = torch.randn(B, M, N, generator=g) * float(r.uniform(0.5, 6.0))
        alpha = float(r.uniform(-1.0, 2.0))
        ref = log_optimal_transport(s, alpha, iters)
        out = E.log_optimal_transport(s.to(gpu), alpha, iters).cpu()
        assert out.shape == ref.shape == (B, M + 1, N + 1)
        assert float((out - ref).abs().max()) < 1e-4, (B, M, N, iters)
        thr = float(r.choice([0.0, 0.2]))
        i0, i1, s0, s1 = extract_matches(ref, thr)
        m0, m1, ms0, ms1 = E.extract_matches(ref.to(gpu), thr)
        assert torch.equal(m0.cpu(), i0) and torch.equal(m1.cpu(), i1), (B, M, N)
        # exp() of the arg-max; with 0 / 1 iterations Z is far from normalised and the "scores" can be in the thousands
        for a, b in ((ms0.cpu(), s0), (ms1.cpu(), s1)):
            assert float(((a - b).abs() / b.abs().clamp(min=1.0)).max()) < 2e-6


@pytest.mark.parametrize("seed", range(6))
def test_w8pt_random_shapes(gpu, seed):
    import e2e_multi_view_matching_amd as E
    from oracle import w8pt as OW
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    r = _rng(100 + seed)
    for _ in range(3):
        B, N = int(r.integers(1, 6)), int(r.integers(9, 400))
        kdim = int(r.choice([3, 4]))
        # a synthetic two-view scene (same construction as the golden generator, re-stated here)
        g = torch.Generator().manual_seed(int(r.integers(0, 10 ** 6)))
        X = torch.stack([torch.rand(B, N, generator=g) * 4 - 2, torch.rand(B, N, generator=g) * 4 - 2, torch.rand(B, N, generator=g) * 4 + 3], -1)
        ang = torch.randn(B, 3, generator=g) * 0.15
        Kx = torch.zeros(B, 3, 3)
        Kx[:, 0, 1], Kx[:, 0, 2], Kx[:, 1, 0], Kx[:, 1, 2], Kx[:, 2, 0], Kx[:, 2, 1] = -ang[:, 2], ang[:, 1], ang[:, 2], -ang[:, 0], -ang[:, 1], ang[:, 0]
        R = torch.matrix_exp(Kx)
        t = torch.randn(B, 3, generator=g) * 0.4
        K = torch.eye(kdim).repeat(B, 1, 1)
        K[:, 0, 0] = K[:, 1, 1] = 600.0
        K[:, 0, 2], K[:, 1, 2] = 320.0, 240.0
        p0 = X[..., :2] / X[..., 2:] * 600 + torch.tensor([320.0, 240.0])
        Xc = X @ R.transpose(1, 2) + t[:, None]
        p1 = Xc[..., :2] / Xc[..., 2:] * 600 + torch.tensor([320.0, 240.0]) + torch.randn(B, N, 2, generator=g) * 0.5
        conf = torch.rand(B, N, 1, generator=g)
        conf[torch.rand(B, N, 1, generator=g) < 0.2] = 0.0
        share_K = bool(r.integers(0, 2))
        K0 = K[:1] if share_K else K
        Tr, iref = OW.estimate_relative_pose_w8pt(p0.double(), p1.double(), K0.double(), K0.double(), conf.double(), determine_inliers=True)
        T, info = E.estimate_relative_pose_w8pt(p0.to(gpu), p1.to(gpu), K0.to(gpu), K0.to(gpu), conf.to(gpu), determine_inliers=True)
        assert float((T.cpu().double() - Tr).abs().max()) < 1e-4, (B, N, kdim)
        # decisions equal except inside a stated margin of the fp64 decision boundary (0.015 px of the 3 px threshold,
        # |depth| < 1e-6)
        near = (iref["epi_err"] - iref["epi_thr"]).abs() < 5e-3 * iref["epi_thr"]
        marg = torch.minimum(iref["depth0"].abs(), iref["depth1"].abs()) < 1e-6
        assert bool(((info["inliers"].cpu() == iref["inliers"]) | near | marg).all()), (B, N)
        assert bool(((info["pos_depth_mask"].cpu() == iref["pos_depth_mask"]) | marg).all()), (B, N)


@pytest.mark.parametrize("seed", range(6))
def test_matcher_random_configs(gpu, seed):
    """Whole forward on drawn configurations: tuple size, layer schedule, keypoint counts (ragged per image), fp16 / fp32
    descriptors, joint vs pairwise mode, both arithmetic modes."""
    from e2e_multi_view_matching_amd import MultiViewMatcher
    from e2e_multi_view_matching_amd.synthetic import identity_like_state, make_tuples
    from oracle.matcher import matcher_forward
    r = _rng(200 + seed)
    T = int(r.choice([2, 2, 3]))
    n_layers = int(r.integers(1, 5))
    layers = [str(r.choice(["self", "cross"])) for _ in range(n_layers)]
    N = int(r.integers(3, 260))
    cfg = {"GNN_layers": layers, "sinkhorn_iterations": int(r.choice([3, 15, 40])), "conf_mlp": bool(r.integers(0, 2)), "tuple_size": T,
           "multi_frame_matching": bool(r.integers(0, 2)) if T > 2 else False, "match_threshold": float(r.choice([0.0, 0.2]))}
    torch.manual_seed(seed)
    model = identity_like_state(MultiViewMatcher(cfg).eval()) if r.integers(0, 2) else MultiViewMatcher(cfg).eval()
    f16 = bool(r.integers(0, 2))
    data = make_tuples(batch=int(r.integers(1, 3)), tuple_size=T, n_kpts=N, seed=300 + seed, desc_dtype=torch.float16 if f16 else torch.float32)
    if r.integers(0, 2) and N > 8:  # ragged: image 1 keeps fewer keypoints
        n1 = int(r.integers(2, N))
        for k in ("keypoints1", "scores1"):
            data[k] = data[k][:, :n1].contiguous()
        data["descriptors1"] = data["descriptors1"][:, :, :n1].contiguous()
    rounded = {k: (v.float() if torch.is_tensor(v) and v.dtype == torch.float16 else v) for k, v in data.items()}
    ref = matcher_forward(rounded, model.state_dict(), {**model.config, "full_output": True})
    model = model.to(gpu)
    dev = {k: (v.to(gpu) if torch.is_tensor(v) else v) for k, v in data.items()}
    pairs = [(i, j) for j in range(T) for i in range(j)]
    for precision in ("f32", "bf16x3", "f16x2"):
        model.config["mfma_precision"] = precision
        with torch.no_grad():
            out = model(dev)
        for i, j in pairs:
            z, zr = out[f"scores_{i}_{j}"].cpu(), ref[f"scores_{i}_{j}"]
            assert z.shape == zr.shape and float((z - zr).abs().max()) < 1e-4, (cfg, N, precision, float((z - zr).abs().max()))
            for key in (f"matches{i}_{i}_{j}", f"matches{j}_{i}_{j}"):
                assert torch.equal(out[key].cpu(), ref[key]), (cfg, N, precision, key)
            assert float((out[f"conf_scores_{i}_{j}"].cpu() - ref[f"conf_scores_{i}_{j}"]).abs().max()) < 1e-4
