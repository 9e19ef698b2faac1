"""Two-view bundle adjustment parity (-m gpu): HIP Schur-complement LM vs the oracle / the reference golden."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_ba_vs_reference_golden_and_fp64_oracle(gpu):
    import e2e_multi_view_matching_amd as E
    from oracle import ba2view as OB
    z = np.load(os.path.join(G, "ba2view_reference.npz"))
    for name in [str(n) for n in z["names"]]:
        t = lambda k: torch.from_numpy(z[f"{name}/{k}"])
        T64, v64 = OB.run_bundle_adjust_2_view(t("kpts0_norm").double(), t("kpts1_norm").double(), t("conf").double(),
                                               t("T_init").double(), 10)
        T, valid = E.run_bundle_adjust_2_view(t("kpts0_norm").to(gpu), t("kpts1_norm").to(gpu), t("conf").to(gpu),
                                              t("T_init").to(gpu), n_iterations=10)
        assert np.array_equal(valid.cpu().numpy(), z[f"{name}/valid"]), name
        noise = torch.from_numpy(z[f"{name}/ref_fp32_noise"])
        d64 = (T.cpu().double() - T64).abs().amax((1, 2))
        dref = (T.cpu() - t("T_refined")).abs().amax((1, 2)).double()
        # fp64 Schur vs fp64 dense LU: same algorithm, different elimination order
        assert bool((d64 < 2e-5 + 0.2 * noise).all()), (name, d64.tolist())
        assert bool((dref < 1e-4 + 2 * noise).all()), (name, dref.tolist(), noise.tolist())


def test_ba_refines_w8pt_pose_on_the_eval_path(gpu):
    """eval_pairs.py:246-255: w8pt -> zero the confidence of non-positive-depth points -> 10 BA iterations."""
    import e2e_multi_view_matching_amd as E
    from e2e_multi_view_matching_amd.synthetic import make_tuples
    d = make_tuples(batch=16, tuple_size=2, n_kpts=512, seed=31, rho=1.0, noise_px=1.0)
    gt = d["gt_matches0_0_1"]
    k0 = d["keypoints0"].to(gpu)
    k1 = d["keypoints1"][torch.arange(16)[:, None], gt].to(gpu)
    K = d["intr0"].to(gpu)
    conf = torch.rand(16, 512, generator=torch.Generator().manual_seed(1)).to(gpu)
    T, info = E.estimate_relative_pose_w8pt(k0, k1, K, K, conf, determine_inliers=True)
    c = info["confidence"].clone()
    c[torch.logical_not(info["pos_depth_mask"])] = 0.0
    Tr, valid = E.run_bundle_adjust_2_view(info["kpts0_norm"], info["kpts1_norm"], c.unsqueeze(-1), T, n_iterations=10)
    assert bool(valid.all()) and Tr.shape == (16, 4, 4)
    Tg = d["T_0to1"].to(gpu)
    r0, t0 = E.pose_errors(T, Tg)
    r1, t1 = E.pose_errors(Tr, Tg)
    assert float(r1.mean()) < float(r0.mean()) and float(t1.mean()) < float(t0.mean())  # BA improves the pose on average
    R = Tr[:, :3, :3]
    assert float((R @ R.transpose(1, 2) - torch.eye(3, device=gpu)).abs().max()) < 1e-5
    T2 = T.clone()
    T2[valid] = Tr  # the reference's write-back idiom (eval_pairs.py:255)
    # zero iterations returns the initial pose; a sample without matches is reported invalid and left untouched
    c[3] = 0
    T0, v0 = E.run_bundle_adjust_2_view(info["kpts0_norm"], info["kpts1_norm"], c, T, n_iterations=0)
    assert not bool(v0[3]) and int(v0.sum()) == 15 and float((T0 - T[v0]).abs().max()) == 0.0
