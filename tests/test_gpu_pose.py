"""w8pt parity (-m gpu): HIP pose kernels vs the oracle (fp32 = reference arithmetic, fp64 = truth)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(B, N, seed, noise=0.5, outlier_frac=0.0, kdim=4):
    from e2e_multi_view_matching_amd.synthetic import make_tuples
    d = make_tuples(batch=B, tuple_size=2, n_kpts=N, seed=seed, rho=1.0, noise_px=noise)
    gt = d["gt_matches0_0_1"]
    bidx = torch.arange(B)[:, None]
    k0, k1 = d["keypoints0"], d["keypoints1"][bidx, gt]
    g = torch.Generator().manual_seed(seed)
    conf = torch.rand(B, N, generator=g)
    if outlier_frac:
        bad = torch.rand(B, N, generator=g) < outlier_frac
        conf = torch.where(bad, torch.zeros_like(conf), conf)
        k1 = torch.where(bad[..., None], torch.rand(B, N, 2, generator=g) * 400, k1)
    K = d["intr0"][:, :kdim, :kdim].contiguous()
    return k0, k1, K, K.clone(), conf, d["T_0to1"]


@pytest.mark.parametrize("B,N,seed", [(4, 256, 0), (2, 1024, 1), (3, 8, 2), (1, 77, 3)])
@pytest.mark.parametrize("closest", [False, True])
def test_w8pt_vs_oracle(gpu, B, N, seed, closest):
    import e2e_multi_view_matching_amd as E
    from oracle import w8pt as O
    k0, k1, K0, K1, conf, Tgt = _scene(B, N, seed, outlier_frac=0.2 if N > 8 else 0.0)
    T32, i32 = O.estimate_relative_pose_w8pt(k0, k1, K0, K1, conf, closest, Tgt, True)
    T64, i64 = O.estimate_relative_pose_w8pt(k0.double(), k1.double(), K0.double(), K1.double(), conf.double(), closest,
                                             Tgt.double(), True)
    T, info = E.estimate_relative_pose_w8pt(k0.to(gpu), k1.to(gpu), K0.to(gpu), K1.to(gpu), conf.to(gpu),
                                            choose_closest=closest, T_021=Tgt.to(gpu), determine_inliers=True)
    T = T.cpu()
    assert float((T.double() - T64).abs().max()) < 2e-5, float((T.double() - T64).abs().max())
    assert float((T - T32).abs().max()) < 1e-4 + 2 * float((T32.double() - T64).abs().max())
    assert float((info["kpts0_norm"].cpu() - i32["kpts0_norm"]).abs().max()) < 1e-6
    assert float((info["confidence"].cpu() - i32["confidence"]).abs().max()) < 1e-6
    # masks: identical except where the fp64 oracle itself is within rounding of the decision boundary
    margin = torch.minimum(i64["depth0"].abs(), i64["depth1"].abs()) < 1e-6
    assert bool(((info["pos_depth_mask"].cpu() == i64["pos_depth_mask"]) | margin).all())
    # inliers = pos_depth & (epipolar error <= thr): may differ from fp64 only where the fp64 error is within 5e-3 of
    # the threshold (0.015 px at the 3 px threshold - the effect of fp32 normalised coordinates + 1e-5 relative F) or
    # where the depth decision itself is marginal
    near = (i64["epi_err"] - i64["epi_thr"]).abs() < 5e-3 * i64["epi_thr"]
    assert bool(((info["inliers"].cpu() == i64["inliers"]) | near | margin).all())
    Fh = info["F"].cpu().double()
    F64 = i64["F"]
    # N == 8 selects the smallest NON-null singular vector (thin-SVD quirk): conditioned by sigma7/sigma8
    assert float((Fh - F64).abs().max() / F64.abs().max()) < (1e-5 if N > 8 else 1e-4)


def test_w8pt_fewer_than_8_points_is_none(gpu):
    import e2e_multi_view_matching_amd as E
    z = torch.zeros(1, 7, 2, device=gpu)
    assert E.estimate_relative_pose_w8pt(z, z, torch.eye(3, device=gpu)[None], torch.eye(3, device=gpu)[None],
                                         torch.ones(1, 7, device=gpu)) == (None, None)


def test_3x3_broadcast_intrinsics_and_conf_shapes(gpu):
    import e2e_multi_view_matching_amd as E
    from oracle import w8pt as O
    k0, k1, K0, K1, conf, _ = _scene(1, 300, 5, kdim=3)
    T32, i32 = O.estimate_relative_pose_w8pt(k0, k1, K0, K1, conf.unsqueeze(-1), determine_inliers=True)
    T, info = E.estimate_relative_pose_w8pt(k0.to(gpu), k1.to(gpu), K0.to(gpu), K1.to(gpu), conf.unsqueeze(-1).to(gpu),
                                            determine_inliers=True)
    assert info["confidence"].shape == (1, 300, 1)
    assert float((T.cpu() - T32).abs().max()) < 1e-4


def test_run_weighted_8_point_gathers_like_the_reference(gpu):
    """-1 matches wrap to the last keypoint with zero weight (estimate_relative_pose.py:26-30)."""
    import e2e_multi_view_matching_amd as E
    from oracle import w8pt as O
    from e2e_multi_view_matching_amd.synthetic import make_tuples
    d = make_tuples(batch=2, tuple_size=2, n_kpts=200, seed=8, rho=0.8)
    res = {"matches0_0_1": d["gt_matches0_0_1"], "conf_scores_0_1": torch.rand(2, 200, 1)}
    T32, _ = O.run_weighted_8_point(d, res, 0, 1, choose_closest=True, target_T_021=d["T_0to1"])
    dg = {k: (v.to(gpu) if torch.is_tensor(v) else v) for k, v in d.items()}
    rg = {k: v.to(gpu) for k, v in res.items()}
    T, info = E.run_weighted_8_point(dg, rg, 0, 1, choose_closest=True, target_T_021=dg["T_0to1"])
    assert float((T.cpu() - T32).abs().max()) < 1e-4
    assert E.run_weighted_8_point(dg, {}, 0, 1) == (None, None)
    r_err = E.compute_rotation_error(T, dg["T_0to1"], reduce=False).cpu()
    assert float((r_err - O.compute_rotation_error(T.cpu(), d["T_0to1"], reduce=False)).abs().max()) < 2e-3


def test_degenerate_zero_confidence_sets_status(gpu):
    import e2e_multi_view_matching_amd as E
    k0, k1, K0, K1, conf, _ = _scene(2, 64, 6)
    conf[1] = 0
    T, info = E.estimate_relative_pose_w8pt(k0.to(gpu), k1.to(gpu), K0.to(gpu), K1.to(gpu), conf.to(gpu))
    st = info["status"].cpu()
    assert int(st[0]) == 0 and int(st[1]) & 1


def test_degenerate_pair_without_matches_stays_finite(gpu):
    """A pair without a single match (all matches -1): the reference's fancy indexing gathers the LAST keypoint for every
    row and gives it weight 0 (estimate_relative_pose.py:26-30) - identical points, zero weights, an essential matrix of rank
    < 2.  Its library SVDs still return a finite (arbitrary) pose; so must the device path (status bit 2 flags the case)."""
    import e2e_multi_view_matching_amd as E
    from oracle import w8pt as OW
    g = torch.Generator().manual_seed(0)
    B, N = 2, 64
    k0 = torch.rand(B, N, 2, generator=g) * 300
    k1 = torch.tensor([123.0, 77.0]).expand(B, N, 2).contiguous()
    K = torch.eye(4).repeat(B, 1, 1)
    K[:, 0, 0] = K[:, 1, 1] = 300.0
    K[:, 0, 2], K[:, 1, 2] = 160.0, 120.0
    conf = torch.zeros(B, N, 1)
    Tr, _ = OW.estimate_relative_pose_w8pt(k0, k1, K, K, conf)
    assert torch.isfinite(Tr).all()
    T, info = E.estimate_relative_pose_w8pt(k0.to(gpu), k1.to(gpu), K.to(gpu), K.to(gpu), conf.to(gpu), determine_inliers=True)
    assert torch.isfinite(T).all() and (info["status"] & 4).all() and (info["status"] & 1).all()
    R = T[:, :3, :3].cpu().double()
    assert (R @ R.transpose(1, 2) - torch.eye(3, dtype=torch.float64)).abs().max() < 1e-6  # still a rotation
    assert (T[:, :3, 3].norm(dim=1).cpu() - 1).abs().max() < 1e-5                            # unit translation
