"""End-to-end evaluation flow of eval_pairs.py:207-278 (mode w8pt_ba) at batch 1 on synthetic pairs (-m gpu):
matcher -> keep valid matches -> w8pt on the MATCHED keypoints only (variable M) -> two-view BA -> pose error -> AUC,
HIP path vs the oracle chain on identical inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _flow(matcher_fn, w8pt_fn, ba_fn, data, dev):
    out = matcher_fn(data)
    k0, k1 = data["keypoints0"][0], data["keypoints1"][0]
    m = out["matches0_0_1"][0]
    conf = out["conf_scores_0_1"][0, :, 0]
    valid = m > -1
    mk0, mk1, mconf = k0[valid], k1[m[valid]], conf[valid]
    K0, K1 = data["intr0"][:, :3, :3], data["intr1"][:, :3, :3]
    T, info = w8pt_fn(mk0[None], mk1[None], K0, K1, mconf[None], determine_inliers=True)
    if T is None:
        return None, int(valid.sum())
    c = info["confidence"].clone()
    c[torch.logical_not(info["pos_depth_mask"])] = 0.0
    Tr, vb = ba_fn(info["kpts0_norm"], info["kpts1_norm"], c.unsqueeze(-1), T, 10)
    T = T.clone()
    T[vb] = Tr.to(T.dtype)
    return T[0].detach().cpu().double().numpy(), int(valid.sum())


def test_pairwise_eval_flow_matches_oracle_chain(gpu):
    import e2e_multi_view_matching_amd as E
    from e2e_multi_view_matching_amd.metrics import compute_pose_error, pose_auc
    from e2e_multi_view_matching_amd.synthetic import identity_like_state, make_tuples
    from oracle import ba2view as OB, w8pt as OW
    from oracle.matcher import matcher_forward
    torch.manual_seed(0)
    cfg = {"GNN_layers": ["self", "cross"] * 2, "sinkhorn_iterations": 50, "conf_mlp": True}
    shell = identity_like_state(E.MultiViewMatcher(cfg).eval())
    sd = {k: v.clone() for k, v in shell.state_dict().items()}
    model = shell.to(gpu)
    errs_h, errs_o = [], []
    for seed, (n0, n1) in enumerate([(300, 260), (256, 256), (190, 333), (400, 280)]):
        d = make_tuples(batch=1, tuple_size=2, n_kpts=max(n0, n1), seed=50 + seed, noise_px=0.7)
        for m, n in ((0, n0), (1, n1)):
            d[f"keypoints{m}"] = d[f"keypoints{m}"][:, :n].contiguous()
            d[f"scores{m}"] = d[f"scores{m}"][:, :n].contiguous()
            d[f"descriptors{m}"] = d[f"descriptors{m}"][:, :, :n].contiguous()
        dg = {k: (v.to(gpu) if torch.is_tensor(v) else v) for k, v in d.items()}
        Th, nh = _flow(lambda x: model(x), E.estimate_relative_pose_w8pt, E.run_bundle_adjust_2_view, dg, gpu)
        To, no = _flow(lambda x: matcher_forward(x, sd, {**cfg, "full_output": True}), OW.estimate_relative_pose_w8pt,
                       OB.run_bundle_adjust_2_view, d, None)
        assert nh == no and nh > 50  # identical match sets
        Tg = d["T_0to1"][0].double().numpy()
        eh = max(compute_pose_error(Tg, Th[:3, :3], Th[:3, 3]))
        eo = max(compute_pose_error(Tg, To[:3, :3], To[:3, 3]))
        assert abs(eh - eo) < 0.05, (eh, eo)  # degrees; the fp32 oracle chain (SVD + dense LU) is the noisier side
        errs_h.append(eh)
        errs_o.append(eo)
    ah, ao = pose_auc(errs_h, [5, 10, 20]), pose_auc(errs_o, [5, 10, 20])
    assert np.allclose(ah, ao, atol=5e-3) and ah[2] > 0.8
