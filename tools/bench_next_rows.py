#!/usr/bin/env python
"""Timings of the SURVEY 8(f) "next" rows on one MI355X (not part of bench.py's headline metric):
two-view BA, ground-truth-match builder + match loss, multi-view back-end stages.  Prints a markdown table."""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e2e_multi_view_matching_amd as E  # noqa: E402
from e2e_multi_view_matching_amd import multi_view, synthetic  # noqa: E402


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    rows = []
    # ---- two-view pose + BA at config-2 shape
    B, N = 32, 1024
    d = synthetic.make_tuples(batch=B, tuple_size=2, n_kpts=N, seed=1)
    k0, k1 = d["keypoints0"].to(dev), d["keypoints1"].to(dev)
    gt = d["gt_matches0_0_1"].to(dev)
    k1g = torch.gather(k1, 1, gt.clamp(min=0).unsqueeze(-1).expand(-1, -1, 2))
    conf = (gt >= 0).float().unsqueeze(-1)
    K = d["intr0"].to(dev)
    T, info = E.estimate_relative_pose_w8pt(k0, k1g, K, K, conf, determine_inliers=True)
    rows.append(("w8pt, 32 pairs x 1024 matches", timeit(lambda: E.estimate_relative_pose_w8pt(k0, k1g, K, K, conf, determine_inliers=True))))
    rows.append(("two-view LM BA (10 iterations), 32 pairs x 1024", timeit(
        lambda: E.run_bundle_adjust_2_view(info["kpts0_norm"], info["kpts1_norm"], info["confidence"], T, n_iterations=10))))
    # ---- validation targets
    dp = {k: v.to(dev) for k, v in synthetic.make_depth_pairs(B, n_kpts=N, seed=2).items()}
    f = lambda: E.compute_gt_matches_of_image_pair(dp["keypoints0"], dp["keypoints1"], dp["intr0"], dp["intr1"], dp["T_0to1"],  # noqa: E731
                                                   dp["depth0"], dp["depth1"], 5.0, 15.0)
    rows.append(("GT-match builder, 32 pairs x 1024 x 1024 (480x640 depth)", timeit(f)))
    # ---- multi-view back-end, one 5-tuple of 1024 keypoints (eval_multi_view.py shape)
    Tn = 5
    cfg = {"GNN_layers": ["self", "cross"] * 2, "sinkhorn_iterations": 50, "multi_frame_matching": True, "tuple_size": Tn}
    model = synthetic.identity_like_state(E.MultiViewMatcher(cfg).eval()).to(dev)
    data = synthetic.make_tuples(batch=1, tuple_size=Tn, n_kpts=1024, seed=3)
    dd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data.items()}
    for m in range(Tn):
        dd[f"pose{m}"] = torch.linalg.inv(data[f"pose{m}"])
        dd[f"intr{m}"] = data[f"intr{m}"]
    with torch.no_grad():
        res = model(dd)
    tmp = tempfile.mkdtemp()
    st = {}

    def stage(name, fn, iters=5):
        st[name] = timeit(fn, iters=iters, warm=1)
    pw = multi_view.initialize_bundle_adjust(Tn, dd, res, os.path.join(tmp, "ba_init_in.csv"))
    stage("initialize_bundle_adjust: 10 x (w8pt + two-view BA) + spanning tree + CSV", lambda: multi_view.initialize_bundle_adjust(Tn, dd, res, os.path.join(tmp, "ba_init_in.csv")))
    stage("ba_initializer: rotation averaging + LUD (host C++)", lambda: multi_view.run_ba_initializer(tmp), iters=20)
    extr = np.array(multi_view.read_bundle_adjust_result(os.path.join(tmp, "ba_init_out.csv")))
    stage("write_bundle_adjust_problem: device DLT + CSV text", lambda: multi_view.write_bundle_adjust_problem(Tn, pw, extr, os.path.join(tmp, "ba_in.csv")))
    stage("bundle_adjuster: CSV parse + device LM/Schur + CSV", lambda: multi_view.run_bundle_adjuster(tmp))
    from oracle import mvba
    prob = mvba.read_problem(os.path.join(tmp, "ba_in.csv"))
    f = lambda: multi_view.bundle_adjust(prob["n_cams"], prob["fixed"], prob["intr"], prob["cam_idx"], prob["pt_idx"], prob["obs"],  # noqa: E731
                                         prob["wts"], prob["cams"], prob["pts"])
    _, _, summ = f()
    stage(f"  of which e2emv_mv_bundle_adjust ({len(prob['pts'])} points, {len(prob['obs'])} observations, {summ['iterations']} LM iterations, H2D/D2H included)", f)
    for k, v in st.items():
        rows.append((k, v))
    # ---- a training step: forward with a tape + match loss + pose loss + backward, full-depth network
    for (Bt, Nt) in ((4, 1024), (8, 1024)):
        cfg = {"GNN_layers": ["self", "cross"] * 9, "sinkhorn_iterations": 100, "conf_mlp": True, "full_output": True, "frozen_batchnorm": True}
        torch.manual_seed(0)
        model = synthetic.identity_like_state(E.MultiViewMatcher(cfg)).to(dev).train()
        opt = torch.optim.SGD(model.parameters(), lr=1e-6)  # (a real step: the parameters change, the next forward re-commits them)
        data = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synthetic.make_tuples(batch=Bt, tuple_size=2, n_kpts=Nt, seed=5).items()}
        gt = data["gt_matches0_0_1"]
        idx = torch.full((Bt, Nt + 1), Nt, dtype=torch.int64, device=dev)
        idx[:, :Nt] = torch.where(gt >= 0, gt, torch.full_like(gt, Nt))

        def step():
            model.zero_grad()
            res = model(data)
            nll = -torch.gather(res["scores_0_1"], 2, idx[:, :, None]).mean()
            pred, _ = E.run_weighted_8_point(data, res, 0, 1, choose_closest=True, target_T_021=data["T_0to1"])
            loss = nll + E.compute_rotation_error(pred, data["T_0to1"]) + E.compute_translation_error_as_angle(pred, data["T_0to1"])
            loss.backward()
            opt.step()
        with torch.no_grad():
            model.eval()
            f_inf = timeit(lambda: model(data), iters=5, warm=2)
            model.train()
        rows.append((f"training step (weights re-commit + fwd with tape + match & pose loss + backward + SGD step), {Bt} pairs x {Nt}, 18 layers, 100 iterations", timeit(step, iters=3, warm=1)))
        rows.append((f"  for comparison: inference forward of the same batch", f_inf))
    print("| stage | ms |\n|---|---|")
    for k, v in rows:
        print(f"| {k} | {v:.3f} |")


if __name__ == "__main__":
    main()
