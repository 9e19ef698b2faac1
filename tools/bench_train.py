#!/usr/bin/env python
"""A training step at full depth (18 layers, 1024 keypoints, 100 Sinkhorn iterations): weights re-commit + forward with a tape +
match loss + pose loss + backward + SGD step, with the inference forward of the same batch beside it (the row of
tools/bench_next_rows.py, alone)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e2e_multi_view_matching_amd as E  # noqa: E402
from e2e_multi_view_matching_amd import synthetic  # noqa: E402


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3


dev = torch.device("cuda:0")
print("| stage | ms |\n|---|---|")
for (Bt, Nt) in ((4, 1024), (8, 1024)):
    cfg = {"GNN_layers": ["self", "cross"] * 9, "sinkhorn_iterations": 100, "conf_mlp": True, "full_output": True, "frozen_batchnorm": True}
    torch.manual_seed(0)
    model = synthetic.identity_like_state(E.MultiViewMatcher(cfg)).to(dev).train()
    opt = torch.optim.SGD(model.parameters(), lr=1e-6)
    data = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synthetic.make_tuples(batch=Bt, tuple_size=2, n_kpts=Nt, seed=5).items()}
    gt = data["gt_matches0_0_1"]
    idx = torch.full((Bt, Nt + 1), Nt, dtype=torch.int64, device=dev)
    idx[:, :Nt] = torch.where(gt >= 0, gt, torch.full_like(gt, Nt))

    def fwd():
        model.zero_grad()
        res = model(data)
        nll = -torch.gather(res["scores_0_1"], 2, idx[:, :, None]).mean()
        pred, _ = E.run_weighted_8_point(data, res, 0, 1, choose_closest=True, target_T_021=data["T_0to1"])
        return nll + E.compute_rotation_error(pred, data["T_0to1"]) + E.compute_translation_error_as_angle(pred, data["T_0to1"])

    def step():
        loss = fwd()
        loss.backward()
        opt.step()

    with torch.no_grad():
        model.eval()
        f_inf = timeit(lambda: model(data), iters=5, warm=2)
        model.train()
    t_fwd = timeit(lambda: fwd(), iters=3, warm=1)
    t_step = timeit(step, iters=3, warm=1)
    print(f"| training step, {Bt} pairs x {Nt}, 18 layers, 100 iterations | {t_step:.3f} |")
    print(f"|   of which forward with the tape + losses | {t_fwd:.3f} |")
    print(f"|   for comparison: inference forward of the same batch | {f_inf:.3f} |", flush=True)
