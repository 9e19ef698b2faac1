#!/usr/bin/env python
"""20 batch-1 steps (one pair per call) for a rocprofv3 kernel trace."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import e2e_multi_view_matching_amd as E  # noqa: E402
from e2e_multi_view_matching_amd.synthetic import make_tuples  # noqa: E402

dev = torch.device("cuda", 0)
cfg = {"GNN_layers": ["self", "cross"] * 9, "sinkhorn_iterations": 100, "conf_mlp": True, "tuple_size": 2, "multi_frame_matching": False, "match_threshold": 0.2}
torch.manual_seed(1234)
model = E.MultiViewMatcher(cfg).eval().to(dev)
data = make_tuples(batch=1, tuple_size=2, n_kpts=1024, seed=1000)
data = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data.items()}
for _ in range(20):
    with torch.no_grad():
        res = model(data)
        poses = E.run_weighted_8_point_tuple(data, res)
        Tp, info = poses[(0, 1)]
        err = E.pose_errors(Tp, data["T_0to1"])
torch.cuda.synchronize()
