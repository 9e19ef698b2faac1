#!/usr/bin/env python
"""Batch 1 (the eval_pairs.py loop shape): one pair per call, eager against a captured HIP graph of the same calls."""
import os
import sys
import time

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


def step():
    with torch.no_grad():
        res = model(data)
        poses = E.run_weighted_8_point_tuple(data, res)
        Tp, info = poses[(0, 1)]
        err = E.pose_errors(Tp, data["T_0to1"])
    return res, Tp, err


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def fwd_only():
    with torch.no_grad():
        return model(data)


print(f"eager: step {timed(step):.3f} ms   forward only {timed(fwd_only):.3f} ms", flush=True)
# host time of one call (no synchronisation inside): enqueue cost
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    fwd_only()
host = (time.perf_counter() - t) / 20 * 1e3
torch.cuda.synchronize()
print(f"host enqueue time of forward(): {host:.3f} ms", flush=True)

for name, fn in (("forward only", fwd_only), ("step", step)):
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode="relaxed"):
            out = fn()
        torch.cuda.synchronize()
        print(f"graph replay, {name}: {timed(g.replay):.3f} ms", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"graph capture of {name} failed: {type(e).__name__}: {str(e)[:300]}", flush=True)
        torch.cuda.synchronize()
