#!/usr/bin/env python
"""Experiment: one batch of 32 pairs as TWO half batches on two HIP streams (two library contexts, two model copies) against
the single call - does the second stream fill the ramp / epilogue / tail bubbles of the first one's launches and the waits of
the exchange-bound Sinkhorn?  forward() only (GNN + Sinkhorn + matches)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import e2e_multi_view_matching_amd as E  # noqa: E402
from e2e_multi_view_matching_amd import _lib, matcher  # noqa: E402
from e2e_multi_view_matching_amd.synthetic import make_tuples  # noqa: E402

dev = torch.device("cuda", 0)
B, T, N = 32, 2, 1024
cfg = {"GNN_layers": ["self", "cross"] * 9, "sinkhorn_iterations": 100, "conf_mlp": True, "tuple_size": T, "multi_frame_matching": False, "match_threshold": 0.2}
torch.manual_seed(1234)
mA = E.MultiViewMatcher(cfg).eval().to(dev)
mB = E.MultiViewMatcher(cfg).eval().to(dev)
mB.load_state_dict(mA.state_dict())
data = make_tuples(batch=B, tuple_size=T, n_kpts=N, seed=1000)
data = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data.items()}


def part(d, lo, hi):
    return {k: (v[lo:hi].contiguous() if torch.is_tensor(v) and v.shape[:1] == (B,) else v) for k, v in d.items()}


ctx1 = _lib.context(dev)
ctx2 = _lib.Context(dev.index)
real_context = _lib.context


def run(model, ctx, d):
    matcher._lib.context = lambda *_a, **_k: ctx
    try:
        with torch.no_grad():
            return model(d)
    finally:
        matcher._lib.context = real_context


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for nsplit in (2, 4):
    h = B // nsplit
    parts = [part(data, i * h, (i + 1) * h) for i in range(nsplit)]

    def split():
        for i, p in enumerate(parts):
            with torch.cuda.stream(s1 if i % 2 == 0 else s2):
                run(mA if i % 2 == 0 else mB, ctx1 if i % 2 == 0 else ctx2, p)

    def serial():
        for i, p in enumerate(parts):
            run(mA, ctx1, p)

    ms_full = timed(lambda: run(mA, ctx1, data))
    ms_serial = timed(serial)
    ms_split = timed(split)
    print(f"{nsplit} parts of {h} pairs: one call {ms_full:.3f} ms | parts one after the other {ms_serial:.3f} ms | on two streams {ms_split:.3f} ms", flush=True)
