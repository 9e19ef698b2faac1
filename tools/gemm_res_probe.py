#!/usr/bin/env python
"""gemm_p2 at the layer shapes with and without the residual / ReLU / tile exponents: what each costs a launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import e2e_multi_view_matching_amd as E  # noqa: E402
from e2e_multi_view_matching_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
ctx = _lib.context(dev)


def fam(fn, n=10):
    for _ in range(2):
        fn(1)
    ctx.call("e2emv_profile", 1)
    _lib.profile_read(ctx, reset=True)
    fn(n)
    pr = _lib.profile_read(ctx, reset=True)["gemm"]
    ctx.call("e2emv_profile", 0)
    return pr["ms"] / max(pr["launches"], 1) * 1e3


M = 65536
for rnd in range(2):
    for (N, K, K1) in [(256, 512, 512), (512, 512, 256)]:
        A = torch.randn(M, K1, device=dev)
        A2 = torch.randn(M, K - K1, device=dev) if K1 < K else None
        W = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        R = torch.randn(M, N, device=dev)
        line = f"{M} x {N} x {K}:"
        for name, kw in [("plain", {}), ("bias", dict(bias=b)), ("bias+relu", dict(bias=b, relu=True)), ("bias+exp", dict(bias=b, exponents=True)),
                         ("bias+res", dict(bias=b, residual=R))]:
            us = fam(lambda n: E.gemm_p2(A, W, A2=A2, planes_out=True, reps=n, **kw))
            line += f"  {name} {us:6.1f}"
        print(line, flush=True)
