#!/usr/bin/env python
"""gemm_p3 (bf16x3 on P3 planes, round 6) against gemm_x3 (bf16x3, fp32 activations split in the K loop) and gemm_p2 (f16x2 planes):
the kernels alone at the layer shapes of configs[1] (HIP events inside the library, profile slot "gemm")."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import e2e_multi_view_matching_amd as E
from e2e_multi_view_matching_amd import _lib
dev = torch.device("cuda", 0)
ctx = _lib.context(dev)
for (M, N, K) in [(65536, 768, 256), (65536, 512, 512), (65536, 256, 512), (65536, 256, 256)]:
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    row = f"  {M:6d} {N:4d} {K:4d}:"
    for name, kw in (("gemm_x3", {}), ("gemm_p3 f32 out", dict(p3=True, reps=10)), ("gemm_p3 planes out", dict(p3=True, planes_out=True, reps=10))):
        for _ in range(2):
            E.gemm_bf16x3(A, W, **kw)
        ctx.call("e2emv_profile", 1)
        _lib.profile_read(ctx, reset=True)
        n = 1 if "reps" in kw else 10
        for _ in range(n):
            E.gemm_bf16x3(A, W, **kw)
        pr = _lib.profile_read(ctx, reset=True)["gemm"]
        ctx.call("e2emv_profile", 0)
        ms = pr["ms"] / 10
        row += f"   {name} {ms * 1e3:7.1f} us ({2.0 * M * N * K / ms / 1e9:6.1f} TF-eq)"
    print(row, flush=True)
