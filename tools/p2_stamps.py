#!/usr/bin/env python
"""Where the time of gemm_p2 goes: runs the plane GEMM of a measurement build (tools/libe2emv_stamps.bin, -DE2EMV_STAMPS) with
its ablation variants (E2EMV_P2_DBG: 1 no MFMA, 2 no operand loads, 4 no epilogue, 16 loads spread over the MFMA groups,
8 in-kernel timestamps).  `--build` makes the measurement library (no GPU needed); without it every variant runs in its own
process on the GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "libe2emv_stamps.bin")

if "--build" in sys.argv:
    from e2e_multi_view_matching_amd.build import build_library
    print(build_library(defines=["E2EMV_STAMPS"], out=LIB, verbose=True))
    sys.exit(0)

if "--one" in sys.argv:
    import torch
    import e2e_multi_view_matching_amd as E
    from e2e_multi_view_matching_amd import _lib
    dev = torch.device("cuda", 0)
    ctx = _lib.context(dev)
    dbg = int(os.environ.get("E2EMV_P2_DBG", "0"))
    for (M, N, K, K1) in [(65536, 768, 256, 256), (65536, 512, 512, 256), (65536, 256, 512, 512)]:
        A = torch.randn(M, K1, device=dev)
        A2 = torch.randn(M, K - K1, device=dev) if K1 < K else None
        W = torch.randn(N, K, device=dev) / K ** 0.5
        n = 1 if dbg & 8 else 10
        E.gemm_p2(A, W, A2=A2, planes_out=True, reps=2)
        ctx.call("e2emv_profile", 1)
        _lib.profile_read(ctx, reset=True)
        E.gemm_p2(A, W, A2=A2, planes_out=True, reps=n)
        pr = _lib.profile_read(ctx, reset=True)["gemm"]
        ctx.call("e2emv_profile", 0)
        ms = pr["ms"] / max(pr["launches"], 1)
        print(f"dbg={dbg:2d}  {M} {N} {K}: {ms * 1e3:7.1f} us  {2.0 * M * N * K / ms / 1e9:6.1f} TF-eq", flush=True)
    sys.exit(0)

for dbg in [0, 4, 1, 2, 64, 128]:
    env = dict(os.environ, E2EMV_LIBRARY=LIB, E2EMV_P2_DBG=str(dbg))
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    print(r.stdout[-6000:] if dbg & 8 else r.stdout, flush=True)
