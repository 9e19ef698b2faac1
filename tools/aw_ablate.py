#!/usr/bin/env python
"""Where the time of attention_p2w goes: the kernel of a measurement build (tools/libe2emv_stamps.bin, -DE2EMV_STAMPS) with parts of
its main loop removed (wrong results, timing only).  `--build` makes the measurement library (no GPU needed); `--pmc` runs
the unablated kernel a few times (for rocprofv3 --pmc)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "libe2emv_stamps.bin")

if "--build" in sys.argv:
    from e2e_multi_view_matching_amd.build import build_library
    print(build_library(defines=["E2EMV_STAMPS"], out=LIB, verbose=True))
    sys.exit(0)

os.environ.setdefault("E2EMV_LIBRARY", LIB)
import torch  # noqa: E402
import e2e_multi_view_matching_amd as E  # noqa: E402
from e2e_multi_view_matching_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
ctx = _lib.context(dev)
B, T, N = 32, 2, 1024
qkv = torch.randn(B * T, N, 768, device=dev) * 1.5
fl = B * T * 4.0 * N * N * 256
if "--pmc" in sys.argv:
    for waves in (1, 8):
        E.attention_p2(qkv, B, T, N, 4, 0, waves=waves, reps=3)
    torch.cuda.synchronize()
    sys.exit(0)
if "--stamps" in sys.argv:
    E.attention_p2(qkv, B, T, N, 4, 0, waves=1, reps=3, abl=0)
    E.attention_p2(qkv, B, T, N, 4, 0, waves=1, reps=1, abl=14)
    E.attention_p2(qkv, B, T, N, 4, 0, waves=1, reps=1, abl=14)
    torch.cuda.synchronize()
    sys.exit(0)
NAMES = {0: "full", 1: "no softmax", 2: "no MFMA", 3: "no softmax, no MFMA", 4: "no fragment reads", 5: "no softmax, no fragment reads",
         8: "no LDS-direct loads", 13: "MFMAs only (no softmax / reads / loads)", 16: "no barrier", 15: "MFMAs only, no barrier", 12: "MFMAs + softmax only (no reads / loads / barrier)",
         11: "MFMAs + softmax fed from a constant only", 10: "softmax only",
         6: "full, never the slow path", 7: "... and no wait for Q / K(0) in the prologue"}
ORDER = (0, 1, 2, 3, 4, 5, 8, 13, 16, 15, 12, 11, 10, 6, 7)
best = {}
for rnd in range(3):  # the clock moves with what ran before: every variant in every round, the minimum counts
    for abl in ORDER:
        E.attention_p2(qkv, B, T, N, 4, 0, waves=1, reps=2, abl=abl)
        ctx.call("e2emv_profile", 1)
        _lib.profile_read(ctx, reset=True)
        E.attention_p2(qkv, B, T, N, 4, 0, waves=1, reps=10, abl=abl)
        pr = _lib.profile_read(ctx, reset=True)["attention"]
        ctx.call("e2emv_profile", 0)
        ms = pr["ms"] / max(pr["launches"], 1)
        best.setdefault(abl, []).append(ms)
for abl in ORDER:
    ms = min(best[abl])
    print(f"abl={abl:2d} {NAMES[abl]:52s} {ms * 1e3:7.1f} us  {fl / ms / 1e9:6.1f} TF-eq   (rounds: {' '.join(f'{m * 1e3:.0f}' for m in best[abl])})", flush=True)
