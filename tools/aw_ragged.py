#!/usr/bin/env python
"""attention_p2w against attention_p2 (8 waves) on RAGGED key counts: time per launch and the number of (wave, stream, tile)
softmaxes the one-wave kernel redid on its slow path (e2emv_get_stats)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import e2e_multi_view_matching_amd as E  # noqa: E402
from e2e_multi_view_matching_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
ctx = _lib.context(dev)
B, T, N = 32, 2, 1024
qkv = torch.randn(B * T, N, 768, device=dev) * 1.5


def fam(fn, n=10):
    for _ in range(2):
        fn(1)
    ctx.call("e2emv_profile", 1)
    _lib.profile_read(ctx, reset=True)
    fn(n)
    pr = _lib.profile_read(ctx, reset=True)["attention"]
    ctx.call("e2emv_profile", 0)
    return pr["ms"] / max(pr["launches"], 1)


for rnd in range(2):
    for nv in (1024, 1000, 992, 961, 960, 936, 896, 520):
        line = f"valid={nv:5d} |"
        for waves in (8, 1):
            ms = fam(lambda n: E.attention_p2(qkv, B, T, nv, 4, 0, waves=waves, reps=n))
            line += f" {'p2w' if waves == 1 else 'p2/8'} {ms * 1e3:7.1f} us |"
        ctx.stats(reset=True)
        E.attention_p2(qkv, B, T, nv, 4, 0, waves=1)
        line += f" slow tiles per launch {ctx.stats(reset=True)['attention_slow_tiles']}"
        print(line, flush=True)
