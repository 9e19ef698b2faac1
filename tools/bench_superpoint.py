#!/usr/bin/env python
"""SuperPoint front-end timing on one MI355X: images/s at 480x640 (the reference's working resolution), 1024 keypoints."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from e2e_multi_view_matching_amd import _lib  # noqa: E402
from e2e_multi_view_matching_amd.superpoint import SuperPoint  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    sp = SuperPoint({"max_keypoints": 1024, "nms_radius": 4, "remove_borders": 4}).eval().to(dev)
    img = torch.rand(a.batch, 1, a.height, a.width, device=dev)
    for _ in range(2):
        sp({"image": [img]})
    torch.cuda.synchronize()
    ctx = _lib.context(dev)
    ctx.call("e2emv_profile", 1)
    t = time.perf_counter()
    for _ in range(a.iters):
        sp({"image": [img]})
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / a.iters
    prof = _lib.profile_read(ctx, reset=True)
    flops = 0
    hw = a.height * a.width
    for cin, cout, k, div in [(1, 64, 3, 1), (64, 64, 3, 1), (64, 64, 3, 4), (64, 64, 3, 4), (64, 128, 3, 16), (128, 128, 3, 16), (128, 128, 3, 64),
                              (128, 128, 3, 64), (128, 256, 3, 64), (256, 65, 1, 64), (128, 256, 3, 64), (256, 256, 1, 64)]:
        flops += 2 * cin * cout * k * k * hw / div
    print(f"batch {a.batch} x {a.height}x{a.width}: {dt * 1e3:.2f} ms/call, {a.batch / dt:.1f} images/s, "
          f"{flops * a.batch / dt / 1e12:.1f} TFLOP/s algorithmic ({flops / 1e9:.1f} GFLOP/image)")
    print("families (ms per call):", {k: round(v["ms"] / a.iters, 3) for k, v in prof.items() if v["launches"]})


if __name__ == "__main__":
    main()
