"""Development aid: per-phase timing of the resident Sinkhorn kernel (E2EMV_SKR_DEBUG timestamps, 100 MHz clock).
Needs the measurement build (`python tools/p2_stamps.py --build`): the release library compiles these knobs out."""
import os
import sys
import time

os.environ.setdefault("E2EMV_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                    "tools", "libe2emv_stamps.bin"))

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e2e_multi_view_matching_amd as E  # noqa: E402


def run(B, N, iters, flags, path):
    os.environ["E2EMV_SKR_FLAGS"] = str(flags)
    s = torch.randn(B, N, N, device="cuda") * 3
    for _ in range(3):
        E.log_optimal_transport(s, 1.0, iters)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        E.log_optimal_transport(s, 1.0, iters)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    os.environ["E2EMV_SKR_DEBUG"] = path
    if os.path.exists(path):
        os.remove(path)
    E.log_optimal_transport(s, 1.0, iters)
    torch.cuda.synchronize()
    del os.environ["E2EMV_SKR_DEBUG"]
    rows = np.loadtxt(path, comments="#").reshape(-1, 9)
    t = rows[:, 2:].reshape(-1, int(rows[:, 1].max()) + 1, 7)  # [it][g][7]
    t = t[2:14]  # steady-state iterations
    d = np.diff(t, axis=2) * 10.0  # ns
    # stamps (thread 0 of every workgroup): [0] iteration start, [1] after the row half-iteration and the wave's column partials,
    # [2] after the barrier, the fold of the 8 waves and the stage-A publish, [3] after the stage-A wait / reduce and the
    # stage-B publish, [4] after wave 0's dustbin-statistic poll, [5] after the stage-B poll, [6] after the closing barrier.
    # (Until round 3 the labels below were shifted by one interval against the stamps.)
    names = ["rows+cols", "barrier+fold+pubA", "stageA wait+reduce+pubB", "dustbin poll (wave 0)", "stageB poll", "closing barrier"]
    per = d.mean(axis=(0, 1))
    it_time = (t[1:, :, 0] - t[:-1, :, 0]).mean() * 10.0
    print(f"B={B} N={N} flags={flags}: {dt * 1e3:.3f} ms/call incl. final+match ({dt / iters * 1e6:.2f} us/iter); "
          f"iteration {it_time / 1e3:.2f} us = " + ", ".join(f"{n} {v / 1e3:.2f}" for n, v in zip(names[:6], per)) +
          f"; spread of iteration start over workgroups {((t[:, :, 0].max(1) - t[:, :, 0].min(1)).mean()) * 10 / 1e3:.2f} us")


if __name__ == "__main__":
    if "--rows128" in sys.argv:  # the 128-row kernel (taken by itself for 32 problems) against the 64-row one
        for mode in ("rows64", "rows128"):
            from e2e_multi_view_matching_amd import _lib
            _lib.context().set_sinkhorn_kernel(mode)
            for B, N in ((1, 1024), (32, 1024), (1, 2048), (8, 2048)):
                print(mode, end=": ")
                run(B, N, 100, 0, "/tmp/skr.txt")
        sys.exit(0)
    for flags in (0, 1):
        for B in (1, 16, 32):
            run(B, 1024, 100, flags, "/tmp/skr.txt")
    run(64, 256, 100, 0, "/tmp/skr.txt")
