"""Two ranks, two processes, two library contexts running the real kernels (-m gpu): `bench.py --gpus 2` through
torch.distributed.run.  With two or more GPUs this is the driver's launch (one rank per GPU over RCCL); on a 1-GPU box the
ranks share the GPU (E2EMV_BENCH_SHARE_GPU, metric collectives over gloo) - still two processes each owning a context, its
weights, workspace and streams, with the barrier / MAX-over-ranks / all-gather path around the real step."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = [pytest.mark.gpu]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_two_ranks_real_kernels(gpu):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    extra = []
    if torch.cuda.device_count() < 2:
        env["E2EMV_BENCH_SHARE_GPU"] = "1"
        extra = ["--backend", "gloo"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
           "--kpts", "512", "--cpu-pairs", "0", "--no-alt", "--no-latency"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_pairs"] == 8 and out["auc_pairs"] == 8
    assert out["value"] > 0 and out["range"]["fallbacks"] == 0
    # both ranks ran the identity-like model on their own tuples (seed 1000 + rank): matches are real, poses are good
    assert out["auc_5_10_20"][2] > 50.0, out["auc_5_10_20"]
    # ... and the gathered metric equals what ONE process / one context computes over rank 0's then rank 1's tuples
    sys.path.insert(0, ROOT)
    import bench
    import numpy as np
    from e2e_multi_view_matching_amd.metrics import pose_auc
    args = bench.parse_args(["--steps", "2", "--warmup", "1", "--batch", "4", "--kpts", "512"])
    errs = []
    for rk in range(2):
        wl = bench.HipWorkload(args, rk, 0)
        wl.setup(rk)
        errs.append(wl.auc_errors())
    ref = [100.0 * a for a in pose_auc(np.concatenate(errs), [5, 10, 20])]
    assert np.allclose(out["auc_5_10_20"], ref, atol=2e-3), (out["auc_5_10_20"], ref)
