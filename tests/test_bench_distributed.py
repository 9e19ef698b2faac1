"""bench.py's launch and N > 1 branch on CPU (gloo, world_size 2): `--gpus 2` without a launcher re-executes through
torch.distributed.run, every rank joins the process group, the timing is the MAX over ranks, the per-pair errors are
all-gathered and rank 0 prints ONE line with n_gpus = 2.  The step itself is the stub workload (E2EMV_BENCH_STUB) - the
kernels need a GPU; everything around them is the code the driver's `python bench.py --gpus N` runs."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_gpus_2_self_spawns_two_ranks_and_gathers_the_metric():
    from e2e_multi_view_matching_amd.metrics import pose_auc
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "5", "--tuple-size", "3"],
                       capture_output=True, text=True, timeout=600, env=_env(E2EMV_BENCH_STUB="1"), cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    P = 3
    assert out["config"]["pairs_per_gpu"] == 5 * P and out["config"]["global_pairs"] == 2 * 5 * P
    assert out["auc_pairs"] == 2 * 5 * P
    # the gathered errors are rank 0's then rank 1's (the stub draws them from seed 100 + rank)
    e = np.concatenate([np.random.default_rng(100 + rk).uniform(0, 30, 5 * P) for rk in range(2)]).astype(np.float32)
    assert np.allclose(out["auc_5_10_20"], [100 * a for a in pose_auc(e, [5, 10, 20])], atol=2e-3)
    # MAX over ranks: the stub's rank 1 sleeps 4 ms per step, rank 0 2 ms
    assert out["ms_per_step"] >= 3.9
    assert abs(out["value"] - 2 * 5 * P * 3 / (out["ms_per_step"] * 3e-3)) < 0.02 * out["value"]
    # the label is built from the arguments, and this is not one of the BASELINE configs
    w = out["config"]["workload"]
    assert w.startswith("custom") and "tuple_size=3" in w and "batch 5 tuples/GPU" in w and out["config"]["parallelism"] == "tuple-sharded x2"


def test_world_size_must_match_gpus():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300, env=_env(E2EMV_BENCH_STUB="1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0",
                                             MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"), cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stdout + r.stderr)


def test_workload_label_follows_the_arguments():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse_args([])
    assert bench.workload_string(a, 1).startswith("configs[1]:") and "batch 32 tuples/GPU = 32 pairs/GPU" in bench.workload_string(a, 1)
    assert bench.workload_string(a, 8).startswith("configs[2]")
    a = bench.parse_args(["--config", "c4"])
    assert a.tuple_size == 5 and a.batch == 8 and bench.workload_string(a, 1).startswith("configs[3]:")
    a = bench.parse_args(["--config", "c5"])
    assert a.kpts == 2048 and a.desc == "f16" and "f16 descriptors" in bench.workload_string(a, 1)
    a = bench.parse_args(["--tuple-size", "5"])
    assert bench.workload_string(a, 1).startswith("custom") and "tuple_size=5 (10 pairs per tuple)" in bench.workload_string(a, 1)
    a = bench.parse_args(["--config", "c4", "--gnn", "7x3"])
    assert len(a.layers) == 28 and a.layers[:4] == ["self", "cross", "cross", "cross"]


def test_single_rank_stub_line_is_the_contract():
    r = subprocess.run([sys.executable, BENCH, "--steps", "2", "--warmup", "1", "--batch", "2"], capture_output=True, text=True,
                       timeout=300, env=_env(E2EMV_BENCH_STUB="1"), cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config"):
        assert key in out
    assert out["n_gpus"] == 1 and out["vs_baseline"] is None and out["unit"] == "pairs/s"


def test_gpus_8_world_of_eight_ranks_over_gloo():
    """The driver's `bench.py --gpus 8` shape: eight ranks join one process group, every rank contributes its pairs and its
    clock, rank 0 prints one line with the whole-job aggregate (stub step - the launch, barriers and collectives are real)."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), BENCH, "--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "3"],
                       capture_output=True, text=True, timeout=900, env=_env(E2EMV_BENCH_STUB="1", OMP_NUM_THREADS="1"), cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and out["config"]["global_pairs"] == 8 * 3 and out["auc_pairs"] == 8 * 3
    assert out["config"]["parallelism"] == "tuple-sharded x8"
    assert abs(out["value"] - 8 * 3 * 2 / (out["ms_per_step"] * 2e-3)) < 0.02 * out["value"]


def test_rank_cpu_slices_partition_the_host_cores():
    """bench.pin_cpus: the ranks of a node get disjoint, contiguous slices of the cores the process may use (run in a child:
    the affinity of the test process is left alone)."""
    import subprocess
    import sys
    code = ("import os, sys, json; sys.path.insert(0, %r); import bench; "
            "base = sorted(os.sched_getaffinity(0)); out = []\n"
            "for r in range(4):\n"
            "    os.sched_setaffinity(0, base); m = bench.pin_cpus(r, 4); out.append(sorted(m) if m else None)\n"
            "print(json.dumps({'base': base, 'slices': out}))") % ROOT
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    d = json.loads(r.stdout.strip().splitlines()[-1])
    if len(d["base"]) < 4:
        pytest.skip("fewer than 4 usable cores")
    sl = d["slices"]
    assert all(s for s in sl)
    assert len({c for s in sl for c in s}) == sum(len(s) for s in sl)            # disjoint
    assert all(s == d["base"][i * len(s):(i + 1) * len(s)] for i, s in enumerate(sl))  # contiguous, in rank order
