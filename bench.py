#!/usr/bin/env python
"""bench.py - image-pairs/sec of the matcher -> Sinkhorn -> weighted-8-point path on MI355X.

One step = one pass of the hot path over one batch of synthetic tuples, inputs resident in
HBM: MultiViewMatcher.forward (kenc, GNN, final_proj, scores, Sinkhorn, match block, conf
head) -> weighted 8-point pose of every pair of every tuple -> per-pair pose errors.

Workloads (BASELINE.json `configs`, selected with --config, every field overridable):
  c2 (default)  tuple_size 2, 1024 keypoints, 9x(self,cross), 100 Sinkhorn iterations, batch 32 pairs   = configs[1]
  c4            tuple_size 5 (10 pairs per tuple), 1024 keypoints, batch 8 tuples                          = configs[3]
  c5            tuple_size 5, 2048 keypoints, fp16 descriptors, batch 8 tuples per GPU                     = configs[4] per GPU
  c1            tuple_size 2, 256 keypoints, batch 4, 5 Sinkhorn iterations                                = configs[0]
configs[2] is c2 on 8 GPUs: `--gpus 8` (32 pairs per GPU, 256 in total).

Multi-GPU: one process per GPU (reference launch: torch.distributed.launch --nproc_per_node=k, README.md:107,
train.py:270-277).  `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes itself
through `python -m torch.distributed.run --nproc-per-node N`; under a launcher it asserts WORLD_SIZE == --gpus.
Every rank runs the same per-GPU batch on its own tuples (weak scaling, no data-path collective); the only
collective is the final all-gather of per-pair pose errors for the AUC (RCCL over xGMI), the counterpart of the
reference's one-element all_reduce (train.py:102-106).

Prints ONE JSON line on rank 0 with `roofline` for the dominant kernel family (HIP events on the launch stream
inside the timed region) and `cpu_baseline` (the torch-CPU oracle on a bounded sample, rank 0, N=1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA
PEAK_HBM_GBS = 8000.0

CONFIGS = {
    "c1": dict(tuple_size=2, kpts=256, batch=4, sinkhorn_iters=5, desc="f32"),
    "c2": dict(tuple_size=2, kpts=1024, batch=32, sinkhorn_iters=100, desc="f32"),
    "c4": dict(tuple_size=5, kpts=1024, batch=8, sinkhorn_iters=100, desc="f32"),
    "c5": dict(tuple_size=5, kpts=2048, batch=8, sinkhorn_iters=100, desc="f16"),
}
CONFIG_LABEL = {"c1": "configs[0]", "c2": "configs[1]", "c4": "configs[3]", "c5": "configs[4] (per-GPU share)"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--batch", type=int, default=None, help="tuples per GPU per step")
    ap.add_argument("--kpts", type=int, default=None)
    ap.add_argument("--tuple-size", type=int, default=None)
    ap.add_argument("--sinkhorn-iters", type=int, default=None)
    ap.add_argument("--desc", choices=["f32", "f16"], default=None, help="descriptor dtype handed to the matcher")
    ap.add_argument("--gnn", default="9x1", help="GNN schedule gxc = (['self'] + ['cross']*c) * g (train.py:263-268): "
                    "9x1 two-view / MegaDepth, 7x3 multi-view ScanNet")
    ap.add_argument("--precision", choices=["default", "f32", "bf16x3", "f16x2"], default="default",
                    help="arithmetic of the dense contractions for the timed region (default = the library default)")
    ap.add_argument("--cpu-pairs", type=int, default=4, help="pairs of the workload timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernel families with HIP events")
    ap.add_argument("--ba", action="store_true", help="also run the two-view bundle adjustment (10 LM iterations) per pair "
                    "inside the step (the reference's default eval mode w8pt_ba); off by default: SURVEY 8(d) defines the "
                    "metric on matcher -> w8pt -> pose errors")
    ap.add_argument("--front-end", action="store_true", help="extra (reported separately, never part of `value`): image-in "
                    "pipeline = SuperPoint on tuple_size*batch 480x640 images -> matcher -> w8pt per step")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra measurement in the other arithmetic mode")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 latency measurement (eval_pairs.py's loop shape)")
    ap.add_argument("--baseline-1gpu", type=float, default=None, help="the `value` of the --gpus 1 run of the same config on "
                    "this node: rank 0 then adds scaling_check = {efficiency = value / (n_gpus x it)} to its line (the driver "
                    "computes its own from the per-N lines; this is for a manual 1/2/4/8 sweep)")
    ap.add_argument("--backend", default=None, help="process-group backend (default nccl = RCCL; the CPU test uses gloo)")
    args = ap.parse_args(argv)
    preset = CONFIGS[args.config]
    for key, attr in (("tuple_size", "tuple_size"), ("kpts", "kpts"), ("batch", "batch"), ("sinkhorn_iters", "sinkhorn_iters"),
                      ("desc", "desc")):
        if getattr(args, attr) is None:
            setattr(args, attr, preset[key])
    g, c = (int(v) for v in args.gnn.lower().split("x"))
    args.layers = (["self"] + ["cross"] * c) * g
    return args


def workload_string(args, world):
    """`config.workload`, built from the arguments actually used (never a fixed label)."""
    preset = CONFIGS[args.config]
    exact = all(getattr(args, k) == preset[k] for k in ("tuple_size", "kpts", "batch", "sinkhorn_iters", "desc")) \
        and args.gnn.lower() == "9x1"
    P = args.tuple_size * (args.tuple_size - 1) // 2
    label = CONFIG_LABEL[args.config] if exact else "custom (derived from %s)" % CONFIG_LABEL[args.config]
    if exact and args.config == "c2" and world == 8:
        label = "configs[2] (configs[1] per GPU x 8)"
    return (f"{label}: tuple_size={args.tuple_size} ({P} pair{'s' if P > 1 else ''} per tuple), {args.kpts} keypoints, 256-dim "
            f"{args.desc} descriptors, GNN {args.gnn} = {len(args.layers)} layers, {args.sinkhorn_iters} Sinkhorn iters, "
            f"batch {args.batch} tuples/GPU = {args.batch * P} pairs/GPU, w8pt pose per pair")


# ----------------------------------------------------------------------------------------- launch
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def maybe_self_spawn(args, argv):
    """--gpus N > 1 without a launcher: become the launcher (one rank per GPU, rendezvous on 127.0.0.1)."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return None
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


# ----------------------------------------------------------------------------------------- flops model
def algorithmic_flops(B, T, N, D, layers, conf_mlp):
    """Per-step dense flops by kernel family (SURVEY.md 8(d) formula, joint GNN)."""
    kenc = [3, 32, 64, 128, 256, D]
    n_img = B * T
    gemm = n_img * 2 * N * sum(a * b for a, b in zip(kenc[1:-1], kenc[2:]))  # layers 1.. through the GEMM kernel
    attn = 0
    for name in layers:
        n_src = N if (name == "self" or T == 2) else (T - 1) * N
        gemm += n_img * 20 * N * D * D
        attn += n_img * 4 * N * n_src * D
    gemm += n_img * 2 * N * D * D  # final_proj
    P = T * (T - 1) // 2
    score = B * P * 2 * N * N * D
    if conf_mlp:
        gemm += B * P * 2 * N * (2 * D) * D
    return {"gemm": gemm, "attention": attn, "score_gemm": score}


# ----------------------------------------------------------------------------------------- workloads
def pin_cpus(local_rank, local_world):
    """Per-rank CPU affinity: the host cores this process may use are cut into `local_world` contiguous slices (a contiguous
    slice of the id space stays inside one socket / NUMA node on the two-socket hosts these boxes are) and the rank keeps slice
    LOCAL_RANK - launch threads of 8 ranks then do not migrate across sockets.  E2EMV_BENCH_NO_PIN=1 turns it off; a host
    without sched_setaffinity is left alone.  Returns the cpu set (for the log)."""
    if local_world <= 1 or os.environ.get("E2EMV_BENCH_NO_PIN") or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        cpus = sorted(os.sched_getaffinity(0))
        per = len(cpus) // local_world
        if per < 1:
            return None
        mine = set(cpus[local_rank * per:(local_rank + 1) * per])
        os.sched_setaffinity(0, mine)
        return mine
    except OSError:
        return None


class HipWorkload:
    """The product path on this rank's GPU."""

    def __init__(self, args, rank, local_rank):
        import torch
        assert torch.cuda.is_available(), "bench.py needs an MI355X (the HIP path has no CPU fallback)"
        # E2EMV_BENCH_SHARE_GPU=1 (tests only, never a reported number): ranks beyond the box's GPUs share them round-robin,
        # each with its own process and library context, so the N > 1 branch runs the real kernels on a 1-GPU box
        n_dev = torch.cuda.device_count()
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
        if os.environ.get("E2EMV_BENCH_SHARE_GPU"):
            local_rank %= n_dev
        else:
            # one process per GPU, rank -> device by LOCAL_RANK among the devices this process can see (whatever
            # HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES left visible): fail here, with the numbers, not inside RCCL
            assert n_dev >= local_world and local_rank < n_dev, (
                f"bench.py: rank with LOCAL_RANK={local_rank} of {local_world} local ranks sees {n_dev} GPU(s) "
                f"(HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES')!r}): one GPU per local rank is needed")
        torch.cuda.set_device(local_rank)
        pin_cpus(local_rank, local_world)
        self.torch = torch
        self.dev = self.coll_dev = torch.device("cuda", local_rank)
        self.args = args

    def setup(self, rank):
        torch = self.torch
        import e2e_multi_view_matching_amd as E
        from e2e_multi_view_matching_amd import _lib
        from e2e_multi_view_matching_amd.synthetic import identity_like_state, make_tuples
        a = self.args
        self.E, self._lib = E, _lib
        B, T, N = a.batch, a.tuple_size, a.kpts
        self.cfg = {"GNN_layers": a.layers, "sinkhorn_iterations": a.sinkhorn_iters, "conf_mlp": True, "tuple_size": T,
                    "multi_frame_matching": T > 2, "match_threshold": 0.2}
        self.pairs = [(i, j) for j in range(T) for i in range(j)]
        torch.manual_seed(1234)
        self.model = E.MultiViewMatcher(self.cfg).eval().to(self.dev)           # W-rand: timed
        torch.manual_seed(1234)
        self.model_id = identity_like_state(E.MultiViewMatcher(self.cfg).eval()).to(self.dev)  # W-id: AUC leg
        dt = torch.float16 if a.desc == "f16" else torch.float32
        self.data_cpu = make_tuples(batch=B, tuple_size=T, n_kpts=N, seed=1000 + rank, desc_dtype=dt)
        self.data = {k: (v.to(self.dev) if torch.is_tensor(v) else v) for k, v in self.data_cpu.items()}
        self.ctx = _lib.context(self.dev)
        if a.precision != "default":
            self.set_precision(a.precision)

    def set_precision(self, name):
        self.ctx.set_precision(name)  # explicit override for every model on this device that does not pin its own

    def precision(self):
        return {v: k for k, v in self._lib.PRECISION_NAMES.items() if "-" not in k}[self.ctx.precision()]

    def step(self, model=None, data=None):
        E, torch = self.E, self.torch
        m = self.model if model is None else model
        d = self.data if data is None else data
        with torch.no_grad():
            res = m(d)
            poses = E.run_weighted_8_point_tuple(d, res)   # all pairs of all tuples: one batched solve
            errs = []
            for (i, j) in self.pairs:
                Tp, info = poses[(i, j)]
                if self.args.ba:  # eval_pairs.py:250-255
                    c = E.mask_confidence(info["confidence"], info["pos_depth_mask"])
                    Tr, vb = E.run_bundle_adjust_2_view(info["kpts0_norm"], info["kpts1_norm"], c, Tp, n_iterations=10)
                    Tp[vb] = Tr
                errs.append(E.pose_errors(Tp, d[f"T_{i}to{j}"]))
        self.last_poses = poses
        return res, errs

    def sync(self):
        self.torch.cuda.synchronize()

    def profile(self, on):
        self.ctx.call("e2emv_profile", 1 if on else 0)
        pr = self._lib.profile_read(self.ctx, reset=True)
        # the layer GEMMs of the plane kernels are bracketed per kernel instantiation (gemm_qkv / gemm_mlp0 / gemm_mlp1 /
        # gemm_chain): the family "gemm" is their sum + the small GEMMs, the parts stay available for roofline.per_kernel
        parts = {k: pr.pop(k) for k in [k for k in pr if k.startswith("gemm_")]}
        for v in parts.values():
            pr["gemm"]["ms"] += v["ms"]
            pr["gemm"]["launches"] += v["launches"]
        pr["_gemm_parts"] = parts
        return pr

    def auc_errors(self):
        from e2e_multi_view_matching_amd.metrics import pair_errors_deg
        _, errs = self.step(self.model_id)
        self.id_errs = errs
        self.id_T = [self.last_poses[p][0].detach().cpu() for p in self.pairs]  # the poses themselves (AUC parity sample)
        return np.concatenate([pair_errors_deg(r.cpu().numpy(), t.cpu().numpy()) for r, t in errs])


class StubWorkload:
    """E2EMV_BENCH_STUB=1: no GPU, no kernels - a fixed-duration stand-in for the step so that the launcher, the process
    group, the barriers, the MAX-over-ranks timing, the error all-gather and the rank-0 JSON line of the N > 1 branch can
    be exercised on CPU over gloo (tests/test_bench_distributed.py).  Never used for a reported number."""

    def __init__(self, args, rank, local_rank):
        self.args, self.rank, self.dev, self.coll_dev = args, rank, None, None
        self.pairs = [(i, j) for j in range(args.tuple_size) for i in range(j)]
        self.cpus = pin_cpus(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))

    def setup(self, rank):
        self.default_precision = 0

    def set_precision(self, name):
        pass

    def precision(self):
        return "stub"

    def step(self, model=None, data=None):
        time.sleep(0.002 * (1 + self.rank))
        return None, None

    def sync(self):
        pass

    def profile(self, on):
        return None

    def auc_errors(self):
        n = self.args.batch * len(self.pairs)
        return np.random.default_rng(100 + self.rank).uniform(0, 30, n)


# ----------------------------------------------------------------------------------------- CPU baseline
def host_cpu_info():
    model, cores = "unknown", set()
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = None
    try:
        cpu, phys, core = None, 0, None
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("processor"):
                    cpu = int(line.split(":")[1])
                elif line.startswith("model name") and model == "unknown":
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    phys = int(line.split(":")[1])
                elif line.startswith("core id"):
                    core = int(line.split(":")[1])
                elif not line.strip() and cpu is not None:
                    if allowed is None or cpu in allowed:
                        cores.add((phys, core if core is not None else cpu))
                    cpu, core = None, None
    except OSError:
        pass
    logical = len(allowed) if allowed is not None else (os.cpu_count() or 1)
    return {"model": model, "physical_cores": max(len(cores), 1) if cores else logical, "logical_cpus": logical}


def cpu_baseline(args, wl):
    """The oracle (torch CPU fp32, the reference's unfused op sequence) on a bounded sample of the workload - SURVEY 8(d):
    threads = physical cores and 1, CPU model stated, configs[0] verbatim and the bench workload at a small batch."""
    import torch
    from e2e_multi_view_matching_amd.synthetic import make_tuples
    from oracle import w8pt as OW
    from oracle.matcher import matcher_forward
    info = host_cpu_info()
    phys = info["physical_cores"]
    sd = {k: v.detach().cpu() for k, v in wl.model.state_dict().items()}
    P = len(wl.pairs)

    def run(data, cfg, sdict):
        with torch.no_grad():
            ref = matcher_forward(data, sdict, cfg)
            T = data_tuple_size(data)
            for j in range(T):
                for i in range(j):
                    Tr, _ = OW.run_weighted_8_point(data, ref, i, j)
                    if Tr is not None:
                        OW.compute_rotation_error(Tr, data[f"T_{i}to{j}"], reduce=False)
        return ref

    def data_tuple_size(d):
        t = 0
        while f"keypoints{t}" in d:
            t += 1
        return t

    def timed(data, cfg, sdict, threads):
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        run(data, cfg, sdict)
        return time.perf_counter() - t0

    prev = torch.get_num_threads()
    ocfg = {**wl.cfg, "full_output": True}
    one = {k: (v[:1] if torch.is_tensor(v) else v) for k, v in wl.data_cpu.items()}
    nb = max(1, min(args.cpu_pairs // max(P, 1), args.batch)) if P > 1 else min(args.cpu_pairs, args.batch)
    small = {k: (v[:nb] if torch.is_tensor(v) else v) for k, v in wl.data_cpu.items()}
    # thread scan on one tuple: more threads than the memory system feeds make the unfused torch ops slower
    cand = sorted({n for n in (8, 16, 32, 64, phys) if n <= phys} or {phys})
    scan = {}
    for n in cand:
        scan[n] = timed(one, ocfg, sd, n)
    best = min(scan, key=scan.get)
    t_best = timed(small, ocfg, sd, best)
    t_one = scan[1] if 1 in scan else timed(one, ocfg, sd, 1)
    # configs[0] verbatim: tuple_size 2, 256 keypoints, batch 4, 5 Sinkhorn iterations
    c1cfg = {"GNN_layers": ["self", "cross"] * 9, "sinkhorn_iterations": 5, "conf_mlp": True, "tuple_size": 2,
             "multi_frame_matching": False, "match_threshold": 0.2, "full_output": True}
    torch.manual_seed(1234)
    import e2e_multi_view_matching_amd as E
    sd1 = {k: v.detach() for k, v in E.MultiViewMatcher(c1cfg).eval().state_dict().items()}
    d1 = make_tuples(batch=4, tuple_size=2, n_kpts=256, seed=7)
    timed(d1, c1cfg, sd1, best)  # warm
    c1_best, c1_one = timed(d1, c1cfg, sd1, best), timed(d1, c1cfg, sd1, 1)
    torch.set_num_threads(prev)
    out = {"value": round(nb * P / t_best, 3), "unit": "pairs/s", "cores": best, "kind": "port",
           "cpu_model": info["model"], "physical_cores": phys, "logical_cpus": info["logical_cpus"],
           "sample": f"{nb * P} pairs ({nb} tuple{'s' if nb > 1 else ''}) of the bench workload through oracle/ (torch-CPU fp32, "
                     f"{best} threads = the fastest of the scan {cand}, {t_best:.1f} s)",
           "thread_scan_s_per_tuple": {str(k): round(v, 3) for k, v in scan.items()},
           "single_thread": {"value": round(P / t_one, 4), "unit": "pairs/s", "cores": 1,
                             "sample": f"{P} pair(s) (1 tuple) of the bench workload, {t_one:.1f} s"},
           "configs0": {"workload": "configs[0] verbatim: tuple_size=2, 256 keypoints, batch 4, 5 Sinkhorn iters",
                        "value": round(4 / c1_best, 2), "cores": best, "single_thread_value": round(4 / c1_one, 2),
                        "unit": "pairs/s"}}
    return out, small, nb, best


# ----------------------------------------------------------------------------------------- main
def run(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not os.environ.get("E2EMV_BENCH_FORCE_DIST"):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    stub = bool(os.environ.get("E2EMV_BENCH_STUB"))
    wl = (StubWorkload if stub else HipWorkload)(args, rank, local_rank)

    dist = None
    if world > 1 or os.environ.get("E2EMV_BENCH_FORCE_DIST"):  # the env knob exercises the RCCL path on a 1-GPU box
        import torch.distributed as dist
        backend = args.backend or ("gloo" if stub else "nccl")
        kw = {"device_id": wl.dev} if backend == "nccl" else {}
        # one process per GPU over RCCL ("nccl" on ROCm); binding the group to this rank's device up front keeps
        # barrier()/collectives from guessing it
        dist.init_process_group(backend, init_method="env://", **kw)
        assert dist.get_world_size() == world
        if backend != "nccl":
            wl.coll_dev = None  # gloo: the few-KB metric collectives go through host tensors

    from e2e_multi_view_matching_amd.distributed import gather_pair_errors, reduce_max_seconds
    from e2e_multi_view_matching_amd.metrics import pose_auc

    wl.setup(rank)
    B, T, N, D = args.batch, args.tuple_size, args.kpts, 256
    P = len(wl.pairs)

    def barrier():
        if dist is not None:
            dist.barrier()

    def timed_steps(k):
        wl.sync()
        barrier()
        wl.sync()
        t0 = time.perf_counter()
        for _ in range(k):
            wl.step()
        wl.sync()
        barrier()
        return reduce_max_seconds(time.perf_counter() - t0, device=wl.coll_dev)  # MAX over ranks

    for _ in range(args.warmup):
        wl.step()
    wl.sync()
    if not args.no_profile:
        wl.profile(True)
    elapsed = timed_steps(args.steps)
    prof = None
    if not args.no_profile:
        prof = wl.profile(False)
    mode = wl.precision()

    # the same K steps once more without the HIP-event brackets (what the instrumentation costs)
    bare = None
    if not args.no_profile and not stub:
        bare = timed_steps(args.steps)

    # ---- further measurements: the same workload in the other arithmetic modes (reported separately, each with its own
    # kernel family times and roofline)
    alts = []
    if not args.no_alt and not stub:
        for other in ("f32", "bf16x3", "f16x2"):
            if other == mode:
                continue
            wl.set_precision(other)
            for _ in range(2):
                wl.step()
            wl.sync()
            if not args.no_profile:
                wl.profile(True)
            alt_elapsed = timed_steps(args.steps)
            alt_prof = wl.profile(False) if not args.no_profile else None
            alts.append(({"mode": other, "ms_per_step": round(1000.0 * alt_elapsed / args.steps, 3),
                          "value": round(B * P * world * args.steps / alt_elapsed, 2), "unit": "pairs/s"}, alt_prof))
        wl.set_precision(mode)

    # ---- the same workload with the batch as two halves on two HIP streams (config["streams"] = 2, an opt-in of the matcher:
    # one half's launch boundaries filled by the other half's kernels).  Reported beside the headline, never as it: the
    # per-kernel durations of concurrently running launches are not comparable with the single-stream kernel trace.
    two_streams = None
    if not args.no_alt and not stub and world == 1 and B >= 2:
        wl.model.config["streams"] = 2
        try:
            for _ in range(3):
                wl.step()
            ts_elapsed = timed_steps(args.steps)
            two_streams = {"ms_per_step": round(1000.0 * ts_elapsed / args.steps, 3), "value": round(B * P * args.steps / ts_elapsed, 2), "unit": "pairs/s",
                           "note": "config['streams'] = 2: two half batches, two streams, two library contexts; the outputs are those of the "
                                   "two halves bit for bit (tests/test_gpu_round4.py); not the headline"}
        finally:
            wl.model.config["streams"] = 1

    # ---- optional: the image-in pipeline (SuperPoint front-end feeding the same matcher / pose path)
    image_in = None
    if args.front_end and not stub:
        import torch
        from e2e_multi_view_matching_amd.superpoint import SuperPoint
        E = wl.E
        torch.manual_seed(7)
        sp = SuperPoint({"max_keypoints": N, "nms_radius": 4, "remove_borders": 4, "fill_with_random_keypoints": True}).eval().to(wl.dev)
        images = torch.rand(T * B, 1, 480, 640, device=wl.dev)

        def step_images():
            with torch.no_grad():
                pred = sp({"image": [images]})  # helpers.run_super_point's merged batch (helpers.py:73-96)
                d2 = dict(wl.data)
                for key, v in pred.items():
                    res = torch.stack(v).view(T, B, *v[0].shape)
                    for m in range(T):
                        d2[key + str(m)] = res[m]
                E.run_weighted_8_point_tuple(d2, wl.model(d2))
        for _ in range(2):
            step_images()
        wl.sync()
        barrier()
        f0 = time.perf_counter()
        for _ in range(args.steps):
            step_images()
        wl.sync()
        barrier()
        fe = reduce_max_seconds(time.perf_counter() - f0, device=wl.coll_dev)
        image_in = {"ms_per_step": round(1000.0 * fe / args.steps, 3), "value": round(B * P * world * args.steps / fe, 2),
                    "unit": "pairs/s", "note": f"{T * B} random 480x640 images per GPU and step through the SuperPoint front-end "
                    f"(random weights, padded to {N} keypoints) -> matcher -> w8pt"}

    # ---- batch-1 latency: the reference's eval_pairs.py loop is one pair at a time (eval_pairs.py:207-256)
    latency = None
    if not args.no_latency and not stub and world == 1:
        torch = wl.torch
        one = {k: (v[:1].contiguous() if torch.is_tensor(v) else v) for k, v in wl.data.items()}
        for _ in range(3):
            wl.step(data=one)
        wl.sync()
        l0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            wl.step(data=one)
        wl.sync()
        lt = (time.perf_counter() - l0) / reps
        latency = {"ms_per_tuple": round(1000.0 * lt, 3), "pairs_per_s": round(P / lt, 1),
                   "note": "batch 1 (one tuple per call, eval_pairs.py:207-256 loop shape), matcher -> w8pt -> pose errors"}

    # ---- AUC leg (not timed): identity-like weights give real matches; errors gathered over ranks
    e_deg = wl.auc_errors()
    e_all = gather_pair_errors(e_deg, device=wl.coll_dev)  # the path's only data collective (RCCL over xGMI), B*P floats per rank
    from e2e_multi_view_matching_amd import distributed as _dist_mod
    metric_collective = _dist_mod.last_collective if world > 1 else "none (one rank)"
    auc = [100.0 * a for a in pose_auc(e_all, [5, 10, 20])]

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return 0

    pairs_per_step = B * P * world
    value = pairs_per_step * args.steps / elapsed
    dtype = {"f32": "f32", "bf16x3": "bf16x3 (fp32 operands split into 3 bf16 planes, 6 bf16-MFMA products, fp32 accumulate: "
             "fp32-class accuracy, same 1e-4 / bit-exact-index parity bar)", "f16x2": "f16x2 (fp32 operands carried as 2 fp16 planes hi + 2^-11 lo', 22 significant bits, 3 fp16-MFMA products, "
             "fp32 accumulate; same 1e-4 / bit-exact-index parity bar)", "stub": "stub"}[mode]
    out = {
        "metric": "image-pairs/sec @1024 kpts + pose AUC@5/10/20deg vs reference",
        "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000.0 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": workload_string(args, world), "pairs_per_gpu": B * P, "global_pairs": pairs_per_step,
                   "weights": "random init (timed), identity-like for the AUC leg", "parallelism": f"tuple-sharded x{world}"},
        "auc_5_10_20": [round(a, 3) for a in auc], "auc_pairs": int(len(e_all)), "metric_collective": metric_collective,
    }
    if hasattr(wl, "ctx"):
        st = wl.ctx.stats()
        # the default arithmetic has no out-of-range fallback to take (tile exponents): `fallbacks` is 0 by
        # construction; rescaled_blocks = plane blocks outside the exponents' dead zone (0 for an ordinary network),
        # sinkhorn_reports = problems the exponential-domain Sinkhorn reported non-finite
        out["range"] = {"fallbacks": st["sinkhorn_rescued"], "rescaled_blocks": st["rescaled_blocks"], "sinkhorn_reports": st["sinkhorn_bad"],
                        "attention_slow_tiles": st.get("attention_slow_tiles", 0)}
    if args.baseline_1gpu:
        out["scaling_check"] = {"baseline_1gpu_value": args.baseline_1gpu, "ideal_value": round(world * args.baseline_1gpu, 2),
                                "efficiency": round(value / (world * args.baseline_1gpu), 4),
                                "note": "weak scaling: tuples are sharded, no data-path collective; the metric all-gather is outside the timed steps"}
    if bare is not None:
        out["ms_per_step_without_event_brackets"] = round(1000.0 * bare / args.steps, 3)
    if alts:
        out["other_precisions"] = [a for a, _ in alts]
    if two_streams:
        out["two_streams"] = two_streams
    if image_in:
        out["image_in_pipeline"] = image_in
    if latency:
        out["batch1_latency"] = latency

    # ---- roofline of the dominant kernel family, from HIP events recorded in the timed region
    def roofline_of(prof_, mode_):
        fl = algorithmic_flops(B, T, N, D, args.layers, True)
        fam = max(("gemm", "attention"), key=lambda k: prof_[k]["ms"])
        gen = wl.ctx.f16x2_kernels if mode_ == "f16x2" and getattr(wl, "ctx", None) is not None else 0
        gen3 = gen >= 3
        kname = {"f32": {"gemm": "gemm_nt_kernel", "attention": "attention_kernel"},
                 "bf16x3": {"gemm": "gemm_x3_kernel", "attention": "attention3f_kernel"},
                 "f16x2": {"gemm": "gemm_p2_kernel" if gen3 else "gemm_h2_kernel",
                           "attention": ("attention_p2w_kernel" if gen >= 4 and N > 256 else "attention_p2_kernel") if gen3 else "attention_h2f_kernel"}}[mode_][fam]
        parts = {k: v for k, v in prof_.get("_gemm_parts", {}).items() if v["launches"]}
        if fam == "gemm" and parts.get("gemm_chain"):
            kname = "gemm_p2_chain_kernel"
        # f32 mode: exact fp32 MFMA.  bf16x3 mode: every algorithmic flop is 6 bf16-MFMA flops, so the ceiling for
        # ALGORITHMIC flops is the dense bf16 peak / 6.
        # f16x2 mode: 3 fp16-MFMA flops per algorithmic flop.
        products = {"f32": 1, "bf16x3": 6, "f16x2": 3}[mode_]
        peak = PEAK_F32_MFMA_TFLOPS if mode_ == "f32" else PEAK_BF16_MFMA_TFLOPS / products

        def family(f):
            ms, n = prof_[f]["ms"], prof_[f]["launches"]
            return fl[f] * args.steps / (ms * 1e-3) / 1e12, ms, n
        achieved, ms, n = family(fam)
        # HBM bytes per launch of that kernel from the rocprofv3 --pmc passes of this same command (FETCH_SIZE x 2 per
        # MI355X_MICROARCH.md, calibrated on a kernel with a known byte count) - profiles/summarize_pmc.py
        # ... and only while csrc/ still hashes to what those passes ran on (otherwise null: a stale counter is not a measurement)
        traffic, traffic_src = None, "no PMC pass on record for this workload / mode (tools/gpu.sh prof)"
        try:
            from e2e_multi_view_matching_amd.build import _stamp
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                pj = json.load(fh)
            w_ = pj["workloads"][args.config][mode_]
            k = w_["kernels"][kname]
            if w_.get("csrc_stamp") and w_["csrc_stamp"] == _stamp():
                traffic = int(k["read_bytes"] + k["write_bytes"])
                # attention_p2w runs as up to three instantiations per layer (self / cross items, and the key-split parts of the
                # leftover items - a launch of their own): bytes of ALL of them per LAYER launch, not the average over instantiations
                var = {n: e for n, e in w_["kernels"].items() if n.startswith(kname + "<")}
                whole = sum(e["launches"] for n, e in var.items() if not n.rstrip(">").rstrip().endswith("true"))
                if fam == "attention" and len(var) > 1 and whole > 0:
                    traffic = int(sum((e["read_bytes"] + e["write_bytes"]) * e["launches"] for e in var.values()) / whole)
                traffic_src = ("profiles/pmc_traffic.json: rocprofv3 --pmc passes of this command (tools/gpu.sh prof) on these very sources "
                               f"(csrc stamp {w_['csrc_stamp'][:12]}), not measured in this run")
            else:
                traffic_src = ("profiles/pmc_traffic.json holds counters of OTHER sources (csrc stamp "
                               f"{str(w_.get('csrc_stamp'))[:12]} != {_stamp()[:12]}): not reported; re-run tools/gpu.sh prof")
        except Exception:
            pass
        # compulsory HBM bytes of the family per step (activations are 4 bytes per element in every mode: fp32, or a pair of
        # fp16 planes; the weights stay in L2): the floor the operand / result stream sets next to the matrix-core ceiling
        Mt, L = B * T * N, len(args.layers)
        chained = bool(parts.get("gemm_chain"))
        # One [rows x 256] activation = U bytes.  A launch per GEMM moves q|k|v 1 + 3, MLP0 2 + 2, MLP1 2 + 1 + 1 = 12 U per layer.  A
        # CHAINED launch (MLP0 -> MLP1 -> the next layer's q|k|v inside one workgroup) has its own compulsory traffic: read [x | message]
        # 2 U, write x_new 1 U, write q|k|v 3 U = 6 U (the hidden layer and the re-read of x_new are NOT compulsory: whatever of them
        # reaches HBM shows up in `traffic` / `traffic_ratio`); the last layer's chain ends in final_proj: 2 + 1 + 1 = 4 U; the first
        # q|k|v is a launch of its own: 4 U.
        gemm_units = (4 + (L - 1) * 6 + 4) if chained else L * 12
        hbm = {"gemm": Mt * D * 4 * gemm_units, "attention": L * Mt * D * 4 * (3 + 1)}[fam]
        floor_ms = hbm / (PEAK_HBM_GBS * 1e9) * 1e3
        # Which roof binds: a launch per GEMM in the split-operand modes sits BELOW the ridge (f16x2: 84 flop per compulsory byte
        # against 833 TFLOP/s / 6.3 TB/s = 132) - HBM-bound, the matrix-core fraction rides along as `mfma`.  Chained, the same flops
        # stand against half the bytes (≈190 flop / B): ABOVE the ridge, the bound is the matrix pipe.  Attention (and
        # the fp32-MFMA GEMMs, 16x slower matrix pipe) are matrix-core bound.
        hbm_bound = fam == "gemm" and mode_ != "f32" and not chained
        gbs = hbm / (ms / args.steps * 1e-3) / 1e9
        # algorithmic (compulsory) bytes of ONE launch of the named kernel, the figure `traffic` is to be read against
        alg_launch = {"gemm": Mt * D * 4 * (6 if chained else 4), "attention": Mt * D * 4 * 4}[fam]
        common = {"kernel": kname, "mode": mode_, "traffic": traffic, "traffic_source": traffic_src,
                  "algorithmic_bytes_per_launch": int(alg_launch),
                  "traffic_ratio": (round(traffic / alg_launch, 3) if traffic else None),
                  "clock": "HIP events on the launch stream inside the timed steps (the rocprofv3 kernel trace of the same command, "
                           "profiles/r6_kernel_stats_*.md, reads 4 - 6 % longer per launch: it serialises the launches)"}
        if hbm_bound:
            main = dict(common, bound="hbm", achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4),
                        frac_of_achievable=round(gbs / 6290.0, 4),
                        mfma={"achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4)})
        else:
            main = dict(common, bound="mfma", achieved=round(achieved, 2), peak=round(peak, 1), unit="TFLOP/s", frac=round(achieved / peak, 4))
        main.update({
                "limiter": ("compulsory HBM bytes of the family (activations in and out once, weights from L2) / HIP-event time against the 8 TB/s "
                            "spec peak (6.29 TB/s measured copy rate: frac_of_achievable); a launch is 1 - 3 output tiles per workgroup, so the "
                            "operand stream and the epilogue store burst alternate instead of overlapping (DESIGN.md 4d)") if hbm_bound else
                           ("the matrix-core ceiling is the stated peak (f16x2: 2500 / 3 products); see DESIGN.md 4d / 4f / 4g / 4i for what the kernel "
                            "spends beside it (K step above its MFMA floor, epilogues with the matrix pipe idle)"),
                "hbm_floor": {"bytes_per_step": int(hbm), "bytes_per_launch": int(hbm // max(n // args.steps, 1)),
                              "ms_per_step_at_peak_hbm": round(floor_ms, 3), "peak_gbs": PEAK_HBM_GBS,
                              "frac_of_family_time": round(floor_ms / (ms / args.steps), 4),
                              "note": "layer GEMMs / attention only (the family's small launches add < 2 %)"},
                "note": ("algorithmic flops of the family / HIP-event time; " +
                         ("exact fp32 MFMA (v_mfma_f32_32x32x2_f32)" if mode_ == "f32" else
                          f"peak = dense 16-bit MFMA 2500 TFLOP/s / {products} MFMA products per algorithmic flop (split operands)")) +
                        "; traffic = HBM bytes per launch (read + write) from the PMC passes under profiles/",
                "avg_launch_ms": round(ms / max(n, 1), 4), "launches_per_step": n // args.steps})
        if mode_ in ("f16x2", "bf16x3"):
            # what the matrix pipe SUSTAINS on this chip for a stream of nothing but 32x32x16 16-bit MFMAs on random operands (tools/wp_gemm.hip,
            # profiles/r6_wp_gemm.log: 1.66 PFLOP/s = 0.67 of the 2.5 PFLOP/s the `peak` above is derived from - the clock settles at ~1.6 GHz under a
            # saturated pipe).  Supplementary: `frac` stays priced against the guide's nominal peak.
            sustained = 1665.0 / products
            main["sustained_matrix_rate"] = {"tflops": round(sustained, 1), "frac": round(achieved / sustained, 4),
                                             "source": "measured: tools/wp_gemm.hip 'MFMAs only' steady state, profiles/r6_wp_gemm.log (not a spec number)"}
        fam2 = "attention" if fam == "gemm" else "gemm"
        a2, _, _ = family(fam2)
        second = {"kernel": fam2, "achieved": round(a2, 2), "unit": "TFLOP/s", "frac": round(a2 / peak, 4)}
        fams = {k: {"ms_per_step": round(v["ms"] / args.steps, 3), "launches_per_step": v["launches"] // args.steps}
                for k, v in prof_.items() if not k.startswith("_") and v["launches"]}
        # ---- the kernels of the two big families one by one (plane kernels: each GEMM instantiation has its own event slot)
        if parts:
            U = Mt * D * 4  # bytes of one [rows x 256] activation
            fl_row = 2.0 * Mt * D * D  # flops of a 256 x 256 GEMM over all rows
            # per launch: (compulsory bytes, algorithmic flops); the chain of the last layer ends in final_proj instead of q | k | v
            spec = {"gemm_qkv": ("gemm_p2_kernel<QKV>", 4 * U, 3 * fl_row), "gemm_mlp0": ("gemm_p2_kernel<PLANES>", 4 * U, 4 * fl_row),
                    "gemm_mlp1": ("gemm_p2_kernel<PLANES, residual>", 4 * U, 2 * fl_row),
                    "gemm_chain": ("gemm_p2_chain_kernel (MLP0 -> MLP1 -> next q|k|v)", ((L - 1) * 6 + 4) * U / L, ((L - 1) * 9 + 6 + 1) * fl_row / L)}
            per = []
            for k, v in parts.items():
                nm, by, fp = spec[k]
                us = 1e3 * v["ms"] / v["launches"]
                per.append({"kernel": nm, "launches_per_step": v["launches"] // args.steps, "us_per_launch": round(us, 1),
                            "hbm_gbs": round(by / us / 1e3, 1), "frac_hbm": round(by / us / 1e3 / PEAK_HBM_GBS, 4),
                            "tflops_eq": round(fp / us / 1e6, 1), "frac_mfma": round(fp / us / 1e6 / peak, 4)})
            at = prof_["attention"]
            if at["launches"]:
                us = 1e3 * at["ms"] / at["launches"]
                fp = fl["attention"] / L
                per.append({"kernel": "attention_p2w_kernel" if (gen >= 4 and N > 256) else "attention_p2_kernel", "launches_per_step": at["launches"] // args.steps,
                            "us_per_launch": round(us, 1), "hbm_gbs": round(4 * U / us / 1e3, 1), "frac_hbm": round(4 * U / us / 1e3 / PEAK_HBM_GBS, 4),
                            "tflops_eq": round(fp / us / 1e6, 1), "frac_mfma": round(fp / us / 1e6 / peak, 4)})
            main["per_kernel"] = per
            # The line's roofline is the NAMED kernel's own: its algorithmic flops per launch / its own average launch time (HIP events),
            # not the family average (the family holds the small launches too); the family's numbers move to `family`.
            own = next((e for e in per if e["kernel"].startswith(kname)), None)
            if own is not None and main["bound"] == "mfma":
                main["family"] = {"achieved": main["achieved"], "frac": main["frac"], "avg_launch_ms": main["avg_launch_ms"],
                                  "launches_per_step": main["launches_per_step"], "unit": "TFLOP/s"}
                main["achieved"], main["frac"] = own["tflops_eq"], own["frac_mfma"]
                main["avg_launch_ms"], main["launches_per_step"] = round(own["us_per_launch"] / 1e3, 4), own["launches_per_step"]
                if "sustained_matrix_rate" in main:
                    main["sustained_matrix_rate"]["frac"] = round(own["tflops_eq"] / main["sustained_matrix_rate"]["tflops"], 4)
            main["per_kernel_note"] = ("HIP events around each kernel instantiation's launches in the timed steps (the interval of a launch ends where "
                                       "the next family's begins: dispatch gaps included); bytes = compulsory activation traffic of the launch, flops = "
                                       "algorithmic; a chained launch is priced at its OWN compulsory 6 U (read [x | message] 2, write x_new 1, "
                                       "write q|k|v 3; last layer 4 U) - 192 flop / B, above the 132 ridge: its roof is frac_mfma; us_per_launch = HIP events")
        return main, second, fams

    if prof:
        out["roofline"], out["roofline_second"], out["families"] = roofline_of(prof, mode)
        sk = prof["sinkhorn"]
        if sk["ms"] > 0:
            # A resident kernel: the scores are read from HBM once (plus once by the final sweep that writes logZ), every
            # iteration after that is on-chip.  Its floor is iterations x rounds x the exchange between the workgroups of a
            # problem (two dependent store -> poll hops through the device-coherent level, measured alone on the chip:
            # profiles/README.md), not a byte rate; the arithmetic (one packed FMA per element per half-iteration) is the rest.
            ms_call = sk["ms"] / args.steps
            problems = B * P
            # the launcher's own plan for this batch (e2emv_sinkhorn_plan: what launch_sinkhorn runs on this context and device -
            # kernel kinds, residency, rounds; nothing re-derived here)
            segments = [(sg["rows_per_workgroup"], sg["resident_problems"], sg["rounds"]) for sg in wl.ctx.sinkhorn_plan(problems, N, N, args.sinkhorn_iters)]
            if not segments:  # the log-domain chain (iters == 0 / a demoted context): one "round" per call, no resident exchange
                segments = [(0, problems, 1)]
            rounds = sum(sg[2] for sg in segments)
            resident = max(sg[1] for sg in segments)
            hop_us = 0.8  # an idle one-to-one granule hand-off on this chip (MI355X_MICROARCH.md price list): the physical floor of a hop
            exch_ms = rounds * args.sinkhorn_iters * 2 * hop_us * 1e-3
            fma_ms = sum(sg[2] * args.sinkhorn_iters * 2 * (sg[1] * N * N / 2) / (256 * 64 * 2.0e9) * 1e3 for sg in segments)  # fp32 FMA: 64 elements / clk / CU
            physical = problems * 2 * N * N * 4 + problems * (N + 1) ** 2 * 4
            out["sinkhorn_bound"] = {"bound": "inter-workgroup exchange latency (resident kernel)", "ms_per_call": round(ms_call, 3),
                                     "iterations": args.sinkhorn_iters, "problems": problems,
                                     "segments": [{"rows_per_workgroup": a, "resident_problems": b_, "rounds": c} for a, b_, c in segments],
                                     "resident_problems": resident, "rounds": rounds,
                                     "us_per_iteration": round(ms_call * 1e3 / (rounds * max(args.sinkhorn_iters, 1)), 2),
                                     "exchange_floor_ms": round(exch_ms, 3), "arithmetic_floor_ms": round(fma_ms, 3),
                                     "frac": round((exch_ms + fma_ms) / ms_call, 4),
                                     "physical_hbm_gbs": round(physical / (ms_call * 1e-3) / 1e9, 1),
                                     "note": "frac = (2 idle granule hops of 0.8 us per iteration + fp32 FMA time at 64 elements / clk / CU) / measured; one "
                                             "problem alone on the chip runs 4.6 us per iteration on 64-row workgroups (profiles/README.md); the SURVEY "
                                             "8(d) byte model (2 sweeps of the couplings per iteration from HBM) does not describe a kernel that keeps "
                                             "them in registers"}
    for alt, alt_prof in alts:
        if alt_prof:
            alt["roofline"], alt["roofline_second"], alt["families"] = roofline_of(alt_prof, alt["mode"])
    # the reference's own arithmetic (exact fp32 MFMA) as top-level keys: the headline above is the library's default mode
    # (f16x2, parity-green at north_star's tolerance but narrower than fp32 by the letter)
    for alt, _ in alts:
        if alt["mode"] == "f32":
            out["value_f32"], out["ms_per_step_f32"] = alt["value"], alt["ms_per_step"]
            if "roofline" in alt:
                r = alt["roofline"]
                out["roofline_f32"] = {"kernel": r["kernel"], "bound": r["bound"], "achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"],
                                       "frac": r["frac"], "second": alt["roofline_second"]}
    if mode == "f32":
        out["value_f32"], out["ms_per_step_f32"] = out["value"], out["ms_per_step"]

    # ---- CPU baseline: the oracle (torch CPU, same unfused op sequence as the reference) on a bounded sample
    if world == 1 and args.cpu_pairs > 0 and not stub:
        import torch
        from e2e_multi_view_matching_amd.metrics import pair_errors_deg
        from oracle import w8pt as OW
        from oracle.matcher import matcher_forward
        base, small, nb, best = cpu_baseline(args, wl)
        out["cpu_baseline"] = base
        # AUC parity on the sample (identity-like weights): HIP vs oracle on identical inputs
        torch.set_num_threads(best)
        sd_id = {k: v.detach().cpu() for k, v in wl.model_id.state_dict().items()}
        with torch.no_grad():
            ref = matcher_forward(small, sd_id, {**wl.cfg, "full_output": True})
            eo, eo64, eh64, dT = [], [], [], 0.0

            def angles64(Tp, Tg):
                """compute_pose_error.py:3-22 as a function of the pose alone, in fp64: the reference's fp32 arccos resolves a
                sub-degree angle to a few per cent (d arccos = -1 / sqrt(1 - cos^2)) on EITHER implementation, so the fp32 angle legs
                differ by more than the poses do; the parity statement is about the poses."""
                Tp, Tg = Tp.double().numpy(), Tg.double().numpy()
                R = np.einsum("bji,bjk->bik", Tp[:, :3, :3], Tg[:, :3, :3])
                rot = np.abs(np.arccos(np.clip((np.trace(R, axis1=1, axis2=2) - 1.0) / 2.0, -1.0, 1.0)))
                t0, t1 = Tp[:, :3, 3], Tg[:, :3, 3]
                n = np.linalg.norm(t0, axis=1) * np.linalg.norm(t1, axis=1)
                tr = np.where(n > 1e-6, np.abs(np.arccos(np.clip((t0 * t1).sum(1) / np.where(n > 1e-6, n, 1.0), -1.0, 1.0))), 0.0)
                return pair_errors_deg(rot, tr)
            for q, (i, j) in enumerate(wl.pairs):
                Tr, _ = OW.run_weighted_8_point(small, ref, i, j)
                r = OW.compute_rotation_error(Tr, small[f"T_{i}to{j}"], reduce=False)
                t = OW.compute_translation_error_as_angle(Tr, small[f"T_{i}to{j}"], keep_shape=True)
                eo.append(pair_errors_deg(r.numpy(), t.numpy()))
                Th = wl.id_T[q][:nb]
                eo64.append(angles64(Tr, small[f"T_{i}to{j}"]))
                eh64.append(angles64(Th, small[f"T_{i}to{j}"]))
                dT = max(dT, float((Th - Tr).abs().max()))
        eo, eo64, eh64 = np.concatenate(eo), np.concatenate(eo64), np.concatenate(eh64)
        eh = np.concatenate([pair_errors_deg(r.cpu().numpy()[:nb], t.cpu().numpy()[:nb]) for r, t in wl.id_errs])
        out["auc_parity_sample"] = {"pairs": int(nb * P),
                                    "hip": [round(100 * a, 2) for a in pose_auc(eh64, [5, 10, 20])],
                                    "oracle": [round(100 * a, 2) for a in pose_auc(eo64, [5, 10, 20])],
                                    "max_abs_dT": dT, "max_abs_err_deg_diff": float(np.max(np.abs(eh64 - eo64))),
                                    "angle_arithmetic": "fp64 from the fp32 poses of either side (|dT| is the only difference; bar 1e-4)",
                                    "fp32_angle_legs": {"hip": [round(100 * a, 3) for a in pose_auc(eh, [5, 10, 20])],
                                                        "oracle": [round(100 * a, 3) for a in pose_auc(eo, [5, 10, 20])],
                                                        "max_abs_err_deg_diff": float(np.max(np.abs(eh - eo))),
                                                        "note": "the reference's own fp32 arccos on each side: ill-conditioned below a degree"}}
    if dist is not None:
        _dist_mod.close_library_comms()  # (the library's RCCL communicator, if the metric gather used it: before the process group goes)
        dist.destroy_process_group()
        # RCCL writes its version banner through C stdio: flush it now so that the JSON line below is the LAST line
        import ctypes
        ctypes.CDLL(None).fflush(None)
    print(json.dumps(out), flush=True)
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    rc = maybe_self_spawn(args, argv)
    if rc is not None:
        return rc
    return run(args)


if __name__ == "__main__":
    sys.exit(main())
