#!/usr/bin/env python
"""bench.py - image-pairs/sec of the matcher -> Sinkhorn -> weighted-8-point path on MI355X.

One step = one pass of the hot path over one batch of synthetic pairs, inputs resident in
HBM: MultiViewMatcher.forward (kenc, 18 attention layers, final_proj, scores, 100 Sinkhorn
iterations, match block, conf head) -> run_weighted_8_point (get_kpts + w8pt, fixed shape
B x N like helpers.py:254-258) -> per-pair pose errors.  Workload at N=1: BASELINE.json
configs[1] (tuple_size 2, 1024 keypoints, 256-d, 9x(self,cross), 100 Sinkhorn iterations,
batch 32).  With --gpus N (launched by torch.distributed.run) every rank runs the same
per-GPU batch on its own tuples (weak scaling, no data-path collective); the only
collective is the final all-gather of per-pair pose errors for the AUC (RCCL).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the
dominant kernel family (HIP events on the launch stream inside the timed region) and
`cpu_baseline` (the torch-CPU oracle timed on a bounded sample, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="pairs per GPU per step")
    ap.add_argument("--kpts", type=int, default=1024)
    ap.add_argument("--tuple-size", type=int, default=2)
    ap.add_argument("--sinkhorn-iters", type=int, default=100)
    ap.add_argument("--cpu-pairs", type=int, default=4, help="pairs timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernel families with HIP events")
    ap.add_argument("--ba", action="store_true", help="also run the two-view bundle adjustment (10 LM iterations) per pair "
                    "inside the step (the reference's default eval mode w8pt_ba); off by default: SURVEY 8(d) defines the "
                    "metric on matcher -> w8pt -> pose errors")
    ap.add_argument("--front-end", action="store_true", help="extra (reported separately, never part of `value`): image-in "
                    "pipeline = SuperPoint on 2*batch 480x640 images -> matcher -> w8pt per step")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra (untimed-for-value) bf16x3-attention measurement")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    assert torch.cuda.is_available(), "bench.py needs an MI355X (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("E2EMV_BENCH_FORCE_DIST"):  # the env knob exercises the RCCL path on a 1-GPU box
        import torch.distributed as dist
        # one process per GPU over RCCL ("nccl" on ROCm); binding the group to this rank's device up front keeps
        # barrier()/collectives from guessing it
        dist.init_process_group("nccl", init_method="env://", device_id=dev)

    import e2e_multi_view_matching_amd as E
    from e2e_multi_view_matching_amd import _lib
    from e2e_multi_view_matching_amd.distributed import gather_pair_errors, reduce_max_seconds
    from e2e_multi_view_matching_amd.metrics import pair_errors_deg, pose_auc
    from e2e_multi_view_matching_amd.synthetic import identity_like_state, make_tuples

    B, T, N, D = args.batch, args.tuple_size, args.kpts, 256
    layers = ["self", "cross"] * 9
    cfg = {"GNN_layers": layers, "sinkhorn_iterations": args.sinkhorn_iters, "conf_mlp": True, "tuple_size": T,
           "multi_frame_matching": T > 2, "match_threshold": 0.2}
    pairs = [(i, j) for j in range(T) for i in range(j)]
    P = len(pairs)

    torch.manual_seed(1234)
    model = E.MultiViewMatcher(cfg).eval().to(dev)           # W-rand: timed
    torch.manual_seed(1234)
    model_id = identity_like_state(E.MultiViewMatcher(cfg).eval()).to(dev)  # W-id: AUC leg (meaningful matches)
    data_cpu = make_tuples(batch=B, tuple_size=T, n_kpts=N, seed=1000 + rank)
    data = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data_cpu.items()}

    def step(m):
        with torch.no_grad():
            res = m(data)
            errs = []
            for (i, j) in pairs:
                Tp, info = E.run_weighted_8_point(data, res, i, j)
                if args.ba:  # eval_pairs.py:250-255
                    c = info["confidence"] * info["pos_depth_mask"].unsqueeze(-1)
                    Tr, vb = E.run_bundle_adjust_2_view(info["kpts0_norm"], info["kpts1_norm"], c, Tp, n_iterations=10)
                    Tp[vb] = Tr
                errs.append(E.pose_errors(Tp, data[f"T_{i}to{j}"]))
        return res, errs

    ctx = _lib.context(dev)
    for _ in range(args.warmup):
        step(model)
    torch.cuda.synchronize()
    if not args.no_profile:
        ctx.call("e2emv_profile", 1)
        _lib.profile_read(ctx, reset=True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(model)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    prof = None
    if not args.no_profile:
        prof = _lib.profile_read(ctx, reset=True)
        ctx.call("e2emv_profile", 0)
    elapsed = reduce_max_seconds(elapsed, device=dev)  # MAX over ranks

    # ---- optional second measurement: attention on the bf16 pipe with 3-way split operands (fp32-class accuracy,
    # e2emv_set_precision); reported separately, `value` above is always the all-fp32-MFMA path
    alt = None
    if not args.no_alt:
        ctx.call("e2emv_set_precision", _lib.PRECISION_BF16X3)
        for _ in range(2):
            step(model)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        a0 = time.perf_counter()
        for _ in range(args.steps):
            step(model)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        alt_elapsed = time.perf_counter() - a0
        alt_elapsed = reduce_max_seconds(alt_elapsed, device=dev)
        ctx.call("e2emv_set_precision", _lib.PRECISION_F32)
        alt = {"ms_per_step": round(1000.0 * alt_elapsed / args.steps, 3),
               "value": round(B * len(pairs) * world * args.steps / alt_elapsed, 2), "unit": "pairs/s",
               "note": "same workload with e2emv_set_precision(BF16X3): attention and the two MLP GEMMs of every layer run on the "
                       "bf16 matrix pipe with 3-way split operands (6 bf16-MFMA products per block, fp32 accumulation; "
                       "fp32-class accuracy - every parity test runs in both modes at the same 1e-4 / bit-exact-index bar)"}

    # ---- optional: the image-in pipeline (SuperPoint front-end feeding the same matcher / pose path)
    image_in = None
    if args.front_end:
        from e2e_multi_view_matching_amd.superpoint import SuperPoint
        torch.manual_seed(7)
        sp = SuperPoint({"max_keypoints": N, "nms_radius": 4, "remove_borders": 4, "fill_with_random_keypoints": True}).eval().to(dev)
        images = torch.rand(T * B, 1, 480, 640, device=dev)

        def step_images():
            with torch.no_grad():
                pred = sp({"image": [images]})  # helpers.run_super_point's merged batch (helpers.py:73-96)
                d2 = dict(data)
                for key, v in pred.items():
                    res = torch.stack(v).view(T, B, *v[0].shape)
                    for m in range(T):
                        d2[key + str(m)] = res[m]
                res = model(d2)
                for (i, j) in pairs:
                    E.run_weighted_8_point(d2, res, i, j)
        for _ in range(2):
            step_images()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        f0 = time.perf_counter()
        for _ in range(args.steps):
            step_images()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        fe = reduce_max_seconds(time.perf_counter() - f0, device=dev)
        image_in = {"ms_per_step": round(1000.0 * fe / args.steps, 3), "value": round(B * len(pairs) * world * args.steps / fe, 2),
                    "unit": "pairs/s", "note": f"{T * B} random 480x640 images per GPU and step through the SuperPoint front-end "
                    f"(random weights, padded to {N} keypoints) -> matcher -> w8pt"}

    # ---- AUC leg (not timed): identity-like weights give real matches; errors gathered over ranks
    _, errs = step(model_id)
    e_deg = np.concatenate([pair_errors_deg(r.cpu().numpy(), t.cpu().numpy()) for r, t in errs])
    e_all = gather_pair_errors(e_deg, device=dev)  # the path's only data collective (RCCL over xGMI), B*P floats per rank
    auc = [100.0 * a for a in pose_auc(e_all, [5, 10, 20])]

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    pairs_per_step = B * P * world
    value = pairs_per_step * args.steps / elapsed
    out = {
        "metric": "image-pairs/sec @1024 kpts + pose AUC@5/10/20deg vs reference",
        "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000.0 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"configs[1]: tuple_size={T}, {N} keypoints, 256-dim desc, 9x(self,cross) GNN, "
                               f"{args.sinkhorn_iters} Sinkhorn iters, batch {B} pairs/GPU, w8pt pose per pair",
                   "pairs_per_gpu": B * P, "global_pairs": pairs_per_step, "weights": "random init (timed), "
                   "identity-like for the AUC leg", "parallelism": f"tuple-sharded x{world}"},
        "auc_5_10_20": [round(a, 3) for a in auc],
    }
    if alt:
        out["bf16x3_attention"] = alt
    if image_in:
        out["image_in_pipeline"] = image_in

    # ---- roofline of the dominant kernel family, from HIP events recorded in the timed region
    if prof:
        fl = algorithmic_flops(B, T, N, D, layers, True)
        fam = max(("gemm", "attention"), key=lambda k: prof[k]["ms"])
        ms, n = prof[fam]["ms"], prof[fam]["launches"]
        achieved = fl[fam] * args.steps / (ms * 1e-3) / 1e12
        kname = {"gemm": "gemm_nt_kernel", "attention": "attention_kernel"}[fam]
        # HBM bytes per launch of that kernel from the rocprofv3 --pmc passes of this same command (FETCH_SIZE x 2 per
        # MI355X_MICROARCH.md, calibrated on sinkhorn_sweep's known byte count) - profiles/summarize_pmc.py
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                k = json.load(fh)["kernels"][kname]
            traffic = int(k["read_bytes"] + k["write_bytes"])
        except Exception:
            traffic = None
        out["roofline"] = {"bound": "mfma", "kernel": kname,
                           "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                           "traffic_note": "HBM bytes per launch (read + write), PMC passes committed under profiles/; "
                                           "algorithmic flops include the merge conv that is folded into MLP0 (executed "
                                           "flops are 18/20 of algorithmic)",
                           "avg_launch_ms": round(ms / max(n, 1), 4), "launches_per_step": n // args.steps}
        sk = prof["sinkhorn"]
        sk_bytes = B * P * (2 * args.sinkhorn_iters + 2) * (N + 1) ** 2 * 4
        out["families"] = {k: {"ms_per_step": round(v["ms"] / args.steps, 3), "launches_per_step": v["launches"] // args.steps}
                           for k, v in prof.items() if v["launches"]}
        if sk["ms"] > 0:
            gbs = sk_bytes * args.steps / (sk["ms"] * 1e-3) / 1e9
            out["sinkhorn_roofline"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                        "frac": round(gbs / PEAK_HBM_GBS, 4),
                                        "note": "algorithmic bytes (2 sweeps/iter model) / time; the kernel streams S once per iteration"}
        fam2 = "attention" if fam == "gemm" else "gemm"
        a2 = fl[fam2] * args.steps / (prof[fam2]["ms"] * 1e-3) / 1e12
        out["roofline_second"] = {"kernel": fam2, "achieved": round(a2, 2), "unit": "TFLOP/s",
                                  "frac": round(a2 / PEAK_F32_MFMA_TFLOPS, 4)}

    # ---- CPU baseline: the oracle (torch CPU, same unfused op sequence as the reference) on a bounded sample
    if world == 1 and args.cpu_pairs > 0:
        from oracle import w8pt as OW
        from oracle.matcher import matcher_forward
        nb = min(args.cpu_pairs, B)
        small = {k: (v[:nb] if torch.is_tensor(v) else v) for k, v in data_cpu.items()}
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        sd_id = {k: v.detach().cpu() for k, v in model_id.state_dict().items()}
        ocfg = {**cfg, "full_output": True}
        c0 = time.perf_counter()
        with torch.no_grad():
            ref = matcher_forward(small, sd, ocfg)
            for (i, j) in pairs:
                Tr, _ = OW.run_weighted_8_point(small, ref, i, j)
                if Tr is not None:
                    OW.compute_rotation_error(Tr, small[f"T_{i}to{j}"], reduce=False)
        c1 = time.perf_counter()
        out["cpu_baseline"] = {"value": round(nb * P / (c1 - c0), 3), "unit": "pairs/s", "cores": torch.get_num_threads(),
                               "kind": "port", "sample": f"{nb * P} pairs of the same workload (oracle/ torch-CPU fp32, "
                               f"{torch.get_num_threads()} threads, {c1 - c0:.1f} s)"}
        # AUC parity on the sample (identity-like weights): HIP vs oracle on identical inputs
        with torch.no_grad():
            ref = matcher_forward(small, sd_id, ocfg)
            eo = []
            for (i, j) in pairs:
                Tr, _ = OW.run_weighted_8_point(small, ref, i, j)
                r = OW.compute_rotation_error(Tr, small[f"T_{i}to{j}"], reduce=False)
                t = OW.compute_translation_error_as_angle(Tr, small[f"T_{i}to{j}"], reduce=False)
                eo.append(pair_errors_deg(r.numpy(), t.numpy()))
        eo = np.concatenate(eo)
        eh = np.concatenate([pair_errors_deg(r.cpu().numpy()[:nb], t.cpu().numpy()[:nb]) for r, t in errs])
        out["auc_parity_sample"] = {"pairs": int(nb * P), "hip": [round(100 * a, 3) for a in pose_auc(eh, [5, 10, 20])],
                                    "oracle": [round(100 * a, 3) for a in pose_auc(eo, [5, 10, 20])],
                                    "max_abs_err_deg_diff": float(np.max(np.abs(eh - eo)))}
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
