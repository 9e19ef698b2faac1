"""Round-2 additions (-m gpu): weight ownership across model instances, the pose solve batched over the pairs of a tuple
and over ragged pair sets, on-device normalize / error reductions, BASELINE configs[3] / configs[4] at full size."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev(d, gpu):
    return {k: (v.to(gpu) if torch.is_tensor(v) else v) for k, v in d.items()}


def test_two_models_alternating_on_one_device_keep_their_own_weights(gpu):
    """A runs, B runs, A runs again: the context holds one weight set, so every switch must re-push.  Also the
    precision switch: a model with mfma_precision=None must not inherit what another model selected."""
    from e2e_multi_view_matching_amd import MultiViewMatcher, SuperPoint, _lib
    from e2e_multi_view_matching_amd.synthetic import make_tuples
    cfg = {"GNN_layers": ["self", "cross"], "sinkhorn_iterations": 10, "conf_mlp": True}
    torch.manual_seed(1)
    A = MultiViewMatcher(cfg).eval().to(gpu)
    torch.manual_seed(2)
    ctx = _lib.context(gpu)
    other = _lib.PRECISION_F32 if ctx.default_precision == _lib.PRECISION_BF16X3 else _lib.PRECISION_BF16X3
    B = MultiViewMatcher({**cfg, "mfma_precision": {_lib.PRECISION_F32: "f32", _lib.PRECISION_BF16X3: "bf16x3"}[other]}).eval().to(gpu)
    data = _dev(make_tuples(batch=1, tuple_size=2, n_kpts=128, seed=0), gpu)
    with torch.no_grad():
        a1 = A(data)["scores_0_1"].clone()
        assert ctx.precision() == ctx.default_precision
        b1 = B(data)["scores_0_1"].clone()
        assert ctx.precision() == other
        a2 = A(data)["scores_0_1"].clone()
        assert ctx.precision() == ctx.default_precision          # A did not inherit B's mode
        b2 = B(data)["scores_0_1"].clone()
    assert torch.equal(a1, a2) and torch.equal(b1, b2)
    assert float((a1 - b1).abs().max()) > 1e-2                  # they really are different networks
    # in-place weight update of A is picked up (fingerprint), then B still runs on its own
    with torch.no_grad():
        A.bin_score.add_(0.5)
        a3 = A(data)["scores_0_1"]
        assert not torch.equal(a3, a1)
        assert torch.equal(B(data)["scores_0_1"], b1)
    # same for two SuperPoint instances
    torch.manual_seed(3)
    S1 = SuperPoint({"max_keypoints": 64}).eval().to(gpu)
    torch.manual_seed(4)
    S2 = SuperPoint({"max_keypoints": 64}).eval().to(gpu)
    img = torch.rand(1, 1, 120, 160, device=gpu)
    with torch.no_grad():
        s1a = S1({"image": img})["scores"][0].clone()
        s2a = S2({"image": img})["scores"][0].clone()
        s1b = S1({"image": img})["scores"][0].clone()
    assert torch.equal(s1a, s1b) and not (s1a.shape == s2a.shape and torch.equal(s1a, s2a))


def test_superpoint_floors_image_sizes_like_upstream(gpu):
    """An image whose size is not a multiple of 8: keypoints only inside the floored (H//8*8) x (W//8*8) region - and, since
    round 3, the rows / columns beyond it are SEEN by the encoder like upstream (the comparison with the oracle at such sizes is
    tests/test_gpu_superpoint.py::test_matches_oracle): the result differs from the cropped image's near the cut edge only."""
    from e2e_multi_view_matching_amd import SuperPoint
    torch.manual_seed(0)
    sp = SuperPoint({"max_keypoints": -1, "keypoint_threshold": 0.0, "remove_borders": 0, "return_score_map": True}).eval().to(gpu)
    img = torch.rand(1, 1, 123, 165, device=gpu)
    with torch.no_grad():
        a = sp({"image": img})
        b = sp({"image": img[:, :, :120, :160].contiguous()})
    assert float(a["keypoints"][0][:, 0].max()) < 160 and float(a["keypoints"][0][:, 1].max()) < 120
    sa, sb = a["score_map"][0][0], b["score_map"][0][0]
    assert sa.shape == sb.shape == (120, 160)
    # the receptive field of a score pixel reaches 3 convolutions deep at each of 4 levels: far from the cut edges nothing changes
    assert torch.equal(sa[:56, :96], sb[:56, :96])
    assert not torch.equal(sa, sb)


@pytest.mark.parametrize("closest", [False, True])
def test_tuple_batched_w8pt_equals_the_pair_loop(gpu, closest):
    import e2e_multi_view_matching_amd as E
    from e2e_multi_view_matching_amd.synthetic import make_tuples
    T, B, N = 4, 3, 256
    d = make_tuples(batch=B, tuple_size=T, n_kpts=N, seed=5, rho=0.8)
    g = torch.Generator().manual_seed(0)
    res = {}
    pairs = [(i, j) for j in range(T) for i in range(j)]
    for i, j in pairs:
        res[f"matches{i}_{i}_{j}"] = d[f"gt_matches{i}_{i}_{j}"]
        res[f"conf_scores_{i}_{j}"] = torch.rand(B, N, 1, generator=g)
    dg, rg = _dev(d, gpu), _dev(res, gpu)
    targets = {(i, j): dg[f"T_{i}to{j}"] for i, j in pairs} if closest else None
    out = E.run_weighted_8_point_tuple(dg, rg, choose_closest=closest, targets=targets, determine_inliers=True)
    assert list(out) == pairs
    for i, j in pairs:
        Tp, info = E.run_weighted_8_point(dg, rg, i, j, choose_closest=closest, target_T_021=dg[f"T_{i}to{j}"] if closest else None)
        Tb, ib = out[(i, j)]
        assert torch.equal(Tb, Tp), (i, j)                                   # same kernels, same inputs: bit-identical
        assert torch.equal(ib["pos_depth_mask"], info["pos_depth_mask"]) and torch.equal(ib["confidence"], info["confidence"])
        assert torch.equal(ib["kpts1_norm"], info["kpts1_norm"]) and ib["confidence"].shape == (B, N, 1)
        assert ib["inliers"].dtype == torch.bool
        Tg = d[f"T_{i}to{j}"]                                                # and it is the right pose (t up to scale)
        assert float((Tb[:, :3, :3].cpu() - Tg[:, :3, :3]).abs().max()) < 0.02
        tdir = torch.nn.functional.normalize(Tg[:, :3, 3], dim=-1)
        assert float((Tb[:, :3, 3].cpu() * tdir).sum(-1).min()) > 0.99
    # a pair without matches in `result` -> per-pair path, (None, None) for that pair
    rg2 = {k: v for k, v in rg.items() if k != "matches0_0_1"}
    out2 = E.run_weighted_8_point_tuple(dg, rg2)
    assert out2[(0, 1)] == (None, None) and out2[(1, 2)][0] is not None


def test_ragged_pair_batch_equals_one_call_per_pair(gpu):
    """bundle_adjust_io's pairs have different numbers of matches: one ragged launch == the per-pair calls."""
    from e2e_multi_view_matching_amd import multi_view as MV
    from e2e_multi_view_matching_amd.synthetic import make_tuples
    rng = np.random.default_rng(0)
    d = make_tuples(batch=1, tuple_size=4, n_kpts=300, seed=9, rho=0.9)
    problems = []
    for (i, j), n in zip([(0, 1), (0, 2), (1, 2), (0, 3), (1, 3)], [300, 41, 8, 5, 120]):
        gt = d[f"gt_matches{i}_{i}_{j}"][0].numpy()
        keep = np.nonzero(gt >= 0)[0][:n]
        m0 = d[f"keypoints{i}"][0].numpy()[keep]
        m1 = d[f"keypoints{j}"][0].numpy()[gt[keep]]
        conf = rng.uniform(0.1, 1.0, (len(keep), 1)).astype(np.float32)
        problems.append((d[f"intr{i}"][0].numpy(), d[f"intr{j}"][0].numpy(), m0, m1, conf))
    batched = MV.relative_poses_w8pt_ba(problems)
    assert batched[3][0] is False                      # 5 matches: the reference returns (None, None) -> success False
    for q, pr in enumerate(problems):
        ok, R, t, inl = MV.estimate_relative_pose_w8pt_ba(*pr)
        assert ok == batched[q][0]
        if ok:
            assert np.allclose(R, batched[q][1], atol=1e-6) and np.allclose(t, batched[q][2], atol=1e-6), q
            assert np.array_equal(inl, batched[q][3]) and inl.shape == (pr[2].shape[0],)
    R, t = batched[0][1], batched[0][2]
    Tgt = d["T_0to1"][0].numpy()
    assert np.abs(R - Tgt[:3, :3]).max() < 0.02


def test_normalize_and_error_reductions_on_the_device(gpu):
    import e2e_multi_view_matching_amd as E
    from oracle import w8pt as O
    g = torch.Generator().manual_seed(0)
    k = torch.rand(3, 50, 2, generator=g) * 500
    K = torch.eye(4).repeat(3, 1, 1)
    K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2] = 600.0, 590.0, 321.5, 238.25
    assert float((E.normalize(k.to(gpu), K.to(gpu)).cpu() - O.normalize(k, K)).abs().max()) < 1e-6
    assert float((E.normalize(k.to(gpu), K[:1, :3, :3].to(gpu)).cpu() - O.normalize(k, K)).abs().max()) < 1e-6
    assert E.normalize(k[0].to(gpu), K[0].to(gpu)).shape == (50, 2)
    # translation error: entries with |t0||t1| <= 1e-6 are skipped (compute_pose_error.py:17-21)
    T0 = torch.eye(4).repeat(4, 1, 1)
    T1 = torch.eye(4).repeat(4, 1, 1)
    T0[:, :3, 3] = torch.tensor([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 0], [0, 0, 2.0]])
    T1[:, :3, 3] = torch.tensor([[0.0, 1.0, 0], [0, 1.0, 0], [1.0, 0, 0], [0, 0, -1.0]])
    e = E.compute_translation_error_as_angle(T0.to(gpu), T1.to(gpu), reduce=False).cpu()
    ref = O.compute_translation_error_as_angle(T0, T1, reduce=False)
    assert e.shape == ref.shape == (3,) and float((e - ref).abs().max()) < 1e-6
    m = E.compute_translation_error_as_angle(T0.to(gpu), T1.to(gpu))
    assert abs(float(m) - float(O.compute_translation_error_as_angle(T0, T1))) < 1e-6 and m.dim() == 0
    Z = torch.eye(4).repeat(2, 1, 1)
    assert torch.isnan(E.compute_translation_error_as_angle(Z.to(gpu), Z.to(gpu)))      # mean of nothing, like torch
    assert E.compute_translation_error_as_angle(Z.to(gpu), Z.to(gpu), reduce=False).shape == (0,)
    assert float(E.compute_rotation_error(T0.to(gpu), T1.to(gpu))) == 0.0
    c = torch.rand(2, 9, 1, generator=g)
    mk = torch.rand(2, 9, generator=g) > 0.5
    assert torch.equal(E.mask_confidence(c.to(gpu), mk.to(gpu)).cpu(), c * mk.unsqueeze(-1))


# ---------------------------------------------------------------- BASELINE configs[3] / configs[4] at full size
def _full_size(gpu, n_kpts, desc_dtype, precisions):
    """T = 5 joint matching, batch 8 tuples, 18 layers, 100 Sinkhorn iterations: size-independent properties on the whole
    batch + oracle parity on ONE tuple (the CPU oracle needs seconds per tuple at this size), w8pt over all 80 pairs."""
    import e2e_multi_view_matching_amd as E
    from e2e_multi_view_matching_amd.synthetic import make_tuples
    from oracle import w8pt as OW
    from oracle.matcher import matcher_forward
    T, B, N = 5, 8, n_kpts
    pairs = [(i, j) for j in range(T) for i in range(j)]
    cfg = {"GNN_layers": ["self", "cross"] * 9, "sinkhorn_iterations": 100, "conf_mlp": True, "multi_frame_matching": True,
           "tuple_size": T, "match_threshold": 0.2}
    torch.manual_seed(11)
    model = E.MultiViewMatcher(cfg).eval()
    # random GNN (so attention over the 4N concatenated sources matters) with a final projection that keeps descriptor
    # similarity visible: residual branches scaled down instead of zeroed
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("mlp.3.weight") or name.endswith("mlp.3.bias"):
                p.mul_(0.01)
        D = 256
        s = (20.0 * D ** 0.5) ** 0.5
        model.final_proj.weight.copy_((torch.eye(D) * s).unsqueeze(-1))
        model.final_proj.bias.zero_()
        last = max(int(k.split(".")[2]) for k in model.state_dict() if k.startswith("kenc.encoder.") and k.endswith(".bias"))
        model.kenc.encoder[last].weight.mul_(0.01)
    data = make_tuples(batch=B, tuple_size=T, n_kpts=N, seed=77, desc_dtype=desc_dtype)
    one = {k: (v[:1].float() if (torch.is_tensor(v) and v.dtype == torch.float16) else (v[:1] if torch.is_tensor(v) else v))
           for k, v in data.items()}
    ref = matcher_forward(one, {k: v.clone() for k, v in model.state_dict().items()}, {**model.config, "full_output": True})
    # ... and the LAST tuple of the batch (other output tiles, other CUs, another place in the persistent-tile schedules)
    lastt = {k: (v[B - 1:].float() if (torch.is_tensor(v) and v.dtype == torch.float16) else (v[B - 1:] if torch.is_tensor(v) else v))
             for k, v in data.items()}
    ref_last = matcher_forward(lastt, {k: v.clone() for k, v in model.state_dict().items()}, {**model.config, "full_output": True})
    model = model.to(gpu)
    dg = _dev(data, gpu)
    for precision in precisions:
        model.config["mfma_precision"] = precision
        with torch.no_grad():
            out = model(dg)
        assert len([k for k in out if k.startswith("scores_")]) == 10
        n_matched = 0
        for i, j in pairs:
            z = out[f"scores_{i}_{j}"]
            assert z.shape == (B, N + 1, N + 1) and bool(torch.isfinite(z).all())
            # oracle parity on tuple 0: scores within 1e-4, assignment indices bit-exact
            assert float((z[:1].cpu() - ref[f"scores_{i}_{j}"]).abs().max()) < 1e-4, (precision, i, j)
            assert torch.equal(out[f"matches{i}_{i}_{j}"][:1].cpu(), ref[f"matches{i}_{i}_{j}"]), (precision, i, j)
            assert torch.equal(out[f"matches{j}_{i}_{j}"][:1].cpu(), ref[f"matches{j}_{i}_{j}"]), (precision, i, j)
            assert float((z[B - 1:].cpu() - ref_last[f"scores_{i}_{j}"]).abs().max()) < 1e-4, (precision, i, j, "last tuple")
            assert torch.equal(out[f"matches{i}_{i}_{j}"][B - 1:].cpu(), ref_last[f"matches{i}_{i}_{j}"]), (precision, i, j, "last tuple")
            assert torch.equal(out[f"matches{j}_{i}_{j}"][B - 1:].cpu(), ref_last[f"matches{j}_{i}_{j}"]), (precision, i, j, "last tuple")
            # Sinkhorn marginals on the whole batch: the last half-iteration is the column update -> exact columns
            P = z.exp()
            assert float((P[:, :, :N].sum(1) - 1).abs().max()) < 1e-3
            # mutual-consistency of the match block
            m0, m1 = out[f"matches{i}_{i}_{j}"], out[f"matches{j}_{i}_{j}"]
            v = m0 >= 0
            bi = torch.arange(B, device=gpu)[:, None].expand_as(m0)
            assert bool((m1[bi[v], m0[v]] == torch.arange(N, device=gpu)[None].expand_as(m0)[v]).all())
            gt = dg[f"gt_matches{i}_{i}_{j}"]
            n_matched += int((v & (m0 == gt)).sum())
        assert n_matched > 0.25 * 0.7 * N * B * len(pairs), n_matched   # 70 % of the keypoints are shared points
        # batch independence: tuple 3 alone gives the same result as inside the batch
        sub = {k: (v[3:4].contiguous() if torch.is_tensor(v) else v) for k, v in dg.items()}
        with torch.no_grad():
            o1 = model(sub)
        # (a batch of one may take the small-problem tile shapes: same math, different summation order)
        assert float((o1["scores_0_4"] - out["scores_0_4"][3:4]).abs().max()) < 1e-4
        assert float((o1["matches1_1_3"] == out["matches1_1_3"][3:4]).float().mean()) > 0.999
        # pose: all 80 pairs in one batched solve; tuple 0 against the oracle chain
        poses = E.run_weighted_8_point_tuple(dg, out)
        for i, j in pairs:
            Tp, info = poses[(i, j)]
            assert Tp.shape == (B, 4, 4) and bool(torch.isfinite(Tp).all())
            Tr, _ = OW.run_weighted_8_point(one, ref, i, j)
            assert float((Tp[:1].cpu() - Tr).abs().max()) < 1e-4, (precision, i, j)
            r, t = E.pose_errors(Tp, dg[f"T_{i}to{j}"])
            assert float(torch.rad2deg(r).median()) < 2.0


def test_config4_full_size_t5_1024_batch8_18_layers(gpu):
    """BASELINE configs[3]: tuple_size 5 (10 pairwise matches per tuple), 1024 keypoints, batch 8 tuples, 18 layers
    (eval_multi_view.py:122-132,154-162), both arithmetic modes."""
    _full_size(gpu, 1024, torch.float32, ["f32", "bf16x3", "f16x2"])


def test_config5_per_gpu_shape_t5_2048_fp16_batch8(gpu):
    """BASELINE configs[4] per-GPU share: tuple_size 5, 2048 keypoints (MegaDepth shape), fp16 descriptors, batch 8 tuples,
    18 layers; the oracle is fed the same fp16-rounded descriptors."""
    _full_size(gpu, 2048, torch.float16, ["f32", "bf16x3", "f16x2"])


def test_f16x2_out_of_range_activations(gpu, split_always):
    """Activations beyond fp16's range (|x| > 65504 in a GEMM input).  The plane kernels (round 3, the default) carry a tile
    exponent per 64 x 64 block and solve the same input like bf16x3 does - there is nothing to fall back to.  (Round 2's f16x2
    kernels turned them into +-inf for e2emv_sync to report; that generation is selectable in the measurement build only since
    round 6 - non-finite scores as an error: tests/test_gpu_sinkhorn_resident.py.)"""
    from e2e_multi_view_matching_amd import MultiViewMatcher, _lib
    from e2e_multi_view_matching_amd.synthetic import make_tuples
    ctx = _lib.context(gpu)
    torch.manual_seed(3)
    cfg = {"GNN_layers": ["self", "cross"], "sinkhorn_iterations": 10}
    model = MultiViewMatcher(cfg).eval().to(gpu)
    data = make_tuples(batch=1, tuple_size=2, n_kpts=256, seed=4)
    for m in range(2):
        data[f"descriptors{m}"] = data[f"descriptors{m}"] * 3e6     # far outside anything a descriptor network produces
    dg = _dev(data, gpu)
    assert ctx.lib.e2emv_sync(ctx.h, None) == _lib.OK
    model.config["mfma_precision"] = "bf16x3"
    with torch.no_grad():
        ref = model(dg)
    assert bool(torch.isfinite(ref["scores_0_1"]).all())
    ctx.stats(reset=True)
    model.config["mfma_precision"] = "f16x2"
    with torch.no_grad():
        out = model(dg)
    assert bool(torch.isfinite(out["scores_0_1"]).all())
    assert ctx.lib.e2emv_sync(ctx.h, None) == _lib.OK
    z, zr = out["scores_0_1"], ref["scores_0_1"]
    assert float((z - zr).abs().max() / zr.abs().max()) < 1e-6
    assert torch.equal(out["matches0_0_1"], ref["matches0_0_1"])
    assert ctx.stats()["rescaled_blocks"] > 0                        # the exponents were at work
