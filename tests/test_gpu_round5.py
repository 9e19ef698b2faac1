"""Round-5 evidence (-m gpu).

* The chained GEMMs (f16x2 kernel generation 5, csrc/gemm_p2c.hip): MLP0 -> MLP1 -> the next layer's q | k | v (final_proj behind
  the last layer) walked per 256-row block by ONE workgroup in ONE launch.  Tile shape, K order and epilogues are those of the
  per-GEMM launches (generation 4), so every output - logZ, matches, the matched descriptors - must be equal BIT for bit; the
  hand-off between the GEMMs of a row block (stores retired + workgroup barrier, then plain loads) is what these tests pin:
  configs[1] exactly (one row block per workgroup), more row blocks than CUs (a workgroup walks several), fewer (part of the
  chip), a 3-tuple, activations far outside fp16's range (the tile-exponent side-band crosses the same hand-off).
* BASELINE configs[0]'s exact workload through the HIP path (it is the reference's CPU plumbing config; the oracle runs it in
  tests/test_oracle_golden.py) - so that every BASELINE workload has gone through HIP.
* e2emv_get_descriptors after another call reused the workspace: E2EMV_ESTATE, not stale memory.
* config["streams"] = 2 with a tensor-valued image_size{m} whose length equals the batch size."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev(d, gpu):
    return {k: (v.to(gpu) if torch.is_tensor(v) else v) for k, v in d.items()}


def _run(model, data, gpu, mode):
    import e2e_multi_view_matching_amd as E
    model.config["mfma_precision"] = mode
    with torch.no_grad():
        out = model(data)
    md = E.last_descriptors(gpu)
    torch.cuda.synchronize()
    return out, md


@pytest.mark.parametrize("B,T,N,layers,iters", [
    (32, 2, 1024, None, 100),            # BASELINE configs[1]: 256 row blocks = one per CU, 18 layers
    (40, 2, 1024, ["self", "cross"], 10),  # 320 row blocks: 64 workgroups walk two
    (8, 2, 1024, ["self", "cross"] * 2, 10),  # 64 row blocks: a quarter of the chip
    (4, 3, 640, ["self", "cross"] * 2, 10),   # joint 3-tuples, 640 keypoints per image (2.5 row blocks per image)
])
def test_chained_gemms_equal_the_per_gemm_launches_bit_for_bit(gpu, B, T, N, layers, iters):
    import e2e_multi_view_matching_amd as E
    from e2e_multi_view_matching_amd.synthetic import make_tuples
    from test_gpu_matcher import _randomize_bn
    torch.manual_seed(50 + B)
    cfg = {"sinkhorn_iterations": iters, "conf_mlp": True, "match_threshold": 0.2, "tuple_size": T, "multi_frame_matching": T > 2}
    if layers is not None:
        cfg["GNN_layers"] = layers
    model = E.MultiViewMatcher(cfg).eval()
    _randomize_bn(model, 50 + B)
    model = model.to(gpu)
    data = _dev(make_tuples(seed=50 + B, batch=B, tuple_size=T, n_kpts=N), gpu)
    ctx = E._lib.context(gpu)
    ctx.set_split_min_rows(0)
    ctx.stats(reset=True)
    try:
        ref, md_ref = _run(model, data, gpu, "f16x2-r4")
        for rep in range(3):  # (a race in the hand-off would not show on every run)
            out, md = _run(model, data, gpu, "f16x2-chain")
            assert out.keys() == ref.keys()
            assert torch.equal(md, md_ref), (rep, float((md - md_ref).abs().max()))
            for k, v in ref.items():
                if torch.is_tensor(v):
                    assert torch.equal(out[k], v), (rep, k)
        # the default generation decides by tile rounds whether it chains; either way the same bits
        out, md = _run(model, data, gpu, "f16x2")
        assert torch.equal(md, md_ref)
        for k, v in ref.items():
            if torch.is_tensor(v):
                assert torch.equal(out[k], v), k
        assert ctx.stats()["rescaled_blocks"] == 0
    finally:
        ctx.set_split_min_rows(-1)


@pytest.mark.parametrize("s_qk,s_v,s_h", [(2.0 ** 9, 2.0 ** 12, 2.0 ** 14), (2.0 ** 12, 2.0 ** 20, 2.0 ** 24), (2.0 ** -10, 2.0 ** -14, 2.0 ** -16)])
def test_chained_gemms_carry_the_tile_exponents(gpu, s_qk, s_v, s_h):
    """Activations far outside fp16's range (tests/test_gpu_range.py's re-parametrised network, at a size the plane kernels
    take by default): the exponent side-band written by one tile's epilogue is read by the next tile of the same workgroup.
    Chained == per-GEMM launches bit for bit, and both meet the oracle."""
    import e2e_multi_view_matching_amd as E
    from e2e_multi_view_matching_amd.synthetic import make_tuples
    from oracle.matcher import matcher_forward
    from test_gpu_matcher import _randomize_bn
    from test_gpu_range import _rescale
    cfg = {"sinkhorn_iterations": 30, "conf_mlp": True, "match_threshold": 0.0, "GNN_layers": ["self", "cross"] * 3}
    torch.manual_seed(23)
    model = E.MultiViewMatcher(cfg).eval()
    _randomize_bn(model, 23)
    model.load_state_dict(_rescale(model.state_dict(), 6, s_qk, s_v, s_h))
    data = make_tuples(seed=23, batch=2, tuple_size=2, n_kpts=200)
    ocfg = {**model.config, "full_output": True}
    ref = matcher_forward(data, {k: v.clone() for k, v in model.state_dict().items()}, ocfg)
    model = model.to(gpu)
    d = _dev(data, gpu)
    ctx = E._lib.context(gpu)
    ctx.set_split_min_rows(0)
    try:
        ctx.stats(reset=True)
        a, md_a = _run(model, d, gpu, "f16x2-r4")
        b, md_b = _run(model, d, gpu, "f16x2-chain")
        assert ctx.stats()["rescaled_blocks"] > 0  # the side-band was in use
    finally:
        ctx.set_split_min_rows(-1)
    assert torch.equal(md_a, md_b)
    for k, v in a.items():
        if torch.is_tensor(v):
            assert torch.equal(b[k], v), k
    z = b["scores_0_1"].cpu()
    assert torch.isfinite(z).all() and float((z - ref["scores_0_1"]).abs().max()) < 1e-4
    assert torch.equal(b["matches0_0_1"].cpu(), ref["matches0_0_1"])


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
def test_baseline_configs0_exact_workload_through_hip(gpu, precision):
    """BASELINE.json configs[0]: tuple_size 2, 256 keypoints, 256-d descriptors, batch 4, 5 Sinkhorn iterations (the
    reference's CPU plumbing config) - matcher -> w8pt on the GPU against the oracle on the same inputs."""
    import e2e_multi_view_matching_amd as E
    from e2e_multi_view_matching_amd.synthetic import identity_like_state, make_tuples
    from oracle import w8pt as OW
    from oracle.matcher import matcher_forward
    torch.manual_seed(0)
    model = identity_like_state(E.MultiViewMatcher({"sinkhorn_iterations": 5, "conf_mlp": True}).eval())
    assert len(model.config["GNN_layers"]) == 18
    data = make_tuples(batch=4, tuple_size=2, n_kpts=256, seed=11)
    ref = matcher_forward(data, model.state_dict(), {**model.config, "full_output": True})
    Tr, _ = OW.run_weighted_8_point(data, ref, 0, 1)
    model = model.to(gpu)
    model.config["mfma_precision"] = precision
    d = _dev(data, gpu)
    ctx = E._lib.context(gpu)
    ctx.set_split_min_rows(0 if precision != "f32" else -1)  # (the call is small: reach the plane kernels in the default mode)
    try:
        with torch.no_grad():
            out = model(d)
            T, _ = E.run_weighted_8_point(d, out, 0, 1)
        torch.cuda.synchronize()
    finally:
        ctx.set_split_min_rows(-1)
    assert out["scores_0_1"].shape == (4, 257, 257)
    assert float((out["scores_0_1"].cpu() - ref["scores_0_1"]).abs().max()) < 1e-4
    assert torch.equal(out["matches0_0_1"].cpu(), ref["matches0_0_1"]) and torch.equal(out["matches1_0_1"].cpu(), ref["matches1_0_1"])
    assert float((out["conf_scores_0_1"].cpu() - ref["conf_scores_0_1"]).abs().max()) < 1e-4
    assert float((T.cpu() - Tr).abs().max()) < 1e-4


def test_descriptors_die_with_the_workspace(gpu):
    import e2e_multi_view_matching_amd as E
    from e2e_multi_view_matching_amd.synthetic import make_tuples
    model = E.MultiViewMatcher({"GNN_layers": ["self", "cross"], "sinkhorn_iterations": 5, "conf_mlp": True}).eval().to(gpu)
    d = _dev(make_tuples(batch=2, tuple_size=2, n_kpts=128, seed=1), gpu)
    with torch.no_grad():
        out = model(d)
        md = E.last_descriptors(gpu)
        assert md.shape == (4, 128, 256) and torch.isfinite(md).all()
        E.run_weighted_8_point(d, out, 0, 1)  # carves the same arena
        with pytest.raises(RuntimeError, match="get_descriptors"):
            E.last_descriptors(gpu)
        model(d)
        assert torch.equal(E.last_descriptors(gpu), md)


def test_two_streams_with_a_tensor_image_size_of_batch_length(gpu):
    import e2e_multi_view_matching_amd as E
    from e2e_multi_view_matching_amd.synthetic import make_tuples
    cfg = {"GNN_layers": ["self", "cross"], "sinkhorn_iterations": 5, "conf_mlp": True}
    model = E.MultiViewMatcher(cfg).eval().to(gpu)
    d = _dev(make_tuples(batch=2, tuple_size=2, n_kpts=128, seed=2), gpu)
    for m in range(2):
        h, w = d[f"image_size{m}"]
        d[f"image_size{m}"] = torch.tensor([float(h), float(w)])  # shape [2] == batch: must NOT be split with the batch
    with torch.no_grad():
        one = model(d)
        model.config["streams"] = 2
        two = model(d)
    for k, v in one.items():
        if torch.is_tensor(v):
            assert torch.equal(two[k], v), k
