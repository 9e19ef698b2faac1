"""Round-4 parity evidence (-m gpu): the configuration the bench times - BASELINE configs[1] exactly, batch 32 - compared
with the oracle on its FIRST and its LAST pair (different output tiles, different CUs, a different position in the
persistent-tile schedules than batch index 0), in the exact-fp32 and the default arithmetic; and the matched descriptors
(final_proj output, ``e2emv_get_descriptors``) against the oracle's - the quantity the GNN arithmetic modes differ in (logZ is
dominated by the fp32 Sinkhorn)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev(d, gpu):
    return {k: (v.to(gpu) if torch.is_tensor(v) else v) for k, v in d.items()}


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
def test_configs1_batch32_first_and_last_pair_against_the_oracle(gpu, precision):
    import e2e_multi_view_matching_amd as E
    from e2e_multi_view_matching_amd.synthetic import make_tuples
    from oracle.matcher import matcher_forward
    from test_gpu_matcher import _randomize_bn
    B, N = 32, 1024
    torch.manual_seed(41)
    model = E.MultiViewMatcher({"sinkhorn_iterations": 100, "conf_mlp": True, "match_threshold": 0.2}).eval()
    _randomize_bn(model, 41)
    data = make_tuples(seed=41, batch=B, tuple_size=2, n_kpts=N)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(gpu)
    model.config["mfma_precision"] = precision
    E._lib.context(gpu).stats(reset=True)  # (the counters are cumulative over the process: earlier range tests leave theirs)
    with torch.no_grad():
        out = model(_dev(data, gpu))
    md = E.last_descriptors(gpu).cpu().view(B, 2, N, -1)
    assert E._lib.context(gpu).stats()["rescaled_blocks"] == 0  # an ordinary network: the plain paths of the plane kernels
    for b in (0, B - 1):
        one = {k: (v[b:b + 1] if torch.is_tensor(v) else v) for k, v in data.items()}
        ref = matcher_forward(one, sd, {**model.config, "full_output": True})
        z = out["scores_0_1"][b:b + 1].cpu()
        assert float((z - ref["scores_0_1"]).abs().max()) < 1e-4, (precision, b)
        assert torch.equal(out["matches0_0_1"][b:b + 1].cpu(), ref["matches0_0_1"]), (precision, b)
        assert torch.equal(out["matches1_0_1"][b:b + 1].cpu(), ref["matches1_0_1"]), (precision, b)
        assert float((out["matching_scores0_0_1"][b:b + 1].cpu() - ref["matching_scores0_0_1"]).abs().max()) < 1e-4
        # descriptors: [D, N] per image in the oracle; 1e-4 of the largest descriptor entry (measured margins:
        # profiles/r4_parity_margins.txt)
        for t in range(2):
            r = ref["_mdesc"][t][0].transpose(0, 1)
            e = float((md[b, t] - r).abs().max()) / float(r.abs().max())
            assert e < 1e-4, (precision, b, t, e)


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
def test_two_streams_give_the_single_stream_result(gpu, split_always, precision):
    """config["streams"] = 2: the batch as two halves on two HIP streams / two library contexts (launch boundaries of one half
    filled by the other's kernels).  Tuples are independent: the outputs are those of the two halves run one after the other on
    the main context, BIT for bit, and the whole batch's within rounding (kernel shapes follow the batch size) - an odd batch,
    a 3-tuple with joint matching, the plane kernels (1024 keypoints) and the small-call kernels.  The FIRST two-stream call
    comes while the main stream is busy: the second context's weight upload must be complete before its first kernel
    (the null-stream fence of the commit paths)."""
    import e2e_multi_view_matching_amd as E
    from e2e_multi_view_matching_amd.synthetic import make_tuples
    for (B, T, N, layers) in ((5, 2, 1024, ["self", "cross"]), (4, 3, 256, ["self", "cross"] * 2)):
        torch.manual_seed(3 + B)   # (new weights: the peer context has to take them over as well)
        cfg = {"GNN_layers": layers, "sinkhorn_iterations": 20, "conf_mlp": True, "tuple_size": T, "multi_frame_matching": T > 2,
               "mfma_precision": precision}
        model = E.MultiViewMatcher(cfg).eval().to(gpu)
        data = _dev(make_tuples(seed=3, batch=B, tuple_size=T, n_kpts=N), gpu)
        h = (B + 1) // 2

        def part(lo, hi):
            return {k: (v[lo:hi] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == B else v) for k, v in data.items()}

        with torch.no_grad():
            one = model(data)
            model.config["streams"] = 2
            two = model(data)
            model.config["streams"] = 1
            lo, hi = model(part(0, h)), model(part(h, B))
        assert one.keys() == two.keys()
        for k, v in one.items():
            if not torch.is_tensor(v):
                continue
            assert v.shape == two[k].shape, (precision, B, T, k)
            assert torch.equal(two[k], torch.cat([lo[k], hi[k]], 0)), (precision, B, T, k)
            if k.startswith("scores_"):
                assert float((v - two[k]).abs().max()) < 2e-5, (precision, B, T, k)
