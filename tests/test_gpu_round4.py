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
