"""The default arithmetic (f16x2 on plane activations) over activations far outside fp16's range (-m gpu).

The fp16 planes of the mode carry one exponent per block of 64 x 64 activations (csrc/p2.h), so the NETWORK - not just a
kernel - has to behave like the reference's fp32 when a trained checkpoint produces large activations.  The networks here
are re-parametrised with powers of two (exact in fp32): W_q x s, W_k / s (logits unchanged), W_v x s and the merge
conv / s, the MLP's BatchNorm affine x s and its second conv / s (ReLU is positively homogeneous).  The function - and the
fp32 oracle, bit for bit up to over/underflow - is unchanged, the internal activations move by s."""
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("split_always")]


def _rescale(sd, layers, s_qk, s_v, s_h):
    sd = {k: v.clone() for k, v in sd.items()}
    for l in range(layers):
        a = f"gnn.layers.{l}.attn."
        sd[a + "proj.0.weight"] *= s_qk; sd[a + "proj.0.bias"] *= s_qk
        sd[a + "proj.1.weight"] /= s_qk; sd[a + "proj.1.bias"] /= s_qk
        sd[a + "proj.2.weight"] *= s_v; sd[a + "proj.2.bias"] *= s_v
        sd[a + "merge.weight"] /= s_v  # (its bias is added after the scaling is undone)
        m = f"gnn.layers.{l}.mlp."
        sd[m + "1.weight"] *= s_h; sd[m + "1.bias"] *= s_h  # BatchNorm affine of the hidden layer
        sd[m + "3.weight"] /= s_h
    return sd


@pytest.mark.parametrize("s_qk,s_v,s_h", [(1.0, 1.0, 1.0), (2.0 ** 6, 2.0 ** 8, 2.0 ** 10), (2.0 ** 9, 2.0 ** 12, 2.0 ** 14),
                                          (2.0 ** 12, 2.0 ** 20, 2.0 ** 24), (2.0 ** -10, 2.0 ** -14, 2.0 ** -16)])
@pytest.mark.parametrize("mode", ["f16x2", "f16x2-chain"])
def test_network_with_large_activations_matches_the_oracle(gpu, s_qk, s_v, s_h, mode):
    """18 layers; activations up to ~1e3 x s: inside the old limits of the mode, at them, far beyond them (an fp16 plane
    without the exponent would hold inf) and far below them.  Same bar as every matcher parity test."""
    from e2e_multi_view_matching_amd import MultiViewMatcher, _lib
    from e2e_multi_view_matching_amd.synthetic import make_tuples
    from oracle.matcher import matcher_forward
    from test_gpu_matcher import _randomize_bn
    cfg = {"sinkhorn_iterations": 30, "conf_mlp": True, "match_threshold": 0.0, "mfma_precision": mode}
    torch.manual_seed(21)
    model = MultiViewMatcher(cfg).eval()
    _randomize_bn(model, 21)
    sd = _rescale(model.state_dict(), len(model.config["GNN_layers"]), s_qk, s_v, s_h)
    model.load_state_dict(sd)
    data = make_tuples(seed=21, batch=1, tuple_size=2, n_kpts=192)
    ocfg = dict(model.config)
    ocfg["full_output"] = True
    ocfg.pop("mfma_precision", None)
    ref = matcher_forward(data, {k: v.clone() for k, v in model.state_dict().items()}, ocfg)
    assert torch.isfinite(ref["scores_0_1"]).all()
    model = model.to(gpu)
    with torch.no_grad():
        out = model({k: (v.to(gpu) if torch.is_tensor(v) else v) for k, v in data.items()})
    _lib.context(gpu).call("e2emv_sync", _lib.stream_ptr(gpu))  # no sticky device error
    z, zr = out["scores_0_1"].cpu(), ref["scores_0_1"]
    assert torch.isfinite(z).all()
    assert float((z - zr).abs().max()) < 1e-4, float((z - zr).abs().max())
    for key in ("matches0_0_1", "matches1_0_1"):
        assert torch.equal(out[key].cpu(), ref[key]), key
