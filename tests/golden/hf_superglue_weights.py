"""Seeded SuperGlue weights shared by the golden generator (make_golden.py, which loads them into the HuggingFace port of
upstream SuperGlue) and by the tests (which load the SAME tensors, re-laid-out to upstream's names and channel order, into
the oracle / the HIP matcher).  Pure torch: the D = 256 weight set (10 MB) is reproduced from its seed instead of being
committed; the fixture `superglue_hf_d256.npz` holds only inputs and HF outputs."""
import torch


def seeded_hf_tensors(seed, D, kenc_sizes, n_layers):
    """Tensors in the HF port's layout (attention channels head-major: c = h*d + dd), keyed by role."""
    g = torch.Generator().manual_seed(seed)

    def lin(out_f, in_f, scale=1.0):
        return torch.randn(out_f, in_f, generator=g) * (scale / in_f ** 0.5), torch.randn(out_f, generator=g) * 0.1

    def bn(n):
        return {"running_mean": torch.randn(n, generator=g) * 0.2, "running_var": torch.rand(n, generator=g) + 0.5,
                "weight": torch.rand(n, generator=g) + 0.5, "bias": torch.randn(n, generator=g) * 0.2}

    t = {}
    dims = [3] + list(kenc_sizes) + [D]
    for i in range(len(dims) - 1):
        t[f"enc.{i}.w"], t[f"enc.{i}.b"] = lin(dims[i + 1], dims[i])
        if i < len(dims) - 2:
            for k, v in bn(dims[i + 1]).items():
                t[f"enc.{i}.bn.{k}"] = v
    for li in range(n_layers):
        for name in ("q", "k", "v", "o"):
            t[f"gnn.{li}.{name}.w"], t[f"gnn.{li}.{name}.b"] = lin(D, D)
        t[f"gnn.{li}.mlp0.w"], t[f"gnn.{li}.mlp0.b"] = lin(2 * D, 2 * D)
        for k, v in bn(2 * D).items():
            t[f"gnn.{li}.mlp0.bn.{k}"] = v
        t[f"gnn.{li}.mlp1.w"], t[f"gnn.{li}.mlp1.b"] = lin(D, 2 * D)
    t["final.w"], t["final.b"] = lin(D, D, scale=2.5)
    return t


def to_upstream_state(t, D, H, n_enc, n_layers, bin_score):
    """HF-layout tensors -> upstream-named state dict (upstream channel c_up = dd*H + h  <-  HF channel h*d + dd)."""
    d = D // H
    perm = torch.tensor([(c % H) * d + (c // H) for c in range(D)])
    sd = {}
    for i in range(n_enc):
        sd[f"kenc.encoder.{3 * i}.weight"] = t[f"enc.{i}.w"].unsqueeze(-1).clone()
        sd[f"kenc.encoder.{3 * i}.bias"] = t[f"enc.{i}.b"].clone()
        if i < n_enc - 1:
            for k in ("weight", "bias", "running_mean", "running_var"):
                sd[f"kenc.encoder.{3 * i + 1}.{k}"] = t[f"enc.{i}.bn.{k}"].clone()
    for li in range(n_layers):
        for pi, name in enumerate(("q", "k", "v")):
            sd[f"gnn.layers.{li}.attn.proj.{pi}.weight"] = t[f"gnn.{li}.{name}.w"][perm].unsqueeze(-1).clone()
            sd[f"gnn.layers.{li}.attn.proj.{pi}.bias"] = t[f"gnn.{li}.{name}.b"][perm].clone()
        sd[f"gnn.layers.{li}.attn.merge.weight"] = t[f"gnn.{li}.o.w"][:, perm].unsqueeze(-1).clone()
        sd[f"gnn.layers.{li}.attn.merge.bias"] = t[f"gnn.{li}.o.b"].clone()
        sd[f"gnn.layers.{li}.mlp.0.weight"] = t[f"gnn.{li}.mlp0.w"].unsqueeze(-1).clone()
        sd[f"gnn.layers.{li}.mlp.0.bias"] = t[f"gnn.{li}.mlp0.b"].clone()
        for k in ("weight", "bias", "running_mean", "running_var"):
            sd[f"gnn.layers.{li}.mlp.1.{k}"] = t[f"gnn.{li}.mlp0.bn.{k}"].clone()
        sd[f"gnn.layers.{li}.mlp.3.weight"] = t[f"gnn.{li}.mlp1.w"].unsqueeze(-1).clone()
        sd[f"gnn.layers.{li}.mlp.3.bias"] = t[f"gnn.{li}.mlp1.b"].clone()
    sd["final_proj.weight"] = t["final.w"].unsqueeze(-1).clone()
    sd["final_proj.bias"] = t["final.b"].clone()
    sd["bin_score"] = torch.tensor(float(bin_score))
    return sd
