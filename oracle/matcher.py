"""Oracle: attentional GNN matcher forward (SuperGlue-style), torch-CPU, unfused.

TEST INFRASTRUCTURE (see oracle/__init__.py).  A functional restatement over a plain
``state_dict`` (upstream parameter names) so it shares NO code with the
product module.  Op sequence = what the reference's device executes: Conv1d(k=1) as matmul,
BatchNorm1d (eval, running stats, NOT folded), ReLU, einsum attention with materialised
H x N x N_src probabilities, softmax, logsumexp Sinkhorn.

Source of truth: upstream magicleap ``models/superglue.py`` (the reference's matcher lives
in an absent, unpinned submodule - ``.gitmodules:1-3``) + reference call sites
``train.py:343-348``, ``eval_pairs.py:190-194,212-217``, ``eval_multi_view.py:130-132``,
``helpers.py:245-252``, ``estimate_relative_pose.py:16-31``.

Fork-only behaviour DEFINED here (no reference source, parity unpinned):
* multi-frame ``cross`` layers: image t attends to the keypoints of ALL other images of its
  tuple, concatenated in image order (N_src = (T-1) N); every image's update in a layer
  uses the pre-layer descriptors of all images.  T=2 reduces to upstream.
* ``multi_frame_matching=False`` with T>2: every pair (i<j) is run through the 2-view
  network independently.
* ``conf_scores_i_j`` [B,N,1]: without ``conf_mlp`` = the match score exp(max log-assignment)
  of mutual matches above threshold (0 otherwise) - reference quirk E13
  (``eval_multi_view.py:130-132`` + ``bundle_adjust_io.py:81-82``); with ``conf_mlp`` =
  sigmoid(MLP([2D, D, 1])(cat(mdesc_i[:, n], mdesc_j[:, match(n)]))) for matched n, 0 else.
"""
import torch

from .sinkhorn import extract_matches, log_optimal_transport

DEFAULT_CONFIG = {
    "descriptor_dim": 256,
    "keypoint_encoder": [32, 64, 128, 256],
    "GNN_layers": ["self", "cross"] * 9,
    "num_heads": 4,
    "sinkhorn_iterations": 100,
    "match_threshold": 0.2,
    "multi_frame_matching": False,
    "tuple_size": 2,
    "conf_mlp": False,
    "full_output": True,
    # True: keep the autograd graph (the state_dict's tensors may require grad) - the checker of the training path's
    # gradients (tests/test_gpu_backward.py).  BatchNorm stays on its running statistics either way.
    "grad": False,
}

BN_EPS = 1e-5


def normalize_keypoints(kpts, height, width):
    """Upstream ``normalize_keypoints``: (k - [W/2,H/2]) / (0.7 max(W,H))."""
    size = kpts.new_tensor([[float(width), float(height)]])
    center = size / 2
    scaling = size.max(1, keepdim=True).values * 0.7
    return (kpts - center[:, None, :]) / scaling[:, None, :]


def conv1x1(x, w, b):
    """Conv1d(kernel 1) on [B,C,N]: w [O,C,1] or [O,C]."""
    w = w.reshape(w.shape[0], -1)
    return torch.einsum("oc,bcn->bon", w, x) + b[None, :, None]


def batchnorm_eval(x, sd, prefix):
    mean, var = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    g, be = sd[prefix + ".weight"], sd[prefix + ".bias"]
    return (x - mean[None, :, None]) / torch.sqrt(var[None, :, None] + BN_EPS) * g[None, :, None] + be[None, :, None]


def mlp(x, sd, prefix, n_layers):
    """Upstream ``MLP(channels)``: Sequential(conv, BN, ReLU, conv, BN, ReLU, ..., conv).

    Module indices: conv at 3*i, BN at 3*i+1 (absent after the last conv).
    """
    for i in range(n_layers):
        x = conv1x1(x, sd[f"{prefix}.{3 * i}.weight"], sd[f"{prefix}.{3 * i}.bias"])
        if i < n_layers - 1:
            x = torch.relu(batchnorm_eval(x, sd, f"{prefix}.{3 * i + 1}"))
    return x


def attention(q, k, v):
    """q [B,d,H,N], k,v [B,d,H,M] -> [B,d,H,N]; returns probs too."""
    dim = q.shape[1]
    scores = torch.einsum("bdhn,bdhm->bhnm", q, k) / dim ** 0.5
    prob = torch.softmax(scores, dim=-1)
    return torch.einsum("bhnm,bdhm->bdhn", prob, v), prob


def multi_head_attention(x, src, sd, prefix, heads):
    """Upstream ``MultiHeadedAttention``: channel c <-> (dd = c // H, h = c % H)."""
    b, D, _ = x.shape
    d = D // heads
    q = conv1x1(x, sd[prefix + ".proj.0.weight"], sd[prefix + ".proj.0.bias"]).view(b, d, heads, -1)
    k = conv1x1(src, sd[prefix + ".proj.1.weight"], sd[prefix + ".proj.1.bias"]).view(b, d, heads, -1)
    v = conv1x1(src, sd[prefix + ".proj.2.weight"], sd[prefix + ".proj.2.bias"]).view(b, d, heads, -1)
    o, _ = attention(q, k, v)
    return conv1x1(o.contiguous().view(b, D, -1), sd[prefix + ".merge.weight"], sd[prefix + ".merge.bias"])


def propagation(x, src, sd, layer, heads):
    """Upstream ``AttentionalPropagation``: mlp(cat([x, attn(x, src, src)]))."""
    msg = multi_head_attention(x, src, sd, f"gnn.layers.{layer}.attn", heads)
    return mlp(torch.cat([x, msg], dim=1), sd, f"gnn.layers.{layer}.mlp", 2)


def gnn(descs, sd, layer_names, heads):
    """descs: list of T tensors [B,D,N].  Returns the list after all layers."""
    T = len(descs)
    for li, name in enumerate(layer_names):
        new = []
        for t in range(T):
            if name == "cross":
                src = torch.cat([descs[s] for s in range(T) if s != t], dim=2)
            else:
                src = descs[t]
            new.append(descs[t] + propagation(descs[t], src, sd, li, heads))
        descs = new
    return descs


def encode(data, sd, cfg, m):
    """desc + kenc([x, y, score]) for image m (upstream ``KeypointEncoder``)."""
    kpts, scores, desc = data[f"keypoints{m}"], data[f"scores{m}"], data[f"descriptors{m}"]
    if f"image{m}" in data:
        h, w = data[f"image{m}"].shape[-2:]
    else:
        h, w = data[f"image_size{m}"]
    kn = normalize_keypoints(kpts, h, w)
    inp = torch.cat([kn.transpose(1, 2), scores.unsqueeze(1)], dim=1)
    n_layers = len(cfg["keypoint_encoder"]) + 1
    return desc + mlp(inp, sd, "kenc.encoder", n_layers)


def pair_outputs(md0, md1, sd, cfg, out, i, j):
    D = md0.shape[1]
    scores = torch.einsum("bdn,bdm->bnm", md0, md1) / D ** 0.5
    Z = log_optimal_transport(scores, sd["bin_score"].reshape(()), cfg["sinkhorn_iterations"])
    out[f"scores_{i}_{j}"] = Z
    if not cfg.get("full_output", True):
        return
    idx0, idx1, ms0, ms1 = extract_matches(Z, cfg["match_threshold"])
    out[f"matches{i}_{i}_{j}"] = idx0
    out[f"matches{j}_{i}_{j}"] = idx1
    out[f"matching_scores{i}_{i}_{j}"] = ms0
    out[f"matching_scores{j}_{i}_{j}"] = ms1
    valid = idx0 >= 0
    if cfg.get("conf_mlp", False):
        b, _, n = md0.shape
        gather = idx0.clamp(min=0)[:, None, :].expand(b, D, n)
        feat = torch.cat([md0, md1.gather(2, gather)], dim=1)
        conf = torch.sigmoid(mlp(feat, sd, "conf_mlp", 2))[:, 0, :]
        conf = torch.where(valid, conf, conf.new_tensor(0))
    else:
        conf = torch.where(valid, ms0, ms0.new_tensor(0))
    out[f"conf_scores_{i}_{j}"] = conf.unsqueeze(-1)


def count_images(data):
    T = 0
    while f"keypoints{T}" in data:
        T += 1
    return T


def matcher_forward(data, sd, config=None):
    """dict in -> dict out, keys as the reference's call sites read them."""
    cfg = dict(DEFAULT_CONFIG)
    cfg.update(config or {})
    heads = cfg["num_heads"]
    T = count_images(data)
    out = {}
    with torch.set_grad_enabled(bool(cfg.get("grad", False))):
        enc = [encode(data, sd, cfg, m) for m in range(T)]
        if cfg["multi_frame_matching"] or T == 2:
            descs = gnn(enc, sd, cfg["GNN_layers"], heads)
            md = [conv1x1(d, sd["final_proj.weight"], sd["final_proj.bias"]) for d in descs]
            for j in range(T):
                for i in range(j):
                    pair_outputs(md[i], md[j], sd, cfg, out, i, j)
            out["_mdesc"] = md
        else:
            for j in range(T):
                for i in range(j):
                    d = gnn([enc[i], enc[j]], sd, cfg["GNN_layers"], heads)
                    m0 = conv1x1(d[0], sd["final_proj.weight"], sd["final_proj.bias"])
                    m1 = conv1x1(d[1], sd["final_proj.weight"], sd["final_proj.bias"])
                    pair_outputs(m0, m1, sd, cfg, out, i, j)
    return out


def dense_flops_per_tuple(T, N, D, layer_names, kenc=(32, 64, 128, 256)):
    """Algorithmic dense flops per tuple, SURVEY.md 8(d) formula (joint multi-frame GNN)."""
    ch = [3] + list(kenc) + [D]
    f = T * 2 * N * sum(a * b for a, b in zip(ch[:-1], ch[1:]))
    for name in layer_names:
        n_src = N if (name == "self" or T == 2) else (T - 1) * N
        f += T * (20 * N * D * D + 4 * N * n_src * D)
    f += T * 2 * N * D * D
    f += (T * (T - 1) // 2) * 2 * N * N * D
    return f
