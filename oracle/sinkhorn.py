"""Oracle: Sinkhorn log-space optimal transport with dustbins, and mutual-argmax matching.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Unfused torch-CPU ops, the same op sequence
the reference executes on its device (add, logsumexp, max, gather).

Follows upstream SuperGlue ``models/superglue.py`` (``log_sinkhorn_iterations``,
``log_optimal_transport``, the match block of ``SuperGlue.forward``); the reference calls it
inside the absent ``MultiViewMatcher.forward`` (call sites ``helpers.py:246``,
``eval_pairs.py:212``).  Pinned by tests/golden/sinkhorn_hf_*.npz against the HF port
(``modeling_superglue.py:71-142`` and ``:629-642``).
"""
import math

import torch


def log_sinkhorn_iterations(Z, log_mu, log_nu, iters):
    """u,v updates in log space.  Z [B,M+1,N+1]; log_mu [B,M+1]; log_nu [B,N+1]."""
    u, v = torch.zeros_like(log_mu), torch.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - torch.logsumexp(Z + v.unsqueeze(1), dim=2)
        v = log_nu - torch.logsumexp(Z + u.unsqueeze(2), dim=1)
    return Z + u.unsqueeze(2) + v.unsqueeze(1)


def log_optimal_transport(scores, alpha, iters):
    """scores [B,M,N] -> log assignment [B,M+1,N+1] (dustbin row/col = alpha).

    log_mu = [-log(M+N)]*M ++ [log N - log(M+N)], log_nu symmetric; result has
    log(M+N) added back ("multiply probabilities by M+N").
    """
    b, m, n = scores.shape
    one = scores.new_tensor(1)
    ms, ns = (m * one).to(scores), (n * one).to(scores)
    alpha = torch.as_tensor(alpha, dtype=scores.dtype)
    bins0 = alpha.expand(b, m, 1)
    bins1 = alpha.expand(b, 1, n)
    alpha_ = alpha.expand(b, 1, 1)
    couplings = torch.cat([torch.cat([scores, bins0], -1), torch.cat([bins1, alpha_], -1)], 1)
    norm = -(ms + ns).log()
    log_mu = torch.cat([norm.expand(m), ns.log()[None] + norm])
    log_nu = torch.cat([norm.expand(n), ms.log()[None] + norm])
    log_mu, log_nu = log_mu[None].expand(b, -1), log_nu[None].expand(b, -1)
    Z = log_sinkhorn_iterations(couplings, log_mu, log_nu, iters)
    return Z - norm


def extract_matches(Z, match_threshold):
    """Mutual nearest neighbour in the (M x N) core of the log assignment.

    Returns indices0 [B,M] int64 (-1 = unmatched), indices1 [B,N] int64, mscores0 [B,M],
    mscores1 [B,N].  torch's CPU ``max`` returns the FIRST maximal index - the tie rule
    the HIP kernel must reproduce.
    """
    core = Z[:, :-1, :-1]
    max0, max1 = core.max(2), core.max(1)
    idx0, idx1 = max0.indices, max1.indices
    ar0 = torch.arange(idx0.shape[1])[None]
    ar1 = torch.arange(idx1.shape[1])[None]
    mutual0 = ar0 == idx1.gather(1, idx0)
    mutual1 = ar1 == idx0.gather(1, idx1)
    zero = Z.new_tensor(0)
    ms0 = torch.where(mutual0, max0.values.exp(), zero)
    ms1 = torch.where(mutual1, ms0.gather(1, idx1), zero)
    valid0 = mutual0 & (ms0 > match_threshold)
    valid1 = mutual1 & valid0.gather(1, idx1)
    idx0 = torch.where(valid0, idx0, idx0.new_tensor(-1))
    idx1 = torch.where(valid1, idx1, idx1.new_tensor(-1))
    return idx0, idx1, ms0, ms1


def sinkhorn_bytes_per_pair(n, iters):
    """Algorithmic byte model of SURVEY.md 8(d): (2*iters+2)*(N+1)^2*4."""
    return (2 * iters + 2) * (n + 1) ** 2 * 4


def _selfcheck():  # pragma: no cover
    s = torch.randn(2, 5, 7)
    z = log_optimal_transport(s, 1.0, 50)
    p = z.exp()
    assert abs(p[:, :-1].sum(2) - 1).max() < 1e-3 and math.isfinite(float(z.sum()))
