import os, sys, math
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import e2e_multi_view_matching_amd as E
torch.manual_seed(0)
M = N = 128
s = torch.randn(1, M, N) * 3
os.environ["E2EMV_SKR_FLAGS"] = "4"; os.environ["E2EMV_SKR_DEBUG"] = "/tmp/dump.bin"
out = E.log_optimal_transport(s.cuda(), 1.0, 2).cpu()
d = np.fromfile("/tmp/dump.bin", dtype=np.float32).reshape(2, 3 * M + N + 8)
alpha = 1.0; mu = 1 / (M + N); muM = N / (M + N); nuN = M / (M + N)
m = torch.maximum(s.max(2).values, torch.tensor(alpha))[0]
K = torch.exp(s[0] - m[:, None]); r = torch.exp(alpha - m)
b = torch.ones(N); bN = torch.tensor(1.0)
for it in range(2):
    a = mu / ((K * b[None]).sum(1) + r * bN)
    aM = muM / (b.sum() + bN)
    b = mu / ((K * a[:, None]).sum(0) + aM)
    bN = nuN / ((r * a).sum() + aM)
    da, dK, dm, db = d[it, :M], d[it, M:2 * M], d[it, 2 * M:3 * M], d[it, 3 * M:3 * M + N]
    print("it", it, "a relerr", float(np.abs(da / a.numpy() - 1).max()), "K0 relerr", float(np.abs(dK / K[:, 0].numpy() - 1).max()),
          "m err", float(np.abs(dm - m.numpy()).max()), "b relerr", float(np.abs(db / b.numpy() - 1).max()),
          "aM", d[it, 3 * M + N], float(aM), "bN", d[it, 3 * M + N + 1], float(bN))
    bad = np.nonzero(np.abs(da / a.numpy() - 1) > 1e-4)[0]
    print("  bad a rows:", bad[:20], da[bad[:5]], a.numpy()[bad[:5]])
    badb = np.nonzero(np.abs(db / b.numpy() - 1) > 1e-4)[0]
    print("  bad b cols:", badb[:20])
from oracle.sinkhorn import log_optimal_transport
print("final err", float((out - log_optimal_transport(s, 1.0, 2)).abs().max()))
