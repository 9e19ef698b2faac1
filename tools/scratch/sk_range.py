import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import e2e_multi_view_matching_amd as E
from e2e_multi_view_matching_amd import _lib
from oracle.sinkhorn import log_optimal_transport
ctx = _lib.context(torch.device("cuda", 0))
for scale in (5, 10, 20, 40, 80, 160):
    g = torch.Generator().manual_seed(scale)
    s = torch.randn(2, 300, 280, generator=g) * scale
    ref = log_optimal_transport(s.double(), 1.0, 100).float()
    ref32 = log_optimal_transport(s, 1.0, 100)
    os.environ.pop("E2EMV_SINKHORN", None)
    a = E.log_optimal_transport(s.cuda(), 1.0, 100).cpu()
    rc = ctx.lib.e2emv_sync(ctx.h, None)
    os.environ["E2EMV_SINKHORN"] = "stream"
    b = E.log_optimal_transport(s.cuda(), 1.0, 100).cpu()
    os.environ.pop("E2EMV_SINKHORN", None)
    ia = (a[:, :-1, :-1].argmax(2) == ref[:, :-1, :-1].argmax(2)).float().mean()
    print(f"scale {scale}: max|logZ| {float(ref.abs().max()):.0f}  resident-vs-fp64 {float((a-ref).abs().max()):.2e}  stream-vs-fp64 {float((b-ref).abs().max()):.2e}  "
          f"torch32-vs-fp64 {float((ref32-ref).abs().max()):.2e}  finite {bool(torch.isfinite(a).all())} sync rc {rc} argmax agree {float(ia):.4f}")
