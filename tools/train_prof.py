import os, sys, time, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import e2e_multi_view_matching_amd as E
from e2e_multi_view_matching_amd import synthetic, _lib
dev = torch.device("cuda:0")
Bt, Nt = 4, 1024
cfg = {"GNN_layers": ["self", "cross"] * 9, "sinkhorn_iterations": 100, "conf_mlp": True, "full_output": True, "frozen_batchnorm": True}
torch.manual_seed(0)
model = synthetic.identity_like_state(E.MultiViewMatcher(cfg)).to(dev).train()
opt = torch.optim.SGD(model.parameters(), lr=1e-6)
data = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synthetic.make_tuples(batch=Bt, tuple_size=2, n_kpts=Nt, seed=5).items()}
gt = data["gt_matches0_0_1"]
idx = torch.full((Bt, Nt + 1), Nt, dtype=torch.int64, device=dev)
idx[:, :Nt] = torch.where(gt >= 0, gt, torch.full_like(gt, Nt))
ctx = _lib.context(dev)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for it in range(3):
    t0 = T()
    model.zero_grad()
    t1 = T()
    owner = (model._token, model._fingerprint())
    ctx.sent_owner = None
    model._send_weights(ctx, owner)
    t2 = T()
    md = model._model_desc()
    ctx.call("e2emv_commit_weights", ctypes.byref(md)); ctx.weights_owner = owner
    t3 = T()
    ctx.call("e2emv_train_commit", ctypes.byref(md)); ctx.train_owner = owner
    t4 = T()
    res = model(data)
    nll = -torch.gather(res["scores_0_1"], 2, idx[:, :, None]).mean()
    pred, _ = E.run_weighted_8_point(data, res, 0, 1, choose_closest=True, target_T_021=data["T_0to1"])
    loss = nll + E.compute_rotation_error(pred, data["T_0to1"]) + E.compute_translation_error_as_angle(pred, data["T_0to1"])
    t5 = T()
    loss.backward()
    t6 = T()
    opt.step()
    t7 = T()
    print(f"zero_grad {1e3*(t1-t0):.1f} | send_weights {1e3*(t2-t1):.1f} | commit_weights {1e3*(t3-t2):.1f} | train_commit {1e3*(t4-t3):.1f} | fwd+loss {1e3*(t5-t4):.1f} | backward {1e3*(t6-t5):.1f} | opt.step {1e3*(t7-t6):.1f}", flush=True)
