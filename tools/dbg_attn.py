import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import e2e_multi_view_matching_amd as E
from test_gpu_kernels import _attention_ref
gpu = torch.device('cuda', 0)
def run(tag, qkv, B, T, n_valid, cross, waves=4):
    ref = _attention_ref(qkv, B, T, n_valid, 4, cross)
    out = E.attention_p2(qkv.to(gpu), B, T, n_valid, 4, cross, waves=waves).cpu()
    d = (out[:, :n_valid].double() - ref[:, :n_valid]).abs().view(B * T, n_valid, 4, 64)
    print(f"{tag:40s} n_valid {n_valid:4d} waves {waves}: err d<32 {float(d[..., :32].max()):.2e}  d>=32 {float(d[..., 32:].max()):.2e}", flush=True)
g = torch.Generator().manual_seed(1)
base = torch.randn(2, 256, 768, generator=g) * 1.5
for nv in (5, 32, 33, 64, 65, 96, 128, 129, 256):
    n_rows = 128 if nv <= 128 else 256
    run('random', base[:, :n_rows].contiguous(), 1, 2, nv, 0)
q = base[:, :128].clone()
x = q.clone(); x[..., 512:] = (x[..., 512:] * 16).half().float() / 16
run('V fp16-exact', x, 1, 2, 128, 0)
x = q.clone(); x[..., 256:512] = x[..., 256:512].half().float()
run('K fp16-exact', x, 1, 2, 128, 0)
x = q.clone(); x[..., 512:] = 1.0
run('V == 1', x, 1, 2, 128, 0)
x = q.clone(); x[..., 512:] = torch.arange(128).float()[None, :, None].expand(2, 128, 256) / 128.0
run('V = key index / 128', x, 1, 2, 128, 0)
x = q.clone(); x[..., 512:] = (torch.arange(256).float()[None, None, :] % 64 + 1) / 64.0
run('V = (d + 1) / 64', x, 1, 2, 128, 0)
run('random waves 8', q, 1, 2, 128, 0, waves=8)
