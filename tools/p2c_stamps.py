#!/usr/bin/env python
"""Where the time of the chained plane GEMMs goes (gemm_p2c.hip): the forward of configs[1] on a measurement build
(tools/libe2emv_stamps.bin, -DE2EMV_STAMPS) with the chain kernel's variants (E2EMV_P2C_DBG: 4 no epilogues, 512 no wait at the
hard hand-off, 516 both, 8 in-kernel timestamps per tile: K loop | epilogue | hand-off of two workgroups).  Variants other than
0 and 8 compute wrong results - only their time is read.  `--build` makes the measurement library (no GPU needed)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "libe2emv_stamps.bin")

if "--build" in sys.argv:
    from e2e_multi_view_matching_amd.build import build_library
    print(build_library(defines=["E2EMV_STAMPS"], out=LIB, verbose=True))
    sys.exit(0)

if "--one" in sys.argv:
    import torch
    import e2e_multi_view_matching_amd as E
    from e2e_multi_view_matching_amd import _lib
    from e2e_multi_view_matching_amd.synthetic import make_tuples
    dev = torch.device("cuda", 0)
    ctx = _lib.context(dev)
    dbg = int(os.environ.get("E2EMV_P2C_DBG", "0"))
    torch.manual_seed(0)
    model = E.MultiViewMatcher({"sinkhorn_iterations": 100, "conf_mlp": True, "match_threshold": 0.2}).eval().to(dev)
    data = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in make_tuples(seed=1, batch=32, tuple_size=2, n_kpts=1024).items()}
    with torch.no_grad():
        for _ in range(2):
            model(data)
        torch.cuda.synchronize()
        ctx.call("e2emv_profile", 1)
        _lib.profile_read(ctx, reset=True)
        n = 5
        for _ in range(n):
            model(data)
        torch.cuda.synchronize()
        pr = _lib.profile_read(ctx, reset=True)
        ctx.call("e2emv_profile", 0)
    ch, at = pr["gemm_chain"], pr["attention"]
    print(f"dbg={dbg:3d}  chain {1e3 * ch['ms'] / max(ch['launches'], 1):7.1f} us per launch ({ch['launches'] // n} per forward)   attention "
          f"{1e3 * at['ms'] / max(at['launches'], 1):7.1f} us   gemm_qkv {1e3 * pr['gemm_qkv']['ms'] / max(pr['gemm_qkv']['launches'], 1):6.1f} us", flush=True)
    sys.exit(0)

for dbg in ([int(a) for a in sys.argv[1:] if a.isdigit()] or [0, 4, 512, 516, 8]):
    env = dict(os.environ, E2EMV_LIBRARY=LIB, E2EMV_P2C_DBG=str(dbg))
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    out = "\n".join(l for l in r.stdout.splitlines() if "amdgpu.ids" not in l)
    print(out[-9000:], flush=True)
