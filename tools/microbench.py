#!/usr/bin/env python
"""Micro-benchmarks of the building-block kernels through the C ABI (HIP events via torch)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e2e_multi_view_matching_amd as E  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="gemm,attn,sinkhorn")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    if "gemm" in args.what:
        print("== gemm_nt  (M, N, K) -> us, TFLOP/s")
        for (M, N, K) in [(65536, 256, 256), (65536, 256, 512), (65536, 256, 1024), (65536, 256, 2048), (65536, 512, 512),
                          (65536, 768, 256), (65536, 256, 128), (32768, 256, 256), (131072, 256, 256), (65536, 128, 256)]:
            A = torch.randn(M, K, device=dev)
            W = torch.randn(N, K, device=dev)
            b = torch.randn(N, device=dev)
            C = torch.empty(M, N, device=dev)
            from e2e_multi_view_matching_amd import _lib
            ctx = _lib.context(dev)

            def run():
                ctx.call("e2emv_gemm_nt", 1, M, N, K, K, _lib.ptr(A), K, 0, None, 0, 0, _lib.ptr(W), K, 0, _lib.ptr(b), None, 0, 0,
                         _lib.ptr(C), N, 0, 1.0, 0, _lib.stream_ptr(dev))
            ms = timeit(run)
            print(f"  {M:7d} {N:5d} {K:5d}  {ms * 1e3:9.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TF")
    if "g3" in args.what:
        print("== gemm_bf16x3 kernel alone (HIP events inside the library) (M, N, K) -> us, TFLOP/s fp32-equivalent")
        from e2e_multi_view_matching_amd import _lib
        ctx = _lib.context(dev)
        for (M, N, K) in [(65536, 256, 256), (65536, 256, 512), (65536, 512, 512), (65536, 768, 256), (65536, 256, 2048), (32768, 256, 8192)]:
            A = torch.randn(M, K, device=dev)
            W = torch.randn(N, K, device=dev)
            for v1 in ((0, 4) if os.environ.get('E2EMV_X3_DEBUG') else (0, 4, 2)):
                kw = dict(all_planes=v1 == 2, f16x2=v1 == 4)
                for _ in range(2):
                    E.gemm_bf16x3(A, W, **kw)
                ctx.call("e2emv_profile", 1)
                _lib.profile_read(ctx, reset=True)
                for _ in range(10):
                    E.gemm_bf16x3(A, W, **kw)
                pr = _lib.profile_read(ctx, reset=True)["gemm"]
                ctx.call("e2emv_profile", 0)
                ms = pr["ms"] / pr["launches"]
                print(f"  {M:7d} {N:5d} {K:5d}  {ms * 1e3:9.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TF  ({ {0: 'gemm_x3 bf16x3', 4: 'gemm_x3 f16x2', 2: 'gemm3 all-planes'}[v1]})")
    if "a3" in args.what:
        print("== attention_bf16x3 kernel alone (B pairs, N) -> us, TFLOP/s fp32-equivalent")
        from e2e_multi_view_matching_amd import _lib
        ctx = _lib.context(dev)
        for (B, N) in [(32, 1024), (8, 2048)]:
            qkv = torch.randn(B * 2, N, 768, device=dev)
            for cross in (0, 1):
              for kernel in ("planes", "fused", "f16x2"):
                for _ in range(2):
                    E.attention_bf16x3(qkv, B, 2, N, 4, cross, kernel=kernel)
                ctx.call("e2emv_profile", 1)
                _lib.profile_read(ctx, reset=True)
                for _ in range(5):
                    E.attention_bf16x3(qkv, B, 2, N, 4, cross, kernel=kernel)
                pr = _lib.profile_read(ctx, reset=True)["attention"]
                ctx.call("e2emv_profile", 0)
                ms = pr["ms"] / pr["launches"]
                print(f"  B={B:3d} N={N:5d} cross={cross}  {ms * 1e3:9.1f} us  {B * 2 * 4.0 * N * N * 256 / ms / 1e9:7.1f} TF  ({kernel})")
    if "p2" in args.what:
        print("== f16x2 GEMM: gemm_h2 (fp32 activations, round 2) vs gemm_p2 (plane activations)  (M, N, K) -> us, TFLOP/s fp32-equivalent")
        from e2e_multi_view_matching_amd import _lib
        ctx = _lib.context(dev)

        def family(fn, slot, n=10):
            for _ in range(2):
                fn(1)
            ctx.call("e2emv_profile", 1)
            _lib.profile_read(ctx, reset=True)
            fn(n)
            pr = _lib.profile_read(ctx, reset=True)[slot]
            ctx.call("e2emv_profile", 0)
            return pr["ms"] / max(pr["launches"], 1)
        for (M, N, K, K1) in [(65536, 768, 256, 256), (65536, 512, 512, 256), (65536, 256, 512, 512), (65536, 256, 256, 256)]:
            A = torch.randn(M, K1, device=dev)
            A2 = torch.randn(M, K - K1, device=dev) if K1 < K else None
            W = torch.randn(N, K, device=dev) / K ** 0.5
            Afull = torch.cat([A, A2], 1) if A2 is not None else A
            ms_h2 = family(lambda n: [E.gemm_bf16x3(Afull, W, f16x2=True) for _ in range(n)], "gemm")
            for planes_out, ex in ((False, False), (True, False), (True, True)):
                ms = family(lambda n: E.gemm_p2(A, W, A2=A2, planes_out=planes_out, reps=n, exponents=ex), "gemm")
                print(f"  {M:7d} {N:5d} {K:5d}  gemm_h2 {ms_h2 * 1e3:7.1f} us {2.0 * M * N * K / ms_h2 / 1e9:6.1f} TF | gemm_p2 "
                      f"{'planes' if planes_out else 'fp32  '} out{' + tile exponents' if ex else ''} {ms * 1e3:7.1f} us {2.0 * M * N * K / ms / 1e9:6.1f} TF")
        print("== f16x2 attention: attention_h2f (fp32 q|k|v) vs attention_p2 (plane operands)  -> us, TFLOP/s fp32-equivalent")
        for (B, N) in [(32, 1024), (8, 2048)]:
            qkv = torch.randn(B * 2, N, 768, device=dev)
            fl = B * 2 * 4.0 * N * N * 256
            for cross in (0, 1):
                ms_h = family(lambda n: [E.attention_bf16x3(qkv, B, 2, N, 4, cross, kernel="f16x2") for _ in range(n)], "attention", 5)
                line = f"  B={B:3d} N={N:5d} cross={cross}  h2f {ms_h * 1e3:7.1f} us {fl / ms_h / 1e9:6.1f} TF |"
                for waves in (4, 8):
                    ms = family(lambda n: E.attention_p2(qkv, B, 2, N, 4, cross, waves=waves, reps=n), "attention", 5)
                    line += f" p2/{waves}w {ms * 1e3:7.1f} us {fl / ms / 1e9:6.1f} TF |"
                print(line)
    if "ap2" in args.what:
        print("== attention on planes: attention_p2 (8 waves, 2 per SIMD) vs attention_p2w (4 waves, 1 per SIMD) -> us, TFLOP/s fp32-equivalent; max |diff|")
        from e2e_multi_view_matching_amd import _lib
        ctx = _lib.context(dev)

        def fam(fn, n=10):
            for _ in range(2):
                fn(1)
            ctx.call("e2emv_profile", 1)
            _lib.profile_read(ctx, reset=True)
            fn(n)
            pr = _lib.profile_read(ctx, reset=True)["attention"]
            ctx.call("e2emv_profile", 0)
            return pr["ms"] / max(pr["launches"], 1)
        for (B, T, N, nv) in [(32, 2, 1024, 1024), (8, 5, 1024, 1024), (8, 5, 2048, 2048), (32, 2, 1024, 1000), (32, 2, 512, 512)]:
            qkv = torch.randn(B * T, N, 768, device=dev) * 1.5
            for cross in (0, 1):
                fl = B * T * 4.0 * nv * nv * (T - 1 if cross else 1) * 256
                line = f"  B={B:3d} T={T} N={N:5d} valid={nv:5d} cross={cross} |"
                outs = {}
                for waves in (8, 1):
                    ms = fam(lambda n: E.attention_p2(qkv, B, T, nv, 4, cross, waves=waves, reps=n))
                    outs[waves] = E.attention_p2(qkv, B, T, nv, 4, cross, waves=waves)
                    line += f" {'p2w' if waves == 1 else 'p2/8'} {ms * 1e3:7.1f} us {fl / ms / 1e9:6.1f} TF |"
                d = float((outs[8][:, :nv] - outs[1][:, :nv]).abs().max())
                print(line + f" max|diff| {d:.2e}")
    if "attn" in args.what:
        print("== attention (B pairs, N) -> us, TFLOP/s")
        for (B, N) in [(32, 1024), (8, 1024), (32, 512), (8, 2048)]:
            qkv = torch.randn(B * 2, N, 768, device=dev)
            for cross in (0, 1):
                ms = timeit(lambda: E.attention(qkv, B, 2, N, 4, cross), iters=10)
                fl = B * 2 * 4.0 * N * N * 256
                print(f"  B={B:3d} N={N:5d} cross={cross}  {ms * 1e3:9.1f} us  {fl / ms / 1e9:7.1f} TF")
    if "sinkhorn" in args.what:
        print("== sinkhorn 100 iters (B, N) -> ms, algorithmic GB/s")
        for (B, N) in [(32, 1024), (16, 1024), (8, 1024), (4, 1024), (2, 1024), (8, 2048), (64, 512)]:
            s = torch.randn(B, N, N, device=dev)
            ms = timeit(lambda: E.log_optimal_transport(s, 1.0, 100), iters=5, warm=1)
            by = B * 202 * (N + 1) ** 2 * 4
            print(f"  B={B:3d} N={N:5d}  {ms:8.3f} ms  {by / ms / 1e6:8.1f} GB/s   S = {B * N * N * 4 / 2**20:.0f} MiB, one pass per iteration: "
                  f"{B * 100 * (N + 1) ** 2 * 4 / ms / 1e6:8.1f} GB/s")


if __name__ == "__main__":
    main()
