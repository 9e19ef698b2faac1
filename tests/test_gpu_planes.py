"""Plane-format kernels of the f16x2 mode (-m gpu): gemm_p2.hip / attention_p2.hip through the C ABI vs torch fp64.

The building-block entry points take fp32 tensors, convert them to the plane format (csrc/p2.h) with helper kernels, run
the kernel under test and convert back; the bars are those of the round-2 f16x2 kernels (tests/test_gpu_kernels.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gemm_case(g, M, N, K, act_scale=1.0, K1=None):
    A = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g) * 2) * act_scale
    A = A.clamp(-6e4, 6e4)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * act_scale
    return A, W, b


def _err(out, ref, scale):
    return float(((out.double() - ref).abs() / scale).max())


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 200, 64), (1024, 512, 512), (65, 36, 256), (777, 768, 256)])
@pytest.mark.parametrize("act_scale", [1.0, 1e-3, 3e3])
def test_gemm_p2_has_fp32_class_accuracy(gpu, M, N, K, act_scale):
    import e2e_multi_view_matching_amd as E
    g = torch.Generator().manual_seed(M + N + K)
    A, W, b = _gemm_case(g, M, N, K, act_scale)
    ref = A.double() @ W.double().T + b.double()
    scale = (A.double().abs() @ W.double().abs().T) + b.double().abs()
    out = E.gemm_p2(A.to(gpu), W.to(gpu), bias=b.to(gpu)).cpu()
    out32 = E.gemm_nt(A.to(gpu), W.to(gpu), bias=b.to(gpu)).cpu()
    e, e32 = _err(out, ref, scale), _err(out32, ref, scale)
    # rows of typical magnitude below 6e-5 (act_scale 1e-3 x e^-4) sit under the range where the low plane is a normal fp16
    # number: the representation degrades gracefully to an ABSOLUTE 2^-36 there (the same in gemm_h2)
    assert e < (1e-6 if act_scale < 1e-2 else 5e-7), (e, e32)
    assert e < 4 * e32 + (6e-7 if act_scale < 1e-2 else 2e-7), (e, e32)
    outr = E.gemm_p2(A.to(gpu), W.to(gpu), bias=b.to(gpu), relu=True).cpu()
    assert torch.equal(outr, out.clamp_min(0))


@pytest.mark.parametrize("M,N,K1,K2", [(257, 512, 256, 256), (1000, 256, 512, 0), (513, 96, 64, 32)])
def test_gemm_p2_plane_epilogue_two_segments_residual(gpu, M, N, K1, K2):
    """The forward pass's shapes of use: two K segments (x | attention), ReLU, residual read from its planes, plane output.
    A plane output carries 22 significant bits: the bar adds 2^-22 of |value| to the contraction bar."""
    import e2e_multi_view_matching_amd as E
    g = torch.Generator().manual_seed(M + N)
    K = K1 + K2
    A, W, b = _gemm_case(g, M, N, K)
    R = torch.randn(M, N, generator=g) * 3
    A1, A2 = A[:, :K1].contiguous(), (A[:, K1:].contiguous() if K2 else None)
    for relu in (False, True):
        core = A.double() @ W.double().T + b.double()
        ref = (core.clamp_min(0) if relu else core) + R.double()
        scale = (A.double().abs() @ W.double().abs().T) + b.double().abs() + R.double().abs()
        for planes_out in (False, True):
            out = E.gemm_p2(A1.to(gpu), W.to(gpu), bias=b.to(gpu), relu=relu, A2=A2.to(gpu) if K2 else None, residual=R.to(gpu),
                            planes_out=planes_out).cpu()
            e = _err(out, ref, scale)
            assert e < 5e-7 + 2.5e-7 + (2.5e-7 if planes_out else 0.0), (relu, planes_out, e)


def test_gemm_p2_random_shapes(gpu):
    import e2e_multi_view_matching_amd as E
    g = torch.Generator().manual_seed(2025)
    for case in range(24):
        M = int(torch.randint(1, 1500, (1,), generator=g))
        N = [4 * int(torch.randint(1, 200, (1,), generator=g)), 256 * int(torch.randint(1, 4, (1,), generator=g))][case % 2]
        K = 32 * int(torch.randint(1, 25, (1,), generator=g))
        relu = bool(case & 2)
        A = torch.randn(M, K, generator=g) * float(torch.exp(torch.randn(1, generator=g) * 2))
        W = torch.randn(N, K, generator=g) / K ** 0.5
        b = torch.randn(N, generator=g) if case % 3 else None
        ref = A.double() @ W.double().T + (b.double() if b is not None else 0.0)
        if relu:
            ref = ref.clamp_min(0)
        out = E.gemm_p2(A.to(gpu), W.to(gpu), bias=b.to(gpu) if b is not None else None, relu=relu, planes_out=(N % 32 == 0 and case % 4 == 1)).cpu()
        assert out.shape == (M, N)
        scale = (A.double().abs() @ W.double().abs().T) + (b.double().abs() if b is not None else 0.0)
        e = _err(out, ref, scale)
        assert e < 7.5e-7, (case, M, N, K, relu, e)


@pytest.mark.parametrize("n_img,n_rows", [(2, 128), (3, 256), (1, 1024)])
def test_qkv_projection_attention_operand_epilogue(gpu, n_img, n_rows):
    """q | k plain planes and the transposed, key-permuted V planes, read back through the helper that mirrors the attention
    kernel's addressing: every element of x W^T + b must come back (to the 22 bits the planes carry)."""
    import e2e_multi_view_matching_amd as E
    g = torch.Generator().manual_seed(n_rows)
    D = 256
    X = torch.randn(n_img * n_rows, D, generator=g) * torch.exp(torch.randn(n_img * n_rows, 1, generator=g))
    W = torch.randn(3 * D, D, generator=g) / D ** 0.5
    b = torch.randn(3 * D, generator=g)
    ref = X.double() @ W.double().T + b.double()
    scale = (X.double().abs() @ W.double().abs().T) + b.double().abs()
    out = E.qkv_p2(X.to(gpu), W.to(gpu), b.to(gpu), n_rows).cpu()
    e = _err(out, ref, scale)
    assert e < 1e-6, e


def _attention_ref(qkv, B, T, n_valid, H, cross):
    from test_gpu_kernels import _attention_ref as r
    return r(qkv, B, T, n_valid, H, cross)


@pytest.mark.parametrize("B,T,n_rows,n_valid,cross", [(2, 2, 128, 128, 0), (2, 2, 256, 200, 1), (1, 3, 256, 131, 1),
                                                       (1, 2, 128, 5, 0), (1, 2, 512, 300, 0), (1, 2, 512, 512, 1), (1, 4, 640, 577, 1)])
@pytest.mark.parametrize("waves", [4, 8, 1])  # 1 = attention_p2w.hip (one wave per SIMD, 256 queries per workgroup)
def test_attention_p2(gpu, B, T, n_rows, n_valid, cross, waves):
    import e2e_multi_view_matching_amd as E
    g = torch.Generator().manual_seed(n_valid)
    qkv = torch.randn(B * T, n_rows, 3 * 256, generator=g) * 1.5
    ref = _attention_ref(qkv, B, T, n_valid, 4, cross)
    out = E.attention_p2(qkv.to(gpu), B, T, n_valid, 4, cross, waves=waves).cpu()
    out32 = E.attention(qkv.to(gpu), B, T, n_valid, 4, cross).cpu()
    err = float((out[:, :n_valid].double() - ref[:, :n_valid]).abs().max())
    err32 = float((out32[:, :n_valid].double() - ref[:, :n_valid]).abs().max())
    assert err < 2e-5 and err < 3 * err32 + 1e-6, (err, err32)


@pytest.mark.parametrize("waves", [0, 1])
def test_attention_p2_spiked_key_forces_rescale(gpu, waves):
    """waves = 1 (attention_p2w): the spike sits in the FOURTH key tile of one query of each stream's block - the fast path's
    sum check must send exactly those tiles through the slow path (true maximum, O and l rescaled)."""
    import e2e_multi_view_matching_amd as E
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(2, 256, 768, generator=g)
    qkv[0, 200, 256:512] = qkv[0, 17, 0:256] * 6.0  # key 200 aligned with query 17, all heads
    qkv[1, 77, 256:512] = qkv[1, 40, 0:256] * 9.0   # key 77 (second tile) aligned with query 40 (second stream of its wave)
    ref = _attention_ref(qkv, 1, 2, 256, 4, 0)
    out = E.attention_p2(qkv.to(gpu), 1, 2, 256, 4, 0, waves=waves).cpu()
    assert float((out.double() - ref).abs().max()) < 2e-5


@pytest.mark.parametrize("waves", [0, 1])
@pytest.mark.parametrize("qs,ks,vs", [(1.0, 1.0, 1.0), (0.05, 0.05, 0.01), (6.0, 6.0, 300.0), (30.0, 0.2, 1e-3)])
def test_attention_p2_over_operand_magnitudes(gpu, qs, ks, vs, waves):
    import e2e_multi_view_matching_amd as E
    B, T, n_rows, n_valid, cross = 1, 2, 256, 256, 1
    g = torch.Generator().manual_seed(7)
    qkv = torch.randn(B * T, n_rows, 3 * 256, generator=g)
    qkv[..., :256] *= qs
    qkv[..., 256:512] *= ks
    qkv[..., 512:] *= vs
    ref = _attention_ref(qkv, B, T, n_valid, 4, cross)
    out = E.attention_p2(qkv.to(gpu), B, T, n_valid, 4, cross, waves=waves).cpu()
    out32 = E.attention(qkv.to(gpu), B, T, n_valid, 4, cross).cpu()
    err = float((out.double() - ref).abs().max()) / vs
    err32 = float((out32.double() - ref).abs().max()) / vs
    assert err < 3 * err32 + 2e-6, (err, err32)


@pytest.mark.parametrize("kernel", ["p2", "p2w", "f16x2"])
def test_attention_constant_v_exposes_operand_hazards(gpu, kernel):
    """V == 1 makes every output element exactly sum(p) / sum(p) = 1, whatever the keys: the two 32-dim halves of a head
    go through separate MFMA chains fed from the SAME softmax-numerator registers, so any difference between them is an
    operand hazard, not arithmetic.  Found in round 3: the inline-asm v_fma_mix pair that makes the low plane of P was
    followed within one wait state by the first MFMA reading it (hipcc does not pad hazards of an asm's outputs): dims
    0-31 of every head were off by 2^-11 of P's low plane as soon as a tile had more than 32 keys, dims 32-63 were exact."""
    import e2e_multi_view_matching_amd as E
    g = torch.Generator().manual_seed(5)
    for n_valid in (33, 64, 128, 200):
        n_rows = 128 if n_valid <= 128 else 256
        qkv = torch.randn(2, n_rows, 768, generator=g) * 1.5
        qkv[..., 512:] = 1.0
        if kernel in ("p2", "p2w"):
            out = E.attention_p2(qkv.to(gpu), 1, 2, n_valid, 4, 0, waves=1 if kernel == "p2w" else 0).cpu()
        else:
            out = E.attention_bf16x3(qkv.to(gpu), 1, 2, n_valid, 4, 0, kernel="f16x2").cpu()
        err = (out[:, :n_valid] - 1.0).abs().view(2, n_valid, 4, 64)
        assert float(err[..., :32].max()) < 2e-6 and float(err[..., 32:].max()) < 2e-6, (n_valid, float(err[..., :32].max()), float(err[..., 32:].max()))


# ---------------------------------------------------------------- range side-band: tile exponents
def _block_scales(g, rows, cols, choices):
    """one magnitude per 64 x 64 block"""
    idx = torch.randint(len(choices), (rows // 64, cols // 64), generator=g)
    sc = torch.tensor(choices, dtype=torch.float32)[idx]
    return sc.repeat_interleave(64, 0).repeat_interleave(64, 1)


@pytest.mark.parametrize("choices", [(1.0,), (1e-8, 1.0), (1.0, 3e6), (1e-9, 1e-3, 1.0, 1e5, 2e9), (7e4, 3e12)])
def test_gemm_p2_tile_exponents_carry_fp32_range(gpu, choices):
    """Activations far outside fp16's range (and mixed by 64 x 64 blocks inside one contraction): the plane GEMM with tile
    exponents keeps the fp32-class error bound relative to sum |a||w|; residual and plane output included."""
    import e2e_multi_view_matching_amd as E
    g = torch.Generator().manual_seed(len(choices))
    M, N, K1, K2 = 320, 256, 256, 256
    K = K1 + K2
    A = torch.randn(M, K, generator=g) * _block_scales(g, M, K, choices)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * max(choices)
    R = torch.randn(M, N, generator=g) * _block_scales(g, M, N, choices)
    A1, A2 = A[:, :K1].contiguous(), A[:, K1:].contiguous()
    for relu, planes_out in ((False, False), (True, True)):
        core = A.double() @ W.double().T + b.double()
        ref = (core.clamp_min(0) if relu else core) + R.double()
        scale = (A.double().abs() @ W.double().abs().T) + b.double().abs() + R.double().abs()
        out = E.gemm_p2(A1.to(gpu), W.to(gpu), bias=b.to(gpu), relu=relu, A2=A2.to(gpu), residual=R.to(gpu), planes_out=planes_out,
                        exponents=True).cpu()
        assert torch.isfinite(out).all()
        e = _err(out, ref, scale)
        # a plane output is 22 bits relative to the largest element of its 64 x 64 block: the bar is taken against the block maximum
        bar = 1.5e-6
        if planes_out:
            blk = ref.abs().view(M // 64, 64, N // 64, 64).amax((1, 3), keepdim=True).expand(M // 64, 64, N // 64, 64).reshape(M, N)
            e = float(((out.double() - ref).abs() / (scale + blk)).max())
        assert e < bar, (choices, relu, planes_out, e)


@pytest.mark.parametrize("waves", [0, 1])
@pytest.mark.parametrize("qs,ks,vs", [(1.0, 1.0, 1.0), (300.0, 1.0, 1e6), (1e-3, 2e3, 1e-7), (1e4, 1e-4, 3e9), (50.0, 50.0, 7e4)])
def test_attention_p2_tile_exponents_carry_fp32_range(gpu, qs, ks, vs, waves):
    """q, k, v magnitudes beyond the old limits of the mode (|q| < 5.6e3, |v| < 4e3): exponents in the logit scale and in
    the O accumulator; peaked and flat softmaxes."""
    import e2e_multi_view_matching_amd as E
    B, T, n_rows, n_valid, cross = 1, 2, 256, 200, 1
    g = torch.Generator().manual_seed(11)
    qkv = torch.randn(B * T, n_rows, 3 * 256, generator=g)
    qkv[..., :256] *= qs
    qkv[..., 256:512] *= ks
    qkv[..., 512:] *= vs
    qkv[0, 64:128, 512:] *= 1e-3  # one key block of one image three decades below the others: the O accumulator is rescaled
    ref = _attention_ref(qkv, B, T, n_valid, 4, cross)
    out = E.attention_p2(qkv.to(gpu), B, T, n_valid, 4, cross, waves=waves).cpu()
    out32 = E.attention(qkv.to(gpu), B, T, n_valid, 4, cross).cpu()
    assert torch.isfinite(out[:, :n_valid]).all()
    err = float((out[:, :n_valid].double() - ref[:, :n_valid]).abs().max()) / vs
    err32 = float((out32[:, :n_valid].double() - ref[:, :n_valid]).abs().max()) / vs
    assert err < 3 * err32 + 3e-6, (err, err32)


@pytest.mark.parametrize("B,T,n_rows,n_valid,cross", [(8, 5, 1024, 1024, 1),   # configs[3]'s cross layers: 640 items, the last 128 in two parts each
                                                       (8, 5, 1024, 1024, 0),   # ... and its self layers
                                                       (3, 2, 1024, 1000, 0),   # fewer items than CUs (96 -> 2 parts each), ragged last tile
                                                       (1, 3, 640, 577, 1),     # 36 items -> 7 parts of 19 tiles: uneven parts over two sources
                                                       (2, 2, 384, 300, 1)])    # 5 tiles in 5 parts of one tile each
def test_attention_p2w_key_split_equals_the_unsplit_kernel(gpu, B, T, n_rows, n_valid, cross):
    """Round 6: the items of a half-empty last round of workgroups are split along the keys (parts leave (m, l, O), a second launch
    merges them).  Same operands through the same kernel with the split off: equal up to the order of the softmax sums - and both
    against the fp64 reference at the one size where that is cheap."""
    import e2e_multi_view_matching_amd as E
    from e2e_multi_view_matching_amd import _lib
    ctx = _lib.context(gpu)
    g = torch.Generator().manual_seed(n_valid + T)
    qkv = (torch.randn(B * T, n_rows, 3 * 256, generator=g) * 1.5).to(gpu)
    try:
        ctx.set_attention_key_split(False)
        whole = E.attention_p2(qkv, B, T, n_valid, 4, cross, waves=1).cpu()
        ctx.set_attention_key_split(True)
        split = E.attention_p2(qkv, B, T, n_valid, 4, cross, waves=1).cpu()
    finally:
        ctx.set_attention_key_split(True)
    assert bool(torch.isfinite(split[:, :n_valid]).all())
    d = float((split[:, :n_valid] - whole[:, :n_valid]).abs().max())
    assert d < 8e-6, d  # (outputs of magnitude ~4: the order of fp32 sums; the kernel itself sits 2e-5 from fp64)
    if B * T <= 6:
        ref = _attention_ref(qkv.cpu(), B, T, n_valid, 4, cross)
        assert float((split[:, :n_valid].double() - ref[:, :n_valid]).abs().max()) < 2e-5
