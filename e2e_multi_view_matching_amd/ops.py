"""Stand-alone operators of the hot path (thin ctypes wrappers, torch = memory only)."""
import torch

from . import _lib


def _ctx(t):
    if not t.is_cuda:
        raise RuntimeError("e2e_multi_view_matching_amd operators need CUDA/HIP tensors on an MI355X (no CPU fallback)")
    return _lib.context(t.device)


def log_optimal_transport(scores, bin_score, iters):
    """Upstream ``log_optimal_transport``: scores [B,M,N] -> log assignment [B,M+1,N+1]."""
    ctx = _ctx(scores)
    s = scores.to(torch.float32).contiguous()
    B, M, N = s.shape
    Z = torch.empty((B, M + 1, N + 1), dtype=torch.float32, device=s.device)
    with torch.cuda.device(s.device):
        ctx.call("e2emv_sinkhorn", B, M, N, _lib.ptr(s), float(bin_score), int(iters), _lib.ptr(Z), _lib.stream_ptr(s.device))
    return Z


def extract_matches(logZ, match_threshold):
    """Mutual arg-max block of ``SuperGlue.forward`` on logZ [B,M+1,N+1]."""
    ctx = _ctx(logZ)
    z = logZ.to(torch.float32).contiguous()
    B, M, N = z.shape[0], z.shape[1] - 1, z.shape[2] - 1
    dev = z.device
    m0 = torch.empty((B, M), dtype=torch.int64, device=dev)
    m1 = torch.empty((B, N), dtype=torch.int64, device=dev)
    s0 = torch.empty((B, M), dtype=torch.float32, device=dev)
    s1 = torch.empty((B, N), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        ctx.call("e2emv_extract_matches", B, M, N, _lib.ptr(z), float(match_threshold), _lib.ptr(m0), _lib.ptr(m1),
                 _lib.ptr(s0), _lib.ptr(s1), _lib.stream_ptr(dev))
    return m0, m1, s0, s1


def gemm_nt(A, W, bias=None, residual=None, A2=None, scale=1.0, relu=False):
    """C = act(scale * [A | A2] W^T + bias) (+ residual); A [M,K1] or [Bt,M,K1], W [N,K] or [Bt,N,K]."""
    ctx = _ctx(A)
    batched = A.dim() == 3
    A_ = A.contiguous().float()
    W_ = W.contiguous().float()
    A2_ = A2.contiguous().float() if A2 is not None else None
    if not batched:
        A_, W_ = A_.unsqueeze(0), W_.unsqueeze(0)
        A2_ = A2_.unsqueeze(0) if A2_ is not None else None
    Bt, M, K1 = A_.shape
    N, K = W_.shape[1], W_.shape[2]
    C = torch.empty((Bt, M, N), dtype=torch.float32, device=A.device)
    R = residual.contiguous().float().reshape(Bt, M, N) if residual is not None else None
    b = bias.contiguous().float() if bias is not None else None
    with torch.cuda.device(A.device):
        ctx.call("e2emv_gemm_nt", Bt, M, N, K, K1, _lib.ptr(A_), K1, M * K1, _lib.ptr(A2_), (K - K1), M * (K - K1),
                 _lib.ptr(W_), K, (N * K if W.dim() == 3 else 0), _lib.ptr(b), _lib.ptr(R), N, M * N, _lib.ptr(C), N, M * N,
                 float(scale), 1 if relu else 0, _lib.stream_ptr(A.device))
    return C if batched else C[0]


def attention(qkv, B, T, n_valid, H, cross):
    """qkv [B*T, n_rows, 3D] head-major -> out [B*T, n_rows, D]."""
    ctx = _ctx(qkv)
    q = qkv.contiguous().float()
    n_img, n_rows, D3 = q.shape
    D = D3 // 3
    out = torch.zeros((n_img, n_rows, D), dtype=torch.float32, device=q.device)
    with torch.cuda.device(q.device):
        ctx.call("e2emv_attention", B, T, n_rows, n_valid, D, H, _lib.ptr(q), 1 if cross else 0, _lib.ptr(out),
                 _lib.stream_ptr(q.device))
    return out


def gemm_bf16x3(A, W, bias=None, relu=False, all_planes=False, f16x2=False):
    """bf16x3 split-operand GEMM building block on fp32 tensors: act(A W^T + bias).  Default: gemm_x3.hip (activations
    split on the way into LDS); ``all_planes=True``: the first-generation kernel that reads pre-split planes (gemm3.hip); ``f16x2=True``: the
    fp16 x 2 form of gemm_x3.hip (two activation planes, three products)."""
    ctx = _ctx(A)
    A_, W_ = A.contiguous().float(), W.contiguous().float()
    M, K = A_.shape
    N = W_.shape[0]
    C = torch.empty((M, N), dtype=torch.float32, device=A.device)
    b = bias.contiguous().float() if bias is not None else None
    with torch.cuda.device(A.device):
        ctx.call("e2emv_gemm_bf16x3", M, N, K, _lib.ptr(A_), _lib.ptr(W_), _lib.ptr(b), _lib.ptr(C), (1 if relu else 0) | (2 if all_planes else 0) | (4 if f16x2 else 0),
                 _lib.stream_ptr(A.device))
    return C


def attention_bf16x3(qkv, B, T, n_valid, H, cross, kernel="planes"):
    """Split-operand attention building block; same contract as `attention`.  kernel: "planes" (pre-split bf16x3 planes,
    attention3_kernel), "fused" (the forward pass's bf16x3 kernel: fp32 q|k|v in, planes made in the kernel) or "f16x2"
    (its fp16 x 2 form)."""
    ctx = _ctx(qkv)
    q = qkv.contiguous().float()
    n_img, n_rows, D3 = q.shape
    D = D3 // 3
    out = torch.empty((n_img, n_rows, D), dtype=torch.float32, device=q.device)
    with torch.cuda.device(q.device):
        ctx.call("e2emv_attention_bf16x3", B, T, n_rows, n_valid, D, H, _lib.ptr(q),
                 (1 if cross else 0) | {"planes": 0, "fused": 2, "f16x2": 6}[kernel], _lib.ptr(out), _lib.stream_ptr(q.device))
    return out


def gemm_p2(A, W, bias=None, relu=False, A2=None, residual=None, planes_out=False, reps=1, exponents=False):
    """gemm_p2.hip on fp32 tensors: act([A | A2] W^T + bias) (+ residual).  The operands are converted to the plane format
    (p2.h) by helper kernels, the kernel writes fp32 or (``planes_out``) planes that are converted back; ``exponents``: with
    the tile exponents of the range side-band (shapes in multiples of 64)."""
    ctx = _ctx(A)
    A_, W_ = A.contiguous().float(), W.contiguous().float()
    A2_ = A2.contiguous().float() if A2 is not None else None
    M, K1 = A_.shape
    N, K = W_.shape
    C = torch.empty((M, N), dtype=torch.float32, device=A.device)
    b = bias.contiguous().float() if bias is not None else None
    R = residual.contiguous().float() if residual is not None else None
    flags = (1 if relu else 0) | (2 if planes_out else 0) | (4 if exponents else 0) | (int(reps) << 8 if reps > 1 else 0)
    with torch.cuda.device(A.device):
        ctx.call("e2emv_gemm_p2", M, N, K, K1, _lib.ptr(A_), _lib.ptr(A2_), _lib.ptr(W_), _lib.ptr(b), _lib.ptr(R), _lib.ptr(C), flags,
                 _lib.stream_ptr(A.device))
    return C


def qkv_p2(X, W, bias, n_rows, H=4):
    """q|k|v projection through gemm_p2's attention-operand epilogue, read back as fp32 [n_img*n_rows, 3D]."""
    ctx = _ctx(X)
    X_, W_ = X.contiguous().float(), W.contiguous().float()
    M, D = X_.shape
    out = torch.empty((M, 3 * D), dtype=torch.float32, device=X.device)
    b = bias.contiguous().float() if bias is not None else None
    with torch.cuda.device(X.device):
        ctx.call("e2emv_qkv_p2", M // n_rows, n_rows, D, H, _lib.ptr(X_), _lib.ptr(W_), _lib.ptr(b), _lib.ptr(out), _lib.stream_ptr(X.device))
    return out


def attention_p2(qkv, B, T, n_valid, H, cross, waves=0, reps=1, abl=0):
    """attention_p2.hip on an fp32 q|k|v matrix (split into the plane operands by a helper kernel); same contract as
    `attention`.  waves: 0 = by key count, 4 / 8 = attention_p2 with that workgroup size, 1 = attention_p2w (one wave per SIMD)."""
    ctx = _ctx(qkv)
    q = qkv.contiguous().float()
    n_img, n_rows, D3 = q.shape
    D = D3 // 3
    out = torch.empty((n_img, n_rows, D), dtype=torch.float32, device=q.device)
    flags = (1 if cross else 0) | {0: 0, 4: 2, 8: 4, 1: 8}[waves] | (int(reps) << 8 if reps > 1 else 0) | ((int(abl) & 15) << 4)
    with torch.cuda.device(q.device):
        ctx.call("e2emv_attention_p2", B, T, n_rows, n_valid, D, H, _lib.ptr(q), flags, _lib.ptr(out), _lib.stream_ptr(q.device))
    return out
