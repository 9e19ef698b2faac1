// Test/bench entry points of the bf16x3 building blocks on plain fp32 buffers: the split into S3 planes,
// the V transpose and the merge back to fp32 are done here by small helper kernels so that the kernels of
// gemm3.hip / attention3.hip can be checked in isolation against an fp64 reference.
#include <vector>

#include "common.h"

namespace e2emv {

__device__ __forceinline__ void split3h(float v, __bf16& a, __bf16& b, __bf16& c) {
    a = (__bf16)v;
    const float r1 = v - (float)a;
    b = (__bf16)r1;
    const float r2 = r1 - (float)b;
    c = (__bf16)r2;
}

// qkv fp32 [rows][3D] -> qk S3 [rows][3][2D] (q scaled) + V^T [img][3][D][n_rows]
__global__ void split_qkv_kernel(const float* qkv, int n_rows, int D, float q_scale, uint16_t* qk, uint16_t* vt) {
    const int64_t row = blockIdx.x;
    const int img = (int)(row / n_rows), ml = (int)(row % n_rows);
    const float* src = qkv + row * 3 * D;
    __bf16* qo = reinterpret_cast<__bf16*>(qk + row * 3 * 2 * D);
    __bf16* vo = reinterpret_cast<__bf16*>(vt);
    for (int c = threadIdx.x; c < 3 * D; c += blockDim.x) {
        float v = src[c];
        if (c < D) v *= q_scale;
        __bf16 a, b, d;
        split3h(v, a, b, d);
        if (c < 2 * D) {
            qo[c] = a; qo[2 * D + c] = b; qo[4 * D + c] = d;
        } else {
            const int n = c - 2 * D;
            vo[((int64_t)(img * 3 + 0) * D + n) * n_rows + ml] = a;
            vo[((int64_t)(img * 3 + 1) * D + n) * n_rows + ml] = b;
            vo[((int64_t)(img * 3 + 2) * D + n) * n_rows + ml] = d;
        }
    }
}

__global__ void merge3_kernel(const uint16_t* src, int64_t rows, int C, float* dst) {
    const int64_t r = blockIdx.x;
    const __bf16* s = reinterpret_cast<const __bf16*>(src + r * 3 * C);
    for (int c = threadIdx.x; c < C; c += blockDim.x) dst[r * C + c] = ((float)s[2 * C + c] + (float)s[C + c]) + (float)s[c];
}

}  // namespace e2emv

using namespace e2emv;

extern "C" int e2emv_gemm_bf16x3(e2emv_ctx* ctx, int M, int Nout, int K, const float* d_A, const float* d_W,
                                 const float* d_bias, float* d_C, int flags, void* stream) {
    if (!ctx || !d_A || !d_W || !d_C) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    if (M <= 0 || Nout <= 0 || K <= 0 || K % 32 || Nout % 4) return set_err(ctx, E2EMV_ESHAPE, "gemm_bf16x3: M=%d N=%d K=%d", M, Nout, K);
    hipStream_t s = (hipStream_t)stream;
    auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
    const size_t szA = al((size_t)M * 3 * K * 2), szW = al((size_t)Nout * 3 * K * 2);
    int rc = ws_reserve(ctx, szA + szW);
    if (rc) return rc;
    uint16_t* A3 = (uint16_t*)ctx->d_ws;
    uint16_t* W3 = (uint16_t*)(ctx->d_ws + szA);
    if (flags & 4) {  // f16x2 GEMM (gemm_h2.hip); the weight planes are made on the host as e2emv_commit_weights makes them
        std::vector<float> hw((size_t)Nout * K);
        E2EMV_HIP(ctx, hipStreamSynchronize(s));
        E2EMV_HIP(ctx, hipMemcpy(hw.data(), d_W, hw.size() * sizeof(float), hipMemcpyDeviceToHost));
        std::vector<uint16_t> planes;
        float out_scale = 0.f;
        const size_t off = add_split_h2(planes, hw, Nout, K, &out_scale);
        E2EMV_HIP(ctx, hipMemcpy(W3, planes.data() + off, (size_t)Nout * 2 * K * 2, hipMemcpyHostToDevice));
        GemmArgs g;
        g.M = M; g.N = Nout; g.K = K; g.K1 = K; g.A = d_A; g.lda = K; g.bias = d_bias; g.C = d_C; g.ldc = Nout; g.relu = (flags & 1) != 0;
        prof_begin(ctx, PS_GEMM, s);
        rc = launch_gemm_x3(ctx, g, W3, K, s, out_scale);
        prof_end(ctx, s);
        return rc;
    }
    if ((rc = launch_split3(ctx, d_W, Nout, K, K, W3, K, s))) return rc;
    if (!(flags & 2)) {  // second-generation kernel: activations stay fp32, split on the way into LDS (gemm_x3.hip)
        GemmArgs g;
        g.M = M; g.N = Nout; g.K = K; g.K1 = K; g.A = d_A; g.lda = K; g.bias = d_bias; g.C = d_C; g.ldc = Nout; g.relu = (flags & 1) != 0;
        prof_begin(ctx, PS_GEMM, s);
        rc = launch_gemm_x3(ctx, g, W3, K, s);
        prof_end(ctx, s);
        return rc;
    }
#ifndef E2EMV_STAMPS
    return set_err(ctx, E2EMV_EINVAL, "gemm_bf16x3: the all-planes first-generation kernel (flags bit1, gemm3.hip) is part of the measurement build only");
#else
    if ((rc = launch_split3(ctx, d_A, M, K, K, A3, K, s))) return rc;
    Gemm3Args g;
    g.M = M; g.N = Nout; g.K = K; g.K1 = K; g.A = A3; g.lda = K; g.W = W3; g.ldw = K; g.bias = d_bias;
    g.C32 = d_C; g.ldc32 = Nout; g.relu = (flags & 1) != 0;
    prof_begin(ctx, PS_GEMM, s);
    rc = launch_gemm3(ctx, g, s);
    prof_end(ctx, s);
    return rc;
#endif
}

extern "C" int e2emv_attention_bf16x3(e2emv_ctx* ctx, int B, int T, int n_rows, int n_valid, int D, int H, const float* d_qkv,
                                      int cross, float* d_out, void* stream) {
    if (!ctx || !d_qkv || !d_out) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    if (B <= 0 || T <= 0 || n_rows <= 0) return set_err(ctx, E2EMV_ESHAPE, "attention_bf16x3: empty problem");
    hipStream_t s = (hipStream_t)stream;
    const int64_t rows = (int64_t)B * T * n_rows;
    if (T < 1 || T > E2EMV_MAX_TUPLE) return E2EMV_EINVAL;
    int nv[E2EMV_MAX_TUPLE];
    for (int t = 0; t < E2EMV_MAX_TUPLE; ++t) nv[t] = n_valid;
    int rc;
    if (cross & 6) {  // the kernels of the forward pass: fp32 q|k|v in, planes made in the kernel (bit2: fp16 x 2 form)
        E2EMV_HIP(ctx, hipMemsetAsync(d_out, 0, (size_t)rows * D * sizeof(float), s));
        prof_begin(ctx, PS_ATTN, s);
        rc = launch_attention3f(ctx, B, T, n_rows, nv, D, H, d_qkv, cross & 1, d_out, s, (cross & 4) != 0);
        prof_end(ctx, s);
        return rc;
    }
    auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
    const size_t sz_qk = al((size_t)rows * 3 * 2 * D * 2), sz_v = al((size_t)rows * 3 * D * 2);
    rc = ws_reserve(ctx, sz_qk + sz_v);
    if (rc) return rc;
    uint16_t* qk = (uint16_t*)ctx->d_ws;
    uint16_t* vt = (uint16_t*)(ctx->d_ws + sz_qk);
    hipLaunchKernelGGL(split_qkv_kernel, dim3((unsigned)rows), dim3(256), 0, s, d_qkv, n_rows, D, 0.125f * 1.4426950408889634f, qk, vt);
    E2EMV_HIP(ctx, hipMemsetAsync(d_out, 0, (size_t)rows * D * sizeof(float), s));
    prof_begin(ctx, PS_ATTN, s);
    rc = launch_attention3(ctx, B, T, n_rows, nv, D, H, qk, vt, cross & 1, nullptr, d_out, s);
    prof_end(ctx, s);
    E2EMV_CHECK_LAUNCH(ctx, "bf16x3 helper kernels");
    return rc;
}
