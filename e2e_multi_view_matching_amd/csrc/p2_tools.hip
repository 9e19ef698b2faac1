// Plane format helpers (p2.h): fp32 <-> planes conversion kernels, the host-side weight split, and the building-block
// entry points that let tests and micro-benchmarks drive gemm_p2.hip / attention_p2.hip on plain fp32 buffers.
#include <cmath>
#include <cstring>
#include <vector>

#include "p2.h"

namespace e2emv {

// fp32 [rows][C] -> scaled planes; one thread = 8 consecutive columns
__global__ __launch_bounds__(256) void to_planes_kernel(const float* src, int64_t rows, int C, int64_t ld_src, uint16_t* dst) {
    const int per_row = C / 8;
    const int64_t total = rows * per_row;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / per_row;
        const int n = (int)(i - m * per_row) * 8;
        const float* sp = src + m * ld_src + n;
        const p2_f32x4 v0 = *reinterpret_cast<const p2_f32x4*>(sp), v1 = *reinterpret_cast<const p2_f32x4*>(sp + 4);
        p2_u32x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const P2Pair a = p2_split_scaled(v0[2 * e], v0[2 * e + 1]), c = p2_split_scaled(v1[2 * e], v1[2 * e + 1]);
            hi[e] = a.hi; lo[e] = a.lo; hi[2 + e] = c.hi; lo[2 + e] = c.lo;
        }
        uint16_t* dp = dst + p2_index(m, n, C);
        *reinterpret_cast<p2_u32x4*>(dp) = hi;
        *reinterpret_cast<p2_u32x4*>(dp + 32) = lo;
    }
}

// fp32 [rows][C] -> scaled planes WITH tile exponents: one workgroup per block of 64 rows x 64 columns (thread = one row,
// 16 consecutive columns): block maximum -> exponent e (p2_pick_exponent) -> planes of x 2^-e
__global__ __launch_bounds__(256) void to_planes_exp_kernel(const float* src, int C, int64_t ld_src, uint16_t* dst, int* E, float* AM, unsigned* stats) {
    __shared__ float wmax[4];
    const int cb = blockIdx.x, rb = blockIdx.y;
    const int t = threadIdx.x;
    const int64_t m = (int64_t)rb * 64 + (t >> 2);
    const int n = cb * 64 + (t & 3) * 16;
    const float* sp = src + m * ld_src + n;
    p2_f32x4 v[4];
    float am = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] = *reinterpret_cast<const p2_f32x4*>(sp + 4 * i);
#pragma unroll
        for (int e = 0; e < 4; ++e) am = fmaxf(am, fabsf(v[i][e]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
    if ((t & 63) == 0) wmax[t >> 6] = am;
    __syncthreads();
    am = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    const int ex = p2_pick_exponent(am);
    const float f = p2_exp2i(-ex);
    if (t == 0) {
        E[(int64_t)rb * gridDim.x + cb] = ex;
        if (AM) AM[(int64_t)rb * gridDim.x + cb] = am;
        if (ex != 0 && stats) atomicAdd(stats, 1u);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        p2_u32x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const P2Pair a = p2_split_scaled(v[2 * h][2 * e] * f, v[2 * h][2 * e + 1] * f), c = p2_split_scaled(v[2 * h + 1][2 * e] * f, v[2 * h + 1][2 * e + 1] * f);
            hi[e] = a.hi; lo[e] = a.lo; hi[2 + e] = c.hi; lo[2 + e] = c.lo;
        }
        uint16_t* dp = dst + p2_index(m, n + 8 * h, C);
        *reinterpret_cast<p2_u32x4*>(dp) = hi;
        *reinterpret_cast<p2_u32x4*>(dp + 32) = lo;
    }
}

// The conf head's input [mdesc_i[n] | mdesc_j[match n]] ([B][n_rows][2 D]) made as planes with tile exponents in ONE pass: the gather
// of forward.hip's conf_gather2_kernel resolved in the source address of to_planes_exp_kernel's loads (round 6: one launch and a
// 2 x 67 MB fp32 round trip less per forward at configs[1]; the arithmetic, and so the planes, are to_planes_exp_kernel's bit for bit).
// Unmatched (-1) and padding rows take keypoint 0 of image j, as the gather kernel did (their confidence is 0 whatever the head says).
__global__ __launch_bounds__(256) void conf_gather_planes_kernel(const float* mdesc_i, const float* mdesc_j, int64_t tuple_stride, const int64_t* matches, int N,
                                                                 int n_rows, int D, uint16_t* dst, int* E, unsigned* stats) {
    __shared__ float wmax[4];
    const int cb = blockIdx.x, rb = blockIdx.y;  // 64-column block of the 2 D columns, 64-row block of the B * n_rows rows
    const int t = threadIdx.x;
    const int64_t m = (int64_t)rb * 64 + (t >> 2);
    const int b = (int)(m / n_rows), row = (int)(m - (int64_t)b * n_rows);
    const int n = cb * 64 + (t & 3) * 16;
    const float* sp;
    if (n < D) sp = mdesc_i + b * tuple_stride + (int64_t)row * D + n;
    else {
        int64_t j = row < N ? matches[(int64_t)b * N + row] : 0;
        if (j < 0) j = 0;
        sp = mdesc_j + b * tuple_stride + j * D + (n - D);
    }
    p2_f32x4 v[4];
    float am = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] = *reinterpret_cast<const p2_f32x4*>(sp + 4 * i);
#pragma unroll
        for (int e = 0; e < 4; ++e) am = fmaxf(am, fabsf(v[i][e]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
    if ((t & 63) == 0) wmax[t >> 6] = am;
    __syncthreads();
    am = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    const int ex = p2_pick_exponent(am);
    const float f = p2_exp2i(-ex);
    if (t == 0) {
        E[(int64_t)rb * gridDim.x + cb] = ex;
        if (ex != 0 && stats) atomicAdd(stats, 1u);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        p2_u32x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const P2Pair a = p2_split_scaled(v[2 * h][2 * e] * f, v[2 * h][2 * e + 1] * f), c = p2_split_scaled(v[2 * h + 1][2 * e] * f, v[2 * h + 1][2 * e + 1] * f);
            hi[e] = a.hi; lo[e] = a.lo; hi[2 + e] = c.hi; lo[2 + e] = c.lo;
        }
        uint16_t* dp = dst + p2_index(m, n + 8 * h, 2 * D);
        *reinterpret_cast<p2_u32x4*>(dp) = hi;
        *reinterpret_cast<p2_u32x4*>(dp + 32) = lo;
    }
}

// planes -> fp32; plain != 0: plain planes (hi + lo) divided by `unscale`
__global__ __launch_bounds__(256) void from_planes_kernel(const uint16_t* src, int64_t rows, int C, float* dst, int64_t ld_dst, int plain, float unscale,
                                                          const int* E) {
    const int per_row = C / 8;
    const int64_t total = rows * per_row;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / per_row;
        const int n = (int)(i - m * per_row) * 8;
        const uint16_t* sp = src + p2_index(m, n, C);
        const p2_u32x4 hi = *reinterpret_cast<const p2_u32x4*>(sp), lo = *reinterpret_cast<const p2_u32x4*>(sp + 32);
        float o[8];
        const float us = E ? unscale * p2_exp2i(E[(m >> 6) * (C / 64) + (n >> 6)]) : unscale;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const p2_f32x2 a = plain ? p2_join_plain(hi[e], lo[e]) : p2_join_scaled(hi[e], lo[e]);
            o[2 * e] = a[0] * us; o[2 * e + 1] = a[1] * us;
        }
        float* dp = dst + m * ld_dst + n;
        *reinterpret_cast<p2_f32x4*>(dp) = p2_f32x4{o[0], o[1], o[2], o[3]};
        *reinterpret_cast<p2_f32x4*>(dp + 4) = p2_f32x4{o[4], o[5], o[6], o[7]};
    }
}

// the attention operands back to one fp32 q|k|v matrix [rows][3D] (pre-scales undone): test helper, one thread per element pair
__global__ __launch_bounds__(256) void qkv_from_planes_kernel(const uint16_t* qk, const uint16_t* vt, int64_t rows, int n_rows, int D, int H,
                                                              float q_unscale, float v_unscale, float* dst, const int* EQK, const int* EVt) {
    const int64_t total = rows * (3 * D / 2);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / (3 * D / 2);
        const int n = (int)(i - m * (3 * D / 2)) * 2;
        float a, c;
        if (n < 2 * D) {
            const uint16_t* sp = qk + p2_index(m, n, 2 * D);
            const p2_f32x2 v = p2_join_plain(*reinterpret_cast<const unsigned*>(sp), *reinterpret_cast<const unsigned*>(sp + 32));
            const float u = (n < D ? q_unscale : 1.f) * (EQK ? p2_exp2i(EQK[(m >> 6) * 8 + (n >> 6)]) : 1.f);
            a = v[0] * u; c = v[1] * u;
        } else {
            const int64_t img = m / n_rows;
            const int key = (int)(m - img * n_rows);
            const int pos = (key & ~15) | p2_vt_pos(key & 15);
            float o[2];
            for (int e = 0; e < 2; ++e) {
                const int nv = n + e - 2 * D, head = nv >> 6, dd = nv & 63;
                const _Float16* sp = reinterpret_cast<const _Float16*>(vt) + ((img * H + head) * 64 + dd) * (2 * (int64_t)n_rows) + (pos >> 5) * 64 + (pos & 31);
                o[e] = ((float)sp[0] + (float)sp[32]) * v_unscale * (EVt ? p2_exp2i(EVt[(m >> 6) * 4 + head]) : 1.f);
            }
            a = o[0]; c = o[1];
        }
        dst[m * 3 * D + n] = a;
        dst[m * 3 * D + n + 1] = c;
    }
}

// fp32 q|k|v [rows][3D] -> the attention operands (what gemm_p2's P2_OUT_QKV epilogue writes): test helper
// tile exponents of the attention operands of an fp32 q|k|v matrix: one workgroup per 64 rows x 64 columns
__global__ __launch_bounds__(256) void qkv_exponents_kernel(const float* src, int D, float q_scale, float v_scale, int* EQK, int* EVt) {
    __shared__ float wmax[4];
    const int cb = blockIdx.x, rb = blockIdx.y, t = threadIdx.x;
    const float* sp = src + ((int64_t)rb * 64 + (t >> 2)) * 3 * D + cb * 64 + (t & 3) * 16;
    float am = 0.f;
    for (int i = 0; i < 16; ++i) am = fmaxf(am, fabsf(sp[i]));
    for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
    if ((t & 63) == 0) wmax[t >> 6] = am;
    __syncthreads();
    am = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    if (t) return;
    const int nb = D / 64;
    if (cb < nb) EQK[rb * 2 * nb + cb] = p2_pick_exponent(am * q_scale);
    else if (cb < 2 * nb) EQK[rb * 2 * nb + cb] = p2_pick_exponent(am);
    else EVt[rb * nb + cb - 2 * nb] = p2_pick_exponent(am * v_scale);
}

__global__ __launch_bounds__(256) void qkv_to_planes_kernel(const float* src, int64_t rows, int n_rows, int D, int H, float q_scale, float v_scale,
                                                            uint16_t* qk, uint16_t* vt, const int* EQK, const int* EVt) {
    const int64_t total = rows * (3 * D / 2);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / (3 * D / 2);
        const int n = (int)(i - m * (3 * D / 2)) * 2;
        const float x0 = src[m * 3 * D + n], x1 = src[m * 3 * D + n + 1];
        if (n < 2 * D) {
            const float u = (n < D ? q_scale : 1.f) * (EQK ? p2_exp2i(-EQK[(m >> 6) * 8 + (n >> 6)]) : 1.f);
            const P2Pair pr = p2_split_plain(x0 * u, x1 * u);
            uint16_t* dp = qk + p2_index(m, n, 2 * D);
            *reinterpret_cast<unsigned*>(dp) = pr.hi;
            *reinterpret_cast<unsigned*>(dp + 32) = pr.lo;
        } else {
            const int64_t img = m / n_rows;
            const int key = (int)(m - img * n_rows);
            const int pos = (key & ~15) | p2_vt_pos(key & 15);
            const float u = v_scale * (EVt ? p2_exp2i(-EVt[(m >> 6) * 4 + ((n - 2 * D) >> 6)]) : 1.f);
            const P2Pair pr = p2_split_plain(x0 * u, x1 * u);
            for (int e = 0; e < 2; ++e) {
                const int nv = n + e - 2 * D, head = nv >> 6, dd = nv & 63;
                uint16_t* dp = vt + ((img * H + head) * 64 + dd) * (2 * (int64_t)n_rows) + (pos >> 5) * 64 + (pos & 31);
                dp[0] = (uint16_t)(e ? pr.hi >> 16 : pr.hi & 0xffffu);
                dp[32] = (uint16_t)(e ? pr.lo >> 16 : pr.lo & 0xffffu);
            }
        }
    }
}

static int grid_for(int64_t items) { return (int)std::min<int64_t>((items + 255) / 256, 256 * 8); }

int launch_to_planes(e2emv_ctx* ctx, const float* src, int64_t rows, int C, int64_t ld_src, uint16_t* dst, hipStream_t s, int* E, float* AM) {
    if (rows <= 0 || C <= 0 || C % 32 || ld_src % 4 || (uintptr_t)src % 16 || (uintptr_t)dst % 16)
        return set_err(ctx, E2EMV_ESHAPE, "to_planes: C=%d must be a multiple of 32, rows 16-byte aligned", C);
    if (E) {
        if (int rc = ensure_flags(ctx)) return rc;
        if (rows % 64 || C % 64 || rows / 64 > 65535) return set_err(ctx, E2EMV_ESHAPE, "to_planes: tile exponents need 64 x 64 blocks (rows=%lld C=%d)", (long long)rows, C);
        hipLaunchKernelGGL(to_planes_exp_kernel, dim3(C / 64, (unsigned)(rows / 64)), dim3(256), 0, s, src, C, ld_src, dst, E, AM, ctx->d_flags + 2);
        E2EMV_CHECK_LAUNCH(ctx, "to_planes_exp_kernel");
        return E2EMV_OK;
    }
    hipLaunchKernelGGL(to_planes_kernel, dim3(grid_for(rows * (C / 8))), dim3(256), 0, s, src, rows, C, ld_src, dst);
    E2EMV_CHECK_LAUNCH(ctx, "to_planes_kernel");
    return E2EMV_OK;
}

int launch_conf_gather_planes(e2emv_ctx* ctx, const float* mdesc_i, const float* mdesc_j, int64_t tuple_stride, const int64_t* matches, int N, int n_rows,
                              int B, int D, uint16_t* dst, int* E, hipStream_t s) {
    const int64_t rows = (int64_t)B * n_rows;
    if (rows % 64 || D % 64 || rows / 64 > 65535 || tuple_stride % 4 || (uintptr_t)mdesc_i % 16 || (uintptr_t)mdesc_j % 16 || !E)
        return set_err(ctx, E2EMV_ESHAPE, "conf_gather_planes: needs 64 x 64 blocks (rows=%lld D=%d)", (long long)rows, D);
    if (int rc = ensure_flags(ctx)) return rc;
    hipLaunchKernelGGL(conf_gather_planes_kernel, dim3(2 * D / 64, (unsigned)(rows / 64)), dim3(256), 0, s, mdesc_i, mdesc_j, tuple_stride, matches, N, n_rows, D, dst,
                       E, ctx->d_flags + 2);
    E2EMV_CHECK_LAUNCH(ctx, "conf_gather_planes_kernel");
    return E2EMV_OK;
}

int launch_from_planes(e2emv_ctx* ctx, const uint16_t* src, int64_t rows, int C, float* dst, int64_t ld_dst, hipStream_t s, const int* E) {
    if (rows <= 0 || C <= 0 || C % 32 || ld_dst % 4 || (uintptr_t)src % 16 || (uintptr_t)dst % 16)
        return set_err(ctx, E2EMV_ESHAPE, "from_planes: C=%d must be a multiple of 32, rows 16-byte aligned", C);
    hipLaunchKernelGGL(from_planes_kernel, dim3(grid_for(rows * (C / 8))), dim3(256), 0, s, src, rows, C, dst, ld_dst, 0, 1.f, E);
    E2EMV_CHECK_LAUNCH(ctx, "from_planes_kernel");
    return E2EMV_OK;
}

namespace {
inline uint16_t f2h(float f) {
    const _Float16 h = (_Float16)f;  // round to nearest even
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
inline float h2f(uint16_t u) {
    _Float16 h;
    memcpy(&h, &u, 2);
    return (float)h;
}
}  // namespace

// weights [rows][cols] fp32 -> P2 planes [rows][cols / 32 blocks of {32 hi, 32 lo}] of 2^s W, appended to `out`.  Same
// numbers as add_split_h2 (ctx.hip): s brings max |w| into [2^13, 2^14), lo is the UNSCALED residual fp16(v - hi).
size_t add_split_p2(std::vector<uint16_t>& out, const std::vector<float>& w, int rows, int cols, float* out_scale) {
    float mx = 0.f;
    for (float v : w) mx = std::max(mx, std::fabs(v));
    int e = 0;
    if (mx > 0.f && std::isfinite(mx)) (void)std::frexp(mx, &e);
    const int sh = 14 - e;
    const float sc = std::ldexp(1.f, sh);
    *out_scale = std::ldexp(1.f, -sh);
    const size_t off = (out.size() + 127) & ~size_t(127);
    out.resize(off + (size_t)rows * 2 * cols);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            const float v = w[(size_t)r * cols + c] * sc;
            const uint16_t hi = f2h(v);
            uint16_t* o = &out[off + (size_t)p2_index(r, c, cols)];
            o[0] = hi;
            o[32] = f2h(v - h2f(hi));
        }
    return off;
}

}  // namespace e2emv

using namespace e2emv;

static size_t al256(size_t b) { return (b + 255) & ~size_t(255); }

// weights fp32 [N][K] on the device -> P2 planes at d_dst (host-synchronising: the split is the host code of the commit path)
static int weights_to_planes(e2emv_ctx* ctx, const float* d_W, int N, int K, uint16_t* d_dst, float* out_scale, hipStream_t s) {
    std::vector<float> hw((size_t)N * K);
    E2EMV_HIP(ctx, hipStreamSynchronize(s));
    E2EMV_HIP(ctx, hipMemcpy(hw.data(), d_W, hw.size() * sizeof(float), hipMemcpyDeviceToHost));
    std::vector<uint16_t> planes;
    const size_t off = add_split_p2(planes, hw, N, K, out_scale);
    E2EMV_HIP(ctx, hipMemcpy(d_dst, planes.data() + off, (size_t)N * 2 * K * 2, hipMemcpyHostToDevice));
    return E2EMV_OK;
}

extern "C" int e2emv_gemm_p2(e2emv_ctx* ctx, int M, int Nout, int K, int K1, const float* d_A, const float* d_A2, const float* d_W,
                             const float* d_bias, const float* d_R, float* d_C, int flags, void* stream) {
    if (!ctx || !d_A || !d_W || !d_C) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    const bool planes_out = (flags & 2) != 0, use_e = (flags & 4) != 0;
    if (use_e && (M % 64 || K1 % 64 || (K - K1) % 64 || ((planes_out || d_R) && Nout % 64)))
        return set_err(ctx, E2EMV_ESHAPE, "gemm_p2 with tile exponents: M, K1, K - K1 (and N for plane output / residual) must be multiples of 64");
    if (M <= 0 || Nout <= 0 || K <= 0 || K % 32 || K1 % 32 || K1 <= 0 || K1 > K || (K1 < K && !d_A2) || Nout % 4 || (planes_out && Nout % 32) || (d_R && Nout % 32))
        return set_err(ctx, E2EMV_ESHAPE, "gemm_p2: M=%d N=%d K=%d K1=%d (plane output / residual need N %% 32 == 0)", M, Nout, K, K1);
    hipStream_t s = (hipStream_t)stream;
    const int K2 = K - K1;
    const size_t szA = al256((size_t)M * K1 * 4), szA2 = al256((size_t)M * K2 * 4), szW = al256((size_t)Nout * K * 4),
                 szR = d_R ? al256((size_t)M * Nout * 4) : 0, szC = planes_out ? al256((size_t)M * Nout * 4) : 0;
    const size_t szE = use_e ? al256((size_t)(M / 64) * ((K + 3 * Nout) / 64 + 4) * sizeof(int)) : 0;
    int rc = ws_reserve(ctx, szA + szA2 + szW + szR + szC + szE);
    if (rc) return rc;
    char* w = ctx->d_ws;
    uint16_t* Ap = (uint16_t*)w; w += szA;
    uint16_t* A2p = (uint16_t*)w; w += szA2;
    uint16_t* Wp = (uint16_t*)w; w += szW;
    uint16_t* Rp = (uint16_t*)w; w += szR;
    uint16_t* Cp = (uint16_t*)w; w += szC;
    int* EA = use_e ? (int*)w : nullptr;
    int* EA2 = use_e && K2 ? EA + (M / 64) * (K1 / 64) : nullptr;
    int* ER = use_e && d_R ? EA + (M / 64) * (K / 64) : nullptr;
    int* EC = use_e && planes_out ? EA + (M / 64) * ((K + Nout) / 64) : nullptr;
    float* AR = use_e && d_R ? (float*)(EA + (M / 64) * ((K + 2 * Nout) / 64)) : nullptr;
    float out_scale = 1.f;
    if ((rc = weights_to_planes(ctx, d_W, Nout, K, Wp, &out_scale, s))) return rc;
    if ((rc = launch_to_planes(ctx, d_A, M, K1, K1, Ap, s, EA))) return rc;
    if (K2 && (rc = launch_to_planes(ctx, d_A2, M, K2, K2, A2p, s, EA2))) return rc;
    if (d_R && (rc = launch_to_planes(ctx, d_R, M, Nout, Nout, Rp, s, ER, AR))) return rc;
    GemmP2Args g;
    g.M = M; g.N = Nout; g.K = K; g.K1 = K1;
    g.A = Ap; g.lda = K1;
    if (K2) { g.A2 = A2p; g.lda2 = K2; }
    g.W = Wp; g.out_scale = out_scale; g.bias = d_bias;
    if (d_R) { g.Rp = Rp; g.ldr = Nout; }
    g.relu = (flags & 1) != 0;
    g.EA = EA; g.EA2 = EA2; g.ER = ER; g.EC = EC; g.AR = AR;
    if (d_bias) {  // the bound the commit path computes on the host
        std::vector<float> hb(Nout);
        E2EMV_HIP(ctx, hipStreamSynchronize(s));
        E2EMV_HIP(ctx, hipMemcpy(hb.data(), d_bias, hb.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (float v : hb) g.bias_amax = std::max(g.bias_amax, std::fabs(v));
    }
    if (planes_out) { g.out = P2_OUT_PLANES; g.Cp = Cp; g.ldc = Nout; }
    else { g.out = P2_OUT_F32; g.C32 = d_C; g.ldc = Nout; }
    const int reps = (flags >> 8) > 0 ? (flags >> 8) : 1;  // bits 8+: repeat the launch (micro-benchmarks time the family slot)
    if (use_e && d_R && planes_out && reps > 1) return set_err(ctx, E2EMV_EINVAL, "gemm_p2: repetitions with residual exponents");
    for (int i = 0; i < reps; ++i) {
        prof_begin(ctx, PS_GEMM, s);
        rc = launch_gemm_p2(ctx, g, s);
        prof_end(ctx, s);
        if (rc) return rc;
    }
    if (planes_out) rc = launch_from_planes(ctx, Cp, M, Nout, d_C, Nout, s, EC);
    return rc;
}

extern "C" int e2emv_qkv_p2(e2emv_ctx* ctx, int n_img, int n_rows, int D, int H, const float* d_X, const float* d_W, const float* d_bias,
                            float* d_qkv, void* stream) {
    if (!ctx || !d_X || !d_W || !d_qkv) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    if (n_img <= 0 || n_rows <= 0 || n_rows % 128 || D != 256 || H != 4) return set_err(ctx, E2EMV_ESHAPE, "qkv_p2: n_rows %% 128 == 0, D = 256, H = 4");
    hipStream_t s = (hipStream_t)stream;
    const int64_t M = (int64_t)n_img * n_rows;
    const size_t szX = al256((size_t)M * D * 4), szW = al256((size_t)3 * D * D * 4), szQK = al256((size_t)M * 2 * D * 4), szV = al256((size_t)M * D * 4);
    const size_t szE = al256((size_t)(M / 64) * 16 * sizeof(int));
    int rc = ws_reserve(ctx, szX + szW + szQK + szV + szE);
    if (rc) return rc;
    char* w = ctx->d_ws;
    uint16_t* Xp = (uint16_t*)w; w += szX;
    uint16_t* Wp = (uint16_t*)w; w += szW;
    uint16_t* QK = (uint16_t*)w; w += szQK;
    uint16_t* VT = (uint16_t*)w; w += szV;
    int* EX = (int*)w;
    int* EQK = EX + (M / 64) * 4;
    int* EVt = EQK + (M / 64) * 8;
    float out_scale = 1.f;
    if ((rc = weights_to_planes(ctx, d_W, 3 * D, D, Wp, &out_scale, s))) return rc;
    if ((rc = launch_to_planes(ctx, d_X, M, D, D, Xp, s, EX))) return rc;
    GemmP2Args g;
    g.M = (int)M; g.N = 3 * D; g.K = D; g.K1 = D; g.A = Xp; g.lda = D; g.W = Wp; g.out_scale = out_scale; g.bias = d_bias;
    g.out = P2_OUT_QKV; g.Cp = QK; g.Vt = VT; g.n_rows = n_rows; g.heads = H;
    g.EA = EX; g.EC = EQK; g.EVt = EVt;
    if (d_bias) {
        std::vector<float> hb(3 * D);
        E2EMV_HIP(ctx, hipStreamSynchronize(s));
        E2EMV_HIP(ctx, hipMemcpy(hb.data(), d_bias, hb.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (float v : hb) g.bias_amax = std::max(g.bias_amax, std::fabs(v));
    }
    prof_begin(ctx, PS_GEMM, s);
    rc = launch_gemm_p2(ctx, g, s);
    prof_end(ctx, s);
    if (rc) return rc;
    hipLaunchKernelGGL(qkv_from_planes_kernel, dim3(grid_for(M * (3 * D / 2))), dim3(256), 0, s, QK, VT, M, n_rows, D, H,
                       1.f / (0.125f * 1.4426950408889634f * P2_QS), 1.f / P2_VS, d_qkv, EQK, EVt);
    E2EMV_CHECK_LAUNCH(ctx, "qkv_from_planes_kernel");
    return E2EMV_OK;
}

extern "C" int e2emv_attention_p2(e2emv_ctx* ctx, int B, int T, int n_rows, int n_valid, int D, int H, const float* d_qkv, int flags,
                                  float* d_out, void* stream) {
    if (!ctx || !d_qkv || !d_out) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    if (B <= 0 || T <= 0 || T > E2EMV_MAX_TUPLE || n_rows <= 0 || n_rows % 128 || D != 256 || H != 4)
        return set_err(ctx, E2EMV_ESHAPE, "attention_p2: n_rows %% 128 == 0, D = 256, H = 4");
    hipStream_t s = (hipStream_t)stream;
    const int64_t M = (int64_t)B * T * n_rows;
    const size_t szQK = al256((size_t)M * 2 * D * 4), szV = al256((size_t)M * D * 4), szO = al256((size_t)M * D * 4);
    const size_t szE = al256((size_t)(M / 64) * 16 * sizeof(int));
    int rc = ws_reserve(ctx, szQK + szV + szO + szE);
    if (rc) return rc;
    char* w = ctx->d_ws;
    uint16_t* QK = (uint16_t*)w; w += szQK;
    uint16_t* VT = (uint16_t*)w; w += szV;
    uint16_t* OP = (uint16_t*)w; w += szO;
    int* EQK = (int*)w;
    int* EVt = EQK + (M / 64) * 8;
    int* EO = EVt + (M / 64) * 4;
    const float qs = 0.125f * 1.4426950408889634f * P2_QS;
    hipLaunchKernelGGL(qkv_exponents_kernel, dim3(3 * D / 64, (unsigned)(M / 64)), dim3(256), 0, s, d_qkv, D, qs, P2_VS, EQK, EVt);
    hipLaunchKernelGGL(qkv_to_planes_kernel, dim3(grid_for(M * (3 * D / 2))), dim3(256), 0, s, d_qkv, M, n_rows, D, H, qs, P2_VS, QK, VT, EQK, EVt);
    E2EMV_HIP(ctx, hipMemsetAsync(OP, 0, (size_t)M * D * 4, s));
    E2EMV_HIP(ctx, hipMemsetAsync(EO, 0, (size_t)(M / 64) * 4 * sizeof(int), s));
    int nv[E2EMV_MAX_TUPLE];
    for (int t = 0; t < E2EMV_MAX_TUPLE; ++t) nv[t] = n_valid;
    const int save_nw = ctx->attn_p2_nw;
    if (flags & 2) ctx->attn_p2_nw = 4;
    if (flags & 4) ctx->attn_p2_nw = 8;
    if (flags & 8) ctx->attn_p2_nw = 1;  // attention_p2w.hip
    ctx->attn_abl = (flags >> 4) & 15;   // (bits 4..7: its ablations, measurement build only; 15 stands for 29)
    if (ctx->attn_abl == 15) ctx->attn_abl = 29;
    const int reps = (flags >> 8) > 0 ? (flags >> 8) : 1;
    for (int i = 0; i < reps && !rc; ++i) {
        prof_begin(ctx, PS_ATTN, s);
        rc = launch_attention_p2(ctx, B, T, n_rows, nv, D, H, QK, VT, flags & 1, OP, s, EQK, EVt, EO);
        prof_end(ctx, s);
    }
    ctx->attn_p2_nw = save_nw;
    if (rc) return rc;
    return launch_from_planes(ctx, OP, M, D, d_out, D, s, EO);
}
