// fp32-accurate "NT" GEMM on the gfx950 bf16 matrix pipe (16x the fp32-MFMA rate), by operand splitting:
//
//   x = x1 + x2 + x3,  x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)   (|x - x1 - x2 - x3| <= 2^-26 |x|)
//   a*b ~= a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1            (dropped terms <= 2^-25 |ab|)
//
// Every bf16 x bf16 product is exact in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in fp32, so the
// result carries fp32-class rounding (measured against an fp64 reference in tests/test_gpu_kernels.py)
// at 6 MFMAs of 32 cycles per 32x32x16 block instead of 8 fp32 MFMAs of 64 cycles: 2.67x the
// fp32-MFMA ceiling (2.5 PF / 6 = 417 TFLOP/s fp32-equivalent).  bf16 keeps fp32's exponent range, so
// unlike an fp16 split there is no overflow/underflow hazard.
//
// Operands live in HBM already split ("S3" layout: row r = [plane0 | plane1 | plane2], each ld bf16
// wide), written by the producing kernel's epilogue, so no conversion sits in the inner loop.
// Used for the four big GEMMs of a GNN layer (q|k|v, MLP0 over [x | attention], MLP1); the small ones
// (keypoint encoder, final_proj, conf head, score matrix) stay on the fp32 kernel (gemm.hip).
//
// Tile: 128x128x32, 8 waves x (2x1) 32x32 MFMA tiles, LDS rows padded to 80 B (conflict-free
// ds_read_b128: 20*i mod 64 distinct for 16 rows), register prefetch, two barriers per K tile,
// persistent with cross-tile pipelining and XCD-aware tile ranges like gemm.hip.  Accumulators are
// transposed (lane = output row) for 16-byte epilogues; the V third of q|k|v swaps the MFMA operand
// roles instead, so its epilogue can store V^T (keys contiguous) - the layout the attention kernel's
// P.V MFMA needs as an operand.
#include <algorithm>

#include "common.h"

namespace e2emv {
__device__ __forceinline__ void split3(float v, __bf16& a, __bf16& b, __bf16& c) {
    a = (__bf16)v;
    const float r1 = v - (float)a;
    b = (__bf16)r1;
    const float r2 = r1 - (float)b;
    c = (__bf16)r2;
}

#ifdef E2EMV_STAMPS  // round 6: the all-planes kernel is an A/B arm of the measurement build; the product keeps the plane splitter below

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int G3_BM = 128, G3_BN = 128, G3_BK = 32, G3_LD = 40;  // LDS row = 40 bf16 = 80 B
constexpr int G3_PLANE = 128 * G3_LD;                            // bf16 elements per LDS plane tile

struct Gemm3Params {
    const uint16_t* A;   // S3 [M][3][lda], first K1 columns
    const uint16_t* A2;  // S3 [M][3][lda2], remaining K - K1 columns (or null)
    const uint16_t* W;   // S3 [N][3][ldw]
    const float* bias;   // [N] or null
    const float* R;      // fp32 residual [M][ldr] or null
    float* C32;          // fp32 output [M][ldc32] or null
    uint16_t* C3;        // S3 output [M][3][ldc3] or null (columns n < c3_cols only)
    uint16_t* Vt;        // transposed split output for columns n >= vt_n0: [img][3][N - vt_n0][n_rows]
    int64_t lda, lda2, ldw, ldr, ldc32, ldc3;
    int M, N, K, K1;
    int tiles_m, tiles_n, total;
    int relu;
    int q_cols;          // columns n < q_cols are multiplied by q_scale (attention query pre-scale)
    float q_scale;
    int vt_n0;           // first column that goes to Vt (multiple of 128), or N
    int n_rows;          // rows per image (Vt addressing)
};


// 512 threads = 8 waves as 2 (rows) x 4 (columns): each wave owns a 64 x 32 output block (two 32x32
// MFMA tiles sharing one weight fragment).  Halving the per-wave tile halves accumulators AND staging
// registers (~110 VGPRs), so two workgroups = 16 waves = 4 per SIMD share a CU.
__global__ __launch_bounds__(512, 4) void gemm3_kernel(Gemm3Params p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t smem3[];
    uint16_t* As = smem3;                // [3][128][G3_LD]
    uint16_t* Ws = smem3 + 3 * G3_PLANE;

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int per_xcd = (p.total + 7) / 8;
    const int t_begin = xcd * per_xcd;
    const int t_end = min(t_begin + per_xcd, p.total);
    int tile = t_begin + slot;
    if (tile >= t_end) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int l31 = lane & 31, lh = lane >> 5;
    const int nk = p.K / G3_BK;
    // staging: thread -> (row = tid / 4, 16-byte chunk tid & 3) of each of the 3 planes
    const int st_row = tid >> 2, st_ch = (tid & 3) * 8;

    const uint16_t* a_row;
    const uint16_t* a2_row;
    const uint16_t* w_row;
    auto setup = [&](int t) {
        const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
        const int ra = min(tm * G3_BM + st_row, p.M - 1);
        const int rw = min(tn * G3_BN + st_row, p.N - 1);
        a_row = p.A + (int64_t)ra * 3 * p.lda + st_ch;
        a2_row = p.A2 ? p.A2 + (int64_t)ra * 3 * p.lda2 + st_ch : nullptr;
        w_row = p.W + (int64_t)rw * 3 * p.ldw + st_ch;
    };
    u32x4 ra[3], rb[3];
    auto gload = [&](int kt) {
        const int k = kt * G3_BK;
        const bool first = k < p.K1;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const uint16_t* src = first ? a_row + pl * p.lda + k : a2_row + pl * p.lda2 + (k - p.K1);
            ra[pl] = *reinterpret_cast<const u32x4*>(src);
            rb[pl] = *reinterpret_cast<const u32x4*>(w_row + pl * p.ldw + k);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const int off = pl * G3_PLANE + st_row * G3_LD + st_ch;
            *reinterpret_cast<u32x4*>(As + off) = ra[pl];
            *reinterpret_cast<u32x4*>(Ws + off) = rb[pl];
        }
    };

    f32x16 acc[2];  // activation tiles i = 0, 1 (rows wr*64 + 32 i ..) x this wave's 32 output channels
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    };
    // SWAP = false: weights are the MFMA A operand (rows -> registers), activations B (rows -> lanes)
    // SWAP = true : roles exchanged (lane = output channel, registers = runs of 4 rows) for V^T tiles
    auto compute = [&](auto swap_tag) {
        constexpr bool SWAP = decltype(swap_tag)::value;
        const uint16_t* as = As + (wr * 64 + l31) * G3_LD + lh * 8;
        const uint16_t* ws = Ws + (wc * 32 + l31) * G3_LD + lh * 8;
#pragma unroll 1
        for (int s = 0; s < 2; ++s) {
            bf16x8 x[3][2], w[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                w[pl] = *reinterpret_cast<const bf16x8*>(ws + pl * G3_PLANE + s * 16);
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    x[pl][t] = *reinterpret_cast<const bf16x8*>(as + pl * G3_PLANE + t * 32 * G3_LD + s * 16);
            }
            // smallest terms first: (w3,x1) (w2,x2) (w1,x3) (w2,x1) (w1,x2) (w1,x1)
            constexpr int PW[6] = {2, 1, 0, 1, 0, 0};
            constexpr int PX[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (SWAP)
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[PX[q]][i], w[PW[q]], acc[i], 0, 0, 0);
                    else
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[PW[q]], x[PX[q]][i], acc[i], 0, 0, 0);
                }
        }
    };
    auto epilogue = [&](int t) {
        const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = tm * G3_BM + wr * 64 + i * 32 + l31;
            if (m >= p.M) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = tn * G3_BN + wc * 32 + 8 * g + 4 * lh;
                if (n >= p.N) continue;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][4 * g + e];
                if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = relu_nan(v[e]);
                }
                if (p.R) v += *reinterpret_cast<const f32x4*>(p.R + (int64_t)m * p.ldr + n);
                if (n < p.q_cols) v *= p.q_scale;
                if (p.C32) *reinterpret_cast<f32x4*>(p.C32 + (int64_t)m * p.ldc32 + n) = v;
                if (p.C3) {
                    bf16x4 h0, h1, h2;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        __bf16 a, b, c;
                        split3(v[e], a, b, c);
                        h0[e] = a; h1[e] = b; h2[e] = c;
                    }
                    uint16_t* dst = p.C3 + (int64_t)m * 3 * p.ldc3 + n;
                    *reinterpret_cast<bf16x4*>(dst) = h0;
                    *reinterpret_cast<bf16x4*>(dst + p.ldc3) = h1;
                    *reinterpret_cast<bf16x4*>(dst + 2 * p.ldc3) = h2;
                }
            }
        }
    };
    // V^T tiles: lane = channel n, registers = 4 consecutive rows m -> 8-byte stores along the keys
    auto epilogue_vt = [&](int t) {
        const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
        const int nv = p.N - p.vt_n0;
        const int n = tn * G3_BN + wc * 32 + l31;
        if (n >= p.N) return;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int m = tm * G3_BM + wr * 64 + i * 32 + 8 * g + 4 * lh;
                if (m >= p.M) continue;
                const int img = m / p.n_rows, ml = m - img * p.n_rows;
                bf16x4 h0, h1, h2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    __bf16 a, b, c;
                    split3(acc[i][4 * g + e] + bv, a, b, c);
                    h0[e] = a; h1[e] = b; h2[e] = c;
                }
                uint16_t* dst = p.Vt + ((int64_t)(img * 3) * nv + (n - p.vt_n0)) * p.n_rows + ml;
                *reinterpret_cast<bf16x4*>(dst) = h0;
                *reinterpret_cast<bf16x4*>(dst + (int64_t)nv * p.n_rows) = h1;
                *reinterpret_cast<bf16x4*>(dst + 2 * (int64_t)nv * p.n_rows) = h2;
            }
    };

    zero_acc();
    setup(tile);
    gload(0);
    for (;;) {
        const int tn_cur = tile % p.tiles_n;
        const bool vt = tn_cur * G3_BN >= p.vt_n0;  // workgroup-uniform
        for (int kt = 0; kt < nk; ++kt) {
            __syncthreads();
            lstore();
            __syncthreads();
            const int next = tile + slots;
            if (kt + 1 < nk) {
                gload(kt + 1);
            } else if (next < t_end) {
                setup(next);
                gload(0);
            }
            if (vt) compute(std::true_type{}); else compute(std::false_type{});
        }
        if (vt) epilogue_vt(tile); else epilogue(tile);
        const int next = tile + slots;
        if (next >= t_end) break;
        zero_acc();
        tile = next;
    }
}

int launch_gemm3(e2emv_ctx* ctx, const Gemm3Args& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return set_err(ctx, E2EMV_ESHAPE, "gemm3: empty problem");
    const int K1 = a.A2 ? a.K1 : a.K;
    if (a.K % G3_BK || K1 % G3_BK || K1 > a.K || (K1 < a.K && !a.A2))
        return set_err(ctx, E2EMV_ESHAPE, "gemm3: K=%d K1=%d must be multiples of %d", a.K, K1, G3_BK);
    if (a.N % 4 || (a.lda % 8) || (a.ldw % 8) || (a.A2 && a.lda2 % 8) || (a.C3 && a.ldc3 % 4) || (a.C32 && a.ldc32 % 4) ||
        (a.R && a.ldr % 4))
        return set_err(ctx, E2EMV_ESHAPE, "gemm3: leading dimensions must keep 16-byte alignment");
    const int vt_n0 = a.Vt ? a.vt_n0 : a.N;
    if (a.Vt && (vt_n0 % G3_BN || a.n_rows % 128 || a.M % a.n_rows))
        return set_err(ctx, E2EMV_ESHAPE, "gemm3: V^T output needs vt_n0 %% 128 == 0 and whole images");
    Gemm3Params p{};
    p.A = a.A; p.A2 = a.A2; p.W = a.W; p.bias = a.bias; p.R = a.R; p.C32 = a.C32; p.C3 = a.C3; p.Vt = a.Vt;
    p.lda = a.lda; p.lda2 = a.lda2; p.ldw = a.ldw; p.ldr = a.ldr; p.ldc32 = a.ldc32; p.ldc3 = a.ldc3;
    p.M = a.M; p.N = a.N; p.K = a.K; p.K1 = K1;
    p.tiles_m = (a.M + G3_BM - 1) / G3_BM;
    p.tiles_n = (a.N + G3_BN - 1) / G3_BN;
    p.total = p.tiles_m * p.tiles_n;
    p.relu = a.relu ? 1 : 0;
    p.q_cols = a.q_cols; p.q_scale = a.q_scale;
    p.vt_n0 = vt_n0; p.n_rows = a.n_rows > 0 ? a.n_rows : a.M;
    const int per_xcd = (p.total + 7) / 8;
    const int slots = std::min(per_xcd, std::max(1, ctx->num_cus * 2 / 8));
    const size_t lds = sizeof(uint16_t) * 6 * G3_PLANE;
    hipLaunchKernelGGL(gemm3_kernel, dim3(8 * slots), dim3(512), lds, s, p);
    E2EMV_CHECK_LAUNCH(ctx, "gemm3_kernel");
    return E2EMV_OK;
}

#endif  // E2EMV_STAMPS
// fp32 [rows][C] -> S3 [rows][3][ld] (test / ingest helper; producers normally emit S3 in their epilogue)
__global__ void split3_rows_kernel(const float* src, int64_t rows, int C, int64_t lds_, uint16_t* dst, int64_t ld) {
    const int64_t r = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        __bf16 a, b, d;
        split3(src[r * lds_ + c], a, b, d);
        __bf16* o = reinterpret_cast<__bf16*>(dst + r * 3 * ld);
        o[c] = a; o[ld + c] = b; o[2 * ld + c] = d;
    }
}

int launch_split3(e2emv_ctx* ctx, const float* src, int64_t rows, int C, int64_t ld_src, uint16_t* dst, int64_t ld,
                  hipStream_t s) {
    if (rows <= 0 || C <= 0) return E2EMV_OK;
    hipLaunchKernelGGL(split3_rows_kernel, dim3((unsigned)rows), dim3(256), 0, s, src, rows, C, ld_src, dst, ld);
    E2EMV_CHECK_LAUNCH(ctx, "split3_rows_kernel");
    return E2EMV_OK;
}

}  // namespace e2emv
