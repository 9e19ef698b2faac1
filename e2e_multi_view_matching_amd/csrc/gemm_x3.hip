// fp32-in / fp32-out "NT" GEMM on the gfx950 bf16 matrix pipe with 3-way split operands (fp32-class accuracy):
//   C[m][n] = act(sum_k A[m][k] W[n][k] + bias[n]) (+ R[m][n])
// Second generation of the split-operand GEMM (gemm3.hip keeps the all-planes-in-HBM variant as a tested building block).
// What changed, after measuring that the operand pipeline of the fp32 kernel (gemm.hip) moves 10.8 TB/s from L2 into LDS
// when the matrix cores are taken out - i.e. the first-generation kernel was not L2-bound, it was badly shaped:
//  * same skeleton as gemm.hip: 128x128x32 tile, 4 waves (2x2) x (2x2) 32x32 MFMA tiles, transposed accumulators
//    (lane = output row -> 16-byte epilogue), single LDS buffer + register prefetch, persistent workgroups with cross-tile
//    pipelining and XCD-aware tile ranges;
//  * ACTIVATIONS stay fp32 in HBM (4 B/element instead of 6 B of planes, and no producer has to emit planes): the split
//    x = x1 + x2 + x3 happens once per workgroup on the way from the prefetch registers into LDS (16 elements per thread
//    and K tile - noise next to 48 MFMAs per wave), LDS then holds three bf16 planes per operand;
//  * WEIGHTS are split once at commit time (S3 planes, ctx.hip);
//  * a product is the 6 largest of the 9 plane products (dropped terms <= 2^-25 |ab|), each exact in fp32, accumulated
//    in fp32 by v_mfma_f32_32x32x16_bf16: 48 MFMAs x 32 cycles per wave and K tile against 64 x 64 cycles for fp32.
// LDS rows are 40 bf16 (80 B): 16 rows x 16 B of a ds_read_b128 fall in 16 distinct 16-byte bank groups.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace e2emv {

typedef __attribute__((ext_vector_type(16))) float x3_f32x16;
typedef __attribute__((ext_vector_type(4))) float x3_f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 x3_bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 x3_bf16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int x3_u32x4;

constexpr int X3_BM = 128, X3_BN = 128, X3_BK = 32, X3_LD = 40;
constexpr int X3_PLANE = 128 * X3_LD;  // bf16 elements of one plane of one operand tile

struct GemmX3Params {
    const float* A;
    const float* A2;
    const uint16_t* W3;  // S3 [N][3][ldw3]
    const float* bias;
    const float* R;
    float* C;
    int64_t lda, lda2, ldw3, ldr, ldc;
    int M, N, K, K1;
    int tiles_m, tiles_n, total;
    int relu;
};

template <int DBG>
__global__ __launch_bounds__(256, 2) void gemm_x3_kernel(GemmX3Params p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t smem3[];
    uint16_t* As = smem3;                 // [3][128][40] activations (planes)
    uint16_t* Bs = smem3 + 3 * X3_PLANE;  // [3][128][40] weights

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int per_xcd = (p.total + 7) / 8;
    const int t_begin = xcd * per_xcd;
    const int t_end = min(t_begin + per_xcd, p.total);
    int tile = t_begin + slot;
    if (tile >= t_end) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    // activation tile: 128 rows x 32 fp32 = 8 chunks of 16 B per row -> 4 chunks per thread
    const int a_row = tid >> 3, a_c4 = (tid & 7) * 4;  // rows a_row + 32 i
    // weight tile: 128 rows x 3 planes x 32 bf16 = 12 chunks of 16 B per row -> 6 chunks per thread
    int w_lds[6];
    const int nk = p.K / X3_BK;

    const float* a_ptr[4];
    const float* a2_ptr[4];
    const uint16_t* w_ptr[6];
    auto setup = [&](int t) {
        const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ra = min(tm * X3_BM + a_row + 32 * i, p.M - 1);
            a_ptr[i] = p.A + (int64_t)ra * p.lda + a_c4;
            a2_ptr[i] = p.A2 ? p.A2 + (int64_t)ra * p.lda2 + a_c4 : nullptr;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int c = tid + 256 * i, row = c / 12, rem = c - row * 12, plane = rem >> 2, k8 = (rem & 3) * 8;
            const int rw = min(tn * X3_BN + row, p.N - 1);
            w_ptr[i] = p.W3 + ((int64_t)rw * 3 + plane) * p.ldw3 + k8;
            w_lds[i] = (plane * 128 + row) * X3_LD + k8;
        }
    };

    x3_f32x4 ra[4];
    x3_u32x4 rw[6];
    auto gload = [&](int kt) {
        if ((DBG & 2) && kt > 0) return;  // profiling: no operand traffic after the first K tile
        const int k = kt * X3_BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* src = (k < p.K1) ? a_ptr[i] + k : a2_ptr[i] + (k - p.K1);
            ra[i] = *reinterpret_cast<const x3_f32x4*>(src);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) rw[i] = *reinterpret_cast<const x3_u32x4*>(w_ptr[i] + k);
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x3_bf16x4 h0, h1, h2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = ra[i][e];
                const __bf16 a = (__bf16)v;
                const float r1 = v - (float)a;
                const __bf16 b = (__bf16)r1;
                h0[e] = a; h1[e] = b; h2[e] = (__bf16)(r1 - (float)b);
            }
            uint16_t* dst = &As[(a_row + 32 * i) * X3_LD + a_c4];
            *reinterpret_cast<x3_bf16x4*>(dst) = h0;
            *reinterpret_cast<x3_bf16x4*>(dst + X3_PLANE) = h1;
            *reinterpret_cast<x3_bf16x4*>(dst + 2 * X3_PLANE) = h2;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) *reinterpret_cast<x3_u32x4*>(&Bs[w_lds[i]]) = rw[i];
    };

    x3_f32x16 acc[2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
    };
    auto compute = [&]() {
        const uint16_t* as = &As[(wr * 64 + l31) * X3_LD + lh * 8];
        const uint16_t* bs = &Bs[(wc * 64 + l31) * X3_LD + lh * 8];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            x3_bf16x8 x[2][3], w[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    x[t][pl] = *reinterpret_cast<const x3_bf16x8*>(as + pl * X3_PLANE + t * 32 * X3_LD + ks * 16);
                    w[t][pl] = *reinterpret_cast<const x3_bf16x8*>(bs + pl * X3_PLANE + t * 32 * X3_LD + ks * 16);
                }
            if (DBG & 1) {  // profiling: operand pipeline only
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) acc[0][0][t] += (float)x[t][pl][0] + (float)w[t][pl][0];
                continue;
            }
            // smallest terms first; weights are the MFMA A operand (rows -> registers), activations B (rows -> lanes).
            // Consecutive MFMAs go to DIFFERENT accumulators: no back-to-back dependency on one accumulator tile.
            constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PX[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[j][PW[q]], x[i][PX[q]], acc[j][i], 0, 0, 0);
        }
    };
    auto epilogue = [&](int t) {
        const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = tm * X3_BM + wr * 64 + i * 32 + l31;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = tn * X3_BN + wc * 64 + j * 32 + 8 * g + 4 * lh;
                    if (n >= p.N) continue;
                    x3_f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[j][i][4 * g + e];
                    if (p.bias) v += *reinterpret_cast<const x3_f32x4*>(p.bias + n);
                    if (p.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = relu_nan(v[e]);
                    }
                    if (p.R) v += *reinterpret_cast<const x3_f32x4*>(p.R + (int64_t)m * p.ldr + n);
                    *reinterpret_cast<x3_f32x4*>(p.C + (int64_t)m * p.ldc + n) = v;
                }
        }
    };

    zero_acc();
    setup(tile);
    gload(0);
    for (;;) {
        for (int kt = 0; kt + 1 < nk; ++kt) {
            if (!(DBG & 4) || kt == 0) {  // (DBG & 4: profiling - LDS filled once, no barriers / LDS stores in the loop)
                __syncthreads();
                lstore();
                __syncthreads();
            }
            gload(kt + 1);
            compute();
        }
        const int next = tile + slots;
        const bool more = next < t_end;
        __syncthreads();
        lstore();
        __syncthreads();
        if (more) {
            setup(next);
            gload(0);
        }
        compute();
        epilogue(tile);
        if (!more) break;
        zero_acc();
        tile = next;
    }
}

int launch_gemm_x3(e2emv_ctx* ctx, const GemmArgs& a, const uint16_t* W3, int64_t ldw3, hipStream_t s, float h2_out_scale) {
    if (h2_out_scale != 0.f) return launch_gemm_h2(ctx, a, W3, ldw3, h2_out_scale, s);  // f16x2 planes: gemm_h2.hip
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || a.batch != 1) return set_err(ctx, E2EMV_ESHAPE, "gemm_x3: empty problem or batch != 1");
    const int K1 = a.A2 ? a.K1 : a.K;
    if (a.K % 32 || K1 % 32 || K1 > a.K || (K1 < a.K && !a.A2)) return set_err(ctx, E2EMV_ESHAPE, "gemm_x3: K=%d K1=%d must be multiples of 32", a.K, K1);
    if (!W3 || !a.C || a.N % 4 || a.ldc % 4 || (uintptr_t)a.C % 16 || (a.bias && (uintptr_t)a.bias % 16) ||
        (a.R && (a.ldr % 4 || (uintptr_t)a.R % 16)) || a.lda % 4 || (a.A2 && a.lda2 % 4) || ldw3 % 8 || a.scale != 1.f)
        return set_err(ctx, E2EMV_ESHAPE, "gemm_x3: needs 16-byte aligned rows (N %% 4, ld %% 4, ldw3 %% 8) and scale 1");
    GemmX3Params p;
    p.A = a.A; p.A2 = a.A2; p.W3 = W3; p.bias = a.bias; p.R = a.R; p.C = a.C;
    p.lda = a.lda; p.lda2 = a.lda2; p.ldw3 = ldw3; p.ldr = a.ldr; p.ldc = a.ldc;
    p.M = a.M; p.N = a.N; p.K = a.K; p.K1 = K1;
    p.tiles_m = (a.M + X3_BM - 1) / X3_BM;
    p.tiles_n = (a.N + X3_BN - 1) / X3_BN;
    p.total = p.tiles_m * p.tiles_n;
    p.relu = a.relu ? 1 : 0;
    const int per_xcd = (p.total + 7) / 8;
    const int sl = std::min(per_xcd, std::max(1, ctx->num_cus * 2 / 8));
    const size_t lds = sizeof(uint16_t) * 6 * X3_PLANE;
    static int dbg = -1;  // profiling knob E2EMV_X3_DEBUG: bit0 no MFMA, bit1 no operand loads after the first K tile
    if (dbg < 0) dbg = dbg_knob("E2EMV_X3_DEBUG", 0);
    if (dbg == 1) hipLaunchKernelGGL(gemm_x3_kernel<1>, dim3(8 * sl), dim3(256), lds, s, p);
    else if (dbg == 2) hipLaunchKernelGGL(gemm_x3_kernel<2>, dim3(8 * sl), dim3(256), lds, s, p);
    else if (dbg == 6) hipLaunchKernelGGL(gemm_x3_kernel<6>, dim3(8 * sl), dim3(256), lds, s, p);
    else hipLaunchKernelGGL(gemm_x3_kernel<0>, dim3(8 * sl), dim3(256), lds, s, p);
    E2EMV_CHECK_LAUNCH(ctx, "gemm_x3_kernel");
    return E2EMV_OK;
}

}  // namespace e2emv
